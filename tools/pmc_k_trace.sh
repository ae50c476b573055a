#!/bin/bash
# PMC passes over `bench.py --steps 4` (k_trace). Each --pmc group is its own run (no trace
# domains besides --kernel-trace). Usage on the GPU box: bash tools/pmc_k_trace.sh <tag>
# Other configs: PMC_KERNEL='k_trace<false, false, 256, 6>' PMC_ARGS='--scene straight-hair ...' bash tools/pmc_k_trace.sh <tag>
set -u
TAG=${1:-pmc}
cd "$(dirname "$0")/.." && ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum"; do   # (round 5: MemUnitBusy = TA busy / GPU active, occupancy = wave cycles / GPU active; where the L2's read requests go)
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/p$i" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline ${PMC_ARGS:-} --steps 4 --warmup ${PMC_WARMUP:-1} > "$OUT/p$i.log" 2>&1)
done
python3 - "$OUT" "${PMC_KERNEL:-k_trace<false, false, 512, 4>}" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # counter -> dispatch -> value
        for r in csv.DictReader(open(f)):
            if sys.argv[2] not in r["Kernel_Name"]:  # the launch shape of the config under test only
                continue
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, d in per.items():
            vals = sorted(d.values())
            med = vals[len(vals) // 2]
            full = [v for v in vals if v >= 0.2 * med]  # without the 1-spp probe launch of yh_init_state
            res[k] = sum(full) / max(1, len(full))
json.dump(res, open(out + "/k_trace_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
