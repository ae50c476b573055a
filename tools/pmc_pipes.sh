#!/bin/bash
# Where a wave's cycles go, per pipe (GPU box): two counter passes over bench.py with the kernel forced.
# usage: tools/pmc_pipes.sh TAG KERNEL_FILTER bench-args...
set -u
TAG=$1; KERN=$2; shift 2
cd "$(dirname "$0")/.." && ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/p$i" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-project-scaling --no-other-configs --steps 4 --warmup 1 "$@" > "$OUT/p$i.log" 2>&1) || { tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(out + "/p*/")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, dd in per.items():
        vals = sorted(dd.values()); med = vals[len(vals) // 2]
        full = [v for v in vals if v >= 0.2 * med]
        res.setdefault(k, sum(full) / max(1, len(full)))
wc = res["SQ_WAVE_CYCLES"]
print(kern, " ".join(f"{k[3:]}={v / wc:.3f}" for k, v in res.items() if k.startswith("SQ_") and k not in ("SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS")),
      f"| per wave cycle: VALU {res['SQ_INSTS_VALU'] / wc:.4f} SALU {res['SQ_INSTS_SALU'] / wc:.4f} LDS {res['SQ_INSTS_LDS'] / wc:.4f} instructions; wave cycles {wc:.4g}")
PY
