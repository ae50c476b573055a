#!/bin/bash
# Instruction-cache counters of the sample-loop kernel (GPU box). usage: tools/pmc_icache.sh TAG KERNEL_FILTER bench-args...
set -u
TAG=$1; KERN=$2; shift 2
cd "$(dirname "$0")/.." && ROOT=$PWD
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d "$OUT/p" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-project-scaling --no-other-configs --steps 4 --warmup 1 "$@" > "$OUT/p.log" 2>&1) || { tail -5 "$OUT/p.log"; exit 1; }
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
res = {}
for k, d in per.items():
    vals = sorted(d.values()); med = vals[len(vals) // 2]
    full = [v for v in vals if v >= 0.2 * med]
    res[k] = sum(full) / max(1, len(full))
req = res.get("SQC_ICACHE_REQ", 0)
print(kern, {k: f"{v:.4g}" for k, v in res.items()})
if req:
    print(f"  i-cache: hit rate {res['SQC_ICACHE_HITS'] / req:.4f}, misses {res['SQC_ICACHE_MISSES'] / req:.4f} (+ duplicate {res['SQC_ICACHE_MISSES_DUPLICATE'] / req:.4f}) of requests; "
          f"wave cycles waiting for an instruction {res['SQ_WAIT_INST_ANY'] / res['SQ_WAVE_CYCLES']:.3f}")
PY
