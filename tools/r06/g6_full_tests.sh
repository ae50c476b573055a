#!/bin/bash
# round 6, GPU call 6: the whole GPU suite on the device-side upload + regrouped path slot, the command's laps again (2 s between processes), and
# FETCH_SIZE / WRITE_SIZE of k_stream on C3 for the slot's sectors (item 3b)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g6; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $out/pytest_gpu.txt
S=$(python3 -c "
import sys; sys.path.insert(0,'tools'); import make_scenes
print(make_scenes.ensure_scene('sphere-hairblock','/tmp/yhair_scenes',scale=1.0))" | tail -1)
export YHAIR_CACHE_DIR=/tmp/yh_e2e_cache; rm -rf $YHAIR_CACHE_DIR
for k in cold warm warm2; do sleep 2
  echo "--- $k" | tee -a $out/e2e_laps.txt
  ( time YHAIR_TIMING=1 yocto-hair_amd/yscenetrace $S -r 720 -s 1536 -o /tmp/out.pfm --timing ) 2>&1 | grep -v "amdgpu.ids\|kernel times\|launch shape: worth" | tee -a $out/e2e_laps.txt
done
unset YHAIR_CACHE_DIR
# the counters of k_stream on C3 (FETCH_SIZE, WRITE_SIZE): the passes of tools/pmc_k_trace.sh that matter for the slot
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && YHAIR_SHAPE=3 timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $GRAFT_REPO_ROOT/$out/p$i -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --config C3 --spp-per-step 32 --no-project-scaling --no-other-configs --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/$out/p$i.log 2>&1) || echo "pmc pass $i failed"
done
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_stream<" not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in per.items():
        vals = sorted(d.values()); med = vals[len(vals) // 2]
        full = [v for v in vals if v >= 0.2 * med]
        res[k] = sum(full) / max(1, len(full))
json.dump(res, open(out + "/C3_slot_pmc.json", "w"), indent=1); print(json.dumps(res))
PY
rm -rf $out/p1 $out/p2
