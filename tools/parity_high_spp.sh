#!/bin/bash
# The dense configs against the reference at MORE samples than the bench's budgeted CPU leg renders: C2 at 512 spp, C3 at 64, C4 at 128 —
# two seeds of the reference each (the seed-to-seed floor) on the box's host threads. usage: tools/parity_high_spp.sh TAG   (about ten minutes)
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-parity_hi}; mkdir -p $out; export TMPDIR=/tmp
common="--steps 8 --no-other-configs --no-project-scaling --no-end-to-end --no-trial-cache"
timeout -k 10 500 python3 bench.py --config C2 --beta-m 0.25 --cpu-spp 512 $common > $out/parity_high_spp_C2.json 2> $out/C2.err || { tail -3 $out/C2.err; exit 1; }
echo "C2 done"
timeout -k 10 400 python3 bench.py --config C3 --cpu-spp 64 $common > $out/parity_high_spp_C3.json 2> $out/C3.err || { tail -3 $out/C3.err; exit 1; }
echo "C3 done"
timeout -k 10 400 python3 bench.py --config C4 --cpu-spp 128 $common > $out/parity_high_spp_C4.json 2> $out/C4.err || { tail -3 $out/C4.err; exit 1; }
python3 - $out <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/parity_high_spp_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['parity']
    print(f.split('/')[-1], d['value'], {k:p.get(k) for k in ('spp','rel_rmse','seed_floor','ratio_to_floor','pixels_within_4_sigma','alpha_equal')}, d['cpu_baseline']['value'])
PY
