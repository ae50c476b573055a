#!/bin/bash
# round 6, GPU call 17: the bench lines this round had not refreshed: C2 at the ends of its beta_m sweep, and the three scenes of SURVEY 8(f) (lobes, volumes, textured)
set -o pipefail
cd $GRAFT_REPO_ROOT; out=gpurun_out/g17; mkdir -p $out; export TMPDIR=/tmp
for bm in 0.1 0.6; do timeout -k 10 300 python3 bench.py --config C2 --beta-m $bm --steps 8 --no-other-configs --no-end-to-end > $out/bench_C2_betam$bm.json 2> $out/C2_$bm.err || echo "C2 $bm failed"; done
for sc in lobes volumes textured; do timeout -k 10 300 python3 bench.py --scene $sc --steps 8 --no-other-configs --no-end-to-end --no-project-scaling > $out/bench_$sc.json 2> $out/$sc.err || echo "$sc failed"; done
python3 - $out <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d.get('parity') or {}
    print(os.path.basename(f), d['value'], d['roofline']['frac'], d['roofline']['kernel'][:40], 'parity', p.get('spp'), p.get('ratio_to_floor'), p.get('share_within_4_sigma'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
