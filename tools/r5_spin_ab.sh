cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/spin
for r in 1 2 3; do
  for v in spin nospin; do
    if [ $v = nospin ]; then export YHAIR_NO_SPIN=1; else unset YHAIR_NO_SPIN; fi
    timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-project-scaling --no-end-to-end > gpurun_out/spin/$v$r.json 2> gpurun_out/spin/$v$r.err || exit 1
    python3 - $v $r <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/spin/{sys.argv[1]}{sys.argv[2]}.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))
PY
  done
done
