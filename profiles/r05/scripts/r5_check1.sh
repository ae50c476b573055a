#!/bin/bash
# Round 5: the new GPU tests, the bench line with its new fields (wall-clock of the whole command), the counters rocprofv3 offers on this box.
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-r5check1}; mkdir -p $out; export TMPDIR=/tmp
(time timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "deadline or persist_on_disk or cut_off or cli_ or bare_command" --durations=6) > $out/gputests_subset.log 2>&1; tail -12 $out/gputests_subset.log
(cd /tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$out/list_avail.txt 2>&1); grep -i -c "gfx950\|counter" $out/list_avail.txt
(time timeout -k 10 900 python3 bench.py > $out/bench_C1_no_flags.json 2> $out/bench_C1_no_flags.err) 2> $out/bench_time.txt || { tail -5 $out/bench_C1_no_flags.err; exit 1; }
cat $out/bench_time.txt; python3 - $out/bench_C1_no_flags.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['parity']['spp'], d['parity']['ratio_to_floor'], d['cpu_baseline']['value'])
for k,v in d['config']['other_configs']['runs'].items(): print(k, v.get('value'), (v.get('parity') or {}).get('spp'), (v.get('parity') or {}).get('ratio_to_floor'), (v.get('parity') or {}).get('share_within_4_sigma'), (v.get('parity') or {}).get('alpha_identical'), (v.get('cpu_baseline') or {}).get('value'), v.get('error'))
print(json.dumps(d['config'].get('end_to_end'), indent=1)[:3000])
PY
