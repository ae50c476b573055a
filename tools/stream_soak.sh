#!/bin/bash
# Developer tool (GPU box): the product's k_stream over and over, every run its own process under a timeout, output streamed — 10 x the seven small check
# scenes (bitwise against the quad kernel), 10 x C3 and 10 x C2 at the configs' sizes. usage: tools/stream_soak.sh TAG
set -o pipefail  # (a pipeline fails when ANY stage does: `|| exit` then sees timeout's and python's status, not tail's)
cd $GRAFT_REPO_ROOT; out=gpurun_out/${1:-soak}; mkdir -p $out; export TMPDIR=/tmp YHAIR_NO_DISK_CACHE=1
log=$out/k_stream_repeat.log; : > $log
for i in $(seq 1 10); do
  echo "run $i small" | tee -a $log
  WF_SHAPE=3 timeout -k 10 60 python3 tools/shape_check.py check 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i small)" | tee -a $log; exit 1; }
done
for i in $(seq 1 10); do
  echo "run $i C3" | tee -a $log
  timeout -k 10 90 python3 tools/shape_check.py curly-hair 1280 32 3 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i C3)" | tee -a $log; exit 1; }
  echo "run $i C2" | tee -a $log
  timeout -k 10 90 python3 tools/shape_check.py straight-hair 720 64 3 2>&1 | tee -a $log | tail -1 || { echo "FAILED or TIMED OUT (run $i C2)" | tee -a $log; exit 1; }
done
# ... and the log has to SAY so: 10 x 7 bitwise-equal small scenes, 20 re-renders of the first image at the configs' sizes
small=$(grep -c "images equal True  rng equal True" $log); big=$(grep -c "same image as first: True" $log); bad=$(grep -c "equal False\|as first: False" $log)
if [ "$small" -ne 70 ] || [ "$big" -ne 20 ] || [ "$bad" -ne 0 ]; then echo "INCOMPLETE: $small of 70 small-scene checks, $big of 20 same-image checks, $bad mismatches" | tee -a $log; exit 1; fi
echo "all 30 runs completed ($small bitwise checks of the small scenes, $big same-image checks at the configs' sizes, 0 mismatches)" | tee -a $log
