#!/bin/bash
# developer tool: k_gram_tiled against k_gram_wide (PROMP_GRAM_UNTILED=1) -- parity tests of the wide baseline fits, Stage A wall
# time at Ant's and Humanoid's widths, per-kernel times from a kernel trace.  Output: gpurun_out/gram_ab/
R=gpurun_out/gram_ab
rm -rf $R && mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sample_processing or humanoid_width or fit_" > $R/pytest.log 2>&1; tail -3 $R/pytest.log
for O in 111 376; do
  for rep in 1 2; do
    timeout 200 python tools/stage_a_timing.py 40 $O 2>&1 | sed 's/^/tiled   /' | tee -a $R/stage_a.txt
    PROMP_GRAM_UNTILED=1 timeout 200 python tools/stage_a_timing.py 40 $O 2>&1 | sed 's/^/untiled /' | tee -a $R/stage_a.txt
  done
done
ROOT=$GRAFT_REPO_ROOT
cd /tmp
for O in 111 376; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/trace$O -o t -- python $ROOT/tools/stage_a_timing.py 40 $O > /dev/null 2> $ROOT/$R/trace$O.err
  f=$(find $ROOT/$R/trace$O -name "*kernel_stats.csv" | head -1); echo "--- O=$O"; head -8 $f | cut -d, -f1-4 | tee -a $ROOT/$R/kernels.txt
  find $ROOT/$R/trace$O -name "*kernel_trace.csv" -delete
done
