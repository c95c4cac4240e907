#!/bin/bash
# The measured round in its short form (the GPU budget of a session's last hour): parity tests, smoke, the driver's bench line,
# kernel traces of configs 3 and 4, the four PMC passes the bench line's `traffic` comes from, Stage A alone, Humanoid's step.
# Every command under its own timeout and with stdin closed.  Output: gpurun_out/round/ (tools/summarize_round.py copies it).
R=gpurun_out/round
rm -rf $R && mkdir -p $R
export TMPDIR=/tmp
exec < /dev/null
ROOT=$GRAFT_REPO_ROOT
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $R/device.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -s > $R/pytest_gpu_verbose.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu_verbose.log
grep -E "passed|failed|pytest rc" $R/pytest_gpu_verbose.log > $R/pytest_gpu.log; grep -E "full-size parity|adam golden|trpo golden" $R/pytest_gpu_verbose.log > $R/full_size_parity.txt; cat $R/pytest_gpu.log; grep -E "^FAILED|^ERROR" $R/pytest_gpu_verbose.log | head; rm -f $R/pytest_gpu_verbose.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log; tail -2 $R/smoke.log
timeout 400 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; head -c 300 $R/bench.json; echo
timeout 300 python bench.py --config 4 --steps 10 --warmup 2 > $R/bench_config4.json 2> $R/bench_config4.err; echo "bench config4 rc=$?"; head -c 300 $R/bench_config4.json; echo
(timeout 100 python tools/stage_a_timing.py; timeout 100 python tools/stage_a_timing.py 40 111) > $R/stage_a.txt 2>&1; cat $R/stage_a.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path > $ROOT/$R/trace_bench.json 2> $ROOT/$R/trace.err; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/trace4 -o trace4 -- python $ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/$R/trace4.err; echo "trace4 rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$R/pmc$i -o pmc$i -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/$R/pmc$i.err
  echo "pmc pass $i rc=$?"
done
cd $ROOT
python tools/timeline.py $R/trace > $R/timeline.txt 2>&1
TIMELINE_STEP=3 python tools/timeline.py $R/trace4 > $R/timeline_config4.txt 2>&1
rm -f $R/trace/*kernel_trace.csv $R/trace4/*kernel_trace.csv $R/pmc*/*kernel_trace.csv
timeout 300 python tools/generic_timing.py --case 2 --steps 5 > $R/generic_humanoid.txt 2>&1; head -8 $R/generic_humanoid.txt
ls $R
