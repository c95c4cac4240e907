#!/bin/bash
# One full measured round: GPU parity tests, smoke, bench (with cpu baseline), rocprofv3 kernel trace (csv stats),
# PMC passes for HBM traffic.  Everything lands under gpurun_out/round/.
R=gpurun_out/round
rm -rf $R && mkdir -p $R
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > $R/device.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s > $R/pytest_gpu_verbose.log 2>&1; echo "pytest rc=$?" >> $R/pytest_gpu_verbose.log
grep -E "passed|failed|pytest rc" $R/pytest_gpu_verbose.log > $R/pytest_gpu.log; grep -E "full-size parity|adam golden|trpo golden" $R/pytest_gpu_verbose.log > $R/full_size_parity.txt; cat $R/pytest_gpu.log $R/full_size_parity.txt; rm -f $R/pytest_gpu_verbose.log
python tools/split_accuracy_gpu.py > $R/split_accuracy.txt 2>&1; PROMP_WIDE_FP32=1 python tools/split_accuracy_gpu.py 2>&1 | tail -2 | sed 's/^/exact-FP32 cooperative kernels (PROMP_WIDE_FP32=1): /' >> $R/split_accuracy.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke rc=$?" >> $R/smoke.log; tail -2 $R/smoke.log
timeout 900 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; cat $R/bench.json | head -c 400; echo
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/trace -o trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path > $ROOT/$R/trace_bench.json 2> $ROOT/$R/trace.err; echo "trace rc=$?"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$R/pmc$i -o pmc$i -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/$R/pmc$i.err
  echo "pmc pass $i rc=$?"
done
# BASELINE config 4 (Ant shapes, cooperative kernels): bench line + kernel trace
timeout 600 python $ROOT/bench.py --config 4 --steps 10 --warmup 2 > $ROOT/$R/bench_config4.json 2> $ROOT/$R/bench_config4.err; echo "bench config4 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$R/trace4 -o trace4 -- python $ROOT/bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-plugin-path > /dev/null 2> $ROOT/$R/trace4.err; echo "trace4 rc=$?"
cd $ROOT; TIMELINE_STEP=2 python tools/timeline.py $R/trace4 > $R/timeline_config4.txt 2>&1; rm -f $R/trace4/*kernel_trace.csv
# config 4 with the exact-FP32 cooperative kernels (the A/B of the BF16-pipe kernels, same box), cycle stamps, PMC counters
PROMP_WIDE_FP32=1 timeout 600 python $ROOT/bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path > $ROOT/$R/bench_config4_fp32_kernels.json 2> /dev/null; echo "bench config4 fp32 rc=$?"
bash tools/gpu_wb_stamps.sh 0,1 > $ROOT/$R/wb_stamps.txt 2>&1
bash tools/gpu_pmc_config4.sh > /dev/null 2>&1; cp gpurun_out/pmc4/summary.txt $ROOT/$R/pmc_config4.txt
# BASELINE config 5 (TRPO-MAML on config 3's shapes)
timeout 600 python $ROOT/bench.py --config 5 --steps 5 --warmup 1 > $ROOT/$R/bench_config5.json 2> $ROOT/$R/bench_config5.err; echo "bench config5 rc=$?"
# one rank's share of the fixed 40-task batch at 2 / 4 / 8 ranks, timed on this one GPU (no collective: kernels only)
for n in 2 4 8; do
  timeout 300 python $ROOT/bench.py --shard-of $n --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard-of $n: %d tasks on this GPU, %.4f ms/step' % (d['config']['tasks_per_gpu'], d['ms_per_step']))" >> $ROOT/$R/shard_timings.txt
done
cat $ROOT/$R/shard_timings.txt
# ... and of BASELINE config 4's batch (Ant shapes)
for n in 2 4 8; do
  timeout 300 python $ROOT/bench.py --config 4 --shard-of $n --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shard-of $n: %d tasks on this GPU, %.4f ms/step' % (d['config']['tasks_per_gpu'], d['ms_per_step']))" >> $ROOT/$R/shard_timings_config4.txt
done
cat $ROOT/$R/shard_timings_config4.txt
# primal cache on (default rule) / forced off: full batch and shards, same box, back to back
for n in 0 2 4 8; do for flag in "" "--no-primal-cache"; do
  sh=""; [ $n -gt 0 ] && sh="--shard-of $n"
  timeout 300 python $ROOT/bench.py $sh --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-plugin-path --repeats 1 $flag 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-12s %-18s %d tasks on this GPU, %.4f ms/step' % ('$sh' or 'full batch', '$flag' or 'default', d['config']['tasks_per_gpu'], d['ms_per_step']))" >> $ROOT/$R/primal_cache_ab.txt
done; done
cat $ROOT/$R/primal_cache_ab.txt
# staged uploads: kernel + memory-copy trace of a run fed only by promp_stage_step, and the copy / compute overlap in it
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/$R/trace_staged -o staged -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-plugin-path --staged-only > $ROOT/$R/bench_staged.json 2> $ROOT/$R/trace_staged.err; echo "staged trace rc=$?"
cd $ROOT
python tools/overlap.py $R/trace_staged > $R/h2d_overlap.txt 2>&1; cat $R/h2d_overlap.txt
rm -f $R/trace_staged/*kernel_trace.csv $R/trace_staged/*memory_copy_trace.csv
# policy shapes outside the fused kernels (layer-by-layer kernels), and Stage A standalone (config 3 and Ant's size)
timeout 600 python $ROOT/tools/generic_timing.py > $ROOT/$R/generic_shapes.txt 2>&1; cat $ROOT/$R/generic_shapes.txt
(timeout 300 python $ROOT/tools/stage_a_timing.py; timeout 300 python $ROOT/tools/stage_a_timing.py 40 111) > $ROOT/$R/stage_a.txt 2>&1; cat $ROOT/$R/stage_a.txt
# per-launch timeline of one step (both streams) from the kernel trace
python $ROOT/tools/timeline.py $ROOT/$R/trace > $ROOT/$R/timeline.txt 2>&1
cd $ROOT; rm -f $R/trace/*kernel_trace.csv $R/pmc*/*kernel_trace.csv   # keep the summaries small
ls -R $R | head -40
