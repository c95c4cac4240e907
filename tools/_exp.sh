TIMELINE=1 bash tools/gpu_trace.sh > gpurun_out/trace_out.txt 2>&1
tail -75 gpurun_out/trace_out.txt
for n in 2 4 8; do for flag in "" "--no-primal-cache"; do
python bench.py --shard-of $n --steps 40 --warmup 4 --no-cpu-baseline --no-roofline $flag 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shard-of $n $flag ms %.4f' % d['ms_per_step'])"
done; done
