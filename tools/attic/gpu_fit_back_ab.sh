#!/bin/bash
# developer tool: back substitution with prefetched factor entries (fitw_back_substitute) against the library built before it
# (tools/ablate/lib_base.so): parity tests, Stage A at Ant's width, Humanoid's step, config 4's bench line.  Output: gpurun_out/fit_ab/
exec < /dev/null
R=gpurun_out/fit_ab
rm -rf $R && mkdir -p $R
timeout 400 python -m pytest tests -m gpu -q -x -k "sample_processing or humanoid_width or fit_ or any_layer_table or policy_step or device_rollout" 2>&1 | tail -3 | tee $R/pytest.txt
cp promp_amd/libpromp_hip.so /tmp/lib_new.so
pick='import json,sys; d=json.loads(sys.stdin.read()); print("%.3f ms/step" % d["ms_per_step"])'
for rep in 1 2; do for v in new base; do
  if [ $v = new ]; then cp /tmp/lib_new.so promp_amd/libpromp_hip.so; else cp tools/ablate/lib_base.so promp_amd/libpromp_hip.so; fi
  echo "$v: $(timeout 60 python tools/stage_a_timing.py 40 111)" | tee -a $R/stage_a.txt
  echo "$v: config 4 $(timeout 120 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-plugin-path 2>/dev/null | python -c "$pick")" | tee -a $R/ant.txt
done; done
for v in new base; do
  if [ $v = new ]; then cp /tmp/lib_new.so promp_amd/libpromp_hip.so; else cp tools/ablate/lib_base.so promp_amd/libpromp_hip.so; fi
  echo "== $v" | tee -a $R/humanoid.txt
  timeout 120 python tools/generic_timing.py --steps 5 --case 2 2>&1 | tail -8 | tee -a $R/humanoid.txt
done
cp /tmp/lib_new.so promp_amd/libpromp_hip.so
