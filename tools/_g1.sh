mkdir -p gpurun_out && cd /root/repo
( timeout 60 ./tools/rccl_same_device_probe > gpurun_out/rccl_probe.log 2>&1; echo "probe rc=$?" >> gpurun_out/rccl_probe.log )
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2_default.json 2> gpurun_out/bench_r2_default.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_r2_default.json
PROF_OUT=gpurun_out/prof_cfg1B PASSES="fetch write" timeout 900 bash tools/profile.sh > gpurun_out/prof_cfg1B.log 2>&1; echo "prof rc=$?"
tail -20 gpurun_out/prof_cfg1B.log
cat gpurun_out/rccl_probe.log
