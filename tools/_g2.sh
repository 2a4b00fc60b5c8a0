cd /root/repo; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/pytest_comm.log 2>&1; echo "comm rc=$?"; tail -30 gpurun_out/pytest_comm.log
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_comm.py > gpurun_out/pytest_gpu.log 2>&1; echo "gpu rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --force-merge --rows 100000000 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/bench_fm.json 2> gpurun_out/bench_fm.err; echo "fm rc=$?"; tail -c 1500 gpurun_out/bench_fm.json; tail -5 gpurun_out/bench_fm.err
timeout 300 python bench.py --force-merge --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fm5.json 2> gpurun_out/bench_fm5.err; echo "fm5 rc=$?"; tail -c 800 gpurun_out/bench_fm5.json; tail -5 gpurun_out/bench_fm5.err
