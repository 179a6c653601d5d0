set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04_final_gpu_tests.log
timeout 1200 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1; echo "profile rc=$?"
SUFFIX= timeout 2400 tools/table_rocprof.sh > /dev/null 2>&1; echo "table rc=$?"
timeout 900 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_stderr.log; echo "bench rc=$?"; cat gpurun_out/r04_bench_line.json | head -c 1500
