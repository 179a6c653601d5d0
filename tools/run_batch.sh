set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04_final_gpu_tests.log
ONLY='replay 300' SUFFIX=_route timeout 900 tools/table_rocprof.sh > /dev/null 2>&1; cat gpurun_out/r04_table_rocprof_route.md
