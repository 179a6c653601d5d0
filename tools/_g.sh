mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06/gpu_suite.log
cat gpurun_out/r06/gpu_suite.log
