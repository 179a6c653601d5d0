set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_kernel_random or const_stream_bulk or per_sample" > gpurun_out/r04b_parity.log 2>&1; echo "parity rc=$?" ; tail -5 gpurun_out/r04b_parity.log
