timeout 900 python -m pytest tests/test_gpu_soak.py -m gpu -x -q -s --durations=5 2>&1 | grep -v amdgpu.ids | tail -25
