mkdir -p gpurun_out/r06
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
