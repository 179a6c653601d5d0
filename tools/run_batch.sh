set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r04b_parity.log 2>&1; echo "parity rc=$?" ; tail -3 gpurun_out/r04b_parity.log
ONLY='sincos per sample|replay 300' SUFFIX=_pairs_layout timeout 1500 tools/table_rocprof.sh > /dev/null 2>&1; cat gpurun_out/r04_table_rocprof_pairs_layout.md
