set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r04_final_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>/dev/null | python -c "
import sys, json; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline']['frac'], b['roofline']['traffic'], b['extra']['track']['roofline']['frac'])"
