set -u
mkdir -p gpurun_out
SUFFIX=_box2 timeout 2400 tools/table_rocprof.sh > /dev/null 2>&1; echo "table rc=$?"
timeout 900 python bench.py > gpurun_out/r04_bench_line_box2.json 2> gpurun_out/r04_bench_stderr.log; echo "bench rc=$?"
python -c "
import json; b=json.loads(open('gpurun_out/r04_bench_line_box2.json').read()); print(b['value'], b['roofline']['frac'], b['roofline']['traffic'], b['roofline'].get('frac_rocprof'), b['extra']['track']['roofline']['frac'], b['extra']['track']['roofline'].get('frac_rocprof'), b['per_rank'][0]['pci_bus_id'])"
