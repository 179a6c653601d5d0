set -u
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 900 python bench.py > gpurun_out/r04_bench_line_run$i.json 2> gpurun_out/r04_bench_stderr.log; echo "bench rc=$?"
python -c "
import json; b=json.loads(open('gpurun_out/r04_bench_line_run$i.json').read()); print(b['value'], b['roofline']['frac'], b['roofline']['traffic'], b['roofline'].get('frac_rocprof'), b['extra']['track']['roofline']['frac'], b['extra']['track']['roofline'].get('frac_rocprof'), b['per_rank'][0]['pci_bus_id'])"
done
rocm-smi --showpower --showtemp --showclocks 2>/dev/null | head -30
