timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('track','track_256k','config4_chunk'):
    e=l['extra'][k]; print(k, e['config']['plan_ms'], e['config']['one_shot_ms'], e['config']['plan_parts_us'], e['roofline']['frac'], e['roofline']['frac_settled'])
print(l['config']['plan_ms'], l['config']['one_shot_ms'], l['roofline']['frac'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_ring.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -3
