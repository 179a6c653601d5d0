python bench.py --workload track --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['layout'])"
python tools/track_probe.py 2>&1 | grep "auto"
