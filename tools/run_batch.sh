set -u
mkdir -p gpurun_out
cp doppler_amd/lib/libdoppler_hip.so /tmp/cur.so
cp tools/bin/libdoppler_hip_w1.so doppler_amd/lib/libdoppler_hip.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "span_kernel_plans_vs_oracle" 2>&1 | tail -2
timeout 1200 python tools/ab.py --set w1 --rounds 10 --iters 10 2>gpurun_out/r04b_w1.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s %-8s %-40s %-30s %6.1f' % (d['case'][:24], d['pair'], str(d['opts']), d.get('kernel', '')[:30], d['pct_peak']))" | tee gpurun_out/r04b_w1.log
tail -3 gpurun_out/r04b_w1.err
cp /tmp/cur.so doppler_amd/lib/libdoppler_hip.so
