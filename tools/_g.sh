mkdir -p gpurun_out/r06
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -5 gpurun_out/r06_profile_round.log
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json      # so that the bench line below quotes it (same sources)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06/bench_final.err
tail -2 gpurun_out/r06/bench_final.err
timeout 900 python bench.py > gpurun_out/r06_bench_line_default.json 2>/dev/null
DPX_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r06_bench_line_2ranks_shared_gpu.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_line.json", "gpurun_out/r06_bench_line_default.json"):
    l=json.loads(open(f).read().strip().splitlines()[-1])
    r=l['extra']['stream_ring']
    print(f, l['value'], l['ms_per_step'], l['steps'], {k:l['roofline'].get(k) for k in ('frac','frac_sustained','frac_rocprof','frac_rocprof_sustained','traffic')})
    print('  ring', r['Msamples_per_s'], r['roofline']['achieved'], r['roofline']['peak'], r['roofline']['frac'], r['roofline']['copy_only_ring_GB_per_s'], r['GB_per_s_in_50ms_windows'], r['stream_probe_rounds'])
    for k in ("track","track_256k","config4_chunk"):
        e=l["extra"][k]; print('  ', k, e["roofline"]["frac"], e["roofline"]["frac_settled"], e["roofline"].get("frac_rocprof"), e["roofline"].get("frac_rocprof_settled"), e["config"]["plan_ms"], e["config"]["one_shot_ms"])
    print('  cpu', l['cpu_baseline']['value'], l['cpu_baseline']['all_cores'], l['cpu_baseline']['gpu_output_bit_exact_on_sample'], l['host_round_trip'], l['legs_s'])
l=json.loads(open("gpurun_out/r06_bench_line_2ranks_shared_gpu.json").read().strip().splitlines()[-1])
print(json.dumps(l['gather'].get('product_ring'))[:900]); print(l['gather'].get('product_ring_rccl'))
PY
