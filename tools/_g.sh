mkdir -p gpurun_out/r06
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -3 gpurun_out/r06_profile_round.log
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06/bench_final.err
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r06_bench_line.json").read().strip().splitlines()[-1])
r=l['extra']['stream_ring']
print(l['value'], l['ms_per_step'], {k:l['roofline'].get(k) for k in ('frac','frac_sustained','sustained_s','sustained_launches','frac_rocprof','frac_rocprof_sustained','traffic')})
print('ring', r['Msamples_per_s'], r['roofline']['achieved'], r['roofline']['peak'], r['roofline']['frac'], r['roofline']['copy_only_ring_GB_per_s'], r['GB_per_s_in_50ms_windows'], r['stream_probe_rounds'], r['roofline']['link'])
for k in ("track","track_256k","config4_chunk"):
    e=l["extra"][k]; print(' ', k, e["roofline"]["frac"], e["roofline"]["frac_settled"], e["roofline"].get("frac_rocprof"), e["roofline"].get("frac_rocprof_settled"), e["config"]["plan_ms"], e["config"]["one_shot_ms"], e["config"]["plan_parts_us"])
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline']['all_cores'], l['cpu_baseline']['gpu_output_bit_exact_on_sample'], l['legs_s'])
PY
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
