DPX_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 5 --warmup 2 2>/tmp/err | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=l['gather']; print({k:(v if k!='product_ring' else (v.get('GB_per_s_each_way_aggregate'), v.get('error'))) for k,v in g.items() if k.startswith('product')}, l['legs_s'])"
tail -3 /tmp/err
DPX_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['backend'], l['gather'].keys())"
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_sharding_gloo.py -m gpu -x -q 2>&1 | tail -3
