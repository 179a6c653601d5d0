"""GPU, real processes: the multi-rank path with every rank calling the HIP kernel (the -m "not gpu" twin,
tests/test_sharding_gloo.py, has the oracle standing in for the kernel).

A test box has one GPU, so the ranks share device 0 and talk over gloo (RCCL refuses two ranks on one device); what is
exercised is everything above the transport: block-aligned time chunks, closed-form seeds (const) and per-segment seeds
(track), the kernel at a rank offset, per-rank D2H, the ordered gather, and bench.py's own multi-rank code path.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BPS = {"i16": 4, "f32": 8}


def _stream(intype, n, seed):
    rng = np.random.default_rng(seed)
    if intype == "i16":
        return rng.integers(-32768, 32768, size=2 * n, dtype=np.int16).view(np.uint8)
    return rng.uniform(-1, 1, size=2 * n).astype(np.float32).view(np.uint8)


def _cases():
    rate = 1024000
    track = [((rate // 1024 + 3 * (k % 5)) * 1024, float(np.float32(4000.0 - 37.25 * k))) for k in range(24)]
    track[-1] = (track[-1][0] + 77, track[-1][1])
    return [
        ("const 5001 Hz i16->i16 (odd period, seeds mid-period)", "i16", "i16", [(2048 * 4100 + 333, 5001.0)], rate),
        ("const 815 kHz f32->f32", "f32", "f32", [(1024 * 3001 + 5, 815000.0)], 2400000),
        ("track-shaped, 24 shifts, f32->i16", "f32", "i16", track, rate),
    ]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import doppler_amd
    from doppler_amd import shard
    ctx = doppler_amd.Context(0)                       # every rank on the one GPU of the box
    dev = torch.device("cuda", 0)
    results = []
    for ci, (name, intype, outtype, segs, rate) in enumerate(_cases()):
        n = sum(c for c, _ in segs)
        x = _stream(intype, n, 500 + ci)               # same stream on every rank
        lo, hi = shard.chunk_bounds(n, world, rank, bytes_per_sample=BPS[intype])
        before, inside = shard.segments_for_chunk(segs, lo, hi)
        seed = shard.seed_for_segments(before, rate)
        if len(segs) == 1:
            assert seed == shard.chunk_seed(segs[0][1], rate, lo)
        xd = torch.from_numpy(x[lo * BPS[intype]:hi * BPS[intype]].copy()).to(dev)
        out = torch.empty((hi - lo) * BPS[outtype], dtype=torch.uint8, device=dev)
        plan = ctx.plan_segments(inside, rate, samplenum=seed)
        plan.run_tensors(xd, out, intype, outtype)
        torch.cuda.synchronize(dev)
        plan.close()
        sizes = []
        for r in range(world):
            a, b = shard.chunk_bounds(n, world, r, bytes_per_sample=BPS[intype])
            sizes.append((b - a) * BPS[outtype])
        full = shard.ordered_gather(out.cpu(), sizes, dst=0)          # per-rank D2H, then ordered point-to-point gather
        if rank == 0:
            from oracle import oracle as orc
            want, _ = orc.segments_stream(x, intype, outtype, segs, rate, threads=min(os.cpu_count() or 1, 32))
            got = full.numpy()
            if outtype == "f32":
                g, w = got.view(np.float32), want.view(np.float32)
                ok = bool(np.all((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(w))))
            else:
                ok = bool(np.array_equal(got, want))
            results.append((name, ok))
        else:
            assert full is None
    if rank == 0:
        q.put(results)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_call_the_kernel_then_ordered_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(results) == len(_cases())
    for name, ok in results:
        assert ok, name


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_code_path_on_a_shared_gpu(world):
    """bench.py --gpus 2 and --gpus 8 exactly as the driver launches them, except that the ranks share GPU 0 over gloo
    (DPX_BENCH_SHARE_GPU=1; 8 x 2 GiB of streams + the 8 GiB gather buffer fit one MI355X): chunk seeds, barrier +
    max-over-ranks timing, the gather leg batched and peer by peer, every rank's own kernel time and device identity in
    the ONE JSON line from rank 0.  The numbers of such a run mean nothing and are not asserted."""
    env = dict(os.environ, DPX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + (os.getpid() % 200) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == 3 and line["scaling"] == "weak" and line["world_size_seen"] == world
    assert line["config"]["samples_per_gpu"] == 268435456
    assert "roofline" in line and "gather" in line and "error" not in line["gather"], line.get("gather")
    assert [r_["rank"] for r_ in line["per_rank"]] == list(range(world)) and len(line["per_rank_kernel_ms"]) == world
    assert all(r_["avg_kernel_ms"] > 0 and "pci_bus_id" in r_ for r_ in line["per_rank"]), line["per_rank"]
    assert [p["peer"] for p in line["gather"]["per_peer"]] == list(range(1, world))
    assert "error" not in line["gather"]["per_gpu_d2h"], line["gather"]["per_gpu_d2h"]
    # the N-device ring of the shipped command (dpx_stream_create_multi from rank 0's process; here: one GPU listed N times)
    pr = line["gather"]["product_ring"]
    assert "error" not in pr, pr
    assert pr["devices"] == [0] * world and len(pr["slab_numa_nodes"]) == 3 * world and pr["Msamples_per_s"] > 1000, pr
    # tools/check_scale.py reads the line as it will read the driver's SCALE record, and names what a shared-GPU run gets
    # wrong BY CONSTRUCTION: gloo instead of RCCL, every rank on the same GPU (the numbers it flags besides mean nothing here)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_scale
    findings = " | ".join(check_scale.check_line(line))
    assert "backend 'gloo'" in findings and "distinct GPUs" in findings, findings
    assert "world_size_seen" not in findings and "per_rank has" not in findings, findings


def test_bench_rccl_branch_with_one_rank():
    """The branch the driver's 2/4/8-GPU runs take — init_process_group(backend="nccl", device_id=...), dist.barrier,
    all_reduce(MAX) of the elapsed time, the ordered gather (batch_isend_irecv path) and the per-GPU D2H leg — run under a
    real RCCL with ONE rank (DPX_BENCH_FORCE_DIST=1), launched through torch.distributed.run exactly like the driver does.
    The JSON line names the backend and the world size the process group reported."""
    env = dict(os.environ, DPX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DPX_BENCH_SHARE_GPU", None)
    port = 29400 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu", "--no-extra"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["backend"] == "nccl" and line["world_size_seen"] == 1 and line["n_gpus"] == 1
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["avg_kernel_ms"] > 0 and "pci_bus_id" in line["per_rank"][0]
    assert "error" not in line["gather"] and "error" not in line["gather"]["per_gpu_d2h"], line["gather"]
    assert line["gather"]["ms"] > 0 and line["roofline"]["frac"] > 0.5
    # and the track workload's barrier / all_reduce under the same process group
    cmd = cmd[:-2] + ["--workload", "track"]
    cmd[cmd.index("--master-port") + 1] = str(port + 1)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["roofline"]["frac"] > 0.4
