"""The one JSON line bench.py prints (driver contract): keys, types and arithmetic, checked without a GPU on the
function that builds it."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_result_line_schema():
    import bench
    args = argparse.Namespace(steps=300, warmup=10)
    n = bench.N_SAMPLES
    line = bench.build_result(args, 4, n, elapsed=0.100, avg_kernel_ms=0.3204, gather={"ms": 12.5})
    line = json.loads(json.dumps(line))            # must be JSON-serialisable
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)]:
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None               # BASELINE.md publishes no number for this metric
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["n_gpus"] == 4
    assert "workload" in line["config"] and "model" not in line["config"]
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert line["metric"] == json.load(f)["metric"]
    # whole-job aggregate: all ranks' samples over the max-over-ranks time
    assert abs(line["value"] - 4 * n * 300 / 0.100 / 1e6) < 1.0
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["achieved"] - n * 8 / 0.3204e-3 / 1e9) < 0.1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] == n * 8 and line["gather"] == {"ms": 12.5}
    # no process group in this test: the line says so (under the driver: "nccl" and the world RCCL reported)
    assert line["backend"] is None and line["world_size_seen"] == 1


def test_check_scale_names_what_is_off(tmp_path):
    """tools/check_scale.py against synthetic driver records: a curve as DESIGN.md section 6 predicts passes; a slow rank, two
    ranks on one GPU, a gloo backend, a wrong world size, a serialised gather and a skipped record are each named."""
    import copy
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_scale
    args = argparse.Namespace(steps=300, warmup=10)
    n = bench.N_SAMPLES

    def line(world, kernel_ms=0.320, slow_rank=None, gather_ms=8.0, ring_per_device=46.0, ring_nodes=None, shared=False):
        per_rank = [dict(rank=r, device=r, pci_bus_id="0000:%02x:00.0" % (5 + r), avg_kernel_ms=kernel_ms * (1.12 if r == slow_rank else 1.0),
                         timed_region_s=0.0962) for r in range(world)]
        worst = max(p["avg_kernel_ms"] for p in per_rank)
        ln = bench.build_result(args, world, n, elapsed=300 * worst * 1e-3, avg_kernel_ms=kernel_ms,
                                gather={"ms": gather_ms, "bytes_per_rank": 4 * n, "per_peer": [{"peer": p, "ms": 7.1, "GB_per_s": 151.0} for p in range(1, world)],
                                        "per_gpu_d2h": {"ms": 20.1},
                                        "product_ring": {"devices": list(range(world)), "GB_per_s_each_way_per_device": ring_per_device,
                                                         "GB_per_s_each_way_aggregate": ring_per_device * world, "submit_us_per_slab": 21.0,
                                                         "streams_share_a_queue": shared, "path": "staged",
                                                         "slab_numa_nodes": ring_nodes or [(k % world) * 2 // world for k in range(3 * world)]}}
                                if world > 1 else None, per_rank=per_rank)
        ln["backend"], ln["world_size_seen"] = ("nccl", world) if world > 1 else (None, 1)
        return json.loads(json.dumps(ln))

    good = {"runs": [{"parsed": line(w)} for w in (1, 2, 4, 8)]}
    p = tmp_path / "SCALE_ok.json"
    p.write_text(json.dumps(good))
    assert check_scale.main([str(p)]) == 0
    assert check_scale.check_line(line(8)) == []
    f = check_scale.check_line(line(8, slow_rank=5))
    assert any("rank 5" in x and "0000:0a:00.0" in x for x in f) and any("value" in x for x in f), f
    bad = line(4)
    bad["per_rank"][2]["pci_bus_id"] = bad["per_rank"][1]["pci_bus_id"]
    bad["backend"], bad["world_size_seen"] = "gloo", 3
    f = " | ".join(check_scale.check_line(bad))
    assert "distinct GPUs" in f and "backend 'gloo'" in f and "world_size_seen 3" in f, f
    f = " | ".join(check_scale.check_line(line(8, gather_ms=52.0)))
    assert "RCCL gather 52.0 ms" in f, f
    slow_peer = copy.deepcopy(line(4))
    slow_peer["gather"]["per_peer"][1]["ms"] = 30.0
    assert any("peer 2" in x for x in check_scale.check_line(slow_peer))
    # the N-device ring of the shipped command: slow because its streams share a queue / for another reason, all slabs on one node, missing
    f = " | ".join(check_scale.check_line(line(4, ring_per_device=27.0, shared=True)))
    assert "product ring 27.0 GB/s" in f and "share a hardware queue" in f, f
    f = " | ".join(check_scale.check_line(line(2, ring_per_device=20.0)))
    assert "product ring 20.0 GB/s" in f and "producer thread" in f, f
    assert check_scale.check_line(line(8, ring_per_device=33.0)) == []          # eight GPUs are expected to run into the host's memory
    f = " | ".join(check_scale.check_line(line(8, ring_nodes=[0] * 24)))
    assert "NUMA node 0" in f, f
    gone = line(2)
    del gone["gather"]["product_ring"]
    assert any("product_ring missing" in x for x in check_scale.check_line(gone))
    q = tmp_path / "SCALE_skipped.json"
    q.write_text(json.dumps({"skipped": True, "reason": "no 8-GPU node"}))
    assert check_scale.main([str(q)]) == 2
    assert check_scale.main([str(q), str(p)]) == 0
