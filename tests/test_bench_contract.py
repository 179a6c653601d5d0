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
