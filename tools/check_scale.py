"""Hold a driver record of `bench.py --gpus N` (SCALE_rNN.json, BENCH_rNN.json, or a file of bench lines) against what
DESIGN.md section 6 predicts, and say what is off — so that the first real 1/2/4/8-GPU curve is diagnosed the hour it arrives.

    python tools/check_scale.py SCALE_r05.json [more records ...]

No scaling curve has been measured in any round so far (the driver had no 8-GPU node); the path shards into independent
time chunks (reference src/main.rs:60: the whole carried state is one u32), so the prediction is plain weak scaling:
  value            N x (0.83-0.85) x 10^6 Msamples/s (the N = 1 figure of the same record where there is one)
  per-rank kernel  kernel_ms_spread.max_over_min <= 1.05 (boxes differ by +-2 %)
  process group    world_size_seen == N, backend nccl, N distinct pci_bus_id values
  gather           RCCL into rank 0 within 2x of (N-1) GiB over N-1 xGMI links of ~153 GB/s each, i.e. ~7-10 ms however many
                   peers; per peer within 2x of 1 GiB / 153 GB/s = 7 ms; per-GPU D2H within 2x of 1 GiB / 55 GB/s = 19.5 ms
  product ring     gather.product_ring (dpx_stream_create_multi from one process, what `doppler --gpus N` runs on): N x the one-GPU
                   ring's 45-47 GB/s each way while the GPUs' root complexes and the host's memory keep up — two sockets of
                   DDR5 give ~2 x 300 GB/s of DMA in + out, so N = 8 (8 x 47 x 2 = 750 GB/s) is expected to fall short of 8x, and
                   one Python producer thread (20-40 us per slab) caps the ring near 64 MiB / 30 us; slabs on the GPU's own NUMA node
Exit status 0 when nothing is off, 1 otherwise (2: no bench line found)."""
import json
import sys

PER_GPU = (0.80e6, 0.87e6)         # Msamples/s a single MI355X delivers on the headline (0.83-0.85 measured; margins for box spread)
XGMI_GBPS, PCIE_GBPS = 153.0, 55.0
RING_GBPS = (40.0, 49.0)           # the one-GPU slab ring, each way (profiles/r06_ring.md: 45.6-47.5 at 16-64 MiB slabs)
HOST_DMA_GBPS = 600.0              # in + out, both sockets: where an 8-GPU ring is expected to run into the host's memory


def bench_lines(obj):
    """every dict that looks like a bench line, wherever the record keeps it (parsed / runs / lines / plain)"""
    found = []
    if isinstance(obj, dict):
        if "metric" in obj and "value" in obj and "n_gpus" in obj:
            found.append(obj)
        for v in obj.values():
            found += bench_lines(v)
    elif isinstance(obj, list):
        for v in obj:
            found += bench_lines(v)
    elif isinstance(obj, str) and obj.lstrip().startswith("{") and '"metric"' in obj:
        for ln in obj.splitlines():
            ln = ln.strip()
            if ln.startswith("{"):
                try:
                    found += bench_lines(json.loads(ln))
                except ValueError:
                    pass
    return found


def check_line(line, base_value=None):
    """list of findings (strings) for one bench line"""
    off = []
    n = int(line["n_gpus"])
    v = float(line["value"])
    lo, hi = (PER_GPU[0] * n, PER_GPU[1] * n) if base_value is None else (0.93 * base_value * n, 1.05 * base_value * n)
    if not lo <= v <= hi:
        off.append("value %.0f Msamples/s outside the predicted %.0f-%.0f for N=%d (%s)" % (
            v, lo, hi, n, "N x the 0.83-0.85 M of one GPU" if base_value is None else "N x this record's own N=1 value, -7 %/+5 %"))
    if line.get("scaling") != "weak":
        off.append("scaling is %r, the path is weak-scaled (one 1 GiB chunk per rank)" % line.get("scaling"))
    if n > 1:
        if line.get("world_size_seen") != n:
            off.append("world_size_seen %r != n_gpus %d: the process group did not have one rank per GPU" % (line.get("world_size_seen"), n))
        if line.get("backend") != "nccl":
            off.append("backend %r: the driver's run goes over RCCL ('nccl'); gloo means the development mode on a shared GPU" % line.get("backend"))
        ranks = line.get("per_rank") or []
        if len(ranks) != n:
            off.append("per_rank has %d entries for %d GPUs" % (len(ranks), n))
        ids = [r.get("pci_bus_id") for r in ranks if r]
        if len(set(ids)) != len(ids) or None in ids:
            off.append("ranks do not sit on distinct GPUs: pci_bus_id %r" % ids)
        sp = line.get("kernel_ms_spread") or {}
        if sp.get("max_over_min", 1.0) > 1.05:
            slow = max(ranks, key=lambda r: r.get("avg_kernel_ms", 0)) if ranks else {}
            off.append("kernel time differs by %.1f %% between ranks (slowest: rank %s on %s, %.4f ms): value is the max over ranks, look "
                       "at that GPU's clocks / its neighbours' load" % (100 * (sp["max_over_min"] - 1), slow.get("rank"), slow.get("pci_bus_id"),
                                                                        slow.get("avg_kernel_ms", 0)))
        wall = [r.get("timed_region_s", 0) for r in ranks if r]
        if wall and min(wall) > 0 and max(wall) / min(wall) > 1.10:
            off.append("timed regions differ by %.0f %% between ranks: the barrier or a late rank, not the kernel" % (100 * (max(wall) / min(wall) - 1)))
        g = line.get("gather") or {}
        if "error" in g:
            off.append("gather: %s" % g["error"])
        elif g:
            gib = g.get("bytes_per_rank", 1 << 30) / 1e9
            want_ms = gib / XGMI_GBPS * 1e3                     # all peers at once, each on its own link into rank 0
            if g.get("ms", 0) > 2 * want_ms * max(1.0, (n - 1) / 7.0) + 3:
                off.append("RCCL gather %.1f ms for %d peers, predicted ~%.0f ms (every peer on its own xGMI link at ~%d GB/s): the links "
                           "are not used in parallel, or rank 0's HBM / the copy engines limit" % (g["ms"], n - 1, want_ms, XGMI_GBPS))
            for p in g.get("per_peer", []):
                if p.get("ms", 0) > 2 * want_ms + 2:
                    off.append("peer %s -> rank 0: %.1f ms (%.0f GB/s), predicted ~%.0f ms: an indirect or degraded xGMI path" % (
                        p.get("peer"), p["ms"], p.get("GB_per_s", 0), want_ms))
            d = g.get("per_gpu_d2h") or {}
            if "error" in d:
                off.append("per-GPU D2H: %s" % d["error"])
            elif d and d.get("ms", 0) > 2 * (gib / PCIE_GBPS * 1e3) + 3:
                off.append("per-GPU D2H %.1f ms, predicted ~%.0f ms (each GPU its own PCIe link at ~%d GB/s): shared root complexes or "
                           "slabs on the wrong NUMA node" % (d["ms"], gib / PCIE_GBPS * 1e3, PCIE_GBPS))
        pr = (line.get("gather") or {}).get("product_ring")
        if pr is None:
            off.append("gather.product_ring missing: the N-device ring of the shipped command was not run")
        elif "error" in pr:
            off.append("product ring: %s" % pr["error"])
        else:
            per = pr.get("GB_per_s_each_way_per_device", 0)
            want_lo = min(RING_GBPS[0], HOST_DMA_GBPS / 2 / n * 0.8)
            if per < want_lo:
                off.append("product ring %.1f GB/s each way per device (%.0f aggregate), predicted >= %.0f: %s" % (
                    per, pr.get("GB_per_s_each_way_aggregate", 0), want_lo,
                    "its streams still share a hardware queue" if pr.get("streams_share_a_queue") else
                    "the producer thread (%.0f us per slab), shared root complexes, or the host's memory" % pr.get("submit_us_per_slab", 0)))
            if per > RING_GBPS[1] * 1.1:
                off.append("product ring %.1f GB/s per device is above one PCIe link: the devices listed are not distinct" % per)
            nodes = pr.get("slab_numa_nodes") or []
            if nodes and len(set(nodes)) == 1 and n >= 4 and nodes[0] >= 0:
                off.append("every slab of the %d-GPU ring sits on NUMA node %d: half of the copies cross the socket link" % (n, nodes[0]))
    rf = line.get("roofline") or {}
    if rf and not 0.78 <= rf.get("frac", 0) <= 0.88:
        off.append("roofline.frac %.3f of rank 0 outside 0.78-0.88" % rf.get("frac", 0))
    return off


def main(paths):
    status, seen = 0, 0
    for path in paths:
        with open(path) as f:
            text = f.read()
        try:
            obj = json.loads(text)
        except ValueError:
            obj = text
        lines = [ln for ln in bench_lines(obj) if "1 GB i16 stream" in ln.get("metric", "")]
        if not lines:
            print("%s: no bench line of the headline metric (a skipped record: %s)" % (path, text.strip()[:120].replace("\n", " ")))
            continue
        uniq = {}
        for ln in lines:                               # a record may hold the same line parsed and as raw stdout
            uniq.setdefault((ln["n_gpus"], ln["value"], ln.get("ms_per_step")), ln)
        lines = sorted(uniq.values(), key=lambda ln: ln["n_gpus"])
        base = next((float(ln["value"]) for ln in lines if ln["n_gpus"] == 1), None)
        for ln in lines:
            seen += 1
            off = check_line(ln, base if ln["n_gpus"] > 1 else None)
            eff = "" if base is None or ln["n_gpus"] == 1 else ", efficiency vs this record's N=1: %.3f" % (ln["value"] / (base * ln["n_gpus"]))
            print("%s: N=%d value %.0f Msamples/s%s — %s" % (path, ln["n_gpus"], ln["value"], eff, "as predicted" if not off else "%d finding(s)" % len(off)))
            for o in off:
                print("    * " + o)
                status = 1
    return status if seen else 2


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    sys.exit(main(sys.argv[1:]))
