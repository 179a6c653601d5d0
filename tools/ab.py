"""A/B timing of workloads and kernel-shape options on one MI355X, in ONE process (development tool, not the bench contract).

    python tools/ab.py [--rounds R] [--iters K] "label|CASE key=value ..." ["label|..." ...]

Every case is a tools/prof_case.py workload (same CASE / key=value grammar).  All plans are built first, then the cases are
timed round-robin (R rounds x K back-to-back launches each, one HIP event pair per burst), so that clock and thermal drift
hit every case alike; cases of the same size and formats share their buffers.  Reported: the median burst as algorithmic
GB/s and % of the 8 TB/s HBM peak.  Boxes differ by +-2 points on one binary: only differences inside one run mean anything.
(Rounds 1-4 kept ~45 named case sets in this file; the ones profiles/ cites are spelled out in the logs under profiles/raw.)"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import doppler_amd  # noqa: E402
import prof_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("cases", nargs="+")
    a = ap.parse_args()
    ctx = doppler_amd.Context(0)
    st = torch.cuda.current_stream()
    buffers, built = {}, []
    for spec in a.cases:
        label, _, rest = spec.partition("|")
        c = prof_case.make_case(ctx, rest.split(), buffers)
        c["label"], c["ms"] = label, []
        built.append(c)
    for c in built:
        for _ in range(5):
            prof_case.launch(c, st)
    st.synchronize()
    for _ in range(a.rounds):
        for c in built:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(a.iters):
                prof_case.launch(c, st)
            e1.record(st)
            st.synchronize()
            c["ms"].append(e0.elapsed_time(e1) / a.iters)
    w = max(len(c["label"]) for c in built)
    for c in built:
        med, best = statistics.median(c["ms"]), min(c["ms"])
        print("%-*s  %9.1f us  %7.1f GB/s  %5.1f %%  (best burst %5.1f %%)" % (w, c["label"], med * 1e3, c["bytes"] / med / 1e6,
                                                                              c["bytes"] / med / 1e6 / 80, c["bytes"] / best / 1e6 / 80))


if __name__ == "__main__":
    main()
