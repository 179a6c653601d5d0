"""Turn the rocprofv3 databases written by tools/profile_round.sh into the summaries committed under profiles/.

    python tools/summarize_round.py /tmp/prof_r03 gpurun_out r03
writes <out>/<round>_rocprof_kernel_stats.md and <out>/<round>_pmc_traffic.json (with the hash of the kernel sources,
which bench.py checks before quoting any of it).
"""
import glob
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return f[0] if f else None


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    return [(short(n), c, a / 1e3, mn / 1e3, mx / 1e3, t / 1e3) for n, c, a, mn, mx, t in rows]


def settled(db_path, pattern, after_ms=160.0):
    """(launches, avg us, median us) of the kernel's launches that START more than after_ms after its first one"""
    db = sqlite3.connect(db_path)
    rows = db.execute("select start, duration from kernels where name like ? order by start", ("%" + pattern + "%",)).fetchall()
    if not rows:
        return None
    t0 = rows[0][0]
    d = sorted(r[1] / 1e3 for r in rows if (r[0] - t0) / 1e6 > after_ms)
    if len(d) < 5:
        return None
    return len(d), sum(d) / len(d), d[len(d) // 2]


def head_and_rest(db_path, pattern, n_head):
    """the kernel's launches in start order: (count, avg us) of the first n_head, and (count, avg us, median us) of the rest"""
    db = sqlite3.connect(db_path)
    d = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like ? order by start", ("%" + pattern + "%",)).fetchall()]
    head, rest = d[:n_head], sorted(d[n_head:])
    if not head:
        return None
    return (len(head), sum(head) / len(head)), ((len(rest), sum(rest) / len(rest), rest[len(rest) // 2]) if len(rest) >= 5 else None)


def counter(db_path, name, agg="avg"):
    """per kernel: (launches, avg value) — or the largest value (calibration: the context's 16-byte warm-up launch of the
    copy kernel must not be averaged with the 1 GiB copies)"""
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, count(*), %s(value) from counters_collection where counter_name = ? group by kernel_name" % agg,
                      (name,)).fetchall()
    return {short(n): (c, a) for n, c, a in rows}


def main():
    src, out = sys.argv[1], sys.argv[2]
    rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
    import bench
    lines = ["# Round %s — rocprofv3 --kernel-trace --stats and HBM-traffic counters (1x MI355X, `tools/profile_round.sh %s`)" % (rnd[1:].lstrip("0"), rnd), "",
             "Commands: `python bench.py --workload const --steps 20 --warmup 5 --no-cpu --no-extra` and `--workload track | track_256k | "
             "config4_chunk --steps 300` (kernel trace; the two PMC passes of a workload: 20 steps); kernel sources sha `%s`.  A replay "
             "workload launches W warm-up + K cold + ~150 ms of settling + K timed launches, all of them in the averages below." % bench.kernel_source_sha(), ""]
    result = {"kernel_source_sha": bench.kernel_source_sha()}
    # calibration of the counters on a copy of known size
    cal_f = counter(db_of(os.path.join(src, "cal_fetch")), "FETCH_SIZE", "max")
    cal_w = counter(db_of(os.path.join(src, "cal_write")), "WRITE_SIZE", "max")
    ck = [k for k in cal_f if "copy_kernel" in k]
    fscale = wscale = None
    if ck:
        fscale = (1 << 30) / (cal_f[ck[0]][1] * 1024.0)
        wscale = (1 << 30) / (cal_w[ck[0]][1] * 1024.0)
        lines += ["Calibration on `dpx::copy_kernel` (1 GiB read + 1 GiB written): FETCH_SIZE x %.4f, WRITE_SIZE x %.4f "
                  "(the guide's gfx950 note: FETCH_SIZE reports half the bytes of a wide streaming read)." % (fscale, wscale), ""]
    for wl, pat, alg in (("const", "rows_kernel", 268435456 * 8), ("track", "span_kernel", 614400000 * 8),
                         ("track_256k", "span_kernel", 153600000 * 8), ("config4_chunk", "tile_kernel", 460800000 * 12)):
        if not db_of(os.path.join(src, wl + "_trace")):
            continue
        stats = kernel_stats(db_of(os.path.join(src, wl + "_trace")))
        lines += ["## %s workload" % wl, "", "| kernel | calls | avg us | min us | max us | total us |", "|---|---|---|---|---|---|"]
        for n, c, a, mn, mx, t in stats[:6]:
            lines.append("| `%s` | %d | %.3f | %.3f | %.3f | %.1f |" % (n, c, a, mn, mx, t))
        main_k = [s for s in stats if pat in s[0]]
        r = {"algorithmic_bytes_per_launch": alg}
        # a span launch over a long stream is dealt out as sub-launches of ~2^28 samples (csrc/dpx_types.h, sub_launch_pieces):
        # "per launch" below is per RUN OF THE PLAN — the kernel's dispatches are added up in groups of `pieces`
        n_samples = alg // (8 if wl != "config4_chunk" else 12)
        pieces = max(1, (n_samples + (1 << 27)) >> 28) if pat == "span_kernel" else 1
        r["sub_launches"] = pieces
        if main_k:
            n, c, a, mn, mx, t = main_k[0]
            c, a, mn, mx = c // pieces, a * pieces, mn * pieces, mx * pieces
            r["kernel"] = n
            r["avg_launch_us_kernel_trace"] = round(a, 3)
            r["launches"] = c
            lines += ["", "Dominant kernel `%s`: average %.3f us over %d launches (warm-up included) -> %.1f GB/s algorithmic "
                      "(%d B per launch) = %.1f %% of the 8.0 TB/s HBM3E peak." % (n, a, c, alg / a / 1e3, alg, alg / a / 1e3 / 80.0)]
            if wl == "const":
                # the driver's command: 1 one-shot + 5 warm-up + 20 timed launches, then the sustained leg (bench.py, SUSTAIN_S)
                hr = head_and_rest(db_of(os.path.join(src, wl + "_trace")), pat, 26)
                if hr and hr[1]:
                    (hn, ha), (rn, ra, rm) = hr
                    r["avg_launch_us_kernel_trace"] = round(ha, 3)
                    r["launches"] = hn
                    r["sustained_launches"], r["sustained_avg_us"], r["sustained_median_us"] = rn, round(ra, 3), round(rm, 3)
                    lines += ["Of these, the first %d (one-shot, warm-up, the K timed launches: what `roofline.frac` times): average %.3f us -> %.1f %%; "
                              "the %d launches of the sustained leg (`roofline.frac_sustained`): average %.3f us, median %.3f us -> %.1f %% / %.1f %%." %
                              (hn, ha, alg / ha / 1e3 / 80.0, rn, ra, rm, alg / ra / 1e3 / 80.0, alg / rm / 1e3 / 80.0)]
            st = settled(db_of(os.path.join(src, wl + "_trace")), pat) if wl != "const" else None
            if st:
                st = (st[0] // pieces, st[1] * pieces, st[2] * pieces)
                # avg_launch_us_kernel_trace stays the average over ALL launches (bench.py: frac_rocprof); the settled subset
                # is quoted beside it (frac_rocprof_settled), never instead of it
                r["settled_launches"], r["settled_avg_us"], r["settled_median_us"] = st[0], round(st[1], 3), round(st[2], 3)
                lines += ["Launches that start more than 160 ms after the first one (%d of them: clocks settled, as in the bench line's "
                          "`extra.track`): average %.3f us, median %.3f us -> %.1f %% / %.1f %%." %
                          (st[0], st[1], st[2], alg / st[1] / 1e3 / 80.0, alg / st[2] / 1e3 / 80.0)]
        have_pmc = db_of(os.path.join(src, wl + "_fetch")) and db_of(os.path.join(src, wl + "_write"))
        f = counter(db_of(os.path.join(src, wl + "_fetch")), "FETCH_SIZE") if have_pmc else {}
        w = counter(db_of(os.path.join(src, wl + "_write")), "WRITE_SIZE") if have_pmc else {}
        fk = [k for k in f if pat in k]
        if fk and fscale:
            rd = f[fk[0]][1] * 1024.0 * fscale * pieces
            wr = w[fk[0]][1] * 1024.0 * wscale * pieces
            r.update(fetch_size_raw_kib=round(f[fk[0]][1], 1), write_size_raw_kib=round(w[fk[0]][1], 1), fetch_scale=fscale, write_scale=wscale,
                     hbm_read_bytes_per_launch=int(rd), hbm_write_bytes_per_launch=int(wr), hbm_bytes_per_launch=int(rd + wr))
            lines += ["", "HBM traffic per launch (separate --pmc passes, calibrated): %.1f MiB read + %.1f MiB written = %.1f MiB "
                      "against %.1f MiB algorithmic: ratio %.4f (reads alone against the input bytes: %.4f)." %
                      (rd / 2**20, wr / 2**20, (rd + wr) / 2**20, alg / 2**20, (rd + wr) / alg, rd / (alg * (8 / 12 if wl == "config4_chunk" else 0.5))), ""]
        result[wl] = r
    with open(os.path.join(out, "%s_rocprof_kernel_stats.md" % rnd), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(out, "%s_pmc_traffic.json" % rnd), "w") as fh:
        json.dump(result, fh, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
