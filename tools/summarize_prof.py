"""Turn the rocprofv3 databases written by tools/profile.sh into the committed summaries under profiles/.

    python tools/summarize_prof.py gpurun_out/prof_r01 r01
writes profiles/<tag>_rocprof_kernel_stats.md and profiles/<tag>_pmc_traffic.json.
"""
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    return [(short(n), c, a / 1e3, mn / 1e3, mx / 1e3, t / 1e3) for n, c, a, mn, mx, t in rows]


def counter_per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {short(n): (c, a, mn, mx) for n, c, a, mn, mx in rows}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_md = os.path.join(root, "profiles", "%s_rocprof_kernel_stats.md" % tag)
    lines = ["# rocprofv3 --kernel-trace --stats, `python bench.py --steps 20 --warmup 5 --no-cpu` (1x MI355X)", "",
             "Source: %s/trace/bench_results.db (rocpd SQLite written by rocprofv3; summarised by tools/summarize_prof.py)." % src, "",
             "| kernel | calls | avg us | min us | max us | total us |", "|---|---|---|---|---|---|"]
    stats = kernel_stats(os.path.join(src, "trace", "bench_results.db"))
    for n, c, a, mn, mx, t in stats:
        lines.append("| `%s` | %d | %.3f | %.3f | %.3f | %.1f |" % (n, c, a, mn, mx, t))
    main_k = [s for s in stats if "rows_kernel" in s[0]]
    result = {}
    if main_k:
        n, c, a, mn, mx, t = main_k[0]
        alg = 268435456 * 8
        lines += ["", "Dominant kernel `%s`: average %.3f us over %d launches -> %.1f GB/s algorithmic "
                  "(2 147 483 648 B per launch) = %.1f %% of the 8.0 TB/s HBM3E peak." % (n, a, c, alg / a / 1e3, alg / a / 1e3 / 80.0)]
        result["avg_launch_us_kernel_trace"] = round(a, 3)
    # PMC passes
    f = counter_per_kernel(os.path.join(src, "pmc_fetch", "bench_results.db"), "FETCH_SIZE")
    w = counter_per_kernel(os.path.join(src, "pmc_write", "bench_results.db"), "WRITE_SIZE")
    lines += ["", "## HBM traffic counters (separate --pmc passes: FETCH_SIZE, then WRITE_SIZE)", "",
              "| kernel | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch (raw) |", "|---|---|---|"]
    for k in sorted(set(f) | set(w)):
        lines.append("| `%s` | %s | %s |" % (k, "%.1f" % f[k][1] if k in f else "-", "%.1f" % w[k][1] if k in w else "-"))
    cal_f = cal_w = None
    cf = os.path.join(src, "cal_fetch", "cal_results.db")
    cw = os.path.join(src, "cal_write", "cal_results.db")
    if os.path.exists(cf) and os.path.exists(cw):
        cal_f = counter_per_kernel(cf, "FETCH_SIZE").get("dpx::copy_kernel")
        cal_w = counter_per_kernel(cw, "WRITE_SIZE").get("dpx::copy_kernel")
    rk = [k for k in f if "rows_kernel" in k]
    if rk:
        k = rk[0]
        fetch_raw = f[k][1] * 1024
        write_raw = w[k][1] * 1024 if k in w else None
        # gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a
        # wide coalesced 16 B/lane streaming read; WRITE_SIZE is calibrated on a copy of known size.
        fetch_scale = 2.0
        write_scale = 1.0
        note = "FETCH_SIZE x2 (guide: gfx950 reports 1/2 for 16 B/lane streams)"
        if cal_f and cal_w:
            fetch_scale = (1 << 30) / (cal_f[1] * 1024)
            write_scale = (1 << 30) / (cal_w[1] * 1024)
            note = ("scales calibrated in the same session on dpx::copy_kernel (1 GiB read + 1 GiB written): "
                    "FETCH_SIZE x%.4f, WRITE_SIZE x%.4f" % (fetch_scale, write_scale))
        fetch_b = fetch_raw * fetch_scale
        write_b = write_raw * write_scale if write_raw is not None else None
        result.update({
            "kernel": k, "fetch_size_raw_bytes": fetch_raw, "write_size_raw_bytes": write_raw,
            "fetch_scale": fetch_scale, "write_scale": write_scale, "correction": note,
            "hbm_read_bytes_per_launch": round(fetch_b), "hbm_write_bytes_per_launch": round(write_b) if write_b else None,
            "hbm_bytes_per_launch": round(fetch_b + (write_b or 0)),
            "algorithmic_bytes_per_launch": 268435456 * 8,
        })
        lines += ["", "Corrected per launch of `%s`: read %.1f MiB + written %.1f MiB = %.1f MiB against %.1f MiB algorithmic "
                  "(%s)." % (k, fetch_b / 2**20, (write_b or 0) / 2**20, (fetch_b + (write_b or 0)) / 2**20, 2048.0, note)]
    with open(out_md, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "profiles", "%s_pmc_traffic.json" % tag), "w") as fh:
        json.dump(result, fh, indent=1)
    print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()
