"""gpurun_out/r04_pairs_pmc/raw.tsv (tools/prof_pairs.sh) -> a markdown table, one row per case: per-launch counter averages
turned into per-sample / per-cycle figures."""
import collections, sys
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    p = line.rstrip("\n").split("\t")
    if len(p) < 6:
        continue
    label, pas, name, kern, n, v = p
    d = rows.setdefault(label, {"kernel": kern, "dur": []})
    if name == "duration_us":
        d["dur"].append(float(v))
    elif name != "ERROR":
        d[name] = float(v)
    else:
        d.setdefault("errors", []).append(pas + ": " + kern)
N = {"const": 268435456, "track 600": 614400000, "replay 600": 614400000, "replay 300": 307200000}
def samples(label):
    for k, v in N.items():
        if k in label:
            return v
    return 268435456
BPS = {"i16->i16": 8, "i16:i16": 8, "f32:f32": 16, "i16:f32": 12, "f32:i16": 12, "i16->f32": 12}
cols = ["case", "us (unprofiled-ish: min of passes)", "% of 8 TB/s", "VALU instr / sample", "VALU busy % of wave cycles", "wave cycles parked (WAIT_ANY) %", "issue stall (WAIT_INST_ANY) %",
        "EA wr req / KiB out", "of which 64 B %", "EA rd req / KiB in", "of which 32 B %", "EA wr stall cyc / req", "DRAM wr credit stall / req", "DRAM rd credit stall / req",
        "TCP pending stall / wave cyc %", "TCR->TCP stall / wave cyc %", "avg wr req in flight (LEVEL/GUI)", "avg rd req in flight", "LDS conflict % of LDS cycles", "waves"]
print("| " + " | ".join(cols) + " |")
print("|" + "---|" * len(cols))
for label, d in rows.items():
    n = samples(label)
    bps = next((v for k, v in BPS.items() if k in label), 8)
    inb = {8: 4, 16: 8}.get(bps, 4 if "i16:f32" in label or "i16->f32" in label else 8)
    outb = bps - inb
    dur = min(d["dur"]) if d["dur"] else float("nan")
    g = lambda k: d.get(k, float("nan"))
    wc = g("SQ_WAVE_CYCLES")
    row = [label, "%.1f" % dur, "%.1f" % (n * bps / dur / 1e3 / 80), "%.1f" % (g("SQ_INSTS_VALU") * 64 / n if "SQ_INSTS_VALU" in d else float("nan")),
           "%.0f" % (100 * g("SQ_ACTIVE_INST_VALU") / wc), "%.0f" % (100 * g("SQ_WAIT_ANY") / wc), "%.0f" % (100 * g("SQ_WAIT_INST_ANY") / wc),
           "%.2f" % (g("TCC_EA0_WRREQ_sum") / (n * outb / 1024)), "%.0f" % (100 * g("TCC_EA0_WRREQ_64B_sum") / g("TCC_EA0_WRREQ_sum")),
           "%.2f" % (g("TCC_EA0_RDREQ_sum") / (n * inb / 1024)), "%.0f" % (100 * g("TCC_EA0_RDREQ_32B_sum") / g("TCC_EA0_RDREQ_sum")),
           "%.2f" % (g("TCC_EA0_WRREQ_STALL_sum") / g("TCC_EA0_WRREQ_sum")), "%.2f" % (g("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum") / g("TCC_EA0_WRREQ_sum")),
           "%.2f" % (g("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") / g("TCC_EA0_RDREQ_sum")),
           "%.1f" % (100 * g("TCP_PENDING_STALL_CYCLES_sum") / wc), "%.1f" % (100 * g("TCP_TCR_TCP_STALL_CYCLES_sum") / wc),
           "%.0f" % (g("TCC_EA0_WRREQ_LEVEL_sum") / g("GRBM_GUI_ACTIVE")), "%.0f" % (g("TCC_EA0_RDREQ_LEVEL_sum") / g("GRBM_GUI_ACTIVE")),
           "%.1f" % (100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0), "%.0f" % g("SQ_WAVES")]
    print("| " + " | ".join(row) + " |")
    for e in d.get("errors", []):
        print("| %s: %s |" % (label, e))
print()
print("Raw per-launch averages: raw.tsv (counter values as rocprofv3 reports them: SQ_* cycle counters in quad-cycles, summed over the chip).")
