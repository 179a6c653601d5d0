"""Print the walk/tile kernel rows of the rocprofv3 databases written by tools/profile_track.sh <tag>."""
import sqlite3, sys, glob
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
for sub in ("trace","pmc_fetch","pmc_write","pmc_l2"):
    for db in glob.glob("gpurun_out/prof_track_%s/%s/*.db" % (tag, sub)):
        c = sqlite3.connect(db)
        if sub == "trace":
            for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels group by name order by sum(duration) desc limit 6"): print(sub, r[0][:60], r[1:])
        else:
            for r in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
                if "walk" in r[0] or "tile" in r[0]: print(sub, r[0][:50], r[1:])
