import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["case"], d["pair"], "v%d" % d["variant"], d["kernel"], d["opts"], d["pct_peak"], d["ms_med"], "vs_headline", d.get("vs_headline"))
    elif "rror" in l: print(l.strip())
