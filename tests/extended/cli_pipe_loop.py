"""Loops the `doppler track` invocation of tests/test_gpu_cli.py exactly as the test runs it (subprocess pipes: pipe
timing cuts the stream into different slabs every run).  `python tests/extended/cli_pipe_loop.py 100 [gdb]`.
Found the out-of-bounds hint scan fixed in round 1 (7 crashes in 160 runs before, 0 in 200 after)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_iq
EXE = "doppler_amd/bin/doppler"
rate = 256000
rr = 6.8 * np.tanh((np.arange(16) - 6.0) / 2.0)
n = rate * 9 + 2048 * 2 + 55
x = make_iq("i16", n, 5)
open("/tmp/rr.txt", "w").write("\n".join("%.17g" % v for v in rr))
args = ["track", "-s", str(rate), "-i", "i16", "--range-rate-file", "/tmp/rr.txt", "--frequency", "437505000", "--offset", "-2500", "--time", "2015-01-22T09:07:16"]
use_gdb = len(sys.argv) > 2 and sys.argv[2] == "gdb"
fails = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    for slab in ("33554432", "262144"):
        e = dict(os.environ, DOPPLER_SLAB_BYTES=slab)
        cmd = [EXE] + args
        if use_gdb:
            cmd = ["rocgdb", "-q", "-batch", "-ex", "run", "-ex", "bt", "-ex", "info locals", "-ex", "frame 0", "-ex", "list", "--args"] + cmd
        r = subprocess.run(cmd, input=bytes(x), capture_output=True, timeout=300, env=e)
        if use_gdb:
            if b"SIGSEGV" in r.stdout or b"SIGSEGV" in r.stderr:
                txt = (r.stdout + r.stderr).decode(errors="replace")
                k = txt.find("SIGSEGV")
                print("CRASH run", i, "slab", slab); print(txt[max(0, k - 300):k + 3000]); sys.exit(0)
        elif r.returncode != 0:
            fails += 1
            print("run", i, "slab", slab, "rc", r.returncode, "stdout bytes", len(r.stdout), "of", x.size, "stderr tail:", r.stderr[-300:], flush=True)
print("failures", fails)
