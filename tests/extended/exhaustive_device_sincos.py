"""One-off evidence run (GPU box): the device sincos (dpx_ccexpf_imag, both libm builds) against the oracle's
restated glibc sincosf for ALL 2^32 float bit patterns, and against this host's libm cexpf (through the
reference's complex.c when oracle/_ref is built) for the build the host selects.  Prints one JSON line."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import doppler_amd
from doppler_amd import dsp
from oracle import oracle as orc

ctx = doppler_amd.Context(0)
CH = 1 << 26
variant = orc.libm_variant()
mism = {"fma_vs_restated": 0, "sse2_vs_restated": 0, "host_libm": 0}
t0 = time.time()


def same(a, b):
    x, y = a.view(np.uint32).reshape(-1), b.view(np.uint32).reshape(-1)
    bad = x != y
    if bad.any():
        fa, fb = a.view(np.float32).reshape(-1), b.view(np.float32).reshape(-1)
        bad &= ~(np.isnan(fa) & np.isnan(fb))
    return int(bad.sum())


def host(theta, mode, parts=32):
    step = theta.size // parts
    with ThreadPoolExecutor(parts) as ex:       # the C loops release the GIL (ctypes)
        outs = list(ex.map(lambda i: orc.ccexpf_imag_array(theta[i * step:(i + 1) * step], mode), range(parts)))
    return np.concatenate(outs)


for c in range((1 << 32) // CH):
    bits = (np.arange(CH, dtype=np.uint64) + np.uint64(c) * np.uint64(CH)).astype(np.uint32)
    theta = bits.view(np.float32)
    z = np.zeros(CH, dtype=dsp.complex32)
    z["im"] = theta
    ctx.set_libm_contraction(True)
    g1 = np.array(z)
    doppler_amd.engine.check(ctx._lib.dpx_ccexpf_imag(ctx.handle, g1.ctypes.data, g1.size))
    mism["fma_vs_restated"] += same(g1, host(theta, 1))
    ctx.set_libm_contraction(False)
    g0 = np.array(z)
    doppler_amd.engine.check(ctx._lib.dpx_ccexpf_imag(ctx.handle, g0.ctypes.data, g0.size))
    mism["sse2_vs_restated"] += same(g0, host(theta, 2))
    mism["host_libm"] += same(g1 if variant == 1 else g0, host(theta, 0))
ctx.set_libm_contraction(True)
print(json.dumps({"what": "device sincos vs oracle, all 2^32 float bit patterns", "host_libm_variant": variant,
                  "reference_complex_c_linked": orc.have_ref(), "mismatches": mism, "seconds": round(time.time() - t0, 1)}))
