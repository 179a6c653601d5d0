"""doppler_amd — MI355X-native hot path of cubehub/doppler behind its own `doppler::dsp` interface.

    doppler_amd.dsp        convert_iqi16_to_complex / convert_iqf32_to_complex / shift_frequency
                           (reference src/dsp.rs:85,101,117) + the fused shift_block
    doppler_amd.Context    one GPU; doppler_amd.Plan: device-resident bulk runs

Everything computes in hand-written HIP kernels (doppler_amd/csrc) reached through the C ABI
of include/doppler_hip.h.  There is no CPU implementation in this package.
"""
# PyTorch (when present) ships its own libamdhip64.so.7; import it first so that this
# process ends up with ONE HIP runtime shared by torch tensors/streams and our library.
try:  # pragma: no cover - plumbing only
    import torch as _torch  # noqa: F401
except ImportError:  # the C ABI works without torch
    _torch = None

from . import _lib
from ._lib import FMT_F32, FMT_I16, BUFFER_SIZE  # noqa: F401

lib = _lib.load()

from .engine import Context, Plan, Stream, DspError, default_context, plan_describe, plan_simulate, plan_layout  # noqa: E402,F401
from . import dsp  # noqa: E402,F401
