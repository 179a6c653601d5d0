"""Time-chunk sharding of one IQ stream across the GPUs of a node.

The reference is a single sequential loop whose only carried state is the sample counter
(reference src/main.rs:60, src/dsp.rs:125-130).  Because that counter has a closed form
(doppler_amd.engine.samplenum_after / the planner in csrc/dpx_planner.cpp), rank r can start
in the middle of the stream: it takes a contiguous, block-aligned chunk and seeds its counter
with the value the sequential loop would have reached there.  No data moves between ranks while
the chunks are processed; the only exchange is the ordered gather of the outputs to the rank
that owns stdout, done with point-to-point sends (RCCL over xGMI on GPUs, gloo in CPU tests).
"""
from . import engine

BLOCK_BYTES = 8192  # reference src/main.rs:49: shift may only change at these boundaries


def chunk_bounds(n_samples, world, rank, bytes_per_sample=4):
    """[lo, hi) in samples of rank's chunk: equal numbers of whole 8192-byte blocks per rank,
    the last rank also takes the ragged tail."""
    spb = BLOCK_BYTES // bytes_per_sample
    blocks = n_samples // spb
    per = (blocks + world - 1) // world
    lo = min(rank * per * spb, n_samples)
    hi = n_samples if rank == world - 1 else min((rank + 1) * per * spb, n_samples)
    return lo, max(lo, hi)


def chunk_seed(shift_hz, samplerate, lo, samplenum0=0):
    """Counter value at global sample `lo` of a constant-shift stream that starts at samplenum0."""
    return engine.samplenum_after(shift_hz, samplerate, samplenum0, lo)


def segments_for_chunk(segments, lo, hi):
    """Restrict a list of (n_samples, shift_hz) segments to the sample range [lo, hi).
    Returns (segments_before, segments_inside): the first list is what the counter has to be
    advanced over to obtain the chunk's seed."""
    before, inside = [], []
    pos = 0
    for n, hz in segments:
        a, b = pos, pos + n
        if b <= lo:
            before.append((n, hz))
        elif a >= hi:
            pass
        else:
            if a < lo:
                before.append((lo - a, hz))
            inside.append((min(b, hi) - max(a, lo), hz))
        pos = b
    return before, inside


def seed_for_segments(segments_before, samplerate, samplenum0=0):
    """Counter after a piecewise-constant prefix (track mode), via the closed form per segment."""
    return engine.samplenum_after_segments(segments_before, samplerate, samplenum0)


def ordered_gather(local, sizes, dst=0, group=None):
    """Gather every rank's output chunk, in rank order, into one tensor on `dst`.

    `local`: this rank's 1-D output tensor; `sizes`: element counts of all ranks' chunks.
    Point-to-point (dist.send / dist.irecv) rather than all_gather: only `dst` needs the data, and on
    xGMI each peer then uses exactly its own link into `dst`.  Returns the full tensor on `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    assert len(sizes) == world and local.numel() == sizes[rank]
    # One batch (ncclGroupStart/End under RCCL): the receives from all peers are in flight together, each on its own
    # xGMI link into `dst`, instead of one after the other.
    if rank != dst:
        if local.numel():
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
                q.wait()
        return None
    full = torch.empty(sum(sizes), dtype=local.dtype, device=local.device)
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + s)
    full[offs[rank]:offs[rank + 1]].copy_(local)
    ops = [dist.P2POp(dist.irecv, full[offs[r]:offs[r + 1]], r, group) for r in range(world) if r != dst and sizes[r]]
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return full
