# -*- coding: utf-8 -*-
"""Batch sharding across GPUs (one process per GPU, `torch.distributed`).

The reference has no multi-GPU code. Signals of a batch are independent end to end
(the reference itself processes them with a broadcast plus a Python loop:
ssqueezepy/_cwt.py:270-271, ssqueezing.py:208-214), so the N > 1 path is pure data
parallelism: rank r transforms a contiguous block of signals on its own GPU with its
own plan (the design step is deterministic, so every rank builds identical tables),
and there is no collective on the data path. Outputs stay device-resident and
sharded -- the full `Tx` + `Wx` of BASELINE config 4 (512 x 768 MB = 393 GB) fits on
no single GPU. The one collective is an `all_gather` of small per-signal summaries
(RCCL over xGMI with the 'nccl' backend; 'gloo' in the CPU tests).
"""

__all__ = ['shard_bounds', 'shard_signals', 'gather_summaries', 'signal_summary']


def shard_bounds(n_signals, world_size, rank):
    """Contiguous block [lo, hi) of signal indices owned by `rank`; the first
    ``n_signals % world_size`` ranks get one extra signal."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, extra = divmod(int(n_signals), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_signals(x, world_size, rank):
    """Rows of the (B, N) batch owned by `rank` (a view; may be empty)."""
    lo, hi = shard_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def signal_summary(Tx, Wx, cols=4096):
    """Per-signal checksums (2 floats each): sum |Tx|, sum |Wx| over the first `cols`
    time samples (None = all) -- cheap, order-insensitive fingerprints of the
    device-resident results (a full pass would move more bytes than a transform)."""
    import torch
    if Tx.ndim == 2:
        Tx, Wx = Tx[None], Wx[None]
    if cols is not None:
        Tx, Wx = Tx[..., :cols], Wx[..., :cols]
    return torch.stack([Tx.abs().sum(dim=(1, 2)).double(),
                        Wx.abs().sum(dim=(1, 2)).double()], dim=1)


def gather_summaries(local, n_signals, group=None):
    """all_gather the (b_local, k) summaries of every rank into the (n_signals, k)
    table, in signal order, on every rank. Ranks may own different counts."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    k = local.shape[1] if local.ndim == 2 else 1
    cap = -(-int(n_signals) // world)                     # max block size
    buf = torch.zeros((cap, k), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local.reshape(-1, k)
    dev = buf.device
    if buf.is_cuda and dist.get_backend(group) == 'gloo':
        buf = buf.cpu()                                   # gloo gathers host tensors
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    out = [o.to(dev) for o in out]
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_signals, world, r)
        rows.append(out[r][:hi - lo])
    return torch.cat(rows, dim=0)
