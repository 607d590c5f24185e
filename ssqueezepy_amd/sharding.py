# -*- coding: utf-8 -*-
"""Batch sharding across GPUs (one process per GPU, `torch.distributed`).

The reference has no multi-GPU code. Signals of a batch are independent end to end
(the reference itself processes them with a broadcast plus a Python loop:
ssqueezepy/_cwt.py:270-271, ssqueezing.py:208-214), so the N > 1 path is pure data
parallelism: rank r transforms a contiguous block of signals on its own GPU with its
own plan (the design step is deterministic, so every rank builds identical tables),
and there is no collective on the data path. Outputs stay device-resident and
sharded -- the full `Tx` + `Wx` of BASELINE config 4 (512 x 768 MB = 393 GB) fits on
no single GPU. The one collective at the end is an `all_gather` -- of small per-signal
summaries (`gather_summaries`), or, when a caller does want every rank's `Tx` in one place,
of the transforms themselves in chunks of a few signals (`gather_tx`): RCCL over xGMI with
the 'nccl' backend, 'gloo' in the CPU tests.
"""

__all__ = ['shard_bounds', 'shard_signals', 'gather_summaries', 'signal_summary', 'gather_tx']


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


def gather_tx(Tx, chunk=8, group=None, consume=None):
    """The "single gather at the end" for full results: every rank's (b, na, N) block of `Tx`
    (or `Wx`: any device tensor whose first axis is the rank's signals; every rank must hold the same
    `b` -- pad the last shard) to every rank, `chunk` signals per collective.

    One collective moves ``(world - 1) * chunk * na * N * itemsize`` bytes into each rank over the
    point-to-point xGMI links, so its time is per-link bound and the receive buffer --
    ``world * chunk`` transforms -- stays small next to the rank's own outputs (BASELINE config 4:
    24.6 GB of `Tx` per GPU; all of it gathered would be 197 GB per rank, hence `consume`).

    `consume(c0, block)`, when given, is called after every collective with the first local signal
    index of the chunk and ``block`` = a (world, chunk', na, N) view of the receive buffer (rank r's
    signals c0 .. c0 + chunk' - 1), valid until the next collective; nothing is kept then and the
    function returns None. Without it the gathered array (world * b, na, N), ranks' blocks in rank
    order, is returned -- only for sizes that fit.

    Replaces nothing in the reference (it has no multi-GPU path: its batch loop is
    ssqueezepy/ssqueezing.py:208-214); 'nccl' uses `all_gather_into_tensor`, 'gloo' gathers host
    copies."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    b = int(Tx.shape[0])
    chunk = max(1, min(int(chunk), b)) if b else 1
    tail = tuple(Tx.shape[1:])
    nccl = dist.get_backend(group) == 'nccl'
    dev = Tx.device
    big = torch.empty((world, chunk) + tail, dtype=Tx.dtype, device=dev)
    full = None if consume is not None else torch.empty((world, b) + tail, dtype=Tx.dtype, device=dev)
    for c0 in range(0, b, chunk):
        part = Tx[c0:c0 + chunk]
        n = int(part.shape[0])
        if n < chunk:                                     # the last, short chunk: padded to the collective's shape
            part = torch.cat([part, part.new_zeros((chunk - n,) + tail)])
        if nccl:
            dist.all_gather_into_tensor(big, part.contiguous(), group=group)
        else:
            parts = [torch.empty((chunk,) + tail, dtype=Tx.dtype) for _ in range(world)]
            dist.all_gather(parts, part.cpu().contiguous(), group=group)
            for r in range(world):
                big[r].copy_(parts[r])
        if consume is not None:
            consume(c0, big[:, :n])
        else:
            full[:, c0:c0 + n] = big[:, :n]
    return None if consume is not None else full.reshape((world * b,) + tail)
