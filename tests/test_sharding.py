# -*- coding: utf-8 -*-
"""The N > 1 path (ssqueezepy_amd/sharding.py) with world_size 2 on CPU ('gloo'):
partition of the batch, independence of the shards, and the one collective. The
transform itself is stood in for by the CPU oracle -- what is under test is the
sharding logic the GPU job uses unchanged with the 'nccl' (RCCL) backend. CPU-only."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ssqueezepy_amd.sharding import shard_bounds, shard_signals


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 8, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conftest import two_chirps
    from pipeline import oracle_ssq_cwt
    from oracle import oracle as orc
    from ssqueezepy_amd.sharding import gather_summaries, signal_summary
    B, N = 5, 256
    x = np.stack([two_chirps(N, seed=s) for s in range(B)])
    mine = shard_signals(x, world, rank)
    Tx, Wx = [], []
    for xi in mine:
        r = oracle_ssq_cwt(orc, xi, 'float32', scales='log', nv=8)
        Tx.append(r['Tx']); Wx.append(r['Wx'])
    na = oracle_ssq_cwt(orc, x[0], 'float32', scales='log', nv=8)['Tx'].shape[0]
    Tx = torch.as_tensor(np.stack(Tx)) if Tx else torch.zeros((0, na, N), dtype=torch.complex64)
    Wx = torch.as_tensor(np.stack(Wx)) if Wx else torch.zeros((0, na, N), dtype=torch.complex64)
    table = gather_summaries(signal_summary(Tx, Wx), B)
    q.put((rank, table.numpy(), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch_matches_single_process():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert [g[2] for g in got] == [3, 2]                      # 5 signals over 2 ranks
    assert np.array_equal(got[0][1], got[1][1])               # every rank has the table
    # single-process reference
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    from conftest import two_chirps
    from pipeline import oracle_ssq_cwt
    from oracle import oracle as orc
    ref = []
    for s in range(5):
        r = oracle_ssq_cwt(orc, two_chirps(256, seed=s), 'float32', scales='log', nv=8)
        ref.append([np.abs(r['Tx']).sum(dtype=np.float64), np.abs(r['Wx']).sum(dtype=np.float64)])
    assert np.allclose(got[0][1], np.array(ref), rtol=1e-6)


def _worker_emulated(rank, world, port, q):
    """As `_worker`, but every rank runs the product itself -- `ssq_cwt` on its shard of the
    batch through the kernels under the CPU emulator (tests/emu_backend.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['SSQ_EMU_THREADS'] = '2'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import emu_backend
    from conftest import two_chirps
    from ssqueezepy_amd.sharding import gather_summaries, signal_summary
    B, N = 5, 256
    x = np.stack([two_chirps(N, seed=s) for s in range(B)]).astype(np.float32)
    mine = shard_signals(x, world, rank)
    with emu_backend.emulated() as S:
        Tx, Wx, *_ = S.ssq_cwt(np.ascontiguousarray(mine), S.Wavelet(), scales='log', nv=8)
        table = gather_summaries(signal_summary(Tx, Wx), B)
    q.put((rank, table.numpy(), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch_through_the_emulated_kernels():
    import emu_backend
    if not emu_backend.available():
        pytest.skip("no clang++ under $ROCM_PATH/lib/llvm/bin")
    emu_backend.build()
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_emulated, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert [g[2] for g in got] == [3, 2]
    assert np.array_equal(got[0][1], got[1][1])
    from conftest import two_chirps
    from pipeline import oracle_ssq_cwt
    from oracle import oracle as orc
    ref = []
    for s in range(5):
        r = oracle_ssq_cwt(orc, two_chirps(256, seed=s), 'float32', scales='log', nv=8)
        ref.append([np.abs(r['Tx']).sum(dtype=np.float64), np.abs(r['Wx']).sum(dtype=np.float64)])
    assert np.allclose(got[0][1], np.array(ref), rtol=1e-5)


def _worker_gather_tx(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch
    from ssqueezepy_amd.sharding import gather_tx
    b, na, N = 5, 7, 33
    g = torch.Generator().manual_seed(100 + rank)
    Tx = torch.complex(torch.randn(b, na, N, generator=g), torch.randn(b, na, N, generator=g))
    full = gather_tx(Tx, chunk=2)                            # chunks of 2, 2, 1 signals: the last one short
    blocks = []
    none = gather_tx(Tx, chunk=3, consume=lambda c0, blk: blocks.append((c0, blk.clone())))
    q.put((rank, full.numpy(), none is None, [(c0, blk.numpy()) for c0, blk in blocks]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_tx_two_ranks_chunked():
    """`sharding.gather_tx` (the full-`Tx` gather at the end of a sharded job, round 6: an API of the package instead
    of a loop inside bench.py): two gloo ranks, 5 signals each, chunks that do not divide the count -- the gathered
    array is every rank's block in rank order on every rank, and the `consume` form hands out the same data chunk by
    chunk without keeping it."""
    import torch
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gather_tx, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, na, N = 5, 7, 33
    want = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        want.append(torch.complex(torch.randn(b, na, N, generator=g), torch.randn(b, na, N, generator=g)).numpy())
    want = np.concatenate(want)
    for rank, full, none, blocks in got:
        assert none and full.shape == (world * b, na, N) and np.array_equal(full, want)
        assert [c0 for c0, _ in blocks] == [0, 3] and [blk.shape[1] for _, blk in blocks] == [3, 2]
        for c0, blk in blocks:
            for r in range(world):
                assert np.array_equal(blk[r], want[r * b + c0: r * b + c0 + blk.shape[1]])
