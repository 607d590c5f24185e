# -*- coding: utf-8 -*-
"""bench.py -- ssq_cwt throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[1]: ssq_cwt('gmw'), N=160 000, 300 scales
(`process_scales('log', N, wavelet, nv=32)[:300]`), float32,
synthetic two-chirp + noise signals, already resident in HBM when the clock
starts. One step = one batched `ssq_cwt` call over `--batch` independent signals
per GPU (weak scaling: every rank transforms its own signals; there is no data-path
collective -- signals are independent end to end -- and one tiny RCCL all_gather of
per-signal checksums closes the timed region when N > 1).

Prints ONE JSON line on rank 0: transforms/s (whole job), ms per step, plus
  roofline     algorithmic HBM bytes (x in + Tx, Wx out = N*4 + 2*na*N*8 per
               transform) / measured time, against the 8 TB/s
               HBM3E peak. Measured with HIP events on the launch stream over the
               timed region, for the whole transform (all of its kernels); `traffic` is
               the PMC figure of the same command (profiles/pmc_traffic.json).
  dominant_kernel
               the column-tile kernel (interpolation + reassignment, ~2/3 of the time; round 4:
               ssq::tile2_kernel -- float64 tile, unordered ds_add_f64; SSQ_TILE_ORDER=ordered in the
               environment selects the ticketed ssq::tile_kernel): its own bytes / its own time
               from the plan's HIP-event stage timing; rocprofv3 --kernel-trace --stats of this
               command: profiles/r4*_kernel_stats.txt.
  per_gpu      transforms/s, signals and output bytes per step and GPU (--batch 64 is
               BASELINE config 4's per-GPU shape: 49 GB of Tx + Wx per step).
`--scales log-piecewise` runs the reference's default scales (float64 per-row weights) instead.
  cpu_baseline the CPU oracle pipeline (scipy.fft on 64 threads + the OpenMP C
               restatement of the reference's loop nests, oracle/) on a bounded sample of
               the same workload, same box, core count stated (kind "port").
`--gpus N` without a launcher starts the N ranks itself (torch.distributed.run).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)


def two_chirps(N, seed, noise=0.1):
    rng = np.random.default_rng(seed)
    f0 = rng.uniform(0.01, 0.05)
    f1 = rng.uniform(0.30, 0.45)
    t = np.arange(N) / N
    ph = f0 * N * t + 0.5 * (f1 - f0) * N * t**2
    return (np.cos(2 * np.pi * ph) + np.cos(2 * np.pi * (ph + 0.04 * N * t))
            + noise * rng.standard_normal(N))


def cpu_baseline(N, na, seconds_budget=25.0):
    """CPU oracle on the host cores: the same transform, one signal at a time."""
    from oracle import oracle as orc
    from pipeline import oracle_ssq_cwt
    from ssqueezepy_amd.wavelets import Wavelet
    from ssqueezepy_amd.scales import process_scales
    import scipy.fft as sfft
    cores = os.cpu_count() or 1
    # pocketfft's thread pool stops scaling long before 256 threads on a (300, 2^18) batch
    # (round 1 ran it with workers = all cores: 0.83 s per transform); 64 is the knee
    fft_workers = max(1, min(cores, 64))
    orc.lib()
    wav = Wavelet()
    scales = process_scales('log', N, wav, nv=32)[:na]
    x = two_chirps(N, 0)

    # the design step (bank, grids) is cached by the reference too
    # (examples/benchmarks.py:30-37) -> build once, time only the transform
    from ssqueezepy_amd.padding import pad_geometry
    from ssqueezepy_amd.ssqueezing import (_compute_associated_frequencies,
                                           ssq_grid_params, ssq_const)
    sc = np.asarray(scales, dtype='float32')
    M, n1, _ = pad_geometry(N)
    Psih = wav(scale=sc, N=M, nohalf=False)
    xi = wav.xifn(1., M).reshape(-1)
    sc_ssq, st2, _, nv2 = process_scales(sc.squeeze(), N, get_params=True)
    ssq_freqs = _compute_associated_frequencies(sc_ssq, N, wav, st2, 'peak', True,
                                                1., 'cwt')
    const = ssq_const('cwt', st2, nv2, sc_ssq, ssq_freqs)
    grid, p = ssq_grid_params(ssq_freqs, True)
    gamma = 10 * np.finfo(np.float32).eps

    def one():
        Wx, dWx = orc.cwt(x, Psih, xi, 1., n1, N, derivative=True, workers=fft_workers)
        return orc.ssqueeze(Wx, dWx, 'log', p, const, gamma, True, parallel=True)

    one()
    t0 = time.perf_counter()
    runs = 0
    while runs < 10 and time.perf_counter() - t0 < seconds_budget:
        one()
        runs += 1
    dt = (time.perf_counter() - t0) / runs
    return {"value": 1.0 / dt, "unit": "transforms/s", "cores": cores,
            "kind": "port",
            "sample": "%d ssq_cwt transforms of the same workload (N=%d, %d scales, "
                      "float32; scipy.fft workers=%d + OpenMP loop nests on %d cores; wavelet "
                      "bank cached as in examples/benchmarks.py). kind 'port': the oracle's "
                      "restatement of the reference's CPU algorithm -- ssqueezepy itself "
                      "(numba, SSQ_PARALLEL=1) is not installable on the GPU box. Calibration "
                      "against the reference itself (build container, 8 cores, same input: "
                      "tools/r5/cpu_calibrate.py, profiles/r5_cpu_calibration.json): ssqueezepy's "
                      "own ssq_cwt with its loop nests bound to the OpenMP restatement 0.63 "
                      "transforms/s, this port 0.48 -- the port runs at 0.76 of the reference"
                      % (runs, N, na, fft_workers, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='signals per GPU per step')
    ap.add_argument('--n', '--length', dest='n', type=int, default=160000)
    ap.add_argument('--na', type=int, default=300)
    ap.add_argument('--scales', default='log', choices=['log', 'log-piecewise'],
                    help="'log' = BASELINE config 2 (the first --na of nv=32 log scales); "
                         "'log-piecewise' = the reference's default scales (ssq_cwt(x) without "
                         "arguments; float64 per-row reassignment weights; --na is ignored)")
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--gather-tx', action='store_true',
                    help='N > 1: also time one step followed by an all_gather of the full Tx '
                         '(reported separately; `value` never includes it)')
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU,
    # torch.distributed.run, rendezvous on 127.0.0.1); under a launcher the ranks are already
    # there and --gpus must agree with WORLD_SIZE
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
               '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)]
        # (torch.distributed.run's parser takes a bare `--n` for an abbreviation of its own options, even behind the script)
        cmd += ['--length' + a[3:] if a == '--n' or a.startswith('--n=') else a for a in sys.argv[1:]]
        sys.exit(subprocess.call(cmd, env=env))
    if int(os.environ.get('WORLD_SIZE', 1)) != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get('WORLD_SIZE', '1')))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # validation aids (a 1-GPU box cannot host two RCCL ranks): SSQ_BENCH_BACKEND=gloo with
    # SSQ_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 to exercise the N > 1 code path
    backend = os.environ.get('SSQ_BENCH_BACKEND', 'nccl')
    if os.environ.get('SSQ_BENCH_ONE_DEVICE'):
        local_rank = 0
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if not os.environ.get('SSQ_BENCH_ONE_DEVICE'):
            assert torch.cuda.device_count() >= world, \
                "%d ranks but %d GPUs visible" % (world, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device('cuda', local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    if world > 1:
        # every rank designs its own plan (deterministic, no exchange): its share of the host cores
        os.environ.setdefault('SSQ_HOST_THREADS', str(max(1, (os.cpu_count() or 1) // world)))
    import ssqueezepy_amd as S
    N, na, B = args.n, args.na, args.batch
    wav = S.Wavelet()
    if args.scales == 'log':
        scales = S.process_scales('log', N, wav, nv=32)[:na]
    else:
        scales = 'log-piecewise'
        na = len(S.process_scales(scales, N, wav, nv=32))
    # weak scaling: the job's batch is B*world signals; rank r owns a contiguous block
    # (ssqueezepy_amd/sharding.py) -- its own signals, its own plan, no exchange
    from ssqueezepy_amd.sharding import shard_bounds, gather_summaries, signal_summary
    lo, hi = shard_bounds(B * world, world, rank)
    x_host = np.stack([two_chirps(N, seed=s) for s in range(lo, hi)])
    x = torch.as_tensor(x_host, dtype=torch.float32, device=dev)

    def step():
        return S.ssq_cwt(x, wav, scales=scales)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    if world > 1:
        # the single collective of the job: per-signal summaries to every rank (RCCL)
        table = gather_summaries(signal_summary(out[0], out[1]), B * world)
        assert table.shape[0] == B * world
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall, gpu_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, gpu_ms = t.tolist()

    # optional: what a caller pays to collect every rank's Tx on every rank (SURVEY 8e:
    # config 4's Tx is 24.6 GB per GPU, so the gather -- per-link xGMI bound -- not the
    # transform, sets the rate). Outside the timed region of `value`.
    gather_ms = None
    if world > 1 and args.gather_tx:
        # in chunks of a few signals: one collective moves (world - 1) x chunk x na x N x 8 bytes per
        # rank over the xGMI links (point to point, ~153 GB/s each), the receive buffer stays small
        # enough to sit next to the transform's own outputs (config 4: 24.6 GB of Tx per GPU)
        from ssqueezepy_amd.sharding import gather_tx
        chunk = max(1, min(B, int(os.environ.get('SSQ_GATHER_CHUNK', '8'))))
        seen = [0]
        def consume(c0, block):                           # (a real caller reduces / stores the block here)
            seen[0] += int(block.shape[0] * block.shape[1])
        Tx = step()[0]
        gather_tx(Tx[:chunk], chunk=chunk, consume=consume)   # (warm-up of the collective and its buffer)
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        Tx = step()[0]
        gather_tx(Tx, chunk=chunk, consume=consume)
        torch.cuda.synchronize(); dist.barrier()
        tg = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = tg.item() * 1e3

    # per-stage times of the same workload, HIP events inside the plan on the launch
    # stream (outside the timed region: the plan synchronises while it measures)
    stages = None
    if rank == 0:
        from ssqueezepy_amd._cwt import _PLAN_CACHE
        plan = next(iter(_PLAN_CACHE.values()))
        plan.timing(1)
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
        ms, nsig = plan.timing(0)
        if nsig:
            stages = {k: v / nsig * 1e3 for k, v in zip(
                ("pad_fft_spectra_us", "block_rows_us", "exact_rows_us", "reassignment_us"), ms)}

    if rank == 0:
        transforms = args.steps * B * world
        value = transforms / wall
        bytes_alg = N * 4 + 2 * na * N * 8          # per transform
        t_transform = (gpu_ms / 1e3) / (args.steps * B)   # per GPU, event-timed
        achieved = bytes_alg / t_transform / 1e9
        traffic = traffic_sha = None
        tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.isfile(tfile) and (N, na) == (160000, 300):
            # HBM bytes per transform from rocprofv3 PMC passes of this same command
            # (tools/pmc_collect.sh + tools/pmc_traffic.py; FETCH_SIZE doubled as the
            # gfx950 guide prescribes, calibrated on the reassignment kernel's known
            # read volume). Measured offline, not in this run.
            with open(tfile) as fh:
                tj = json.load(fh)
            traffic = tj.get('bytes_per_transform')
            traffic_sha = tj.get('git_sha')
        # the device code this run executed (stamped into the library at build time: the last commit that
        # touched csrc/ or include/) against the one the PMC traffic figure was collected on
        from ssqueezepy_amd import _lib
        build_sha = _lib.load(build_if_missing=False).ssq_build_sha().decode()
        line = {
            "metric": "ssq_cwt transforms/sec (N=160k, 300 scales, f32)",
            "value": value, "unit": "transforms/s", "n_gpus": world, "ranks": (dist.get_world_size() if world > 1 else 1),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "per_gpu": {"transforms_per_s": value / world, "signals_per_step": B,
                        "output_bytes_per_step": int(2 * B * na * N * 8)},
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ssq_cwt('gmw'), N=%d, %d %s scales (nv=32), "
                                   "float32, two-chirp+noise" % (N, na, args.scales),
                       "signals_per_gpu_per_step": B,
                       "sharding": "independent signals per rank, no data-path "
                                   "collective; one all_gather of checksums",
                       "algo": plan.algo, "build_sha": build_sha,
                       "tile_kernel": {1: "ordered (ticketed float32 tile)",
                                       2: "float64 tile, unordered ds_add_f64, one column per lane (tile2_kernel)",
                                       3: "float64 tile, unordered ds_add_f64, column pair per lane (tile3_kernel)"
                                       }.get(plan.tile_kernel) if 'tiles' in plan.algo else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_measured_at": traffic_sha,
                         "traffic_stale": (traffic is not None and traffic_sha != build_sha),
                         "scope": "whole transform (all kernels), HIP-event timed",
                         "bytes_alg_per_transform": bytes_alg,
                         "us_per_transform": t_transform * 1e6},
        }
        if gather_ms is not None:
            line["with_full_tx_gather"] = {
                "ms_per_step": gather_ms,
                "transforms_per_s": B * world / (gather_ms * 1e-3),
                "gathered_bytes_per_rank": int((world - 1) * B * na * N * 8)}
        if stages:
            line["stages_us_per_transform"] = stages
            grp = min(plan.group, B)
            tiles = 'tiles' in plan.algo
            if tiles:
                # the column-tile kernel: writes Wx of the rows it interpolates and all of Tx,
                # reads Wx + 2-byte bin of the rows the block / exact kernels left in HBM
                kname = {1: "ssq::tile_kernel", 2: "ssq::tile2_kernel", 3: "ssq::tile3_kernel"}.get(plan.tile_kernel, "?")
                acc_bytes = N * (plan.tile_rows * 8 + na * 8 + (na - plan.tile_rows) * 10)
            else:
                # the reassignment: Wx (8 B) + bin map (2 B) in and Tx (8 B) out per point
                kname = "ssq::accumulate_tile16_kernel"
                acc_bytes = na * N * 18
            line["dominant_kernel"] = {
                "name": kname, "us": stages["reassignment_us"],
                "signals_per_launch": grp,
                "us_per_launch": stages["reassignment_us"] * grp,
                "bytes_moved": acc_bytes,
                "GBps": acc_bytes / (stages["reassignment_us"] * 1e-6) / 1e9,
                "frac_of_hbm_peak": acc_bytes / (stages["reassignment_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu and args.scales == 'log':
            try:
                line["cpu_baseline"] = cpu_baseline(N, na)
            except Exception as e:              # never lose the GPU number
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
