# -*- coding: utf-8 -*-
"""Timings of the other BASELINE.json configurations (the bench line is config 2):
C1 cwt N=10k; C3 ssq_stft n_fft=1024 hop=256 (single + batched); C5 ssq_cwt float64
N=1 048 576, 512 scales. Prints one JSON line per config (HIP-event timed, inputs in
HBM). Size-independent parity properties are asserted on the way."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import two_chirps


HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)


def roofline(bytes_alg, ms, what):
    """the block bench.py prints for config 2, for another BASELINE configuration: algorithmic bytes (SURVEY 8d:
    inputs read once + returned outputs written once) over the HIP-event time of the whole call"""
    gbs = bytes_alg / ms / 1e6
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": None, "bytes_alg": int(bytes_alg), "scope": what}


def build_sha():
    from ssqueezepy_amd import _lib
    return _lib.load(build_if_missing=False).ssq_build_sha().decode()


def timeit(fn, n):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def main():
    which = sys.argv[1:] or ['c1', 'c3', 'c5', 'inv']
    dev = torch.device('cuda')
    if 'c1' in which:
        N, na = 10000, 300
        wav = S.Wavelet()
        scales = S.process_scales('log', N, wav, nv=32)[:na]
        x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float32, device=dev)
        ms, out = timeit(lambda: S.cwt(x, wav, scales=scales), 50)
        print(json.dumps({"config": "C1 cwt N=10k 300 scales f32", "ms": ms,
                          "transforms_per_s": 1e3 / ms, "build_sha": build_sha(),
                          "roofline": roofline(N * 4 + na * N * 8, ms, "cwt: x in, Wx out; one call, launch-bound")}))
    if 'c3' in which:
        N = 160000
        for B in (1, 64, 512):
            xb = np.stack([two_chirps(N, s) for s in range(min(B, 8))])
            xb = np.tile(xb, (max(1, B // 8), 1))[:B]
            x = torch.as_tensor(xb if B > 1 else xb[0], dtype=torch.float32, device=dev)
            ms, out = timeit(lambda: S.ssq_stft(x, n_fft=1024, hop_len=256, dtype='float32'), 10)
            Tx, Sx = out[0], out[1]
            bytes_alg = B * (N * 4 + 2 * Sx.shape[-2] * Sx.shape[-1] * 8)
            print(json.dumps({"config": "C3 ssq_stft N=160k n_fft=1024 hop=256 f32", "batch": B,
                              "ms": ms, "transforms_per_s": B * 1e3 / ms,
                              "shape": list(Sx.shape), "GBps_alg": bytes_alg / ms / 1e6, "build_sha": build_sha(),
                              "roofline": roofline(bytes_alg, ms, "ssq_stft: x in, Tx + Sx out; %d signals per call" % B)}))
    if 'c5' in which:
        N, na = 1048576, 512
        wav = S.Wavelet(('gmw', {'dtype': 'float64'}))
        t0 = time.time()
        scales = S.process_scales('log', N, wav, nv=32)[:na]
        x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float64, device=dev)
        ms, out = timeit(lambda: S.ssq_cwt(x, wav, scales=scales), 2)
        Tx, Wx = out[0], out[1]
        # assignment-invariant checksum at full size (float64: 1e-12 relative)
        const = np.log(2) / 32
        lhs, rhs = Tx.sum(0), (Wx * const).sum(0)
        err = ((lhs - rhs).abs().max() / rhs.abs().max()).item()
        bytes_alg = N * 8 + 2 * na * N * 16
        print(json.dumps({"config": "C5 ssq_cwt N=1048576 512 scales f64", "ms": ms,
                          "transforms_per_s": 1e3 / ms, "colsum_rel_err": err,
                          "GBps_alg": bytes_alg / ms / 1e6, "build_sha": build_sha(),
                          "roofline": roofline(bytes_alg, ms, "ssq_cwt float64: x in, Tx + Wx out; one signal per call"),
                          "setup_s": time.time() - t0}))
        assert err < 1e-11, err

    if 'inv' in which:
        # inverses at config-2 size: one streaming pass over the (300, 160000) array
        N, na = 160000, 300
        wav = S.Wavelet()
        scales = S.process_scales('log', N, wav, nv=32)[:na]
        x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float32, device=dev)
        Tx, Wx, *_ = S.ssq_cwt(x, wav, scales=scales)
        from ssqueezepy_amd import algos
        ms, _ = timeit(lambda: algos.colsum_real(Tx), 20)
        print(json.dumps({"config": "colsum kernel (icwt / issq_cwt core) 300x160000 c64",
                          "ms": ms, "GBps": na * N * 8 / ms / 1e6}))
        ms, xr = timeit(lambda: S.issq_cwt(Tx, wav), 10)
        print(json.dumps({"config": "issq_cwt N=160k 300 scales f32 (incl. host design)",
                          "ms": ms}))
        ms, xr = timeit(lambda: S.icwt(Wx, wav, scales=scales, nv=32), 10)
        print(json.dumps({"config": "icwt N=160k 300 scales f32 (incl. host design)",
                          "ms": ms}))
        Sx = S.stft(x, n_fft=1024, hop_len=256, dtype='float32')
        ms, xr = timeit(lambda: S.istft(Sx, n_fft=1024, hop_len=256, N=N), 10)
        print(json.dumps({"config": "istft N=160k n_fft=1024 hop=256 f32", "ms": ms}))

    if 'ridges' in which:
        # ridge extraction at config-2 size (sequential in time: one workgroup)
        N, na = int(os.environ.get('RIDGE_N', 160000)), 300
        wav = S.Wavelet()
        scales = S.process_scales('log', N, wav, nv=32)[:na]
        x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float32, device=dev)
        Tx, Wx, ssq_freqs, sc = S.ssq_cwt(x, wav, scales=scales)
        ms, r = timeit(lambda: S.extract_ridges(Tx, ssq_freqs, penalty=2.0, n_ridges=1, bw=4), 2)
        print(json.dumps({"config": "extract_ridges ssq_cwt N=%d 300 scales f32, 1 ridge" % N,
                          "ms": ms, "us_per_step": ms * 1e3 / N}))
        # ... and over a batch of transforms in one call (one workgroup per transform)
        B = int(os.environ.get('RIDGE_B', 16))
        Tb = Tx[None].expand(B, -1, -1).contiguous()
        msb, rb = timeit(lambda: S.extract_ridges(Tb, ssq_freqs, penalty=2.0, n_ridges=1, bw=4), 2)
        assert torch.equal(rb[0], r) and torch.equal(rb[B - 1], r)
        print(json.dumps({"config": "extract_ridges batch of %d, N=%d 300 scales f32, 1 ridge" % (B, N),
                          "ms": msb, "ms_per_transform": msb / B, "speedup_vs_one_by_one": ms * B / msb}))


if __name__ == '__main__':
    main()
