# -*- coding: utf-8 -*-
"""The reference-side binding of libssq_hip's kernel-level C ABI -- the file a maintainer of
ssqueezepy would add next to ``ssqueezepy/algos.py`` (Option B), executed.

ssqueezepy dispatches its loop nests through one table, ``algos._cpu_fns`` (algos.py:986-1005:
``ssqueeze_fast`` / ``indexed_sum_onfly`` look the function up by name and call it with the
arrays and the keyword parameters ``_process_ssq_params`` marshalled, algos.py:139-149,
160-168), and ``phase_cwt_cpu`` / ``phase_stft_cpu`` / ``replace_under_abs`` call their
``_par`` variants by name (algos.py:716-718, 790-792, 500-502). `install` puts functions with
the very same signatures in those places; each one hands plain pointers, sizes and the
reference's own parameters to the C ABI (include/ssq_hip.h) through ctypes -- no torch, no
ssqueezepy_amd Python.

Pointers: the ABI takes *device* pointers. In the build container there is no GPU; the
library bound here is then ``libssq_hip_emu.so`` (tests/emu: the same kernels compiled for the
host), for which host memory is device memory, so NumPy buffers are passed as they are. On a
GPU box the same calls take ``tensor.data_ptr()`` at the reference's GPU seam
(algos.py:139-143: ``_run_on_gpu(kernel, *args)``), see INTEGRATION.md.

What runs through it: the reference's own tests of these kernels, unmodified
(tests/fft_test.py:141-377), with its own thresholds -- see tests/test_reference_binding.py.
TEST INFRASTRUCTURE of this repository; the product never imports it.
"""
import ctypes
import numpy as np

F32, F64 = 0, 1
GRID_LOG, GRID_LOG_PIECEWISE, GRID_LIN = 0, 1, 2
_vp, _i64, _dbl, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int


class Binding:
    def __init__(self, lib_path):
        lib = ctypes.CDLL(lib_path)
        P5 = ctypes.POINTER(_dbl)
        lib.ssq_ssqueeze.argtypes = [_int, _vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _dbl,
                                     _int, P5, _int, _vp, _vp]
        lib.ssq_indexed_sum.argtypes = [_int, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _int, P5,
                                        _int, _vp]
        lib.ssq_phase_cwt.argtypes = [_int, _vp, _vp, _vp, _i64, _i64, _i64, _dbl, _vp]
        lib.ssq_phase_stft.argtypes = [_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _dbl, _vp]
        lib.ssq_replace_under_abs.argtypes = [_int, _vp, _vp, _i64, _dbl, _dbl, _vp]
        lib.ssq_last_error.restype = ctypes.c_char_p
        self.lib = lib

    # ---- helpers
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libssq_hip: " + self.lib.ssq_last_error().decode())

    @staticmethod
    def _code(a):
        return F32 if a.dtype in (np.float32, np.complex64) else F64

    @staticmethod
    def _ptr(a):
        assert a.flags['C_CONTIGUOUS']
        return a.ctypes.data

    def _const(self, Wx, const):
        # the reference hands over a vector in the data dtype, or a float64 vector with float32
        # data (algos.py:66-79): the latter makes the sums go through double
        # (a scalar `const` arrives as np.full(na, const, dtype=Wx.dtype), i.e. complex with a zero
        # imaginary part, algos.py:70-72: Wx * const is then the real scaling the ABI does)
        if np.iscomplexobj(const):
            assert not const.imag.any()
            const = const.real
        const = np.ascontiguousarray(const)
        f64 = int(self._code(Wx) == F32 and const.dtype == np.float64)
        return const, f64

    @staticmethod
    def _params(*v):
        p = (_dbl * 5)()
        for i, x in enumerate(v):
            p[i] = float(x)
        return p

    def _ssq(self, Wx, dWx, Sfs, out, const, gamma, grid, params, flipud):
        const, f64 = self._const(Wx, const)
        na, n = Wx.shape
        # (the library overwrites Tx; the reference accumulates into a zeroed `out`: same thing)
        self._check(self.lib.ssq_ssqueeze(self._code(Wx), self._ptr(Wx), self._ptr(dWx),
                                          self._ptr(Sfs) if Sfs is not None else None,
                                          self._ptr(out), self._ptr(const), f64, 1, na, n,
                                          float(gamma), grid, params, int(bool(flipud)), None, None))

    def _isum(self, Wx, w, out, const, grid, params, flipud):
        const, f64 = self._const(Wx, const)
        na, n = Wx.shape
        self._check(self.lib.ssq_indexed_sum(self._code(Wx), self._ptr(Wx), self._ptr(w),
                                             self._ptr(out), self._ptr(const), f64, 1, na, n,
                                             grid, params, int(bool(flipud)), None))

    # ---- the table entries: the reference's signatures (algos.py:859-984, 172-250)
    def ssq_cwt_log(self, Wx, dWx, out, const, gamma, vlmin, dvl, omax, flipud=False):
        self._ssq(Wx, dWx, None, out, const, gamma, GRID_LOG, self._params(vlmin, dvl), flipud)

    def ssq_cwt_log_piecewise(self, Wx, dWx, out, const, gamma, vlmin0, vlmin1, dvl0, dvl1, idx1,
                              omax, flipud=False):
        self._ssq(Wx, dWx, None, out, const, gamma, GRID_LOG_PIECEWISE,
                  self._params(vlmin0, vlmin1, dvl0, dvl1, idx1), flipud)

    def ssq_cwt_lin(self, Wx, dWx, out, const, gamma, vmin, dv, omax, flipud=False):
        self._ssq(Wx, dWx, None, out, const, gamma, GRID_LIN, self._params(vmin, dv), flipud)

    def ssq_stft(self, Wx, dWx, Sfs, out, const, gamma, vmin, dv, omax, flipud=False):
        self._ssq(Wx, dWx, np.ascontiguousarray(Sfs), out, const, gamma, GRID_LIN,
                  self._params(vmin, dv), flipud)

    def indexed_sum_log(self, Wx, w, out, const, vlmin, dvl, omax, flipud=False):
        self._isum(Wx, w, out, const, GRID_LOG, self._params(vlmin, dvl), flipud)

    def indexed_sum_log_piecewise(self, Wx, w, out, const, vlmin0, vlmin1, dvl0, dvl1, idx1, omax,
                                  flipud=False):
        self._isum(Wx, w, out, const, GRID_LOG_PIECEWISE,
                   self._params(vlmin0, vlmin1, dvl0, dvl1, idx1), flipud)

    def indexed_sum_lin(self, Wx, w, out, const, vmin, dv, omax, flipud=False):
        self._isum(Wx, w, out, const, GRID_LIN, self._params(vmin, dv), flipud)

    def phase_cwt(self, Wx, dWx, out, gamma):
        na, n = Wx.shape
        self._check(self.lib.ssq_phase_cwt(self._code(Wx), self._ptr(Wx), self._ptr(dWx),
                                           self._ptr(out), 1, na, n, float(gamma), None))

    def phase_stft(self, Wx, dWx, Sfs, out, gamma):
        na, n = Wx.shape
        self._check(self.lib.ssq_phase_stft(self._code(Wx), self._ptr(Wx), self._ptr(dWx),
                                            self._ptr(np.ascontiguousarray(Sfs)), self._ptr(out),
                                            1, na, n, float(gamma), None))

    def replace_under_abs(self, x, ref, value=0., replacement=0.):
        self._check(self.lib.ssq_replace_under_abs(self._code(ref), self._ptr(x), self._ptr(ref),
                                                   x.size, float(value), float(replacement), None))


def install(algos, lib_path):
    """Route the reference's *parallel* CPU entries (what ``parallel=True`` / SSQ_PARALLEL=1
    select) to the library; the serial entries keep the reference's own loop nests, so its
    tests compare the two. Returns the names replaced."""
    b = Binding(lib_path)
    table = {
        'ssq_cwt_log_par': b.ssq_cwt_log, 'ssq_cwt_log_piecewise_par': b.ssq_cwt_log_piecewise,
        'ssq_cwt_lin_par': b.ssq_cwt_lin,
        # (the reference selects 'ssq_stft' for the STFT form whatever `parallel` says --
        # algos.py:131-133 drops the '_par' suffix, 'ssq_stft_par' is never looked up -- so it is
        # the serial entry that is bound)
        'ssq_stft': b.ssq_stft,
        'indexed_sum_log_par': b.indexed_sum_log,
        'indexed_sum_log_piecewise_par': b.indexed_sum_log_piecewise,
        'indexed_sum_lin_par': b.indexed_sum_lin,
    }
    for k, f in table.items():
        assert k in algos._cpu_fns, k
        algos._cpu_fns[k] = f
    algos._phase_cwt_par = b.phase_cwt
    algos._phase_stft_par = b.phase_stft
    algos._replace_under_abs_par = b.replace_under_abs
    return sorted(table) + ['_phase_cwt_par', '_phase_stft_par', '_replace_under_abs_par']
