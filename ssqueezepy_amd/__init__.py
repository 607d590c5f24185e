# -*- coding: utf-8 -*-
"""ssqueezepy_amd -- MI355X-native synchrosqueezed CWT / STFT engine.

Drop-in for the transform path of ssqueezepy (`cwt`, `stft`, `ssq_cwt`, `ssq_stft`,
`ssqueeze`, `phase_cwt`, `phase_stft`, their inverses `icwt`, `issq_cwt`, `istft`,
`issq_stft`, `extract_ridges`, `trigdiff`, `phase_ssqueeze`, `Wavelet` and the scale-design
utilities): same Python API, computed by hand-written HIP kernels (gfx950) behind a
C ABI (include/ssq_hip.h, libssq_hip.so). The design step (scales, filter bank,
frequency grid, windows) is host NumPy and value-exact with the reference; all
O(na * N) work is on the GPU and there is no CPU fallback.
"""
__version__ = '0.1.0'

from .configs import EPS32, EPS64, USE_GPU, IS_PARALLEL
from .padding import p2up, padsignal
from .wavelets import Wavelet, center_frequency
from .scales import (process_scales, make_scales, cwt_scalebounds, infer_scaletype,
                     logscale_transition_idx, adm_ssq, adm_cwt)


def __getattr__(name):
    # the compute layer needs torch + the HIP library; import it lazily so that the
    # host-side design modules stay importable on a machine without a GPU
    _lazy = {
        'cwt': '_cwt', 'cwt_higher_order': '_cwt', 'ssq_cwt': '_ssq_cwt', 'phase_cwt': '_ssq_cwt',
        'stft': '_stft', 'get_window': '_stft', 'ssq_stft': '_ssq_stft',
        'phase_stft': '_ssq_stft', 'ssqueeze': 'ssqueezing',
        'ssqueeze_fast': 'algos', 'indexed_sum_onfly': 'algos', 'buffer': 'algos',
        'replace_under_abs': 'algos', 'phase_cwt_gpu': 'algos',
        'phase_stft_gpu': 'algos',
        'icwt': '_inverse', 'issq_cwt': '_inverse', 'istft': '_inverse',
        'issq_stft': '_inverse', 'extract_ridges': 'ridge_extraction',
        'freq_to_scale': 'experimental', 'scale_to_freq': 'experimental', 'trigdiff': 'common',
        'phase_ssqueeze': 'experimental', 'phase_transform': 'experimental',
    }
    if name in _lazy:
        import importlib
        mod = importlib.import_module('.' + _lazy[name], __name__)
        obj = getattr(mod, name)
        globals()[name] = obj          # later look-ups find it directly (the import costs ~5 us per call)
        return obj
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
