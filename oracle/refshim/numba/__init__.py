"""Minimal stand-in for `numba`, used ONLY by oracle/gen_golden.py in the build
container so the pure-Python reference under /root/reference can be imported
(numba is not installed and there is no network). Decorators are identity, so the
reference's loop nests run as plain Python. This mirrors what the reference's own
test-suite does to get coverage (tests/z_all_test.py:8-20). Test infrastructure,
never imported by the product path.
"""


def _identity_decorator(*dargs, **dkw):
    # supports both `@jit` and `@jit(nopython=True, cache=True, parallel=True)`
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


jit = njit = vectorize = guvectorize = _identity_decorator
prange = range
__version__ = "0.0-stub"
