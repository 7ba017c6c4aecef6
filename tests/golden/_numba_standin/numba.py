"""Identity stand-in for the `numba` package, used ONLY by make_golden.py.

The reference's kernels are plain Python functions decorated with
``numba.njit(cache=True, fastmath=False[, parallel=True])`` and use
``numba.prange``.  With fastmath off, the JIT does not change IEEE results, so
running the undecorated functions in CPython yields the same values.
This module is test infrastructure; nothing in the product imports it.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
prange = range
