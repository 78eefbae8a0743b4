"""No-op stand-in for `numba` (TEST INFRASTRUCTURE, build container only).

The upstream reference decorates ten tiny functions with ``@jit(nopython=True)``.
numba is not installable in this image, so the fixture generator puts this
directory ahead of the reference on ``sys.path``; the decorated functions then
run as plain numpy code.  Nothing in the product imports this.
"""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


njit = jit
prange = range
