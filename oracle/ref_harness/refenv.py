"""Import the upstream Python reference from /root/reference (TEST INFRASTRUCTURE).

Only usable in the build container (the reference does not travel to the GPU
box).  Used by ``tests/golden/make_golden.py`` to produce the committed golden
vectors (the oracle is then checked against those vectors by
``tests/test_oracle_golden.py``, which needs no reference).

Two adjustments make the import faithful to the numba-compiled original:

* a no-op ``numba`` stand-in (``standin/numba``);
* ``np.sum`` inside the two jit modules is replaced by a strict left-to-right
  float64 accumulation, which is what numba's lowering of ``np.sum`` does
  (numpy itself sums pairwise for n >= 8).  The reference files are not
  touched: the module-level name ``np`` of the two modules is rebound to a
  proxy object after import.

Known deviation from a real numba run: ``np.linalg.norm`` in
``predeconmc_functions.calculate_euclidean_dist`` evaluates as numpy's
sqrt(dot) here, while numba lowers it to BLAS nrm2.  The oracle and the HIP
kernel use sqrt of the left-to-right sum; eps-neighbourhood membership exactly
on the ``<= upsilon * epsilon`` boundary can therefore differ from a numba run
by one ulp ("known unpinned spot").  No golden hits it.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("CHRONOCLUST_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "chronoclust"))


class _SequentialSumNumpy(object):
    """Forwards everything to numpy except ``sum`` (1-D, left to right)."""

    def __init__(self, np_module):
        self._np = np_module

    def __getattr__(self, name):
        return getattr(self._np, name)

    def sum(self, a, *args, **kwargs):
        np = self._np
        arr = np.asarray(a)
        if args or kwargs or arr.ndim != 1 or arr.dtype != np.float64:
            return np.sum(a, *args, **kwargs)
        acc = np.float64(0.0)
        for v in arr:
            acc = acc + v
        return acc


_loaded = None


def load(sequential_sum=True):
    """Returns the imported reference package ``chronoclust``."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    standin = os.path.join(_HERE, "standin")
    for p in (REFERENCE_ROOT, standin):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, standin)
    import numpy
    import chronoclust  # noqa: F401
    import chronoclust.app  # noqa: F401
    import chronoclust.utilities.mc_functions as mcf
    import chronoclust.utilities.predeconmc_functions as pdf
    if sequential_sum:
        proxy = _SequentialSumNumpy(numpy)
        mcf.np = proxy
        pdf.np = proxy
    _loaded = chronoclust
    return chronoclust
