"""ctypes face of the CPU oracle (oracle/chrono_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (chronoclust_amd/) never imports it.

`OracleHDDStream` follows the call sequence of the reference's
``HDDStream.online_microcluster_maintenance`` (clustering/hddstream.py:166-245):
dataset-dependent parameters, decay + downgrade when the daystamp changes, the
per-point online loop, then the offline PreDeCon phase.  Parameter derivation
uses the reference's own Python expressions so that the C side only ever sees
finished doubles.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libchrono_oracle.so")

PCORE, OUTLIER = 0, 1


class CoParams(C.Structure):
    _fields_ = [("eps_sq", C.c_double), ("delta_sq", C.c_double), ("k", C.c_double), ("beta", C.c_double),
                ("mu", C.c_double), ("omicron", C.c_double), ("ups_eps", C.c_double),
                ("ups_eps_sq", C.c_double), ("delta", C.c_double), ("pi", C.c_int32), ("pad", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "chrono_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.co_create.restype = C.c_void_p
        L.co_destroy.argtypes = [C.c_void_p]
        L.co_set_params.argtypes = [C.c_void_p, C.POINTER(CoParams)]
        L.co_inject_mc.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, C.c_int64, C.c_int64]
        L.co_decay_downgrade.argtypes = [C.c_void_p, C.c_double]
        L.co_online.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int8)]
        L.co_count.argtypes = [C.c_void_p, C.c_int]
        L.co_dim.argtypes = [C.c_void_p]
        L.co_pcore_last_id.argtypes = [C.c_void_p]
        L.co_pcore_last_id.restype = C.c_int64
        L.co_outlier_last_id.argtypes = [C.c_void_p]
        L.co_outlier_last_id.restype = C.c_int64
        L.co_export.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), dp, dp, dp, dp, dp]
        L.co_offline.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32)]
        L.co_num_core.argtypes = [C.c_void_p]
        L.co_num_clusters.argtypes = [C.c_void_p]
        L.co_cluster_size.argtypes = [C.c_void_p, C.c_int]
        L.co_cluster_export.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), dp, dp, dp, dp, dp]
        L.co_assoc_argmin.argtypes = [dp, dp, C.c_int, dp, C.c_int, C.c_int, C.POINTER(C.c_int32), dp]
        for name in ("co_projected_distance", "co_projected_radius_sq", "co_euclidean",
                     "co_variance_along_dimension", "co_weighted_dist_sq"):
            getattr(L, name).restype = C.c_double
        L.co_projected_distance.argtypes = [dp, dp, dp, C.c_int]
        L.co_projected_radius_sq.argtypes = [dp, dp, dp, C.c_double, C.c_int]
        L.co_update_pref.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, dp, C.c_int]
        L.co_is_core.argtypes = [dp, dp, dp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int]
        L.co_euclidean.argtypes = [dp, dp, C.c_int]
        L.co_variance_along_dimension.argtypes = [C.c_double, dp, C.c_int]
        L.co_weighted_dist_sq.argtypes = [dp, dp, dp, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# ---- vector-level functions (known-answer tests) ---------------------------

def projected_distance(centroid, pref, point):
    c, p, x = _f64(centroid), _f64(pref), _f64(point)
    return lib().co_projected_distance(_dp(c), _dp(p), _dp(x), len(c))


def projected_radius_sq(cf1, cf2, pref, w):
    a, b, p = _f64(cf1), _f64(cf2), _f64(pref)
    return lib().co_projected_radius_sq(_dp(a), _dp(b), _dp(p), float(w), len(a))


def update_pref(cf1, cf2, w, delta_sq, k):
    a, b = _f64(cf1), _f64(cf2)
    out = np.empty_like(a)
    lib().co_update_pref(_dp(a), _dp(b), float(w), float(delta_sq), float(k), _dp(out), len(a))
    return out


def is_core(cf1, cf2, pref, w, radius_thr_sq, density_thr, max_pdim):
    a, b, p = _f64(cf1), _f64(cf2), _f64(pref)
    return bool(lib().co_is_core(_dp(a), _dp(b), _dp(p), float(w), len(a), float(radius_thr_sq),
                                 float(density_thr), int(max_pdim)))


def euclidean(a, b):
    a, b = _f64(a), _f64(b)
    return lib().co_euclidean(_dp(a), _dp(b), len(a))


def variance_along_dimension(point, neighbours):
    n = _f64(neighbours)
    return lib().co_variance_along_dimension(float(point), _dp(n), len(n))


def weighted_dist_sq(pref, p, q):
    w, p, q = _f64(pref), _f64(p), _f64(q)
    return lib().co_weighted_dist_sq(_dp(w), _dp(p), _dp(q), len(p))


def assoc_argmin(cur_cen, cur_pref, prev_cen):
    cc, cp, pc = _f64(cur_cen), _f64(cur_pref), _f64(prev_cen)
    mc, d = cc.shape
    mp = pc.shape[0]
    idx = np.empty(mc, dtype=np.int32)
    dist = np.empty(mc, dtype=np.float64)
    lib().co_assoc_argmin(_dp(cc), _dp(cp), mc, _dp(pc), mp, d, idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(dist))
    return idx, dist


# ---- stateful oracle ---------------------------------------------------------

class OracleHDDStream(object):
    """Restates HDDStream (clustering/hddstream.py:29-549) on top of the C oracle."""

    def __init__(self, config):
        self.config = dict(config)
        # hddstream.py:45-52
        self.epsilon = float(config['epsilon'])
        self.epsilon_squared = self.epsilon ** 2
        self.upsilon = float(config['upsilon']) * self.epsilon
        self.delta = float(config['delta'])
        if self.delta > 1 or self.delta < 0:
            raise SystemExit("Given delta ({}) is out of range. Must be within 0-1.".format(self.delta))
        self.delta_squared = self.delta ** 2
        self.beta = float(config['beta'])
        self.k = float(config['k'])
        self.lambbda = float(config['lambda'])
        self.pi = None
        self.mu = None
        self.omicron = None
        self.dataset_size = 0
        self.dataset_dimensionality = 0
        self.last_data_timestamp = 0
        self._h = C.c_void_p(lib().co_create())
        self.labels_uid = None
        self.paths = None

    def __del__(self):
        try:
            if self._h:
                lib().co_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _push_params(self):
        p = CoParams(self.epsilon_squared, self.delta_squared, self.k, self.beta, float(self.mu),
                     float(self.omicron), self.upsilon, self.upsilon ** 2, self.delta, int(self.pi), 0)
        lib().co_set_params(self._h, C.byref(p))

    def set_dataset_dependent_parameters(self, X):
        # hddstream.py:89-128
        d = X.shape[1]
        self.dataset_dimensionality = d
        config_pi = float(self.config['pi'])
        self.pi = d if config_pi <= 0 else round(config_pi)
        self.omicron = self.config['omicron'] * self.dataset_size
        self.dataset_size = X.shape[0]
        self.mu = float(self.config['mu']) * self.dataset_size

    def online_microcluster_maintenance(self, X, daystamp, reset_param=True, offline=True):
        X = _f64(X)
        if reset_param:
            self.set_dataset_dependent_parameters(X)
        self._push_params()
        if (self.last_data_timestamp - daystamp) != 0:  # hddstream.py:199-205
            interval = daystamp - self.last_data_timestamp
            lib().co_decay_downgrade(self._h, 2 ** (-self.lambbda * interval))
        n, d = X.shape
        uid = np.empty(n, dtype=np.int64)
        path = np.empty(n, dtype=np.int8)
        rc = lib().co_online(self._h, _dp(X), n, d, uid.ctypes.data_as(C.POINTER(C.c_int64)),
                             path.ctypes.data_as(C.POINTER(C.c_int8)))
        if rc != 0:
            raise RuntimeError("co_online failed: %d" % rc)
        self.labels_uid, self.paths = uid, path
        self.last_data_timestamp = daystamp
        if offline:
            self.offline_clustering()

    def offline_clustering(self):
        self._push_params()
        m = lib().co_count(self._h, PCORE)
        core = np.zeros(m, dtype=np.int8)
        pdim = np.zeros(m, dtype=np.int32)
        nn = np.zeros(m, dtype=np.int32)
        nw = np.zeros(m, dtype=np.int32)
        lib().co_offline(self._h, core.ctypes.data_as(C.POINTER(C.c_int8)),
                         pdim.ctypes.data_as(C.POINTER(C.c_int32)), nn.ctypes.data_as(C.POINTER(C.c_int32)),
                         nw.ctypes.data_as(C.POINTER(C.c_int32)))
        self.offline_dump = dict(core=core, pdim=pdim, nn=nn, nw=nw)

    def inject(self, kind, cf1, cf2, cen, pref, w, id, uid):
        cf1, cf2, cen, pref = _f64(cf1), _f64(cf2), _f64(cen), _f64(pref)
        rc = lib().co_inject_mc(self._h, kind, len(cf1), _dp(cf1), _dp(cf2), _dp(cen), _dp(pref), float(w),
                                int(id), int(uid))
        assert rc == 0

    def table(self, kind):
        L = lib()
        n, d = L.co_count(self._h, kind), L.co_dim(self._h)
        out = dict(id=np.empty(n, np.int64), uid=np.empty(n, np.int64), w=np.empty(n, np.float64),
                   cf1=np.empty((n, d)), cf2=np.empty((n, d)), cen=np.empty((n, d)), pref=np.empty((n, d)))
        i64 = C.POINTER(C.c_int64)
        L.co_export(self._h, kind, out['id'].ctypes.data_as(i64), out['uid'].ctypes.data_as(i64), _dp(out['w']),
                    _dp(out['cf1']), _dp(out['cf2']), _dp(out['cen']), _dp(out['pref']))
        return out

    @property
    def counters(self):
        return int(lib().co_pcore_last_id(self._h)), int(lib().co_outlier_last_id(self._h))

    @property
    def clusters(self):
        L = lib()
        d = L.co_dim(self._h)
        res = []
        for c in range(L.co_num_clusters(self._h)):
            n = L.co_cluster_size(self._h, c)
            mem = np.empty(n, np.int64)
            w = C.c_double()
            cf1, cf2, cen, pref = (np.empty(d) for _ in range(4))
            L.co_cluster_export(self._h, c, mem.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(w), _dp(cf1),
                                _dp(cf2), _dp(cen), _dp(pref))
            res.append(dict(members=mem, w=w.value, cf1=cf1, cf2=cf2, cen=cen, pref=pref))
        return res
