"""Thin ctypes binding of include/chronoclust_hip.h.  Fails loudly when the HIP library or a GPU is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

import numpy as np

from . import build as _build

PCORE, OUTLIER = 0, 1
MAX_DIM = 128

_ERRORS = {-1: "no HIP device / HIP runtime error", -2: "bad argument", -3: "non-finite input", -4: "out of memory",
           -5: "internal error", -6: "exchange between ranks failed"}


class ChronoclustHipError(RuntimeError):
    pass


class CcParams(C.Structure):
    _fields_ = [("eps_sq", C.c_double), ("delta_sq", C.c_double), ("k", C.c_double), ("beta", C.c_double),
                ("mu", C.c_double), ("omicron", C.c_double), ("ups_eps", C.c_double), ("ups_eps_sq", C.c_double),
                ("delta", C.c_double), ("pi", C.c_int32), ("pad", C.c_int32)]


class CcTuning(C.Structure):
    _fields_ = [("window", C.c_int32), ("rounds", C.c_int32), ("segments", C.c_int32),
                ("windows_per_sync", C.c_int32), ("time_kernels", C.c_int32), ("dirty_segments", C.c_int32),
                ("early_window", C.c_int32), ("lookahead", C.c_int32), ("sequential", C.c_int32)]


class CcStats(C.Structure):
    _fields_ = [("points", C.c_int64), ("windows", C.c_int64), ("rounds", C.c_int64), ("truncated", C.c_int64),
                ("scan_launches", C.c_int64), ("scan_ms", C.c_double), ("scan_pair_dims", C.c_double),
                ("run_ms", C.c_double), ("rows", C.c_int64), ("table_rows_scanned", C.c_int64),
                ("lookahead_windows", C.c_int64), ("sharded_windows", C.c_int64), ("comm_launches", C.c_int64),
                ("comm_ms", C.c_double), ("seq_points", C.c_int64), ("scan_u_launches", C.c_int64),
                ("scan_p_launches", C.c_int64), ("pruned_scan_rows", C.c_int64), ("pruned_scan_full_rows", C.c_int64),
                ("window", C.c_int64), ("long_chains", C.c_int64), ("long_chain_launches", C.c_int64),
                ("tiles", C.c_int64), ("dirty_tiles", C.c_int64),
                ("scan_launches_pruned", C.c_int64), ("scan_ms_pruned", C.c_double), ("scan_pair_dims_pruned", C.c_double),
                ("scan_g_launches", C.c_int64), ("missed_points", C.c_int64), ("probe_launches", C.c_int64),
                ("seq_r_points", C.c_int64), ("heavy_launches", C.c_int64),
                ("scan_lean_launches", C.c_int64), ("long_prepared", C.c_int64), ("long_replayed", C.c_int64),
                ("seq_g_points", C.c_int64), ("link_launches", C.c_int64),
                ("scan_p2_launches", C.c_int64),
                ("calib_allgather_us", C.c_double), ("calib_scan_ns_per_row_dim", C.c_double),
                ("split_threshold_row_dims", C.c_int64), ("split_threshold_row_dims_pruned", C.c_int64),
                ("missed_plain_launches", C.c_int64), ("seed16_launches", C.c_int64)]


POLICY_MAX_ROUNDS = 8


class CcPolicyConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("window", "rounds_max", "windows_per_sync", "early_window", "lookahead",
                                         "allow_nodirty", "prune_mode", "prune_applicable", "can_shard", "d", "resume",
                                         "allow_sparse", "allow_guess", "allow_probe")] + \
               [("shard_min_row_dims", C.c_int64), ("n_end", C.c_int64), ("shard_min_row_dims_pruned", C.c_int64),
                ("lookahead_pruned", C.c_int32), ("force_prune_rows", C.c_int32)]


class CcPolicyCarry(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("adapt_win", "clean_batches", "since_shrink", "pad")]


class CcPolicyObs(C.Structure):
    _fields_ = [("cursor", C.c_int64), ("m_rows", C.c_int32), ("stall_b", C.c_int32)] + \
               [(k, C.c_int64) for k in ("stat_windows", "stat_truncated", "stat_trunc_unknown", "stat_tiles",
                                         "stat_dirty_tiles", "stat_unsafe", "stat_missed")] + \
               [("round_hist", C.c_int64 * (POLICY_MAX_ROUNDS + 2)), ("prune_rows", C.c_uint64), ("prune_full", C.c_uint64),
                ("after_sequential", C.c_int32), ("tg_ok", C.c_int32)]


class CcPolicyDecision(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("win_cfg", "want", "rounds", "batch_windows", "lookahead", "nodirty", "prune",
                                         "shard", "restart", "bad", "stalled", "sparse", "probe", "pad")] + \
               [(k, C.c_int64) for k in ("wins", "pts", "trunc", "unk", "tiles", "dtiles", "grew", "prune_rows", "prune_full")]


class CcRelaxedStats(C.Structure):
    _fields_ = [("super_steps", C.c_int64), ("minibatch_points", C.c_int64), ("deferred_points", C.c_int64),
                ("reserved", C.c_int64 * 5)]


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_i8p = C.POINTER(C.c_int8)

# name -> (restype, argtypes): every symbol include/chronoclust_hip.h declares
SYMBOLS = {
    "cc_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "cc_destroy": (None, [C.c_void_p]),
    "cc_last_error": (C.c_char_p, [C.c_void_p]),
    "cc_set_tuning": (C.c_int, [C.c_void_p, C.POINTER(CcTuning)]),
    "cc_reset": (C.c_int, [C.c_void_p]),
    "cc_set_params": (C.c_int, [C.c_void_p, C.POINTER(CcParams)]),
    "cc_decay_downgrade": (C.c_int, [C.c_void_p, C.c_double]),
    "cc_points_upload": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32]),
    "cc_points_prefetch": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32, _dp, _dp]),
    "cc_online_run": (C.c_int, [C.c_void_p]),
    "cc_col_minmax": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32, _dp, _dp]),
    "cc_points_upload_scaled": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32, _dp, _dp]),
    "cc_points_download": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "cc_labels_download": (C.c_int, [C.c_void_p, _i64p, _i8p]),
    "cc_online": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32, _i64p, _i8p]),
    "cc_count": (C.c_int, [C.c_void_p, C.c_int]),
    "cc_dim": (C.c_int, [C.c_void_p]),
    "cc_counters": (C.c_int, [C.c_void_p, _i64p, _i64p]),
    "cc_set_counters": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "cc_export": (C.c_int, [C.c_void_p, C.c_int, _i64p, _i64p, _dp, _dp, _dp, _dp, _dp]),
    "cc_inject_mc": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, _dp, _dp, _dp, _dp, C.c_double, C.c_int64,
                               C.c_int64]),
    "cc_inject_bulk": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _i64p, _i64p]),
    "cc_offline": (C.c_int, [C.c_void_p, _i32p, _i8p, _i32p, _i32p, _i32p]),
    "cc_num_core": (C.c_int, [C.c_void_p]),
    "cc_cluster_size": (C.c_int, [C.c_void_p, C.c_int32]),
    "cc_cluster_export": (C.c_int, [C.c_void_p, C.c_int32, _i64p, _dp, _dp, _dp, _dp, _dp]),
    "cc_clusters_total_members": (C.c_int, [C.c_void_p]),
    "cc_clusters_export": (C.c_int, [C.c_void_p, _i64p, _i32p, _dp, _dp, _dp, _dp, _dp]),
    "cc_assoc_argmin": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32, _dp, C.c_int32, C.c_int32, _i32p, _dp]),
    "cc_get_stats": (C.c_int, [C.c_void_p, C.POINTER(CcStats)]),
    "cc_sync": (C.c_int, [C.c_void_p]),
    "cc_policy_replay": (C.c_int, [C.POINTER(CcPolicyConfig), C.POINTER(CcPolicyCarry), C.c_int64, C.c_int32,
                                   C.POINTER(CcPolicyObs), C.c_int32, C.POINTER(CcPolicyDecision)]),
    "cc_point_clusters": (C.c_int, [C.c_void_p, _i32p]),
    "cc_format_points_csv": (C.c_int64, [_dp, C.c_int64, C.c_int32, C.c_int64, _i32p, C.c_char_p, _i32p, C.c_int32,
                                         C.c_void_p, C.c_int64]),
    "cc_comm_unique_id": (C.c_int, [C.c_char_p]),
    "cc_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "cc_comm_init_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "cc_comm_calibrate": (C.c_int, [C.c_void_p]),
    "cc_comm_destroy": (C.c_int, [C.c_void_p]),
    "cc_comm_info": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p]),
    "cc_comm_set_relaxed": (C.c_int, [C.c_void_p, C.c_int32]),
    "cc_get_relaxed_stats": (C.c_int, [C.c_void_p, C.POINTER(CcRelaxedStats)]),
    "cc_policy_seq_rate_guess": (C.c_double, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cc_shard_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]),
    "cc_set_shard_thresholds": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32]),
}

COMM_ID_BYTES = 128

_lib = None


def load():
    """Loads libchronoclust_hip.so (built in-tree by chronoclust_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is None:
        path = os.environ.get("CHRONOCLUST_HIP_LIB") or _build.LIB_PATH  # (a build variant, for kernel experiments)
        if not os.path.exists(path):
            raise ChronoclustHipError(
                "HIP library %s not built; run `python -m chronoclust_amd.build` (needs hipcc). "
                "chronoclust_amd has no CPU fallback." % path)
        if _build.is_stale(path) and os.environ.get("CHRONOCLUST_HIP_ALLOW_STALE") != "1":
            # (keyed to the CONTENT of csrc/ and the header, not to file times: the GPU box receives a copy of the tree)
            raise ChronoclustHipError(
                "HIP library %s was not built from the sources beside it (csrc/ or include/chronoclust_hip.h changed since, or "
                "its .sha256 stamp is missing); run `python -m chronoclust_amd.build`." % path)
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


HOST_ONLY_SYMBOLS = ("cc_policy_replay", "cc_policy_seq_rate_guess", "cc_shard_rows", "cc_format_points_csv")


def load_host_only(path):
    """Binds a library that holds only the entry points that are plain host code (csrc/cc_host_abi.inc built by itself: the
    sanitizer builds of tests/host_san/) in the place of the HIP library, so that policy_replay, shard_rows and
    format_points_csv of this module drive it.  Test infrastructure: nothing that needs a handle works afterwards."""
    global _lib
    lib = C.CDLL(path)
    for name in HOST_ONLY_SYMBOLS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = SYMBOLS[name]
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


def comm_unique_id():
    """The 128-byte RCCL id rank 0 hands to the other ranks of a stream group (cc_comm_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().cc_comm_unique_id(buf)
    if rc != 0:
        raise ChronoclustHipError("cc_comm_unique_id failed (%s): is librccl available?" % _ERRORS.get(rc, rc))
    return buf.raw


def comm_init_local(handles):
    """Makes the Handles of this process one group (rank = position): the in-process transport of the exact
    multi-GPU path.  Each handle must then be driven by its own host thread; collective calls block until every
    member has made them."""
    arr = (C.c_void_p * len(handles))(*[h._h for h in handles])
    rc = load().cc_comm_init_local(arr, len(handles))
    if rc != 0:
        raise ChronoclustHipError("cc_comm_init_local failed: %s" % _ERRORS.get(rc, rc))


def format_points_csv(values, first_id, label_idx, labels, threads=8, chunk=65536, out=None):
    """The body of cluster_points_D{t}.csv (app.py:357-360, DataFrame.to_csv(index=False)) as bytes: one line per row of
    `values` [n, d] - "<id>,<label>,<repr(float)>,..." -, ids counting from first_id, labels[label_idx[r]] as the label
    (already CSV-quoted where needed; index -1 = the last label).  Formatted by cc_format_points_csv in chunks on a few
    threads (ctypes releases the GIL).  With `out` (a binary file object) the chunks are written to it in order as they
    become ready - the threads format ahead of the writer, nothing is joined in memory - and the byte count is returned."""
    from concurrent.futures import ThreadPoolExecutor
    lib = load()
    values = _f64(values)
    if values.ndim != 2:
        raise ValueError("values must be a [n, d] array, got shape %r" % (values.shape,))
    n, d = values.shape
    label_idx = np.ascontiguousarray(label_idx, dtype=np.int32)
    if label_idx.shape != (n,):  # (the C side reads n entries: a short array would be read past its end)
        raise ValueError("label_idx must hold one entry per row: shape %r for %d rows" % (label_idx.shape, n))
    if len(labels) < 1:
        raise ValueError("labels must not be empty")
    enc = [s.encode() for s in labels]
    offs = np.zeros(len(enc) + 1, np.int32)
    offs[1:] = np.cumsum([len(e) for e in enc])
    blob = b"".join(enc)
    per_row = 25 + max(len(e) for e in enc) + 33 * d + 2

    def work(a):
        b = min(n, a + chunk)
        buf = np.empty(per_row * (b - a), np.uint8)  # (not zeroed: the formatter says how much of it it filled)
        got = lib.cc_format_points_csv(values[a:b].ctypes.data_as(_dp), b - a, d, first_id + a,
                                       label_idx[a:b].ctypes.data_as(_i32p), blob, offs.ctypes.data_as(_i32p), len(enc),
                                       buf.ctypes.data, buf.size)
        if got < 0:
            raise ChronoclustHipError("cc_format_points_csv failed: %s" % _ERRORS.get(got, got))
        return memoryview(buf)[:got]

    if n == 0:
        return 0 if out is not None else b""
    starts = list(range(0, n, chunk))
    workers = max(1, threads)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        if out is None:
            return b"".join(ex.map(work, starts))
        # a bounded number of chunks in flight: the formatters run at most 2 x workers chunks ahead of the writer
        total, pending, nxt = 0, [], 0
        while nxt < len(starts) or pending:
            while nxt < len(starts) and len(pending) < 2 * workers:
                pending.append(ex.submit(work, starts[nxt]))
                nxt += 1
            part = pending.pop(0).result()
            out.write(part)
            total += len(part)
        return total


def policy_replay(config, carry, start, observations):
    """cc_policy_replay: the window policy over recorded observations, no GPU.  config: dict of cc_policy_config fields;
    carry: (adapt_win, clean_batches, since_shrink); start: (cursor, rows); observations: dicts of cc_policy_obs fields.
    Returns (decisions as dicts - one more than observations -, carry after the call)."""
    cfg = CcPolicyConfig(**{k: int(v) for k, v in config.items()})
    car = CcPolicyCarry(int(carry[0]), int(carry[1]), int(carry[2]), 0)
    n = len(observations)
    obs = (CcPolicyObs * max(n, 1))()
    for i, o in enumerate(observations):
        for k, v in o.items():
            if k == "round_hist":
                for r, x in enumerate(v):
                    obs[i].round_hist[r] = int(x)
            else:
                setattr(obs[i], k, int(v))
    out = (CcPolicyDecision * (n + 1))()
    rc = load().cc_policy_replay(C.byref(cfg), C.byref(car), int(start[0]), int(start[1]), obs, n, out)
    if rc != 0:
        raise ValueError("cc_policy_replay: %s" % _ERRORS.get(rc, rc))
    keys = [k for k, _ in CcPolicyDecision._fields_ if k != "pad"]
    return [{k: getattr(d, k) for k in keys} for d in out], (car.adapt_win, car.clean_batches, car.since_shrink)


def shard_rows(n, world, rank, unit=1):
    """[lo, hi) of n rows for `rank` of `world` in units of `unit` rows (cc_shard_rows; no GPU needed)."""
    lo, hi = C.c_int32(), C.c_int32()
    rc = load().cc_shard_rows(int(n), int(world), int(rank), int(unit), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise ValueError("cc_shard_rows(%r, %r, %r, %r): %s" % (n, world, rank, unit, _ERRORS.get(rc, rc)))
    return lo.value, hi.value


class Handle(object):
    """One HDDStream state on one GPU."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.cc_create(int(device), C.byref(h))
        if rc != 0:
            raise ChronoclustHipError("cc_create(device=%d) failed: %s. chronoclust_amd needs an MI355X-class HIP "
                                      "device; there is no CPU fallback." % (device, _ERRORS.get(rc, rc)))
        self._h = h
        self.device = device
        self._prefetched = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            msg = self._lib.cc_last_error(self._h)
            exc = ValueError if rc in (-2, -3) else ChronoclustHipError
            raise exc("%s: %s" % (_ERRORS.get(rc, rc), msg.decode() if msg else ""))
        return rc

    def set_tuning(self, window=0, rounds=0, segments=0, windows_per_sync=0, time_kernels=0, dirty_segments=0,
                   lookahead=0, early_window=0, sequential=0):
        t = CcTuning(window, rounds, segments, windows_per_sync, time_kernels, dirty_segments, int(early_window),
                     int(lookahead), int(sequential))
        self._check(self._lib.cc_set_tuning(self._h, C.byref(t)))

    def reset(self):
        self._check(self._lib.cc_reset(self._h))

    def set_params(self, eps_sq, delta_sq, k, beta, mu, omicron, ups_eps, ups_eps_sq, delta, pi):
        p = CcParams(eps_sq, delta_sq, k, beta, mu, omicron, ups_eps, ups_eps_sq, delta, int(pi), 0)
        self._check(self._lib.cc_set_params(self._h, C.byref(p)))

    def decay_downgrade(self, factor):
        self._check(self._lib.cc_decay_downgrade(self._h, float(factor)))

    def points_upload(self, x):
        x = _f64(x)
        if x.ndim != 2:
            raise ValueError("points must be a 2-d array")
        try:
            self._check(self._lib.cc_points_upload(self._h, _ptr(x), x.shape[0], x.shape[1]))
        finally:
            self._prefetched = None  # adopted or discarded by the library: the array is no longer pinned by us
        self._n = x.shape[0]

    def points_prefetch(self, x, scale=None, min_=None):
        """Starts the background upload of the NEXT timepoint's points (cc_points_prefetch).  `x` must be the very
        array (C-contiguous float64) that is later passed to points_upload / points_upload_scaled / online; the
        handle keeps a reference to it until then.  Contract: the array must not be written to between this call and
        that upload - the upload is recognised by pointer, shape and scaling, and the copy already on the device is
        used as it is."""
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags["C_CONTIGUOUS"] and x.ndim == 2):
            raise ValueError("points_prefetch needs a C-contiguous float64 [n, d] array (it is not copied)")
        if x.shape[0] == 0:
            return
        scale = None if scale is None else _f64(scale)
        min_ = None if min_ is None else _f64(min_)
        self._prefetched = (x, scale, min_)  # keeps the buffers alive while the worker reads them
        self._check(self._lib.cc_points_prefetch(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(scale), _ptr(min_)))

    def col_minmax(self, x):
        """Per-column (min, max) of x, NaN ignored, reduced on the device."""
        x = _f64(x)
        mn = np.empty(x.shape[1], dtype=np.float64)
        mx = np.empty(x.shape[1], dtype=np.float64)
        self._check(self._lib.cc_col_minmax(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(mn), _ptr(mx)))
        return mn, mx

    def points_upload_scaled(self, x, scale, min_):
        """points_upload of x * scale + min_ (MinMaxScaler.transform), scaled on the device."""
        x, scale, min_ = _f64(x), _f64(scale), _f64(min_)
        if x.ndim != 2 or scale.shape != (x.shape[1],) or min_.shape != (x.shape[1],):
            raise ValueError("points must be [n, d], scale and min_ [d]")
        try:
            self._check(self._lib.cc_points_upload_scaled(self._h, _ptr(x), x.shape[0], x.shape[1], _ptr(scale),
                                                          _ptr(min_)))
        finally:
            self._prefetched = None
        self._n = x.shape[0]

    def points_download(self, d, scale=None, min_=None):
        """The resident points; with scale / min_, (X - min_) / scale (MinMaxScaler.inverse_transform)."""
        out = np.empty((self._n, d), dtype=np.float64)
        if scale is None:
            self._check(self._lib.cc_points_download(self._h, _ptr(out), None, None))
        else:
            scale, min_ = _f64(scale), _f64(min_)
            self._check(self._lib.cc_points_download(self._h, _ptr(out), _ptr(scale), _ptr(min_)))
        return out

    def online_run(self):
        self._check(self._lib.cc_online_run(self._h))

    def labels_download(self, want_path=True):
        uid = np.empty(self._n, dtype=np.int64)
        path = np.empty(self._n, dtype=np.int8) if want_path else None
        self._check(self._lib.cc_labels_download(self._h, _ptr(uid, _i64p), _ptr(path, _i8p)))
        return uid, path

    def online(self, x):
        self.points_upload(x)
        self.online_run()
        return self.labels_download()

    def count(self, kind):
        return self._check(self._lib.cc_count(self._h, kind))

    def dim(self):
        return self._check(self._lib.cc_dim(self._h))

    def counters(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.cc_counters(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_counters(self, pcore_last_id, outlier_last_id):
        self._check(self._lib.cc_set_counters(self._h, int(pcore_last_id), int(outlier_last_id)))

    def export(self, kind):
        n, d = self.count(kind), self.dim()
        out = dict(id=np.empty(n, np.int64), uid=np.empty(n, np.int64), w=np.empty(n, np.float64),
                   cf1=np.empty((n, d)), cf2=np.empty((n, d)), cen=np.empty((n, d)), pref=np.empty((n, d)))
        self._check(self._lib.cc_export(self._h, kind, _ptr(out["id"], _i64p), _ptr(out["uid"], _i64p), _ptr(out["w"]),
                                        _ptr(out["cf1"]), _ptr(out["cf2"]), _ptr(out["cen"]), _ptr(out["pref"])))
        return out

    def inject(self, kind, cf1, cf2, cen, pref, w, id, uid):
        cf1, cf2, cen, pref = _f64(cf1), _f64(cf2), _f64(cen), _f64(pref)
        self._check(self._lib.cc_inject_mc(self._h, kind, len(cf1), _ptr(cf1), _ptr(cf2), _ptr(cen), _ptr(pref),
                                           float(w), int(id), int(uid)))

    def inject_bulk(self, kind, cf1, cf2, cen, pref, w, id, uid):
        """Appends len(w) microclusters to a list with one upload per column (cc_inject_bulk)."""
        cf1, cf2, cen, pref, w = _f64(cf1), _f64(cf2), _f64(cen), _f64(pref), _f64(w)
        id = np.ascontiguousarray(id, dtype=np.int64)
        uid = np.ascontiguousarray(uid, dtype=np.int64)
        n = w.shape[0]
        if n == 0:
            return
        if cf1.shape != (n, cf1.shape[1]) or any(a.shape != cf1.shape for a in (cf2, cen, pref)) or \
                id.shape != (n,) or uid.shape != (n,):
            raise ValueError("inject_bulk: cf1/cf2/cen/pref must be [n, d], w/id/uid [n]")
        self._check(self._lib.cc_inject_bulk(self._h, kind, cf1.shape[1], n, _ptr(cf1), _ptr(cf2), _ptr(cen),
                                             _ptr(pref), _ptr(w), _ptr(id, _i64p), _ptr(uid, _i64p)))

    def offline_arrays(self, dumps=False):
        """cc_offline + cc_clusters_export: (members, offsets, w, cf1, cf2, cen, pref), info - all clusters as arrays."""
        n = C.c_int32()
        m = self.count(PCORE) if dumps else 0
        core = np.zeros(m, np.int8) if dumps else None
        pdim, nn, nw = ((np.zeros(m, np.int32) for _ in range(3)) if dumps else (None, None, None))
        self._check(self._lib.cc_offline(self._h, C.byref(n), _ptr(core, _i8p), _ptr(pdim, _i32p), _ptr(nn, _i32p),
                                         _ptr(nw, _i32p)))
        d, nc = self.dim(), n.value
        tot = self._check(self._lib.cc_clusters_total_members(self._h))
        mem = np.empty(tot, np.int64)
        off = np.zeros(nc + 1, np.int32)
        w = np.empty(nc, np.float64)
        cf1, cf2, cen, pref = (np.empty((nc, d)) for _ in range(4))
        self._check(self._lib.cc_clusters_export(self._h, _ptr(mem, _i64p), _ptr(off, _i32p), _ptr(w), _ptr(cf1),
                                                 _ptr(cf2), _ptr(cen), _ptr(pref)))
        info = dict(core=core, pdim=pdim, nn=nn, nw=nw) if dumps else None
        return (mem, off, w, cf1, cf2, cen, pref), info

    def offline(self, dumps=False):
        """The same, one dict per cluster."""
        (mem, off, w, cf1, cf2, cen, pref), info = self.offline_arrays(dumps)
        clusters = [dict(members=mem[off[c]:off[c + 1]], w=float(w[c]), cf1=cf1[c], cf2=cf2[c], cen=cen[c],
                         pref=pref[c]) for c in range(len(w))]
        return clusters, info

    def point_clusters(self):
        """Cluster index (order of offline_arrays) of every resident point, -1 outside every cluster (cc_point_clusters)."""
        out = np.empty(self._n, dtype=np.int32)
        if self._n:
            self._check(self._lib.cc_point_clusters(self._h, _ptr(out, _i32p)))
        return out

    def num_core(self):
        return self._check(self._lib.cc_num_core(self._h))

    def assoc_argmin(self, cur_cen, cur_pref, prev_cen):
        cc, cp, pc = _f64(cur_cen), _f64(cur_pref), _f64(prev_cen)
        mc, d = cc.shape
        mp = pc.shape[0]
        idx = np.empty(mc, np.int32)
        dist = np.empty(mc, np.float64)
        self._check(self._lib.cc_assoc_argmin(self._h, _ptr(cc), _ptr(cp), mc, _ptr(pc), mp, d, _ptr(idx, _i32p),
                                              _ptr(dist)))
        return idx, dist

    def comm_init_rccl(self, unique_id, rank, world):
        """Joins the RCCL communicator of a stream group (collective over the `world` processes)."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % COMM_ID_BYTES)
        self._check(self._lib.cc_comm_init_rccl(self._h, bytes(unique_id), int(rank), int(world)))

    def comm_calibrate(self):
        """Measures the all-gather of a window's records and a plain scan and derives the split thresholds from them
        (cc_comm_calibrate; collective: every rank of the group calls it - the members of an in-process group from their
        own threads; cc_comm_init_rccl has done it already).  Returns the measured figures and the thresholds."""
        self._check(self._lib.cc_comm_calibrate(self._h))
        s = self.stats()
        return {k: s[k] for k in ("calib_allgather_us", "calib_scan_ns_per_row_dim", "split_threshold_row_dims",
                                  "split_threshold_row_dims_pruned")}

    def comm_destroy(self):
        self._check(self._lib.cc_comm_destroy(self._h))

    def comm_info(self):
        r, w, t = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.cc_comm_info(self._h, C.byref(r), C.byref(w), C.byref(t)))
        return dict(rank=r.value, world=w.value, transport={0: "none", 1: "rccl", 2: "local"}[t.value])

    def comm_set_relaxed(self, minibatch_points):
        """RELAXED multi-GPU mode (events sharded over the ranks, CF deltas all-reduced per super-step of
        `minibatch_points` points per rank; not the reference's semantics).  0: back to the exact path."""
        self._check(self._lib.cc_comm_set_relaxed(self._h, int(minibatch_points)))

    def relaxed_stats(self):
        s = CcRelaxedStats()
        self._check(self._lib.cc_get_relaxed_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in CcRelaxedStats._fields_ if k != "reserved"}

    def set_shard_thresholds(self, min_row_dims=-1, offline_min_rows=-1):
        self._check(self._lib.cc_set_shard_thresholds(self._h, int(min_row_dims), int(offline_min_rows)))

    def sync(self):
        """Waits for everything enqueued on the handle's HIP streams (cc_sync): the timing bracket of bench.py."""
        self._check(self._lib.cc_sync(self._h))

    def stats(self):
        s = CcStats()
        self._check(self._lib.cc_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in CcStats._fields_ if k != "reserved"}
