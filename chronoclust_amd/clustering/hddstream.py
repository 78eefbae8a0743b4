"""HDDStream on MI355X: the interface of chronoclust/clustering/hddstream.py:29-549, the state in HBM.

`online_microcluster_maintenance(X, daystamp)` performs, in this order and with the reference's semantics,
  1. dataset-dependent parameters                       (hddstream.py:89-128)     host, Python
  2. decay + downgrade + delete when the daystamp moved (hddstream.py:199-213)    cc_decay_downgrade
  3. the per-point online loop, exact                   (hddstream.py:220-237)    cc_online
  4. the offline PreDeCon phase                         (hddstream.py:464-510)    cc_offline
through the C-ABI of include/chronoclust_hip.h.  Derived parameters use the reference's own Python
expressions so thresholds are the same doubles.
"""
import logging
import sys
from decimal import ROUND_HALF_UP, Decimal

import numpy as np

from .. import _lib
from ..objects.microcluster import ClusterView, MicroclusterView


def rounded_weights(w):
    """app.py:184 for a whole array: Decimal(str(x)).quantize(Decimal('1.1'), ROUND_HALF_UP) per weight.  Weights
    whose tenths are not within 1e-6 of a half-way case are rounded in float arithmetic (the decimal value of the
    shortest repr differs from the double by less than an ulp, so it falls on the same side); the rest - and
    anything unusual - goes through the reference's own expression."""
    w = np.asarray(w, dtype=np.float64)
    v = w * 10.0
    frac = v - np.floor(v)
    plain = np.isfinite(v) & (v >= 0.0) & (v < 1e14) & (np.abs(frac - 0.5) > 1e-6)
    tenths = np.floor(np.where(plain, v, 0.0) + 0.5).astype(np.int64).tolist()
    out = [Decimal(t).scaleb(-1) for t in tenths]
    for i in np.nonzero(~plain)[0].tolist():
        out[i] = Decimal(str(float(w[i]))).quantize(Decimal('1.1'), rounding=ROUND_HALF_UP)
    return out


class _TqdmToLogger(object):
    """The reference's TqdmToLogger (hddstream.py:552-573): what tqdm writes goes to the log, one record per refresh."""

    def __init__(self, logger, level=logging.INFO):
        self.logger, self.level, self.buf = logger, level, ""

    def write(self, buf):
        self.buf = buf.strip("\r\n\t ")

    def flush(self):
        self.logger.log(self.level, self.buf)


def _progress_lines(logger, n):
    """The progress records of the reference's per-point loop (`tqdm(range(N), file=TqdmToLogger(logger), mininterval=1)`,
    hddstream.py:218-220): the bar at 0 of N when the loop starts and at N of N when it ends, written by tqdm itself (the
    refreshes in between - one per second of the reference's loop - have no counterpart: the online phase is one call).
    Returns the function that closes the bar at `done` points.  Without tqdm installed nothing is logged."""
    try:
        from tqdm import tqdm
    except ImportError:
        return lambda done: None
    bar = tqdm(total=n, file=_TqdmToLogger(logger), mininterval=1)

    def finish(done):
        bar.update(done)
        bar.close()
    return finish


class HDDStream(object):
    def __init__(self, config, logger=None, device=0, tuning=None):
        self.config = config
        self.logger = logger if logger is not None else logging.getLogger("chronoclust_amd")
        self.pi = None
        self.mu = None
        self.omicron = None
        self.epsilon = float(config['epsilon'])
        self.epsilon_squared = self.epsilon ** 2
        self.upsilon = float(config['upsilon']) * self.epsilon
        self.delta = self._checked_delta()
        self.delta_squared = self.delta ** 2
        self.beta = float(config['beta'])
        self.k = float(config['k'])
        self.lambbda = float(config['lambda'])
        self.last_data_timestamp = 0
        self.dataset_dimensionality = 0
        self.dataset_size = 0

        self._h = _lib.Handle(device)
        if tuning:
            self._h.set_tuning(**tuning)
        self._X = None
        self.labels_uid = None   # per row of the last call: creation number of the MC that holds it
        self.labels_path = None
        self._tables = {}
        self._clusters = None
        self._cl_arrays = None
        self._uid_rows = None

    # ---- parameters -----------------------------------------------------------------------------

    def _checked_delta(self):
        delta = float(self.config['delta'])
        if delta > 1 or delta < 0:  # hddstream.py:148-149
            sys.exit("Given delta ({}) is out of range. Must be within 0-1.".format(delta))
        return delta

    def set_logger(self, logger):
        self.logger = logger

    def set_config(self, config):
        self.config = config

    def _set_dataset_dependent_parameters(self, input_dataset):
        n, d = input_dataset.shape
        self.dataset_dimensionality = d
        config_pi = float(self.config['pi'])
        self.pi = d if config_pi <= 0 else round(config_pi)
        self.omicron = self.config['omicron'] * self.dataset_size  # previous timepoint's size
        self.dataset_size = n
        self.mu = float(self.config['mu']) * self.dataset_size

    def _push_params(self):
        self._h.set_params(self.epsilon_squared, self.delta_squared, self.k, self.beta, float(self.mu),
                           float(self.omicron), self.upsilon, self.upsilon ** 2, self.delta, int(self.pi))

    # ---- the timestep ---------------------------------------------------------------------------

    def online_microcluster_maintenance(self, input_dataset, input_dataset_daystamp, reset_param=True,
                                        device_scaling=None):
        """device_scaling=(scale_, min_): `input_dataset` holds raw values and MinMaxScaler.transform
        (X * scale_ + min_) is applied on the device during the upload (scaling/scaler.py:39-41)."""
        X = np.ascontiguousarray(np.asarray(input_dataset, dtype=np.float64))
        if X.ndim != 2:
            raise ValueError("input_dataset must be 2-d [N, d]")
        if reset_param:
            self._set_dataset_dependent_parameters(X)
        self._push_params()
        log = self.logger
        log.info(f"Setting up online phase for timepoint {input_dataset_daystamp} with following params:\n"
                 f"Pcore density threshold factor(beta) = {self.beta}\n"
                 f"Decay rate(lambda) = {self.lambbda}\n"
                 f"Radius threshold(epsilon) = {self.epsilon}\n"
                 f"Max projected dimensionality(pi) = {self.pi}\n"
                 f"Density threshold(mu) = {self.mu} = {self.mu}\n"
                 f"Variance threshold(delta) = {self.delta}\n"
                 f"K = {self.k}\n"
                 f"PreDeCon epsilon(upsilon) = {self.upsilon}\n"
                 f"Outlier deletion point(omicron) = {self.omicron}\n")
        self._invalidate()
        if (self.last_data_timestamp - input_dataset_daystamp) != 0:
            log.info("Decaying and downgrading microclusters")
            interval = input_dataset_daystamp - self.last_data_timestamp
            self._h.decay_downgrade(2 ** (-self.lambbda * interval))  # hddstream.py:283

        log.info("Starting online microcluster maintenance for timepoint {}".format(input_dataset_daystamp))
        progress = _progress_lines(log, X.shape[0])  # hddstream.py:218-220: tqdm's bar, written to the log
        self._X = X if device_scaling is None else None  # scaled values are fetched from the device on demand
        self._n_points = X.shape[0]
        try:
            if X.shape[0] > 0 and device_scaling is not None:
                self._h.points_upload_scaled(X, device_scaling[0], device_scaling[1])
                self._h.online_run()
                self.labels_uid, self.labels_path = self._h.labels_download()
            elif X.shape[0] > 0:
                self.labels_uid, self.labels_path = self._h.online(X)
            else:
                self.labels_uid, self.labels_path = np.empty(0, np.int64), np.empty(0, np.int8)
        except BaseException:
            # (the reference's loop dies with its bar at the point that raised: the bar is closed where it stands - at 0,
            # the online phase being one call - instead of being left to tqdm's destructor, which would write a stray
            # record into the log at some later moment)
            progress(0)
            raise
        progress(X.shape[0])
        log.info("Finish online microcluster maintenance for timepoint {}".format(input_dataset_daystamp))
        log.info("Online maintenance yield {} pcores and {} outlier".format(
            self._h.count(_lib.PCORE), self._h.count(_lib.OUTLIER)))
        self.last_data_timestamp = input_dataset_daystamp
        self.offline_clustering(input_dataset_daystamp)

    def prefetch(self, input_dataset, device_scaling=None):
        """Starts uploading the NEXT timepoint in the background (cc_points_prefetch) while the caller still works on the
        current one.  Returns the array to pass to online_microcluster_maintenance (the same buffer, so that the upload
        is recognised); results never depend on whether a timepoint was prefetched.  The returned array must not be
        modified before it is passed on: the device copy made here is the one that gets clustered."""
        X = np.ascontiguousarray(np.asarray(input_dataset, dtype=np.float64))
        if X.ndim == 2 and X.shape[0] > 0:
            if device_scaling is None:
                self._h.points_prefetch(X)
            else:
                self._h.points_prefetch(X, device_scaling[0], device_scaling[1])
        return X

    def offline_clustering(self, dataset_daystamp):
        self._push_params()
        self._cl_arrays, _ = self._h.offline_arrays()  # (members, offsets, w, cf1, cf2, cen, pref): all clusters
        self._clusters = None                          # ClusterView objects are built on demand (final_clusters)
        num_core = self._h.num_core()
        num_pcore = self._h.count(_lib.PCORE) - num_core
        self.logger.info(f'Starting offline clustering with {num_core} core clusters and {num_pcore} pcore clusters.')
        self.logger.info('Finish offline clustering for dataset with timepoint: {}'.format(dataset_daystamp))
        self.logger.info("Offline clustering yield {} clusters.".format(len(self._cl_arrays[2])))

    # ---- results ---------------------------------------------------------------------------------

    def _invalidate(self):
        self._tables = {}
        self._clusters = None
        self._cl_arrays = None
        self._uid_rows = None

    def table(self, kind):
        """dict of arrays (id, uid, w, cf1, cf2, cen, pref) for the pcore (0) or outlier (1) list, list order."""
        if kind not in self._tables:
            self._tables[kind] = self._h.export(kind)
        return self._tables[kind]

    def _views(self, kind):
        t = self.table(kind)
        return [MicroclusterView(t["id"][i], t["cf1"][i], t["cf2"][i], t["w"][i], t["cen"][i], t["pref"][i],
                                 t["uid"][i], owner=self) for i in range(len(t["id"]))]

    @property
    def pcore_MC(self):
        return self._views(_lib.PCORE)

    @property
    def outlier_MC(self):
        return self._views(_lib.OUTLIER)

    @property
    def final_clusters(self):
        """hddstream.py:508: the merged PredeconMC objects of the offline phase (views over the exported arrays)."""
        if self._clusters is None:
            if getattr(self, "_cl_arrays", None) is None:
                return []
            mem, off, w, cf1, cf2, cen, pref = self._cl_arrays
            self._clusters = [ClusterView(mem[off[c]:off[c + 1]], w[c], cf1[c], cf2[c], cen[c], pref[c])
                              for c in range(len(w))]
        return self._clusters

    def point_cluster_index(self):
        """For every row of the last timepoint the index (into final_clusters / cluster_records order) of the cluster
        its microcluster belongs to, or -1 for points in outlier MCs and in pcore MCs outside every cluster (what
        app.py:303-332 derives from the per-MC `points` dicts; written as the literal None there)."""
        if self.labels_uid is None or getattr(self, "_cl_arrays", None) is None:
            return np.empty(0, np.int64)
        if not getattr(self, "_n_points", 0):
            return np.empty(0, np.int64)
        # a gather on the device: per-point labels -> creation number -> cluster (cc_point_clusters);
        # chronoclust_amd.multi.point_cluster_index is the same join in numpy (tests compare the two)
        return self._h.point_clusters().astype(np.int64)

    def cluster_records(self):
        """The tracking-side records of this timepoint's clusters: what app.py:181-190 builds one by one -
        Cluster(list(id_set), centroid, weight rounded to one decimal place, preferred dimensions) plus the member
        pcores - assembled from the exported arrays (member centroids / preferred dimensions are gathered from the
        pcore table in one indexing operation instead of one object per pcore)."""
        from ..objects.cluster import Cluster, no_gc_pauses
        if getattr(self, "_cl_arrays", None) is None:
            return []
        with no_gc_pauses():
            return self._cluster_records(Cluster)

    def _cluster_records(self, Cluster):
        mem, off, w, _, _, cen, pref = self._cl_arrays
        pc = self.table(_lib.PCORE)
        order = np.argsort(pc["id"], kind="stable")
        pos = order[np.searchsorted(pc["id"][order], mem)] if len(mem) else np.empty(0, np.int64)
        weights = rounded_weights(w)
        mem_list, off_list = mem.tolist(), off.tolist()
        base = (pc["cen"][pos], pc["pref"][pos], pc["uid"][pos])  # member pcores of all clusters, merge order
        cen_rows, pref_rows = list(cen), list(pref)
        new_cluster = Cluster.__new__
        out = []
        for c in range(len(w)):
            a, b = off_list[c], off_list[c + 1]
            merged = mem_list[a:b]
            if b - a == 1:
                ids, rows = merged, range(a, b)
            else:
                id_set = set()
                for m in merged:  # predecon_mc.py:67: the set is filled in merge order (CPython iteration order depends on it)
                    id_set.add(m)
                ids = list(id_set)  # app.py:186
                where = {m: a + i for i, m in enumerate(merged)}
                rows = [where[m] for m in ids]  # the member rows in pcore_ids order
            # Cluster(ids, centroid, weight, preferred dimensions) + set_pcore_rows(base, rows), the fields filled in one
            # go (5 000 constructor calls cost as much as the offline phase they follow)
            cl = new_cluster(Cluster)
            cl.__dict__ = {"pcore_ids": ids, "id": set(), "parents": set(), "centroid": cen_rows[c],
                           "cumulative_weight": weights[c], "preferred_dimensions": pref_rows[c], "_pcore_objects": None,
                           "_pc_cen": None, "_pc_pref": None, "_pc_uid": None, "_pc_base": base, "_pc_rows": rows,
                           "historical_associates": set(), "historical_associates_pcores": set(),
                           "offline_index": c}  # position among final_clusters (what point_cluster_index returns)
            out.append(cl)
        return out

    @property
    def pcore_MC_last_id(self):
        return self._h.counters()[0]

    @property
    def outlier_MC_last_id(self):
        return self._h.counters()[1]

    def resident_points(self, scale=None, min_=None):
        """The points of the last timepoint as the device holds them; with scale / min_ the inverse transform
        (X - min_) / scale_ is applied on the device first (scaling/scaler.py:43-44)."""
        if not getattr(self, "_n_points", 0):
            return np.empty((0, self.dataset_dimensionality or 0))
        return self._h.points_download(self.dataset_dimensionality, scale, min_)

    def _points_of(self, uid):
        if self.labels_uid is not None and self._X is None and getattr(self, "_n_points", 0):
            self._X = self.resident_points()
        if self.labels_uid is None or self._X is None:
            return {}
        if self._uid_rows is None:
            order = np.argsort(self.labels_uid, kind="stable")
            keys, starts = np.unique(self.labels_uid[order], return_index=True)
            bounds = np.append(starts, len(order))
            self._uid_rows = {int(k): order[bounds[i]:bounds[i + 1]] for i, k in enumerate(keys)}
        rows = self._uid_rows.get(int(uid))
        if rows is None:
            return {}
        return {int(r): self._X[r].tolist() for r in rows}

    def stats(self):
        return self._h.stats()

    # ---- checkpoint (what chronoclust/app.py:402-465 pickles; here plain arrays + the id counters the
    # reference's __getstate__ forgets, hddstream.py:69-81) -------------------------------------------------

    def get_state(self):
        state = {"last_data_timestamp": self.last_data_timestamp, "dataset_size": self.dataset_size,
                 "dataset_dimensionality": self.dataset_dimensionality,
                 "pcore_MC_last_id": self.pcore_MC_last_id, "outlier_MC_last_id": self.outlier_MC_last_id}
        for kind, name in ((_lib.PCORE, "pcore"), (_lib.OUTLIER, "outlier")):
            for key, val in self.table(kind).items():
                state["%s_%s" % (name, key)] = val
        return state

    def set_state(self, state):
        self.last_data_timestamp = int(state["last_data_timestamp"])
        self.dataset_size = int(state["dataset_size"])
        self.dataset_dimensionality = int(state["dataset_dimensionality"])
        d = self.dataset_dimensionality
        self._h.reset()
        # k must be known before rows arrive so that their preferred-dimension entries are recognised
        self._h.set_params(self.epsilon_squared, self.delta_squared, self.k, self.beta, 0.0, 0.0, self.upsilon,
                           self.upsilon ** 2, self.delta, max(d, 1))
        for kind, name in ((_lib.PCORE, "pcore"), (_lib.OUTLIER, "outlier")):
            if len(state["%s_id" % name]):  # the whole list in one upload (cc_inject_bulk)
                self._h.inject_bulk(kind, state["%s_cf1" % name].reshape(-1, d), state["%s_cf2" % name].reshape(-1, d),
                                    state["%s_cen" % name].reshape(-1, d), state["%s_pref" % name].reshape(-1, d),
                                    state["%s_w" % name], state["%s_id" % name], state["%s_uid" % name])
        self._h.set_counters(state["pcore_MC_last_id"], state["outlier_MC_last_id"])
        self._invalidate()
