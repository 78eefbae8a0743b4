"""Tracking-side cluster record: the interface of chronoclust/objects/cluster.py:5-115.

Differences in representation only: member pcores are kept as small snapshots (id, centroid, preferred
dimensions) taken when the cluster is built, instead of deep copies of whole Microcluster objects with all
their points (cluster.py:57)."""


import contextlib
import gc


@contextlib.contextmanager
def no_gc_pauses():
    """Building the records of a timepoint allocates tens of thousands of small containers (sets, dicts, lists) that all
    stay alive; the cyclic collector's threshold-triggered passes over them (2-20 ms each at 5 000 clusters) find nothing
    to free.  Collection is suspended for the duration and left as it was afterwards."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class PcoreSnapshot(object):
    """What the trackers need of a member pcore, frozen at the timepoint the cluster was formed."""

    __slots__ = ("id", "cluster_centroids", "preferred_dimension_vector", "prev_outlier_id")

    def __init__(self, id, centroid, pref, uid=None):
        self.id = [int(id)]
        self.cluster_centroids = centroid
        self.preferred_dimension_vector = pref
        self.prev_outlier_id = uid


class Cluster(object):
    def __init__(self, pcore_ids, cluster_centroid=None, cumulative_weight=None, preferred_dimensions=None):
        self.pcore_ids = pcore_ids
        self.id = set()          # becomes the lineage string after TrackByLineage.assign_child_id
        self.parents = set()
        self.centroid = cluster_centroid
        self.cumulative_weight = cumulative_weight
        self.preferred_dimensions = preferred_dimensions
        self._pcore_objects = None   # list of PcoreSnapshot, built on first access of `pcore_objects`
        self._pc_cen = None          # [members, d] centroids / preferred dimensions of the member pcores as arrays
        self._pc_pref = None         # (set_pcore_arrays: the vectorised form of add_pcore_objects)
        self._pc_uid = None
        self._pc_base = None         # set_pcore_rows: (centroids, preferred dimensions, uids) shared by all clusters of a
        self._pc_rows = None         # timepoint + this cluster's rows of them (a range, or a list in pcore_ids order)
        self.historical_associates = set()
        self.historical_associates_pcores = set()
        self.offline_index = None    # position among HDDStream.final_clusters (set by HDDStream.cluster_records)

    # -- lineage side (cluster.py:35-47) ------------------------------------------------------
    def add_id(self, id):
        self.id.add(id)

    def add_parent(self, id):
        self.parents.add(id)

    def set_parents(self, parent_pcores_to_id):
        for pcore in self.pcore_ids:
            if pcore in parent_pcores_to_id:
                self.parents.add(parent_pcores_to_id[pcore])

    def get_parents(self):
        return self.parents

    # -- association side (cluster.py:50-76) --------------------------------------------------
    @property
    def pcore_objects(self):
        """The member pcores as objects (cluster.py:57 keeps deep copies of whole Microclusters); materialised from
        the arrays of set_pcore_arrays on first use."""
        if self._pcore_objects is None:
            self._pcore_objects = []
            self._materialise_rows()
            if self._pc_cen is not None:
                for i, pcore_id in enumerate(self.pcore_ids):
                    self._pcore_objects.append(PcoreSnapshot(pcore_id, self._pc_cen[i], self._pc_pref[i],
                                                             None if self._pc_uid is None else int(self._pc_uid[i])))
        return self._pcore_objects

    def add_pcore_objects(self, pcore_id_to_object):
        for pcore_id in self.pcore_ids:
            src = pcore_id_to_object[pcore_id]
            self.pcore_objects.append(PcoreSnapshot(pcore_id, src.cluster_centroids, src.preferred_dimension_vector,
                                                    getattr(src, "prev_outlier_id", None)))

    def set_pcore_arrays(self, centroids, preferred_dimensions, uids=None):
        """add_pcore_objects from arrays: row i belongs to pcore_ids[i]."""
        self._pc_cen, self._pc_pref, self._pc_uid = centroids, preferred_dimensions, uids
        self._pc_base = self._pc_rows = None
        self._pcore_objects = None

    def set_pcore_rows(self, base, rows):
        """add_pcore_objects without touching any array: `base` = (centroids, preferred dimensions, uids) of all member
        pcores of the timepoint's clusters, `rows` = this cluster's rows of them in pcore_ids order (a range or a list).
        The per-cluster arrays are sliced on demand; the association tracker reads `base` directly."""
        self._pc_base, self._pc_rows = base, rows
        self._pc_cen = self._pc_pref = self._pc_uid = None
        self._pcore_objects = None

    def _materialise_rows(self):
        if self._pc_cen is None and self._pc_base is not None:
            r = self._pc_rows
            idx = slice(r.start, r.stop) if isinstance(r, range) else r
            self._pc_cen, self._pc_pref, self._pc_uid = (self._pc_base[0][idx], self._pc_base[1][idx], self._pc_base[2][idx])

    def pcore_arrays(self):
        """(ids, centroids [n, d], preferred dimensions [n, d]) of the member pcores, in pcore_ids order."""
        import numpy as np
        if self._pcore_objects is None:
            self._materialise_rows()
        if self._pcore_objects is None and self._pc_cen is not None:
            return list(self.pcore_ids), self._pc_cen, self._pc_pref
        objs = self.pcore_objects
        if not objs:
            return [], np.empty((0, 0)), np.empty((0, 0))
        return ([p.id[0] for p in objs], np.array([np.asarray(p.cluster_centroids, dtype=np.float64) for p in objs]),
                np.array([np.asarray(p.preferred_dimension_vector, dtype=np.float64) for p in objs]))

    def __getstate__(self):
        self._materialise_rows()  # a checkpoint holds this cluster's rows, not the whole timepoint's arrays
        state = dict(self.__dict__)
        state["_pc_base"] = state["_pc_rows"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("_pc_base", None)
        self.__dict__.setdefault("_pc_rows", None)
        if "pcore_objects" in state:  # images written before the arrays existed
            self._pcore_objects = self.__dict__.pop("pcore_objects")
            self._pc_cen = self._pc_pref = self._pc_uid = None

    def add_historical_associate(self, associate):
        self.historical_associates.add(associate)

    def add_historical_associate_pcore(self, pcore_id):
        self.historical_associates_pcores.update(pcore_id)

    # -- result.csv string formats (cluster.py:78-92) -----------------------------------------
    def get_historical_associates_as_str(self):
        return '&'.join(str(s) for s in sorted(self.historical_associates))

    def get_historical_associates_pcore_as_str(self):
        return '&'.join(str(s) for s in self.historical_associates_pcores)

    def get_preferred_dimensions_as_str(self):
        return ';'.join(str(s) for s in self.preferred_dimensions)

    def get_pcore_ids_as_str(self):
        return '|'.join(str(s) for s in self.pcore_ids)

    # -- gating labels (cluster.py:94-115); a handful of clusters x gates per timepoint: host arithmetic --
    def get_projected_dist_to_point(self, other_point):
        dist = 0.0
        for c_i, p_i, d_i in zip(self.centroid, other_point, self.preferred_dimensions):
            dist += ((float(p_i) - float(c_i)) ** 2) / float(d_i)
        return dist

    def get_dist_to_point(self, other_point):
        dist = 0.0
        for i, c in enumerate(self.centroid):
            dist += (float(other_point[i]) - float(c)) ** 2
        return dist
