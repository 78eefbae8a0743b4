"""Read-only views of device-resident microclusters, shaped like the reference's objects
(chronoclust/objects/microcluster.py:18-81, predecon_mc.py:5-42) so code written against
`HDDStream.pcore_MC / outlier_MC / final_clusters` keeps working."""
import numpy as np


class MicroclusterView(object):
    """One row of the HBM microcluster table, copied to the host after a timestep."""

    __slots__ = ("id", "CF1", "CF2", "cumulative_weight", "cluster_centroids", "preferred_dimension_vector",
                 "prev_outlier_id", "prev_pcore_id", "_owner")

    def __init__(self, id, cf1, cf2, weight, centroid, pref, uid, owner=None):
        self.id = [int(id)]  # the reference keeps a one-element list / set (hddstream.py:428, 450-451)
        self.CF1 = cf1
        self.CF2 = cf2
        self.cumulative_weight = float(weight)
        self.cluster_centroids = centroid
        self.preferred_dimension_vector = pref
        self.prev_outlier_id = int(uid)
        self.prev_pcore_id = None  # never set by the reference either (microcluster.py:86 has no callers)
        self._owner = owner

    @property
    def points(self):
        """row index -> list of feature values for the rows of the current timestep held by this MC
        (microcluster.py:149).  Materialised on demand from the per-point label array."""
        if self._owner is None:
            return {}
        return self._owner._points_of(self.prev_outlier_id)

    def get_projected_dist_to_point(self, other_point):
        # microcluster.py:167-181 (host convenience for small inputs; the hot paths never call this)
        dist = 0.0
        for c, p, w in zip(self.cluster_centroids, np.asarray(other_point, dtype=np.float64),
                           self.preferred_dimension_vector):
            t = p - c
            dist = dist + (t * t) / w
        return dist


def _ordered_sum(terms):
    """Left to right in double precision: what numba's np.sum does, not numpy's pairwise sum."""
    acc = 0.0
    for t in terms:
        acc = acc + float(t)
    return acc


class Microcluster(MicroclusterView):
    """The reference's mutable microcluster object (objects/microcluster.py:18-257) for code - and tests - written against
    it: same constructor, attributes and methods, the same operations in the same order (utilities/mc_functions.py:14-77).
    A host-side convenience type for a handful of vectors: the online phase never builds one (the table lives in HBM and
    its rows come back as read-only `MicroclusterView`s); nothing here is a fallback for a device path."""

    __slots__ = ("creation_time_in_hrs", "_points")

    def __init__(self, cf1, cf2, id=None, cumulative_weight=0, preferred_dimension_vector=None, cluster_centroids=None,
                 creation_time_in_hrs=0):
        self.id = set() if id is None else id
        self.CF1 = cf1
        self.CF2 = cf2
        self.cumulative_weight = cumulative_weight
        self.preferred_dimension_vector = preferred_dimension_vector
        self.cluster_centroids = cluster_centroids
        self.creation_time_in_hrs = creation_time_in_hrs
        self.prev_pcore_id = None
        self.prev_outlier_id = None
        self._owner = None
        self._points = {}

    @property
    def points(self):
        return self._points

    @points.setter
    def points(self, value):
        self._points = value

    def update_prev_outlier_id(self, outlier_id):
        self.prev_outlier_id = outlier_id

    def update_prev_pcore_id(self, pcore_id):
        self.prev_pcore_id = pcore_id

    def _squared_variance(self):
        w = self.cumulative_weight  # mc_functions.py:14-22: CF2 / W - (CF1 / W)^2
        cf1, cf2 = np.asarray(self.CF1, dtype=np.float64), np.asarray(self.CF2, dtype=np.float64)
        q = cf1 / w
        return cf2 / w - q * q

    def update_preferred_dimensions(self, variance_threshold_squared, k_constant):
        self.preferred_dimension_vector = np.array(
            [k_constant if v <= variance_threshold_squared else 1.0 for v in self._squared_variance()])

    def add_new_point(self, new_point_values, new_point_timestamp, new_point_idx, new_point_weight=1, update_centroid=True):
        x = np.asarray(new_point_values, dtype=np.float64)
        self.CF1 = np.asarray(self.CF1, dtype=np.float64) + x            # mc_functions.py:24-29
        self.CF2 = np.asarray(self.CF2, dtype=np.float64) + x * x
        self.cumulative_weight += new_point_weight
        self._points[new_point_idx] = x.tolist()
        if update_centroid:
            self.set_centroid()

    def set_centroid(self):
        self.cluster_centroids = np.asarray(self.CF1, dtype=np.float64) / self.cumulative_weight  # mc_functions.py:31-33

    def get_projected_dist_to_point(self, other_point):
        c = np.asarray(self.cluster_centroids, dtype=np.float64)
        w = np.asarray(self.preferred_dimension_vector, dtype=np.float64)
        t = np.asarray(other_point, dtype=np.float64) - c
        return _ordered_sum((t * t) / w)                                   # mc_functions.py:35-43

    def calculate_projected_radius_squared(self):
        w = np.asarray(self.preferred_dimension_vector, dtype=np.float64)
        return _ordered_sum(self._squared_variance() / w)                  # mc_functions.py:45-56

    def get_copy(self):
        return Microcluster(cf1=np.array(self.CF1, dtype=np.float64), cf2=np.array(self.CF2, dtype=np.float64),
                            cumulative_weight=self.cumulative_weight)

    def get_copy_with_new_point(self, datapoint, variance_threshold_squared, k_constant):
        clone = self.get_copy()
        clone.add_new_point(datapoint, -1, -1)
        clone.update_preferred_dimensions(variance_threshold_squared, k_constant)
        return clone

    def is_core(self, radius_threshold_squared, density_threshold, max_subspace_dimensionality):
        n_pref = int(np.count_nonzero(np.asarray(self.preferred_dimension_vector, dtype=np.float64) > 1))
        return bool(self.calculate_projected_radius_squared() <= radius_threshold_squared
                    and self.cumulative_weight >= density_threshold
                    and n_pref <= max_subspace_dimensionality)             # mc_functions.py:64-77

    def reset_points(self):
        self._points = {}


class ClusterView(object):
    """A final cluster of the offline phase (hddstream.py:508): `id` is the Python set of member pcore ids,
    filled in the reference's merge order because CPython set iteration order depends on it."""

    __slots__ = ("id", "members_in_merge_order", "CF1", "CF2", "cumulative_weight", "cluster_centroids",
                 "preferred_dimension_vector")

    def __init__(self, members, weight, cf1, cf2, centroid, pref):
        self.members_in_merge_order = [int(m) for m in members]
        s = set()
        for m in self.members_in_merge_order:
            s.add(m)  # predecon_mc.py:67
        self.id = s
        self.CF1, self.CF2 = cf1, cf2
        self.cumulative_weight = float(weight)
        self.cluster_centroids = centroid
        self.preferred_dimension_vector = pref
