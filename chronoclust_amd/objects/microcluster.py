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
