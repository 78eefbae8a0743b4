"""Tracking-side cluster record: the interface of chronoclust/objects/cluster.py:5-115.

Differences in representation only: member pcores are kept as small snapshots (id, centroid, preferred
dimensions) taken when the cluster is built, instead of deep copies of whole Microcluster objects with all
their points (cluster.py:57)."""


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
        self.pcore_objects = []
        self.historical_associates = set()
        self.historical_associates_pcores = set()

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
    def add_pcore_objects(self, pcore_id_to_object):
        for pcore_id in self.pcore_ids:
            src = pcore_id_to_object[pcore_id]
            self.pcore_objects.append(PcoreSnapshot(pcore_id, src.cluster_centroids, src.preferred_dimension_vector,
                                                    getattr(src, "prev_outlier_id", None)))

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
