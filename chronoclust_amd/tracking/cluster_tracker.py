"""Cluster tracking: the interface of chronoclust/tracking/cluster_tracker.py.

TrackByLineage is integer / string logic on pcore ids and stays on the host.
TrackByHistoricalAssociation's distance argmin (cluster_tracker.py:127-141) is the K9 kernel behind
cc_assoc_argmin: one launch for all current pcores against all previous pcores."""
import string
from collections import deque

import numpy as np

from ..objects.cluster import no_gc_pauses


class TrackByLineage(object):
    def __init__(self):
        self.child_clusters = []
        self.parent_clusters = []
        self._letter_len = 1
        self.letters = deque(string.ascii_uppercase)
        self.split_per_id = {}

    # letters: A..Z, AA..ZZ, AAA.. (cluster_tracker.py:14-24)
    def get_new_letter(self):
        letter = self.letters.popleft()
        if not self.letters:
            self._letter_len = len(letter) + 1
            self.letters = deque(ch * self._letter_len for ch in string.ascii_uppercase)
        return letter

    def add_new_child_cluster(self, cluster):
        self.child_clusters.append(cluster)

    def get_parent_pcore_to_id(self):
        mapping = {}
        for parent in self.parent_clusters:
            for pcore in parent.pcore_ids:
                mapping[pcore] = parent.id
        return mapping

    def calculate_ids(self):
        with no_gc_pauses():
            self._calculate_ids()

    def _calculate_ids(self):
        children = self.child_clusters
        if all(c.cumulative_weight is not None for c in children):
            children.sort(key=lambda c: c.cumulative_weight)  # stable; also the row order of result.csv

        parent_of_pcore = self.get_parent_pcore_to_id()
        offspring = {}
        for cluster in children:
            parents = cluster.parents
            for pcore in cluster.pcore_ids:  # Cluster.set_parents
                label = parent_of_pcore.get(pcore)
                if label is not None:
                    parents.add(label)
            if not parents:
                parents.add(self.get_new_letter())
            for parent in parents:
                offspring.setdefault(parent, []).append(cluster)

        # per parent: the child holding most of its pcores keeps the label, the others become parent|n with a
        # split counter that persists over timepoints (cluster_tracker.py:51-70)
        split_per_id = self.split_per_id
        for parent, kids in offspring.items():
            if len(kids) == 1:  # (the common case: nothing to rank, the counter keeps its value)
                kids[0].id.add(parent)
                split_per_id.setdefault(parent, 0)
                continue
            ranked = sorted(kids, key=lambda c: len(c.pcore_ids), reverse=True)
            splits = split_per_id.get(parent, 0)
            for rank, child in enumerate(ranked):
                if rank == 0:
                    child.add_id(parent)
                else:
                    splits += 1
                    child.add_id('{}|{}'.format(parent, splits))
            split_per_id[parent] = splits
        self.assign_child_id()

    def assign_child_id(self):
        # one label: itself; several (a merge): nested parentheses over the sorted labels (:88-109)
        for child in self.child_clusters:
            if len(child.id) == 1:
                (child.id,) = child.id
                continue
            labels = sorted(child.id)
            text = labels[0]
            for extra in labels[1:]:
                text = '(' + text + ',' + extra + ')'
            child.id = text

    def transfer_child_to_parent(self):
        self.parent_clusters = self.child_clusters
        self.child_clusters = []


def _pcores_of(clusters, want_pref):
    """The member pcores of `clusters`, cluster by cluster in pcore_ids order: (owning cluster per pcore, pcore ids,
    centroids [n, d], preferred dimensions [n, d] or None).  Clusters built by HDDStream.cluster_records share one set of
    arrays per timepoint (Cluster.set_pcore_rows): those are indexed once instead of being concatenated piece by piece."""
    owner, ids = [], []
    base = clusters[0]._pc_base if clusters else None
    shared = base is not None and all(c._pc_base is base and c._pcore_objects is None for c in clusters)
    if shared:
        rows = []
        for c in clusters:
            owner += [c] * len(c.pcore_ids)
            ids += c.pcore_ids
            rows += c._pc_rows
        rows = np.asarray(rows, dtype=np.int64)
        contiguous = len(rows) == len(base[0]) and (len(rows) == 0 or bool((rows == np.arange(len(rows))).all()))
        cen = base[0] if contiguous else base[0][rows]
        pref = (base[1] if contiguous else base[1][rows]) if want_pref else None
        return owner, ids, cen, pref
    cen, pref = [], []
    for c in clusters:
        pid, ce, pr = c.pcore_arrays()
        if pid:
            owner += [c] * len(pid)
            ids += pid
            cen.append(ce)
            pref.append(pr)
    if not owner:
        return owner, ids, np.empty((0, 0)), np.empty((0, 0))
    return owner, ids, np.concatenate(cen), (np.concatenate(pref) if want_pref else None)


class TrackByHistoricalAssociation(object):
    def __init__(self, handle=None):
        self.current_clusters = []
        self.previous_timepoint_clusters = []
        self._handle = handle

    def _hip(self):
        if self._handle is None:
            from .. import _lib
            self._handle = _lib.Handle(0)
        return self._handle

    def __getstate__(self):  # the GPU handle is not part of a checkpoint
        return {"current_clusters": self.current_clusters,
                "previous_timepoint_clusters": self.previous_timepoint_clusters}

    def __setstate__(self, state):
        self.current_clusters = state["current_clusters"]
        self.previous_timepoint_clusters = state["previous_timepoint_clusters"]
        self._handle = None

    def set_current_clusters(self, clusters):
        self.current_clusters = clusters

    def track_cluster_history(self):
        with no_gc_pauses():
            self._track_cluster_history()

    def _track_cluster_history(self):
        if len(self.previous_timepoint_clusters) == 0:
            for cluster in self.current_clusters:
                cluster.add_historical_associate(None)
            return
        prev_owner, prev_pcore, prev_cen = _pcores_of(self.previous_timepoint_clusters, want_pref=False)[:3]
        cur_owner, _, cur_cen, cur_pref = _pcores_of(self.current_clusters, want_pref=True)
        if not cur_owner:
            return
        if len(prev_owner):
            idx, _ = self._hip().assoc_argmin(cur_cen, cur_pref, prev_cen)
            idx = idx.tolist()
        else:
            idx = [-1] * len(cur_owner)
        for cluster, i in zip(cur_owner, idx):
            if i >= 0:
                cluster.historical_associates.add(prev_owner[i].id)      # Cluster.add_historical_associate
                cluster.historical_associates_pcores.add(prev_pcore[i])  # ....add_historical_associate_pcore([id])
            else:
                cluster.historical_associates.add(None)

    def transfer_current_to_previous(self):
        self.previous_timepoint_clusters = self.current_clusters
        self.current_clusters = []
