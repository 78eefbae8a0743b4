"""Cluster tracking: the interface of chronoclust/tracking/cluster_tracker.py.

TrackByLineage is integer / string logic on pcore ids and stays on the host.
TrackByHistoricalAssociation's distance argmin (cluster_tracker.py:127-141) is the K9 kernel behind
cc_assoc_argmin: one launch for all current pcores against all previous pcores."""
import string
from collections import deque

import numpy as np


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
        children = self.child_clusters
        if all(c.cumulative_weight is not None for c in children):
            children.sort(key=lambda c: c.cumulative_weight)  # stable; also the row order of result.csv

        parent_of_pcore = self.get_parent_pcore_to_id()
        offspring = {}
        for cluster in children:
            cluster.set_parents(parent_pcores_to_id=parent_of_pcore)
            if not cluster.parents:
                cluster.add_parent(id=self.get_new_letter())
            for parent in cluster.get_parents():
                offspring.setdefault(parent, []).append(cluster)

        # per parent: the child holding most of its pcores keeps the label, the others become parent|n with a
        # split counter that persists over timepoints (cluster_tracker.py:51-70)
        for parent, kids in offspring.items():
            ranked = sorted(kids, key=lambda c: len(c.pcore_ids), reverse=True)
            splits = self.split_per_id.get(parent, 0)
            for rank, child in enumerate(ranked):
                if rank == 0:
                    child.add_id(parent)
                else:
                    splits += 1
                    child.add_id('{}|{}'.format(parent, splits))
            self.split_per_id[parent] = splits
        self.assign_child_id()

    def assign_child_id(self):
        # one label: itself; several (a merge): nested parentheses over the sorted labels (:88-109)
        for child in self.child_clusters:
            labels = sorted(child.id)
            text = labels[0]
            for extra in labels[1:]:
                text = '(' + text + ',' + extra + ')'
            child.id = text

    def transfer_child_to_parent(self):
        self.parent_clusters = self.child_clusters
        self.child_clusters = []


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
        if len(self.previous_timepoint_clusters) == 0:
            for cluster in self.current_clusters:
                cluster.add_historical_associate(None)
            return
        prev_owner, prev_pcore, prev_cen = [], [], []
        for pc in self.previous_timepoint_clusters:
            ids, cen, _ = pc.pcore_arrays()
            if ids:
                prev_owner += [pc.id] * len(ids)
                prev_pcore += ids
                prev_cen.append(cen)
        cur_owner, cur_cen, cur_pref = [], [], []
        for cl in self.current_clusters:
            ids, cen, pref = cl.pcore_arrays()
            if ids:
                cur_owner += [cl] * len(ids)
                cur_cen.append(cen)
                cur_pref.append(pref)
        if not cur_owner:
            return
        if prev_cen:
            idx, _ = self._hip().assoc_argmin(np.concatenate(cur_cen), np.concatenate(cur_pref),
                                              np.concatenate(prev_cen))
            idx = idx.tolist()
        else:
            idx = [-1] * len(cur_owner)
        for cluster, i in zip(cur_owner, idx):
            if i >= 0:
                cluster.historical_associates.add(prev_owner[i])          # Cluster.add_historical_associate
                cluster.historical_associates_pcores.add(prev_pcore[i])   # ....add_historical_associate_pcore([id])
            else:
                cluster.historical_associates.add(None)

    def transfer_current_to_previous(self):
        self.previous_timepoint_clusters = self.current_clusters
        self.current_clusters = []
