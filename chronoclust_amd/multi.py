"""Multi-GPU layouts of the hot path.  One process per GPU in both.

1. Independent streams ("replicas"): N GPUs cluster N event streams (samples, patients), one per rank, with no
   data-path collective.  The online phase of ONE stream is a sequential chain over its points
   (hddstream.py:220-237), so events of a stream are never sharded.
2. One stream on N GPUs, exact: every rank holds the whole microcluster table and
   receives the same calls; the snapshot scan of a window is split by table rows and the ranks all-gather one
   candidate record per window point over RCCL (cc_comm_init_rccl), the offline and association pair matrices
   are split by rows.  All ranks end with bit-identical results.  `join_stream_group` sets this up.

The host-side group (`dist` below) is plumbing: the barrier / max-over-ranks time of bench.py and the channel that
carries the 128-byte RCCL id from rank 0 to the other ranks.  It is a `chronoclust_amd.rendezvous.HostGroup` (plain
TCP on 127.0.0.1, no torch; what bench.py uses) or, for callers that already run under it, `torch.distributed` -
the helpers below only use what both offer.  The collectives of the data path are RCCL calls made by the C-ABI
library on its own HIP streams."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def stream_seed(base_seed, rank):
    """Layout 1: every rank clusters its own synthetic stream (same shape, different seed)."""
    return int(base_seed) + int(rank)


def max_over_ranks(value, dist=None):
    """The job's step time is the slowest rank's.  `dist`: a HostGroup, any object with torch.distributed's
    get_world_size / all_gather_object (the gloo tests pass that module; nothing here imports it), or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if hasattr(dist, "max_float"):
        return dist.max_float(value)
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, float(value))
    return max(float(v) for v in box)


def whole_job_rate(points_per_rank, steps, world, seconds):
    """value of bench.py, layout 1: units all ranks processed / max-over-ranks time."""
    return world * points_per_rank * steps / seconds


def one_stream_rate(points, steps, seconds):
    """Layout 2: the ranks share ONE stream, so the job processed `points` per step whatever the rank count."""
    return points * steps / seconds


def broadcast_bytes(payload, dist, src=0):
    """`payload` (bytes on rank `src`, ignored elsewhere) to every rank, through the host-side group."""
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def join_stream_group(handle, dist):
    """Layout 2: makes `handle` (a chronoclust_amd._lib.Handle on this rank's GPU) a member of the RCCL group of
    all ranks of `dist`.  Collective.  Afterwards online_run / offline / assoc_argmin are collective calls."""
    from . import _lib
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = broadcast_bytes(_lib.comm_unique_id() if rank == 0 else None, dist)
    handle.comm_init_rccl(uid, rank, world)
    return rank, world


def all_ranks_equal(digest, dist):
    """True when every rank passes the same bytes (e.g. a hash of its labels and tables)."""
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, digest)
    return all(b == box[0] for b in box)


def label_agreement(labels, reference_labels):
    """Agreement of two clusterings of the same points that may name their clusters differently: every cluster of
    `labels` is matched to the cluster of `reference_labels` it shares most points with; returns the fraction of points
    that fall into their cluster's match.  1.0 for identical partitions (whatever the ids), used to report how far the
    relaxed multi-GPU mode is from the exact path."""
    import numpy as np
    a = np.asarray(labels)
    b = np.asarray(reference_labels)
    if a.shape != b.shape:
        raise ValueError("label arrays differ in shape")
    if a.size == 0:
        return 1.0
    _, ai = np.unique(a, return_inverse=True)
    _, bi = np.unique(b, return_inverse=True)
    na, nb = int(ai.max()) + 1, int(bi.max()) + 1
    keys, counts = np.unique(ai.astype(np.int64) * nb + bi, return_counts=True)

    def matched(group, n_groups):
        best = np.zeros(n_groups, dtype=np.int64)
        np.maximum.at(best, group, counts)
        return float(best.sum()) / float(a.size)

    # both directions: a clustering that merges two reference clusters loses in the first, one that splits a
    # reference cluster loses in the second
    return min(matched(keys // nb, na), matched(keys % nb, nb))


def point_cluster_index(labels_uid, pcore_id, pcore_uid, cluster_members, cluster_offsets):
    """For every point the index of the final cluster its microcluster belongs to, or -1 (outlier MCs, pcore MCs
    outside every cluster): labels_uid[n] = creation number of the point's MC; pcore_id / pcore_uid = the pcore list;
    cluster_members / cluster_offsets = member pcore ids of all clusters (cc_clusters_export)."""
    import numpy as np
    labels_uid = np.asarray(labels_uid)
    mem, off = np.asarray(cluster_members), np.asarray(cluster_offsets)
    pcore_id, pcore_uid = np.asarray(pcore_id), np.asarray(pcore_uid)
    if len(labels_uid) == 0:
        return np.empty(0, np.int64)
    if len(pcore_id) == 0 or len(mem) == 0:
        return np.full(len(labels_uid), -1, np.int64)
    cluster_of_member = np.repeat(np.arange(len(off) - 1, dtype=np.int64), np.diff(off))
    order = np.argsort(mem, kind="stable")
    ids_sorted = mem[order]
    pos = np.clip(np.searchsorted(ids_sorted, pcore_id), 0, len(ids_sorted) - 1)
    cluster_of_pcore = np.where(ids_sorted[pos] == pcore_id, cluster_of_member[order][pos], -1)
    uo = np.argsort(pcore_uid, kind="stable")
    uid_sorted = pcore_uid[uo]
    q = np.clip(np.searchsorted(uid_sorted, labels_uid), 0, len(uid_sorted) - 1)
    return np.where(uid_sorted[q] == labels_uid, cluster_of_pcore[uo][q], -1).astype(np.int64)
