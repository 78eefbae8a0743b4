"""chronoclust_amd.app — the reference's entry point (chronoclust/app.py:32-226) on top of the MI355X hot path.

`run()` keeps the signature, the parameter surface (beta, delta, epsilon, lambda, k, mu, pi, omicron,
upsilon) and the output files of the reference: result.csv, cluster_points_D{t}.csv, parameters.csv and
logs/Chronoclust.log, byte for byte.  Per timepoint the clustering, the offline phase and the association
argmin run on the GPU (clustering/hddstream.py, tracking/cluster_tracker.py); the writers are vectorised
(per-point labels come back as one int64 array instead of per-microcluster Python dicts).
"""
import csv
import logging
import os
from collections import defaultdict

import numpy as np
import pandas as pd

from .clustering.hddstream import HDDStream
from .scaling.scaler import Scaler, read_timepoint
from .tracking.cluster_tracker import TrackByHistoricalAssociation, TrackByLineage

# wall-clock seconds per phase and timepoint of the last run() of this process (tools/c3_app.py reports them)
LAST_RUN_TIMINGS = []

HDDSTREAM_OBJ = 'hddstream'
TRACKER_HISTORICAL_ASSOC = 'tracking_by_historical_association'
TRACKER_LINEAGE = 'tracking_by_lineage'


def run(data, output_directory, gating_centroid_file=None, normalise_data=True, restore_program=False,
        param_beta=0.8, param_delta=0.0, param_epsilon=0.03, param_lambda=0, param_k=1,
        param_mu=0.001, param_pi=0, param_omicron=0.0, param_upsilon=1):
    """Runs ChronoClust over the timepoint files in `data` (in time order) and writes the results into
    `output_directory`.  Same arguments and side effects as the reference's `chronoclust.app.run`.

    restore_program: continue from the state saved under `<output_directory>/program_images` after the last
    completed timepoint (app.py:107-114, 165-166).  Unlike the reference's pickles (which lose the id counters
    and whose restart truncates result.csv) the images here hold the microcluster tables
    and counters, and result.csv is appended to.
    """
    program_state_dir = '{}/program_images'.format(output_directory)
    logger = setup_logger('{}/logs'.format(output_directory))
    logger.info("Chronoclust start")
    config = {"beta": param_beta, "delta": param_delta, "epsilon": param_epsilon, "lambda": param_lambda,
              "k": param_k, "mu": param_mu, "pi": param_pi, "omicron": param_omicron, "upsilon": param_upsilon}

    restoring = bool(restore_program) and os.path.exists(os.path.join(program_state_dir, HDDSTREAM_OBJ + '.npz'))
    hddstream = HDDStream(config, logger)
    if restoring:
        logger.info("Restoring Chronoclust state saved in {}".format(program_state_dir))
        tracker_by_association, tracker_by_lineage = restore_program_state(program_state_dir, hddstream)
    else:
        if restore_program:
            logger.warning("Restoring previous Chronoclust state not possible as program_images is not in {}".format(
                output_directory))
        logger.info("Setup new Chronoclust state")
        tracker_by_association = TrackByHistoricalAssociation(handle=hddstream._h)
        tracker_by_lineage = TrackByLineage()

    dataset_attributes = get_dataset_attributes(data[0])
    result_filename = f'{output_directory}/result.csv'
    result_file_header = ['timepoint', 'cumulative_size', 'pcore_ids', 'pref_dimensions'] + dataset_attributes + \
                         ['tracking_by_lineage', 'tracking_by_association']

    gating = defaultdict(dict)
    if gating_centroid_file is not None:
        gating_df = pd.read_csv(gating_centroid_file)
        result_file_header.append('predicted_label')
        for _, gate in gating_df.iterrows():
            centroid = tuple(gate[dataset_attributes].values)
            gating[int(gate['Day'])][centroid] = gate['PopName']
    if restoring and os.path.exists(result_filename):
        drop_rows_after(result_filename, hddstream.last_data_timestamp)
    else:
        write_file_header(result_filename, result_file_header)

    scaler = None
    if normalise_data:
        logger.info("Setting up scaler")
        scaler = Scaler(data, handle=hddstream._h)  # column min / max reduced on the device, files parsed once

    import time
    del LAST_RUN_TIMINGS[:]
    for timepoint, data_file in enumerate(data):
        if restoring and hddstream.last_data_timestamp >= timepoint:
            continue  # already processed before the checkpoint (app.py:165-166)
        logger.info("Processing dataset {}".format(timepoint))
        tm = {"timepoint": timepoint}
        t0 = time.perf_counter()
        raw = scaler.parsed.pop(data_file, None) if scaler is not None else None
        if raw is None:
            raw = read_timepoint(data_file)
        tm["read"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if normalise_data:
            # MinMaxScaler.transform runs on the device as part of the upload (cc_points_upload_scaled)
            logger.info("Scaling dataset {}".format(timepoint))
            hddstream.online_microcluster_maintenance(raw, timepoint, device_scaling=(scaler.scale_, scaler.min_))
        else:
            hddstream.online_microcluster_maintenance(raw, timepoint)
        tm["clustering"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        # app.py:179-190: one Cluster record per final cluster (weight to one decimal place, half up, through the
        # float's shortest repr; member pcores attached), built from the exported arrays
        for cluster in hddstream.cluster_records():
            tracker_by_lineage.add_new_child_cluster(cluster)
        tm["cluster_records"] = time.perf_counter() - t0
        t0 = time.perf_counter()

        tracker_by_lineage.calculate_ids()
        tm["lineage"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        tracker_by_association.set_current_clusters(tracker_by_lineage.child_clusters)
        tracker_by_association.track_cluster_history()
        tm["association"] = time.perf_counter() - t0
        t0 = time.perf_counter()

        write_result_file(gating, result_filename, timepoint, tracker_by_association, scaler=scaler)
        tm["result_rows"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        write_datapoints_details(dataset_attributes, tracker_by_lineage.child_clusters, hddstream,
                                 raw, f'{output_directory}/cluster_points_D{timepoint}.csv', scaler)
        tm["point_details"] = time.perf_counter() - t0
        t0 = time.perf_counter()

        tracker_by_lineage.transfer_child_to_parent()
        tracker_by_association.transfer_current_to_previous()

        logger.info("Saving Chronoclust state for timepoint {}".format(timepoint))
        save_program_state(hddstream, output_directory, tracker_by_association, tracker_by_lineage)
        tm["program_image"] = time.perf_counter() - t0
        LAST_RUN_TIMINGS.append(tm)

    with open(f'{output_directory}/parameters.csv', 'w') as f:
        w = csv.DictWriter(f, config.keys())
        w.writeheader()
        w.writerow(config)
    logger.info('Chronoclust finish')


def write_result_file(gating, result_filename, timepoint, tracker_by_association, scaler):
    """One row per cluster of this timepoint (app.py:229-260); the centroids of all clusters are de-normalised and
    rounded in one array operation (elementwise arithmetic: the same doubles as cluster by cluster)."""
    rows = []
    gating_now = gating.get(timepoint)
    clusters = tracker_by_association.current_clusters
    if clusters:
        cen = np.asarray([cluster.centroid for cluster in clusters], dtype=np.float64)
        if scaler:
            cen = np.asarray(scaler.reverse_scaling(cen), dtype=np.float64)
        centroids = np.round(cen, 5).tolist()
    for i, cluster in enumerate(clusters):
        row = [timepoint, cluster.cumulative_weight, cluster.get_pcore_ids_as_str(),
               cluster.get_preferred_dimensions_as_str()]
        row.extend(centroids[i])
        row.append(cluster.id)
        row.append(cluster.get_historical_associates_as_str())
        if bool(gating_now):
            row.append(find_closest_gating(gating_now, cluster, scaler))
        rows.append(row)
    append_to_file(result_filename, rows)


def write_datapoints_details(dataset_attributes, clusters, hddstream, raw, cluster_points_filename, scaler):
    """id, cluster_id, <features> for every point of the timepoint, in input order (app.py:263-360).

    The reference walks per-microcluster `points` dicts and lets pandas format 10^6 x d floats one by one; here the
    cluster of every point is a gather on the device (cc_point_clusters: per-point microcluster label -> creation
    number -> cluster), the lineage ids are looked up by cluster index, and the text is produced by
    cc_format_points_csv (repr(float) bytes, a few host threads).  Same bytes as DataFrame.to_csv(index=False)."""
    import io
    # (the reference's final DataFrame.to_csv rewrites the whole file, header included, with "\n" line ends)
    head = io.StringIO()
    csv.writer(head, lineterminator="\n").writerow(['id', 'cluster_id'] + dataset_attributes)
    with open(cluster_points_filename, 'w') as f:
        f.write(head.getvalue())
    n = raw.shape[0]
    if n == 0:
        return
    from . import _lib
    # CSV text of every cluster's lineage id (ids like "(A|1,C)" need quotes), by position among the final clusters;
    # last entry: the literal None of points outside every cluster (app.py:396)
    n_final = len(hddstream.final_clusters)
    labels = ["None"] * (n_final + 1)
    for cluster in clusters:
        buf = io.StringIO()
        csv.writer(buf, lineterminator="").writerow([cluster.id])
        labels[cluster.offline_index] = buf.getvalue()
    idx = hddstream.point_cluster_index()  # -1 -> labels[-1]
    if scaler:
        # the reference writes inverse_transform(transform(raw)), not raw: (X - min_) / scale_ on the device
        values = hddstream.resident_points(scaler.scale_, scaler.min_)
    else:
        values = raw
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    with open(cluster_points_filename, 'ab') as f:
        _lib.format_points_csv(values, 0, idx, labels, threads=max(1, min(16, usable)), out=f)


def save_program_state(hddstream, output_dir, tracker_by_association, tracker_by_lineage):
    """State after a completed timepoint (app.py:402-433): ONE bundle holding the microcluster tables + counters
    as arrays and the two trackers pickled (they only hold small Cluster records here), written to a temporary
    file and moved into place with os.replace, so that a crash leaves either the previous image or the new one,
    never the clustering of one timepoint beside the trackers of another."""
    import io
    import pickle
    d = "{}/program_images".format(output_dir)
    os.makedirs(d, exist_ok=True)
    state = hddstream.get_state()
    state[TRACKER_HISTORICAL_ASSOC] = np.frombuffer(pickle.dumps(tracker_by_association), dtype=np.uint8)
    state[TRACKER_LINEAGE] = np.frombuffer(pickle.dumps(tracker_by_lineage), dtype=np.uint8)
    buf = io.BytesIO()
    np.savez(buf, **state)
    final = os.path.join(d, HDDSTREAM_OBJ + '.npz')
    tmp = final + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(buf.getvalue())
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, final)


def restore_program_state(program_state_dir, hddstream):
    """Loads what save_program_state wrote into `hddstream`; returns the two trackers (app.py:436-465)."""
    import pickle
    path = os.path.join(program_state_dir, HDDSTREAM_OBJ + '.npz')
    with np.load(path) as z:
        state = {k: z[k] for k in z.files}
    if TRACKER_HISTORICAL_ASSOC not in state or TRACKER_LINEAGE not in state:
        # an image of the earlier layout: tables in the .npz, the trackers pickled beside it
        legacy = {name: os.path.join(program_state_dir, name + '.pkl') for name in (TRACKER_HISTORICAL_ASSOC, TRACKER_LINEAGE)}
        missing = [f for f in legacy.values() if not os.path.exists(f)]
        if missing:
            raise ValueError("program image {} holds no trackers and {} not found: it was not written by this version "
                             "of chronoclust_amd (or is incomplete); rerun without restore_program".format(
                                 path, " / ".join(missing)))
        for name, f in legacy.items():
            with open(f, 'rb') as fh:
                state[name] = np.frombuffer(fh.read(), dtype=np.uint8)
    needed = ("last_data_timestamp", "dataset_size", "dataset_dimensionality", "pcore_MC_last_id", "outlier_MC_last_id")
    absent = [k for k in needed if k not in state]
    if absent:
        raise ValueError("program image {} lacks {}: not an image of chronoclust_amd".format(path, ", ".join(absent)))
    state.pop("image_timepoint", None)
    tracker_by_association = pickle.loads(state.pop(TRACKER_HISTORICAL_ASSOC).tobytes())
    tracker_by_lineage = pickle.loads(state.pop(TRACKER_LINEAGE).tobytes())
    hddstream.set_state(state)
    tracker_by_association._handle = hddstream._h
    return tracker_by_association, tracker_by_lineage


def drop_rows_after(result_filename, last_timepoint):
    """A run that stopped between writing timepoint t's rows and saving t's image resumes at t: rows of
    timepoints the image does not cover are dropped before result.csv is appended to again.  A last row torn by a
    crash in the middle of a write (short, or without a readable timepoint) is dropped like them."""
    with open(result_filename, newline='') as f:
        rows = list(csv.reader(f))
    width = len(rows[0]) if rows else 0

    def covered(r):
        try:
            return len(r) == width and int(r[0]) <= last_timepoint
        except ValueError:
            return False

    keep = rows[:1] + [r for r in rows[1:] if covered(r)]
    if len(keep) != len(rows):
        tmp = result_filename + '.tmp'
        with open(tmp, 'w') as f:
            csv.writer(f).writerows(keep)
        os.replace(tmp, result_filename)


def setup_logger(log_dir):
    os.makedirs(log_dir, exist_ok=True)
    logging.basicConfig(filename='{}/Chronoclust.log'.format(log_dir),
                        format='%(asctime)s [%(levelname)-8s] %(message)s')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    return logger


def write_file_header(filename, header):
    with open(filename, 'w') as f:
        csv.writer(f).writerow(header)


def append_to_file(filename, content):
    with open(filename, 'a') as f:
        csv.writer(f).writerows(content)


def get_dataset_attributes(dataset_file):
    if str(dataset_file).endswith(".npy"):
        # binary side input: column names from `<file>.columns` (one name per line) if present
        names_file = str(dataset_file) + ".columns"
        if os.path.exists(names_file):
            with open(names_file) as f:
                return [line.strip() for line in f if line.strip()]
        return ["m%d" % i for i in range(np.load(dataset_file, mmap_mode="r").shape[1])]
    return pd.read_csv(dataset_file, sep=',', header=None).iloc[0].values.tolist()


def find_closest_gating(gating_dict, cluster, scaler):
    """Label of the gating centroid with the smallest projected distance to the cluster (app.py:497-512)."""
    best, best_label = None, None
    for centroid, label in gating_dict.items():
        point = scaler.scale_data([centroid])[0].tolist() if scaler else centroid
        dist = cluster.get_projected_dist_to_point(np.array(point))
        if best is None or dist < best:
            best, best_label = dist, label
    return best_label
