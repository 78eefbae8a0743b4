/*
 * chronoclust_hip.h — C-ABI of the MI355X (gfx950) ChronoClust hot path.
 *
 * This is the drop-in boundary.  The upstream project is pure Python and has
 * no FFI of its own; the seam a maintainer would bind
 * is the HDDStream object (chronoclust/clustering/hddstream.py) and the
 * association tracker (chronoclust/tracking/cluster_tracker.py).  Each entry
 * point below names the reference code it replaces.  INTEGRATION.md shows the
 * ctypes stub that goes into the reference.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types; every function returns 0
 *    on success or a negative cc_status; cc_last_error() has the text;
 *  - host pointers unless the name says "device"; row-major float64 [N, d];
 *  - one handle = one HDDStream state on one GPU; a handle is not thread-safe;
 *  - all floating-point work is IEEE double, no FMA contraction, sums over
 *    dimensions strictly left to right, so results are bit-identical to the
 *    reference's numba functions (utilities/mc_functions.py,
 *    utilities/predeconmc_functions.py);
 *  - there is NO CPU fallback: without a HIP device cc_create fails.
 *
 * Derived parameters are computed by the caller with the reference's own
 * Python expressions and passed in finished (so libm's pow never enters a
 * comparison on the device):
 *    eps_sq      = epsilon ** 2                 hddstream.py:46
 *    delta_sq    = delta ** 2                   hddstream.py:49
 *    ups_eps     = upsilon * epsilon            hddstream.py:47
 *    ups_eps_sq  = (upsilon * epsilon) ** 2     predecon.py:40
 *    mu          = mu_cfg * N                   hddstream.py:126,164
 *    omicron     = omicron_cfg * previous N     hddstream.py:119
 *    pi          = d if pi_cfg <= 0 else round  hddstream.py:107-114
 *    decay factor= 2 ** (-lambda * interval)    hddstream.py:283
 */
#ifndef CHRONOCLUST_HIP_H
#define CHRONOCLUST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cc_handle cc_handle;

enum cc_status {
    CC_OK = 0,
    CC_ERR_NO_DEVICE = -1,   /* no HIP device / HIP runtime error            */
    CC_ERR_BAD_ARG = -2,     /* null pointer, d mismatch, d > CC_MAX_DIM ... */
    CC_ERR_NONFINITE = -3,   /* NaN/Inf in the input points                  */
    CC_ERR_OOM = -4,
    CC_ERR_INTERNAL = -5,
    CC_ERR_COMM = -6         /* RCCL / in-process exchange between ranks failed */
};

enum cc_kind { CC_PCORE = 0, CC_OUTLIER = 1 };

/* Dimensions of a point (hddstream.py:107-114 takes any).  Up to CC_WINDOW_MAX_DIM the online phase runs on the windowed,
 * speculative path (and the sequential kernels where the policy prefers them); from there to CC_MAX_DIM on the sequential
 * workgroup kernel alone (k_seq_g: the reference's loop on the table in HBM, ~10-20 us per point) - exact like every path,
 * one GPU only (cc_online* in a multi-GPU group returns CC_ERR_BAD_ARG for such data).  The offline phase and the trackers
 * take any d <= CC_MAX_DIM. */
#define CC_MAX_DIM 128
#define CC_WINDOW_MAX_DIM 64

typedef struct cc_params {
    double eps_sq;
    double delta_sq;
    double k;
    double beta;
    double mu;
    double omicron;
    double ups_eps;
    double ups_eps_sq;
    double delta;
    int32_t pi;
    int32_t pad;
} cc_params;

/* Tuning knobs of the exact windowed online path (0 = library default). */
typedef struct cc_tuning {
    int32_t window;          /* points speculated per window (<= 49152 = the default since round 6) */
    int32_t rounds;          /* max validation rounds per window                 */
    int32_t segments;        /* microcluster-range segments per point tile       */
    int32_t windows_per_sync;/* windows enqueued between host read-backs         */
    int32_t time_kernels;    /* 1: bracket every scan launch with HIP events     */
    int32_t dirty_segments;  /* sub-ranges of the version-row scan (0: = segments)*/
    int32_t early_window;    /* window while the table grows fast (0: 4096)       */
    int32_t lookahead;       /* 0 / 1: scan the next window on a second stream
                              * while this one is validated, as long as windows
                              * commit in full (default); 2: off; 3: always      */
    int32_t sequential;      /* the one-wavefront sequential kernel for small tables:
                              * 0 when the speculative windows keep being cut
                              * short and it measures faster (default); 1 never;
                              * 2 whenever the table fits its LDS image           */
} cc_tuning;

typedef struct cc_stats {
    int64_t points;          /* points processed by the last cc_online_run       */
    int64_t windows;         /* windows committed                                 */
    int64_t rounds;          /* validation rounds executed                        */
    int64_t truncated;       /* windows committed short of their full size       */
    int64_t scan_launches;   /* snapshot-scan launches timed (time_kernels = 1)  */
    double  scan_ms;         /* sum of their HIP-event durations                  */
    double  scan_pair_dims;  /* (point, microcluster, dim) triples they covered   */
    double  run_ms;          /* HIP-event time of the whole cc_online_run         */
    int64_t rows;            /* microcluster rows in the table after the run     */
    int64_t table_rows_scanned; /* sum over windows of the table rows a scan read  */
    int64_t lookahead_windows; /* windows whose snapshot scan ran ahead             */
    int64_t sharded_windows; /* windows whose snapshot scan was split over the ranks (multi-GPU) */
    int64_t comm_launches;   /* merge + all-gather steps timed (time_kernels = 1)  */
    double  comm_ms;         /* sum of their HIP-event durations                  */
    int64_t seq_points;      /* points taken by the sequential kernel             */
    int64_t scan_u_launches; /* snapshot scans launched as k_scan_u (rows as scalar
                              * operands) rather than the LDS-staged k_scan          */
    int64_t scan_p_launches; /* of those, pruned scans (k_seed + k_seed_merge + k_scan_p: rows
                              * abandoned as soon as their partial sums pass a threshold) */
    int64_t pruned_scan_rows;      /* (wave, row) pairs the pruned scans visited (sampled: the
                                    * first point tile of every window) ...              */
    int64_t pruned_scan_full_rows; /* ... and of those, pairs evaluated over all dimensions */
    int64_t window;          /* configured window of the run (cc_tuning.window or the default)  */
    int64_t long_chains;     /* chains of existing microclusters with more than 32 claimants in a window
                              * (per validation round) ...                                      */
    int64_t long_chain_launches; /* ... and launches of k_chain_long over the list of such chains (tables
                              * of more than 1 024 rows; smaller ones always run it)             */
    int64_t tiles;           /* 64-point tiles validated ...                                               */
    int64_t dirty_tiles;     /* ... and of those, tiles whose dirty scan had to run (last round of a window) */
    int64_t scan_launches_pruned;  /* of scan_launches: the timed launches that were pruned chains ...       */
    double  scan_ms_pruned;        /* ... their share of scan_ms ...                                          */
    double  scan_pair_dims_pruned; /* ... and of scan_pair_dims (the rest: plain scans)                       */
    int64_t scan_g_launches;       /* of scan_p_launches: with guessed thresholds (no seed pass over the window) */
    int64_t missed_points;         /* ... points those scans missed (the seeded chain ran for them alone)        */
    int64_t probe_launches;        /* plain scans that carried a probe of the pruned chain (128 points)          */
    int64_t seq_r_points;          /* of seq_points: taken by the register-resident sequential kernel (d <= 4)    */
    int64_t heavy_launches;        /* k_claims_heavy launches (claims of heavy rows gathered without k_decide's atomics) */
    int64_t scan_lean_launches;    /* of scan_g_launches: lean - no list of missed points, no seeded chain behind the scan */
    int64_t long_prepared;         /* of long_chains: chains of pcore microclusters whose running sums were laid out ahead of
                                    * k_chain, every step then evaluated by the step's own 32-lane group ...              */
    int64_t long_replayed;         /* ... and of those, chains replayed from the first step the radius test rejected */
    int64_t seq_g_points;          /* of seq_points: taken by k_seq_g (table in HBM: beyond the sequential kernel's LDS image) */
    int64_t link_launches;         /* windows whose round 0 linked the points that decided "create" among themselves
                                    * (k_link_scan + k_link_apply: while microclusters are being created)                  */
    int64_t scan_p2_launches;      /* of scan_p_launches: the window's pruned scan as k_scan_p2 (two points per lane in phase A,
                                    * phase B from the same residency) rather than k_scan_p                                */
    /* the split of the snapshot scans over the ranks of a group (cc_comm_calibrate): what was measured when the group
     * was formed and the thresholds in force - (table rows x d) from which a plain scan / a pruned chain is split */
    double  calib_allgather_us;    /* all-gather of one full window's records (64 B per point and rank); 0: not measured   */
    double  calib_scan_ns_per_row_dim;  /* plain snapshot scan of a full window: ns per (table row, dimension)             */
    int64_t split_threshold_row_dims;
    int64_t split_threshold_row_dims_pruned;
    int64_t missed_plain_launches; /* plain scans (k_scan_u over a point list) for the points a guessed-threshold scan missed */
    int64_t seed16_launches;       /* seeded pruned chains whose seeds came from the matrix cores (k_seed16) with the tight threshold */
} cc_stats;

/* HDDStream.__init__ (hddstream.py:30-67): one state object on GPU `device`. */
int cc_create(int device, cc_handle** out);
void cc_destroy(cc_handle* h);
const char* cc_last_error(const cc_handle* h);
int cc_set_tuning(cc_handle* h, const cc_tuning* t);
/* Back to the state of a fresh HDDStream (no microclusters, id counters at 0); keeps buffers and parameters. */
int cc_reset(cc_handle* h);

/* hddstream.py:89-128 + 45-52: parameters in force for the following calls. */
int cc_set_params(cc_handle* h, const cc_params* p);

/* hddstream.py:199-213: _decay_clusters_weight(interval) (:247-286), then
 * downgrade_microclusters() (:512-549, including the skip-next-element
 * behaviour of removing from the list being iterated).  `factor` =
 * 2 ** (-lambda * interval). */
int cc_decay_downgrade(cc_handle* h, double factor);

/* The per-point loop of online_microcluster_maintenance (hddstream.py:220-237:
 * _add_to_pcore :288-343, _add_to_outlier :345-395 incl. upgrade :397-430,
 * _create_new_outlier_cluster :434-462), exact sequential semantics.
 *   cc_points_upload : copy X[N,d] to HBM (checks for NaN/Inf on the device)
 *   cc_online_run    : cluster the resident points in row order
 *   cc_labels_download: out_uid[r] = creation number (prev_outlier_id) of the
 *                      microcluster holding row r (microcluster.py:149);
 *                      out_path[r] (optional) = 0 pcore add, 1 outlier add,
 *                      2 new microcluster, |4 if the add promoted it
 *   cc_online        : the three of them in one call */
int cc_points_upload(cc_handle* h, const double* x, int64_t n, int32_t d);
/* Starts uploading the NEXT timepoint's points in the background (a worker thread copies them through page-locked
 * staging buffers on a stream of its own, then scales / checks / transposes them), while the caller still works on
 * the current timepoint (offline phase, trackers, writers - app.py:179-216).  scale / min_: as for
 * cc_points_upload_scaled, or both NULL.  The next cc_points_upload / cc_points_upload_scaled (or cc_online) with the
 * same pointer, shape and scaling adopts the result instead of copying; any other upload discards it.  x must stay
 * valid and unchanged until then.  The clustering results do not depend on whether an upload was prefetched. */
int cc_points_prefetch(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale, const double* min_);
int cc_online_run(cc_handle* h);
int cc_labels_download(cc_handle* h, int64_t* out_uid, int8_t* out_path);
int cc_online(cc_handle* h, const double* x, int64_t n, int32_t d, int64_t* out_uid, int8_t* out_path);

/* HDDStream.pcore_MC / outlier_MC (hddstream.py:56-57) in list order.
 * Any output pointer may be NULL.  cf1/cf2/cen/pref are [count, d]. */
int cc_count(cc_handle* h, int kind);
int cc_dim(cc_handle* h);
int cc_counters(cc_handle* h, int64_t* pcore_last_id, int64_t* outlier_last_id);
/* Checkpoint restore: the id counters of hddstream.py:63-64 (the reference's own pickle drops them). */
int cc_set_counters(cc_handle* h, int64_t pcore_last_id, int64_t outlier_last_id);
int cc_export(cc_handle* h, int kind, int64_t* id, int64_t* uid, double* w,
              double* cf1, double* cf2, double* cen, double* pref);
/* Appends one microcluster to a list (tests, checkpoint restore). */
int cc_inject_mc(cc_handle* h, int kind, int32_t d, const double* cf1, const double* cf2, const double* cen,
                 const double* pref, double w, int64_t id, int64_t uid);

/* Appends n microclusters to a list in one upload (checkpoint restore of a whole table: what unpickling the
 * reference's pcore_MC / outlier_MC lists does, app.py:436-465).  cf1/cf2/cen/pref are [n, d]; w, id, uid [n]. */
int cc_inject_bulk(cc_handle* h, int kind, int32_t d, int32_t n, const double* cf1, const double* cf2,
                   const double* cen, const double* pref, const double* w, const int64_t* id, const int64_t* uid);

/* HDDStream.offline_clustering (hddstream.py:464-510) + PreDeCon.run
 * (clustering/predecon.py:49-267): core flags, eps-neighbourhoods, subspace
 * preference vectors and weighted reachability on the device; the ordered
 * expansion (predecon.py:89-120) on the host.  Returns the number of clusters
 * through *n_clusters.  Optional per-pcore dumps are indexed by list position. */
int cc_offline(cc_handle* h, int32_t* n_clusters, int8_t* out_core, int32_t* out_pdim, int32_t* out_nn,
               int32_t* out_nw);
int cc_num_core(cc_handle* h);
int cc_cluster_size(cc_handle* h, int32_t c);
/* members = pcore ids in merge order (predecon_mc.py:50-68) */
int cc_cluster_export(cc_handle* h, int32_t c, int64_t* members, double* w, double* cf1, double* cf2,
                      double* cen, double* pref);

/* All clusters in one call: offsets[n_clusters + 1] index members[] (pcore ids, merge order);
 * w is [n_clusters]; cf1 / cf2 / cen / pref are [n_clusters, d].  Any output pointer may be NULL. */
int cc_clusters_total_members(cc_handle* h);
int cc_clusters_export(cc_handle* h, int64_t* members, int32_t* offsets, double* w, double* cf1, double* cf2,
                       double* cen, double* pref);

/* MinMax scaling on the device (scaling/scaler.py:27-47, i.e. scikit-learn's
 * MinMaxScaler.partial_fit / transform / inverse_transform, feature range [0, 1]):
 *   cc_col_minmax              : per-column minimum / maximum of x[n, d], NaN ignored
 *                                (np.nanmin / np.nanmax); the caller folds the files of
 *                                all timepoints and forms scale_ = 1 / range
 *                                (range < 10 eps -> 1), min_ = 0 - data_min * scale_
 *   cc_points_upload_scaled    : cc_points_upload of x * scale_ + min_ (two roundings,
 *                                as numpy evaluates transform), scaled on the device
 *   cc_points_download         : the resident points back to the host; with scale_ / min_
 *                                given, (X - min_) / scale_ (inverse_transform), else as held */
int cc_col_minmax(cc_handle* h, const double* x, int64_t n, int32_t d, double* out_min, double* out_max);
int cc_points_upload_scaled(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale,
                            const double* min_);
int cc_points_download(cc_handle* h, double* out, const double* scale, const double* min_);

/* TrackByHistoricalAssociation.track_cluster_history (cluster_tracker.py:127-141):
 * for every current pcore (cur_cen/cur_pref [mc,d]) the index of the previous
 * pcore (prev_cen [mp,d], given in iteration order) with the smallest
 * sum_d (prev - cur)^2 / cur_pref; strict <, first minimum wins. */
int cc_assoc_argmin(cc_handle* h, const double* cur_cen, const double* cur_pref, int32_t mc,
                    const double* prev_cen, int32_t mp, int32_t d, int32_t* out_idx, double* out_dist);

/* Exact multi-GPU path for ONE event stream (the reference is a single Python thread and
 * has no counterpart).  `world` handles - one per GPU, each in its own process (RCCL) or, for verification on a
 * single GPU, several in one process with one host thread each (local) - hold the same state and receive the same
 * calls with the same arguments.  From then on cc_online_run, cc_offline and cc_assoc_argmin are collective:
 *   - the snapshot scan of a window (the loop of hddstream.py:311-328 / :371-375 over the microclusters) is
 *     split by table rows; the ranks all-gather one 64-byte candidate record per window point and every rank
 *     validates and commits the window redundantly and deterministically, so all ranks end with bit-identical
 *     tables, labels and clusters - identical to what one GPU computes;
 *   - the pair matrices of the offline phase (predecon.py:161-188, :219-239) and of the association tracker
 *     (cluster_tracker.py:127-141) are split by rows and all-gathered.
 * Small tables are not split (cc_set_shard_thresholds): below ~1 ms of scan per window the exchange costs more
 * than it saves.
 *   cc_comm_unique_id  : rank 0 obtains the 128-byte RCCL id and hands it to the other ranks (any channel)
 *   cc_comm_init_rccl  : joins the group (collective; librccl is loaded here, not before).  One communicator
 *                        serves both HIP streams of the handle; CHRONOCLUST_HIP_TWO_COMMS=1 adds a second one for
 *                        the lookahead stream (its id travels through the first), so that the all-gathers of
 *                        lookahead scans are not ordered with those of the validation stream - opt-in until a
 *                        multi-GPU run has confirmed it.  No wait for a collective is unbounded: host waits
 *                        poll ncclCommGetAsyncError and a deadline (CHRONOCLUST_HIP_COMM_TIMEOUT_S, default
 *                        120 s); on an error, a missed deadline or any failing call of a member the
 *                        communicators are aborted (ncclCommAbort) and calls return CC_ERR_COMM
 *   cc_comm_init_local : the in-process group of handles[0..world)
 *   cc_comm_info       : transport 0 none, 1 RCCL, 2 local */
#define CC_COMM_ID_BYTES 128
int cc_comm_unique_id(void* out_id);
int cc_comm_init_rccl(cc_handle* h, const void* id_bytes, int rank, int world);
int cc_comm_init_local(cc_handle** handles, int world);
/* Collective (every rank of the group, after joining it; cc_comm_init_rccl calls it itself unless
 * CHRONOCLUST_HIP_CALIBRATE=0; the members of an in-process group call it from their own threads): MEASURES what the split of
 * the snapshot scans over the ranks costs and what it saves, and derives the split thresholds from the measurement instead of
 * a constant -
 *   - three all-gathers of one full window's candidate records (64 B per point and rank) on the handle's stream: the exchange
 *     a split window pays, `calib_allgather_us` (minimum of the three);
 *   - three plain snapshot scans (k_scan_u) of a full window over 4 096 synthetic rows x 20 dimensions on scratch buffers:
 *     what a row costs, `calib_scan_ns_per_row_dim`;
 *   - both figures are exchanged and every rank takes the group's MAXIMUM, so that all ranks derive the same thresholds (they
 *     decide the sequence of collectives): a plain scan is split from rows x d >= allgather x world / (world - 1) / scan per
 *     (row, dim) on - where the time saved, scan x (1 - 1 / world), equals the exchange -, a pruned chain from 3.3 times
 *     that (it spends 0.3 of a plain scan's time on the same rows: 92 against 306 us per full window at 5 000 x 20).
 * A group of ONE rank measures and reports but keeps its thresholds (there is nothing to save; one-rank groups exist to
 * exercise the transport).  cc_set_shard_thresholds afterwards overrides both thresholds.  cc_get_stats reports all four. */
int cc_comm_calibrate(cc_handle* h);
int cc_comm_destroy(cc_handle* h);
int cc_comm_info(cc_handle* h, int32_t* rank, int32_t* world, int32_t* transport);
/* RELAXED multi-GPU mode - not the reference's semantics.  The events of a timepoint are sharded over the ranks in
 * contiguous blocks (what BASELINE.json's north_star sketches: "events within a timestep shard across the GPUs with
 * an RCCL all-reduce of the per-MC CF-vector deltas").  Per super-step every rank clusters `minibatch_points` of its
 * block against the table all ranks share (exact path, but a point that no MC absorbs is set aside instead of
 * creating one), the CF1 / CF2 / weight changes of the existing rows are all-reduced and written back with centroids
 * and preferred dimensions recomputed and promotions decided on the merged rows, and the set-aside points of all
 * ranks are clustered on every rank redundantly (exact path), so that new MCs are created once and all ranks keep
 * identical tables.  Within a super-step a rank does not see the other ranks' adds: labels, ids and CF sums differ
 * from the reference's; bench.py reports the agreement with the exact path beside the number.  Every rank must hold
 * the whole timepoint (cc_points_upload of the same array); cc_labels_download returns all labels on every rank
 * (path code 8 never remains: set-aside points are labelled by the second half of their super-step).
 * minibatch_points = 0 switches back to the exact path. */
int cc_comm_set_relaxed(cc_handle* h, int32_t minibatch_points);
typedef struct cc_relaxed_stats {
    int64_t super_steps;       /* of the last cc_online_run                                   */
    int64_t minibatch_points;  /* points this rank clustered in the sharded halves             */
    int64_t deferred_points;   /* set-aside points (all ranks) clustered in the replicated halves */
    int64_t reserved[5];
} cc_relaxed_stats;
int cc_get_relaxed_stats(cc_handle* h, cc_relaxed_stats* out);
/* The block [lo, hi) of n rows rank `rank` of `world` takes, in whole units of `unit` rows: the partition every
 * kernel of the multi-GPU path uses (unit 1: table rows of a scan, current pcores of the association argmin;
 * 64: rows of the offline pair matrices).  Pure host arithmetic, needs no handle and no GPU. */
int cc_shard_rows(int32_t n, int32_t world, int32_t rank, int32_t unit, int32_t* lo, int32_t* hi);
/* split the scan when rows * d >= min_row_dims, the offline / association pair matrices when rows >=
 * offline_min_rows; a negative value keeps the current setting (defaults 400 000 and 8 192) */
int cc_set_shard_thresholds(cc_handle* h, int64_t min_row_dims, int32_t offline_min_rows);

/* The join of app.py:303-332 (write_datapoints_details walks the `points` dicts of every cluster's pcores): for every
 * resident point the index of the cluster (order of cc_clusters_export, after cc_offline) its microcluster belongs
 * to, -1 for points in outlier microclusters and in pcores outside every cluster (the literal `None` of app.py:396).
 * A gather on the device from the per-point labels through a creation-number -> cluster table. */
int cc_point_clusters(cc_handle* h, int32_t* out_idx);

/* Text of n rows of cluster_points_D{t}.csv as DataFrame.to_csv(index=False) writes them (app.py:357-360):
 * "<first_id + r>,<label>,<v[r,0]>,...,<v[r,d-1]>\n", floats as repr(float) (shortest round-trip digits, CPython's
 * layout rules).  label_idx[r] selects one of n_labels CSV-ready label texts (label_bytes[label_offsets[i] ..
 * label_offsets[i + 1])); an index outside [0, n_labels) selects the last one.  Pure host work, no handle: callers
 * format chunks of rows on several threads.  Returns the number of bytes written to out, or CC_ERR_OOM when `cap`
 * cannot hold them (25 + longest label + 33 d + 2 bytes per row always suffice). */
int64_t cc_format_points_csv(const double* values, int64_t n, int32_t d, int64_t first_id, const int32_t* label_idx,
                             const char* label_bytes, const int32_t* label_offsets, int32_t n_labels, char* out,
                             int64_t cap);

/* Waits until everything the handle has enqueued on its HIP streams is done (the timing bracket of a harness: what
 * torch.cuda.synchronize() would be for work a framework had launched).  Inside a group the wait is bounded like
 * every wait of the library that may hold a collective: CC_ERR_COMM when a peer is gone. */
int cc_sync(cc_handle* h);

int cc_get_stats(cc_handle* h, cc_stats* out);

/* The window policy of the exact online phase as a pure function (csrc/cc_policy.h; no reference counterpart): between
 * two batches of windows the library reads the device's counters and decides window size, validation rounds, windows per
 * batch, lookahead / dirty / pruned scans and the split over the ranks of a group.  No decision can change a result, but
 * the ranks of a group must all take the same ones, so they depend on the counters alone.  cc_policy_replay runs the
 * policy over recorded observations - no handle, no GPU: out[0] is the decision for the first batch of a call that
 * starts at (start_cursor, start_rows), out[i + 1] the one taken after obs[i] (obs[i].after_sequential != 0: the stream
 * came back from the sequential kernel at obs[i].cursor / m_rows instead of finishing a batch).  *carry is updated to
 * what the handle would keep for its next call.  CHRONOCLUST_HIP_POLICY_TRACE=<file> makes the library append the
 * observations and decisions of every call as JSON lines (how the traces under tests/golden/policy/ were recorded). */
#define CC_POLICY_MAX_ROUNDS 8
typedef struct cc_policy_config {
    int32_t window, rounds_max, windows_per_sync, early_window;  /* cc_tuning values in force                        */
    int32_t lookahead;         /* cc_tuning.lookahead                                                              */
    int32_t allow_nodirty;     /* dirty scans may be left out while ruled out (CHRONOCLUST_HIP_NODIRTY != 0)       */
    int32_t prune_mode;        /* CHRONOCLUST_HIP_PRUNE: 0 never, 1 while it pays, 2 always                        */
    int32_t prune_applicable;  /* k = 2^e, pi >= d and d one of the widths the pruned scan is compiled for         */
    int32_t can_shard;         /* the handle belongs to a group and is not inside a relaxed super-step             */
    int32_t d;
    int32_t resume;            /* the call continues a stream this handle was clustering a moment ago              */
    int32_t allow_sparse;      /* sparse dirty scans while at most one point in this many needs them (0: never)     */
    int32_t allow_guess;       /* pruned scans may take table-wide guessed thresholds (CHRONOCLUST_HIP_GUESS != 0); 1: and
                                  run lean - without the list of missed points - after a batch without a miss, 2: never lean */
    int32_t allow_probe;       /* pruned scans come back on a probe's word (CHRONOCLUST_HIP_PROBE != 0), else after a
                                * stretch of points that doubles with every failed try                               */
    int64_t shard_min_row_dims;
    int64_t n_end;             /* end of the range of points the call clusters                                     */
    int64_t shard_min_row_dims_pruned;  /* the split threshold while the scans are pruned chains (0: shard_min_row_dims)   */
    int32_t lookahead_pruned;  /* 1: lookahead scans also while the scans are pruned chains on one GPU (CHRONOCLUST_HIP_LA_PRUNED=1;
                                * round 6: off - such a scan is too short to be worth a stream of its own, cc_policy.h)        */
    int32_t force_prune_rows;  /* > 0: pruned scans whenever the table has at least this many rows, whatever the samples said (round
                                * 6: behind k_seed16's seeds and the tight threshold a pruned chain returns what the plain scan returns,
                                * and from a few thousand rows on at a fraction of its time - also while the table still fills)  */
} cc_policy_config;
typedef struct cc_policy_carry {
    int32_t adapt_win, clean_batches, since_shrink, pad;
} cc_policy_carry;
typedef struct cc_policy_obs {   /* cumulative device counters of the call as read back after a batch              */
    int64_t cursor;
    int32_t m_rows, stall_b;
    int64_t stat_windows, stat_truncated, stat_trunc_unknown, stat_tiles, stat_dirty_tiles;
    int64_t stat_unsafe;       /* (point, round) pairs that needed rows only a dirty scan covers                  */
    int64_t stat_missed;       /* points a guessed threshold missed (the seeded chain ran for them)               */
    int64_t round_hist[CC_POLICY_MAX_ROUNDS + 2];
    uint64_t prune_rows, prune_full;
    int32_t after_sequential;
    int32_t tg_ok;             /* a mean join distance exists: guessed thresholds are available                   */
} cc_policy_obs;
typedef struct cc_policy_decision {
    int32_t win_cfg;        /* window size of the next batch (changes only with `restart`)                         */
    int32_t want;           /* the size the policy is heading for                                                  */
    int32_t rounds, batch_windows, lookahead, nodirty;
    int32_t prune;          /* 0: plain scans, 1: pruned with seeded thresholds, 2: pruned with guessed thresholds,
                             * 3: guessed thresholds, lean (no list of missed points, no seeded chain behind the scan) */
    int32_t shard;
    int32_t restart;        /* the chain of windows restarts: pending lookahead scan dropped, control block pushed */
    int32_t bad;            /* short, truncated windows at a small window size (input of the sequential-kernel rule) */
    int32_t stalled;        /* five batches without progress: the call fails with CC_ERR_INTERNAL                  */
    int32_t sparse;         /* with nodirty: the sparse dirty scans run for the points that need rows of their own  */
    int32_t probe;          /* plain scans, and the pruned chain runs beside the batch's first one on 128 points: its
                             * sample tells the policy whether pruned scans would pay, at 1 / 256 of a scan's cost    */
    int32_t pad;
    int64_t wins, pts, trunc, unk, tiles, dtiles, grew, prune_rows, prune_full;  /* what the batch did (deltas)     */
} cc_policy_decision;
int cc_policy_replay(const cc_policy_config* cfg, cc_policy_carry* carry, int64_t start_cursor, int32_t start_rows,
                     const cc_policy_obs* obs, int32_t n, cc_policy_decision* out);
/* The one wall-clock rule of the library (a stream of short, truncated windows is handed to the sequential kernel when that
 * is faster) compares the windows' measured rate with an ASSUMED rate of the sequential kernel until that has been measured
 * in the call: points per millisecond, as a function of the dimensionality and of the table size - k_seq_g, the kernel for
 * tables beyond the LDS image, slows down in proportion to the rows once they exceed its 1 024 threads.  Pure; < 0 on a
 * bad argument.  No reference counterpart. */
double cc_policy_seq_rate_guess(int32_t d, int32_t m_rows, int32_t allow_seq_r, int32_t allow_seq_g);

#ifdef __cplusplus
}
#endif
#endif
