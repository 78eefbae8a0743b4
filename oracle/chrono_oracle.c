/*
 * chrono_oracle.c — CPU restatement of ChronoClust's per-timestep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (chronoclust_amd/) may
 * link, load or call this file.  It is imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and only as the
 * checker / the CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 * golden vectors produced by importing the upstream Python reference in the
 * build container (tests/golden/make_golden.py): the reference's own
 * unit-test known answers, its integration golden (result.csv + per-point
 * labels on synthetic d0-d4) and per-point / per-microcluster dumps of
 * d = 3 / 14 / 20 scenarios over several timepoints with decay.
 *
 * Every function cites the reference file:line it restates (paths relative
 * to /root/reference/chronoclust/).  All arithmetic is IEEE double; sums over
 * dimensions are strict left-to-right (numba lowers np.sum to a sequential
 * loop); build with -O2 -ffp-contract=off and without -ffast-math.
 *
 * Derived parameters (epsilon^2, delta^2, upsilon*epsilon, mu = mu_cfg*N, the
 * decay factor 2**(-lambda*dt), ...) are computed by the caller in Python with
 * the reference's own expressions (clustering/hddstream.py:45-52, 107-126,
 * 283) and handed in, so that libm's pow() never enters a comparison here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CO_PCORE 0
#define CO_OUTLIER 1

typedef struct {
    double eps_sq;      /* hddstream.py:46  epsilon ** 2                      */
    double delta_sq;    /* hddstream.py:49  delta ** 2                        */
    double k;           /* hddstream.py:51                                    */
    double beta;        /* hddstream.py:50                                    */
    double mu;          /* hddstream.py:126,164  mu_cfg * N                   */
    double omicron;     /* hddstream.py:119 omicron_cfg * previous N          */
    double ups_eps;     /* hddstream.py:47  upsilon * epsilon                 */
    double ups_eps_sq;  /* predecon.py:40   (upsilon*epsilon) ** 2            */
    double delta;       /* predecon.py:42   delta (NOT squared, predecon:213) */
    int32_t pi;         /* hddstream.py:107-114                               */
    int32_t pad;
} co_params;

typedef struct {
    double *cf1, *cf2, *cen, *pref; /* microcluster.py:71-77 */
    double w;                       /* cumulative_weight      */
    int64_t id;                     /* current id (pcore id or outlier id) */
    int64_t uid;                    /* prev_outlier_id: creation number (hddstream.py:449,459) */
} co_mc;

typedef struct {
    int64_t *members;  /* pcore ids in merge order (predecon_mc.py:67 id.add) */
    int32_t n_members;
    double *cf1, *cf2, *cen, *pref;
    double w;
} co_cluster;

typedef struct co_state {
    co_params p;
    int32_t d;
    co_mc **pcore;   int32_t n_pcore, cap_pcore;     /* hddstream.py:56 */
    co_mc **outlier; int32_t n_outlier, cap_outlier; /* hddstream.py:57 */
    int64_t pcore_last_id, outlier_last_id;          /* hddstream.py:63-64 */
    co_cluster *clusters; int32_t n_clusters;        /* hddstream.py:58 final_clusters */
    int32_t n_core_last;                             /* hddstream.py:491 */
} co_state;

/* ------------------------------------------------------------------ */
/* utilities/mc_functions.py restated on raw vectors                   */
/* ------------------------------------------------------------------ */

/* mc_functions.py:35-43  sum_d (p-c)^2 / pref, left to right */
double co_projected_distance(const double *cen, const double *pref, const double *pt, int d)
{
    double acc = 0.0;
    for (int i = 0; i < d; ++i) {
        double t = pt[i] - cen[i];
        t = t * t;
        t = t / pref[i];
        acc = acc + t;
    }
    return acc;
}

/* mc_functions.py:14-22  cf2/W - (cf1/W)^2 for one dimension */
static inline double sq_variance(double cf1, double cf2, double w)
{
    double a = cf2 / w;
    double b = cf1 / w;
    b = b * b;
    return a - b;
}

/* mc_functions.py:45-56 */
double co_projected_radius_sq(const double *cf1, const double *cf2, const double *pref, double w, int d)
{
    double acc = 0.0;
    for (int i = 0; i < d; ++i) {
        double v = sq_variance(cf1[i], cf2[i], w);
        v = v / pref[i];
        acc = acc + v;
    }
    return acc;
}

/* microcluster.py:89-115  pref_d = k if var_d <= delta^2 else 1.0 (NaN -> 1.0) */
void co_update_pref(const double *cf1, const double *cf2, double w, double delta_sq, double k, double *pref, int d)
{
    for (int i = 0; i < d; ++i) {
        double v = sq_variance(cf1[i], cf2[i], w);
        pref[i] = (v <= delta_sq) ? k : 1.0;
    }
}

/* mc_functions.py:64-77 */
int co_is_core(const double *cf1, const double *cf2, const double *pref, double w, int d,
               double radius_thr_sq, double density_thr, int max_pdim)
{
    double r = co_projected_radius_sq(cf1, cf2, pref, w, d);
    int cnt = 0;
    for (int i = 0; i < d; ++i) cnt += (pref[i] > 1.0);
    return (r <= radius_thr_sq) && (w >= density_thr) && (cnt <= max_pdim);
}

/* utilities/predeconmc_functions.py:4-17.  np.linalg.norm is BLAS nrm2 under
 * numba and sqrt(dot) under numpy; both are platform-defined in the last ulp.
 * This restatement (and the HIP kernel) use sqrt of the left-to-right sum. */
double co_euclidean(const double *a, const double *b, int d)
{
    double acc = 0.0;
    for (int i = 0; i < d; ++i) {
        double t = a[i] - b[i];
        acc = acc + t * t;
    }
    return sqrt(acc);
}

/* predeconmc_functions.py:19-42 */
double co_variance_along_dimension(double point, const double *neigh, int n)
{
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = point - neigh[i];
        acc = acc + t * t;
    }
    return acc / (double)n;
}

/* predeconmc_functions.py:44-62 */
double co_weighted_dist_sq(const double *pref, const double *p, const double *q, int d)
{
    double acc = 0.0;
    for (int i = 0; i < d; ++i) {
        double t = p[i] - q[i];
        t = t * t;
        t = pref[i] * t;
        acc = acc + t;
    }
    return acc;
}

/* ------------------------------------------------------------------ */
/* state                                                               */
/* ------------------------------------------------------------------ */

static co_mc *mc_new(int d)
{
    co_mc *m = (co_mc *)calloc(1, sizeof(co_mc));
    m->cf1 = (double *)calloc((size_t)d * 4, sizeof(double));
    m->cf2 = m->cf1 + d;
    m->cen = m->cf2 + d;
    m->pref = m->cen + d;
    return m;
}
static void mc_free(co_mc *m) { if (m) { free(m->cf1); free(m); } }

static void list_push(co_mc ***list, int32_t *n, int32_t *cap, co_mc *m)
{
    if (*n == *cap) {
        *cap = *cap ? *cap * 2 : 64;
        *list = (co_mc **)realloc(*list, sizeof(co_mc *) * (size_t)*cap);
    }
    (*list)[(*n)++] = m;
}
static void list_remove_at(co_mc **list, int32_t *n, int32_t i)
{
    memmove(list + i, list + i + 1, sizeof(co_mc *) * (size_t)(*n - i - 1));
    (*n)--;
}

static void clusters_free(co_state *s)
{
    for (int i = 0; i < s->n_clusters; ++i) {
        free(s->clusters[i].members);
        free(s->clusters[i].cf1);
    }
    free(s->clusters);
    s->clusters = NULL;
    s->n_clusters = 0;
}

co_state *co_create(void) { return (co_state *)calloc(1, sizeof(co_state)); }

void co_destroy(co_state *s)
{
    if (!s) return;
    for (int i = 0; i < s->n_pcore; ++i) mc_free(s->pcore[i]);
    for (int i = 0; i < s->n_outlier; ++i) mc_free(s->outlier[i]);
    free(s->pcore);
    free(s->outlier);
    clusters_free(s);
    free(s);
}

void co_set_params(co_state *s, const co_params *p) { s->p = *p; }

/* Inject a microcluster directly (used by unit-level parity tests). */
int co_inject_mc(co_state *s, int kind, int d, const double *cf1, const double *cf2, const double *cen,
                 const double *pref, double w, int64_t id, int64_t uid)
{
    if (s->d == 0) s->d = d;
    if (s->d != d) return -1;
    co_mc *m = mc_new(d);
    memcpy(m->cf1, cf1, sizeof(double) * d);
    memcpy(m->cf2, cf2, sizeof(double) * d);
    memcpy(m->cen, cen, sizeof(double) * d);
    memcpy(m->pref, pref, sizeof(double) * d);
    m->w = w; m->id = id; m->uid = uid;
    if (kind == CO_PCORE) {
        list_push(&s->pcore, &s->n_pcore, &s->cap_pcore, m);
        if (id >= s->pcore_last_id) s->pcore_last_id = id + 1;
    } else {
        list_push(&s->outlier, &s->n_outlier, &s->cap_outlier, m);
    }
    if (uid >= s->outlier_last_id) s->outlier_last_id = uid + 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* decay + downgrade  (hddstream.py:199-213, 247-286, 512-549)         */
/* ------------------------------------------------------------------ */

static void decay_one(co_mc *m, double f, int d)
{
    /* hddstream.py:283-286; centroid and pref are NOT recomputed */
    for (int i = 0; i < d; ++i) m->cf1[i] = m->cf1[i] * f;
    for (int i = 0; i < d; ++i) m->cf2[i] = m->cf2[i] * f;
    m->w = m->w * f;
}

int co_decay_downgrade(co_state *s, double factor)
{
    int d = s->d;
    for (int i = 0; i < s->n_pcore; ++i) decay_one(s->pcore[i], factor, d);
    for (int i = 0; i < s->n_outlier; ++i) decay_one(s->outlier[i], factor, d);

    double beta_mu = s->p.beta * s->p.mu;
    /* hddstream.py:528-537.  Python iterates the list while removing from it:
     * after a removal at index i the iterator moves to i+1, so the element
     * that slid into i is never examined. */
    for (int i = 0; i < s->n_pcore; ++i) {
        co_mc *m = s->pcore[i];
        int cnt = 0;
        for (int j = 0; j < d; ++j) cnt += (m->pref[j] > 1.0);
        if ((m->w < beta_mu) || (cnt > s->p.pi)) {
            m->id = m->uid; /* :535 id = [prev_outlier_id] */
            list_remove_at(s->pcore, &s->n_pcore, i);
            list_push(&s->outlier, &s->n_outlier, &s->cap_outlier, m);
        }
    }
    /* hddstream.py:545-549, same iteration quirk */
    for (int i = 0; i < s->n_outlier; ++i) {
        co_mc *m = s->outlier[i];
        if (m->w <= s->p.omicron) {
            list_remove_at(s->outlier, &s->n_outlier, i);
            mc_free(m);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* online maintenance (hddstream.py:220-237, 288-462)                  */
/* ------------------------------------------------------------------ */

/* microcluster.py:213-233 get_copy_with_new_point: tentative add into scratch */
static void tentative_add(const co_mc *m, const double *pt, int d, double delta_sq, double k,
                          double *cf1, double *cf2, double *pref, double *w_out)
{
    double w = m->w + 1.0;
    for (int i = 0; i < d; ++i) {
        cf1[i] = m->cf1[i] + pt[i];
        cf2[i] = m->cf2[i] + pt[i] * pt[i];
    }
    co_update_pref(cf1, cf2, w, delta_sq, k, pref, d);
    *w_out = w;
}

/* microcluster.py:117-165 add_new_point + update_preferred_dimensions */
static void commit_add(co_mc *m, const double *pt, int d, double delta_sq, double k)
{
    for (int i = 0; i < d; ++i) {
        m->cf1[i] = m->cf1[i] + pt[i];
        m->cf2[i] = m->cf2[i] + pt[i] * pt[i];
    }
    m->w = m->w + 1.0;
    for (int i = 0; i < d; ++i) m->cen[i] = m->cf1[i] / m->w; /* mc_functions.py:31-33 */
    co_update_pref(m->cf1, m->cf2, m->w, delta_sq, k, m->pref, d);
}

/*
 * out_uid[r]  : uid of the microcluster that absorbed row r (microcluster.py:149 points[idx])
 * out_path[r] : 0 = added to a pcore MC, 1 = added to an outlier MC, 2 = new outlier MC,
 *               | 4 if the add promoted the outlier MC to pcore (optional, may be NULL)
 */
int co_online(co_state *s, const double *X, int64_t N, int d, int64_t *out_uid, int8_t *out_path)
{
    if (s->d == 0) s->d = d;
    if (s->d != d) return -1;
    const double delta_sq = s->p.delta_sq, k = s->p.k, eps_sq = s->p.eps_sq;
    const int pi = s->p.pi;
    const int filter = (pi < d); /* pdim <= pi is vacuous when pi >= d */
    const double beta_mu = s->p.beta * s->p.mu;
    double *t_cf1 = (double *)malloc(sizeof(double) * (size_t)d * 3);
    double *t_cf2 = t_cf1 + d, *t_pref = t_cf2 + d;

    for (int64_t r = 0; r < N; ++r) {
        const double *pt = X + r * d;
        int done = 0;
        /* ---- _add_to_pcore, hddstream.py:288-343 ---- */
        {
            int best = -1;
            double best_d = 0.0;
            for (int i = 0; i < s->n_pcore; ++i) {
                const co_mc *m = s->pcore[i];
                if (filter) {
                    double w;
                    tentative_add(m, pt, d, delta_sq, k, t_cf1, t_cf2, t_pref, &w);
                    int pdim = 0;
                    for (int j = 0; j < d; ++j) pdim += (t_pref[j] != 1.0); /* :319 */
                    if (pdim > pi) continue;
                }
                double dist = co_projected_distance(m->cen, m->pref, pt, d); /* :325 stored centroid/pref */
                if (best < 0 || dist < best_d) { best = i; best_d = dist; }  /* :326 strict < */
            }
            if (best >= 0) {
                co_mc *m = s->pcore[best];
                double w;
                tentative_add(m, pt, d, delta_sq, k, t_cf1, t_cf2, t_pref, &w);
                double r2 = co_projected_radius_sq(t_cf1, t_cf2, t_pref, w, d); /* :335 */
                if (r2 <= eps_sq) {
                    commit_add(m, pt, d, delta_sq, k);
                    out_uid[r] = m->uid;
                    if (out_path) out_path[r] = 0;
                    done = 1;
                }
            }
        }
        /* ---- _add_to_outlier, hddstream.py:345-395 ---- */
        if (!done) {
            int best = -1;
            double best_d = 0.0;
            for (int i = 0; i < s->n_outlier; ++i) {
                const co_mc *m = s->outlier[i];
                double dist = co_projected_distance(m->cen, m->pref, pt, d);
                if (best < 0 || dist < best_d) { best = i; best_d = dist; }
            }
            if (best >= 0) {
                co_mc *m = s->outlier[best];
                double w;
                tentative_add(m, pt, d, delta_sq, k, t_cf1, t_cf2, t_pref, &w);
                double r2 = co_projected_radius_sq(t_cf1, t_cf2, t_pref, w, d);
                if (r2 <= eps_sq) {
                    commit_add(m, pt, d, delta_sq, k);
                    out_uid[r] = m->uid;
                    int8_t path = 1;
                    /* _upgrade_outlier_microcluster, :397-430 (prev_pcore_id is never set: always a fresh id) */
                    int cnt = 0;
                    for (int j = 0; j < d; ++j) cnt += (m->pref[j] > 1.0);
                    if ((m->w >= beta_mu) && (cnt <= pi)) {
                        m->id = s->pcore_last_id++;
                        list_remove_at(s->outlier, &s->n_outlier, best);
                        list_push(&s->pcore, &s->n_pcore, &s->cap_pcore, m);
                        path |= 4;
                    }
                    if (out_path) out_path[r] = path;
                    done = 1;
                }
            }
        }
        /* ---- _create_new_outlier_cluster, hddstream.py:434-462 ---- */
        if (!done) {
            co_mc *m = mc_new(d); /* zeros */
            commit_add(m, pt, d, delta_sq, k);
            m->id = s->outlier_last_id;
            m->uid = s->outlier_last_id;
            s->outlier_last_id++;
            list_push(&s->outlier, &s->n_outlier, &s->cap_outlier, m);
            out_uid[r] = m->uid;
            if (out_path) out_path[r] = 2;
        }
    }
    free(t_cf1);
    return 0;
}

/* ------------------------------------------------------------------ */
/* export                                                              */
/* ------------------------------------------------------------------ */

int co_count(const co_state *s, int kind) { return kind == CO_PCORE ? s->n_pcore : s->n_outlier; }
int co_dim(const co_state *s) { return s->d; }
int64_t co_pcore_last_id(const co_state *s) { return s->pcore_last_id; }
int64_t co_outlier_last_id(const co_state *s) { return s->outlier_last_id; }

/* list order = Python list order */
void co_export(const co_state *s, int kind, int64_t *id, int64_t *uid, double *w,
               double *cf1, double *cf2, double *cen, double *pref)
{
    int n = co_count(s, kind), d = s->d;
    co_mc **l = kind == CO_PCORE ? s->pcore : s->outlier;
    for (int i = 0; i < n; ++i) {
        if (id) id[i] = l[i]->id;
        if (uid) uid[i] = l[i]->uid;
        if (w) w[i] = l[i]->w;
        if (cf1) memcpy(cf1 + (size_t)i * d, l[i]->cf1, sizeof(double) * d);
        if (cf2) memcpy(cf2 + (size_t)i * d, l[i]->cf2, sizeof(double) * d);
        if (cen) memcpy(cen + (size_t)i * d, l[i]->cen, sizeof(double) * d);
        if (pref) memcpy(pref + (size_t)i * d, l[i]->pref, sizeof(double) * d);
    }
}

/* ------------------------------------------------------------------ */
/* offline phase: hddstream.py:464-510 + clustering/predecon.py        */
/* ------------------------------------------------------------------ */

typedef struct { int32_t *v; int32_t n, cap; } ivec;
static void ivec_push(ivec *a, int32_t x)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (int32_t *)realloc(a->v, sizeof(int32_t) * (size_t)a->cap); }
    a->v[a->n++] = x;
}

/*
 * Optional dumps (may be NULL), all indexed by pcore list position:
 *   out_core[M], out_pdim[M] (PreDeCon pdim, predecon_mc.py:70-81), out_nn[M] (|N_eps|), out_nw[M] (|N_w|)
 */
int co_offline(co_state *s, int8_t *out_core, int32_t *out_pdim, int32_t *out_nn, int32_t *out_nw)
{
    clusters_free(s);
    const int M = s->n_pcore, d = s->d;
    const double eps = s->p.ups_eps, eps_sq = s->p.ups_eps_sq, delta = s->p.delta, k = s->p.k;
    const int lam = s->p.pi;
    s->n_core_last = 0;
    if (M == 0) return 0;

    /* hddstream.py:483-496: core flag from the MC itself; dict keyed by pcore id in list order */
    int8_t *core = (int8_t *)calloc((size_t)M, 1);
    for (int i = 0; i < M; ++i) {
        co_mc *m = s->pcore[i];
        core[i] = (int8_t)co_is_core(m->cf1, m->cf2, m->pref, m->w, d, s->p.eps_sq, s->p.mu, s->p.pi);
        s->n_core_last += core[i];
    }

    /* predecon.py:149-152: eps-neighbourhood (includes self) + subspace preference vector */
    ivec *nb = (ivec *)calloc((size_t)M, sizeof(ivec));
    ivec *nw = (ivec *)calloc((size_t)M, sizeof(ivec));
    double *wvec = (double *)malloc(sizeof(double) * (size_t)M * d);
    int32_t *pdim = (int32_t *)calloc((size_t)M, sizeof(int32_t));
    for (int p = 0; p < M; ++p) {
        const double *cp = s->pcore[p]->cen;
        for (int q = 0; q < M; ++q) { /* predecon.py:161-188 */
            if (co_euclidean(s->pcore[q]->cen, cp, d) <= eps) ivec_push(&nb[p], q);
        }
        for (int j = 0; j < d; ++j) { /* predecon.py:190-217 */
            double acc = 0.0;
            for (int t = 0; t < nb[p].n; ++t) {
                double df = cp[j] - s->pcore[nb[p].v[t]]->cen[j];
                acc = acc + df * df;
            }
            double var = acc / (double)nb[p].n;
            wvec[(size_t)p * d + j] = (var <= delta) ? k : 1.0; /* :213 delta NOT squared */
            pdim[p] += (wvec[(size_t)p * d + j] > 1.0);          /* predecon_mc.py:81 */
        }
    }
    /* predecon.py:155-159, 219-239 */
    for (int p = 0; p < M; ++p) {
        const double *cp = s->pcore[p]->cen;
        for (int t = 0; t < nb[p].n; ++t) {
            int q = nb[p].v[t];
            const double *cq = s->pcore[q]->cen;
            double dpq = co_weighted_dist_sq(wvec + (size_t)p * d, cp, cq, d);
            double dqp = co_weighted_dist_sq(wvec + (size_t)q * d, cq, cp, d);
            double dist = dpq > dqp ? dpq : dqp; /* max() */
            if (dist <= eps_sq) ivec_push(&nw[p], q);
        }
    }

    /* predecon.py:62-87 main loop + _expand :89-120 + _find_directly_reachable_points :242-267 */
    int8_t *cls = (int8_t *)calloc((size_t)M, 1); /* 0 = 'u', 1 = 'c', 2 = 'n' */
    ivec queue = {0};
    co_cluster *clusters = NULL;
    int n_clusters = 0, cap_clusters = 0;
    for (int seed = 0; seed < M; ++seed) {
        if (cls[seed] != 0) continue;
        if (!core[seed]) { cls[seed] = 2; continue; }
        co_cluster c;
        memset(&c, 0, sizeof(c));
        c.cf1 = (double *)calloc((size_t)d * 4, sizeof(double));
        c.cf2 = c.cf1 + d; c.cen = c.cf2 + d; c.pref = c.cen + d;
        ivec members = {0};
        queue.n = 0;
        for (int t = 0; t < nw[seed].n; ++t) ivec_push(&queue, nw[seed].v[t]); /* :103 */
        int head = 0;
        while (head < queue.n) {
            int q = queue.v[head++]; /* pop(0) */
            if (!core[q]) continue;  /* :261, R is empty */
            /* R = ids in dict order that are in N_w(q) with pdim <= lambda.  N_w(q) is already in dict order. */
            for (int t = 0; t < nw[q].n; ++t) {
                int x = nw[q].v[t];
                if (pdim[x] > lam) continue;
                if (cls[x] == 0) ivec_push(&queue, x);           /* :116-117 */
                if (cls[x] == 0 || cls[x] == 2) {                /* :118-120 */
                    cls[x] = 1;
                    co_mc *m = s->pcore[x];
                    for (int j = 0; j < d; ++j) c.cf1[j] = c.cf1[j] + m->cf1[j]; /* predecon_mc.py:64-68 */
                    for (int j = 0; j < d; ++j) c.cf2[j] = c.cf2[j] + m->cf2[j];
                    c.w = c.w + m->w;
                    ivec_push(&members, x);
                }
            }
        }
        /* predecon.py:80 update_preferred_dimensions(delta^2, k) on the merged CF; centroid = CF1/W (:68) */
        for (int j = 0; j < d; ++j) c.cen[j] = c.cf1[j] / c.w;
        co_update_pref(c.cf1, c.cf2, c.w, s->p.delta_sq, k, c.pref, d);
        if (c.w > 0) { /* :83 */
            c.n_members = members.n;
            c.members = (int64_t *)malloc(sizeof(int64_t) * (size_t)(members.n ? members.n : 1));
            for (int t = 0; t < members.n; ++t) c.members[t] = s->pcore[members.v[t]]->id;
            if (n_clusters == cap_clusters) {
                cap_clusters = cap_clusters ? cap_clusters * 2 : 16;
                clusters = (co_cluster *)realloc(clusters, sizeof(co_cluster) * (size_t)cap_clusters);
            }
            clusters[n_clusters++] = c;
        } else {
            free(c.cf1);
        }
        free(members.v);
    }
    s->clusters = clusters;
    s->n_clusters = n_clusters;

    for (int i = 0; i < M; ++i) {
        if (out_core) out_core[i] = core[i];
        if (out_pdim) out_pdim[i] = pdim[i];
        if (out_nn) out_nn[i] = nb[i].n;
        if (out_nw) out_nw[i] = nw[i].n;
        free(nb[i].v);
        free(nw[i].v);
    }
    free(nb); free(nw); free(wvec); free(pdim); free(core); free(cls); free(queue.v);
    return n_clusters;
}

int co_num_core(const co_state *s) { return s->n_core_last; }
int co_num_clusters(const co_state *s) { return s->n_clusters; }
int co_cluster_size(const co_state *s, int c) { return s->clusters[c].n_members; }
void co_cluster_export(const co_state *s, int c, int64_t *members, double *w, double *cf1, double *cf2,
                       double *cen, double *pref)
{
    const co_cluster *cl = &s->clusters[c];
    int d = s->d;
    if (members) memcpy(members, cl->members, sizeof(int64_t) * (size_t)cl->n_members);
    if (w) *w = cl->w;
    if (cf1) memcpy(cf1, cl->cf1, sizeof(double) * d);
    if (cf2) memcpy(cf2, cl->cf2, sizeof(double) * d);
    if (cen) memcpy(cen, cl->cen, sizeof(double) * d);
    if (pref) memcpy(pref, cl->pref, sizeof(double) * d);
}

/* ------------------------------------------------------------------ */
/* association tracking: tracking/cluster_tracker.py:127-141           */
/* ------------------------------------------------------------------ */

/* for each current pcore c: argmin over previous pcores (given order) of
 * sum_d (prev_cen - cur_cen)^2 / cur_pref ; strict <, first minimum wins */
void co_assoc_argmin(const double *cur_cen, const double *cur_pref, int mc,
                     const double *prev_cen, int mp, int d, int32_t *out_idx, double *out_dist)
{
    for (int c = 0; c < mc; ++c) {
        int best = -1;
        double bd = 0.0;
        for (int q = 0; q < mp; ++q) {
            double dist = co_projected_distance(cur_cen + (size_t)c * d, cur_pref + (size_t)c * d,
                                                prev_cen + (size_t)q * d, d);
            if (best < 0 || dist < bd) { best = q; bd = dist; }
        }
        out_idx[c] = best;
        if (out_dist) out_dist[c] = bd;
    }
}
