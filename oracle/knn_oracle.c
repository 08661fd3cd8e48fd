/*
 * knn_oracle.c -- CPU restatement of the scan-matching inner loop: exact
 * k-nearest-neighbour correspondence + weighted centroid / 3x3 covariance +
 * SVD rigid solve.  TEST INFRASTRUCTURE ONLY (see gpscal_oracle.h).
 *
 * Third-party algorithm restated (absent from /root/reference):
 *   PCL 1.8.0 pcl::KdTreeFLANN<PointXYZI>::nearestKSearch
 *   (install/install_u1604_basic.sh:32) -> FLANN KDTreeSingleIndex, leaf 15,
 *   exact (eps = 0) L2 search in float32 xyz; call sites
 *   loam/laserOdometry.cpp:603,758 (k=1), loam/laserMapping.cpp:760,867 (k=5).
 *   FLANN leaves tie order unspecified; this restatement orders by (d2, index).
 * The reduction/solve follows trackCalibration::BFTWithWeight
 * (track_calibration.cc:416-526): centroids weighted by w, covariance by w^2.
 */
#include "gpscal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

float orc_sqdist(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* sorted insertion into a k-list ordered by (d2, idx) */
static void klist_insert(int k, int32_t *idx, float *sqd, int *cnt, int32_t j,
                         float d)
{
    int n = *cnt;
    if (n == k) {
        if (d > sqd[k - 1] || (d == sqd[k - 1] && j > idx[k - 1])) return;
    } else {
        ++n;
    }
    int pos = n - 1;
    while (pos > 0 && (sqd[pos - 1] > d || (sqd[pos - 1] == d && idx[pos - 1] > j))) {
        sqd[pos] = sqd[pos - 1];
        idx[pos] = idx[pos - 1];
        --pos;
    }
    sqd[pos] = d;
    idx[pos] = j;
    *cnt = n;
}

int orc_knn_brute(const float *tgt, int m, const float *q, int n, int k,
                  int32_t *idx, float *sqd)
{
    if (k <= 0 || m < 0) return -1;
    for (int i = 0; i < n; ++i) {
        int cnt = 0;
        int32_t *ii = idx + (size_t)i * k;
        float *dd = sqd + (size_t)i * k;
        for (int j = 0; j < m; ++j) {
            const float *c = tgt + 3 * (size_t)j;
            if (!(isfinite(c[0]) && isfinite(c[1]) && isfinite(c[2]))) continue; /* never a neighbour */
            klist_insert(k, ii, dd, &cnt, j, orc_sqdist(q + 3 * i, c));
        }
        for (int r = cnt; r < k; ++r) {
            ii[r] = -1;
            dd[r] = INFINITY;
        }
    }
    return 0;
}

/* ------------------------------------------------------------- kd-tree   */

#define LEAF_MAX 15

typedef struct {
    int left, right; /* children (node ids) or -1 for leaf */
    int begin, end;  /* point range in perm[] for leaves */
    int dim;
    float lo, hi; /* split: left subtree coords <= lo .. right >= hi */
} kd_node;

struct orc_kdtree {
    const float *pts; /* borrowed */
    float *reord;     /* points in perm order (cache-friendly leaves) */
    int *perm;
    kd_node *nodes;
    int nnodes, cap, m;
    float bbmin[3], bbmax[3];
};

static int kd_new_node(orc_kdtree *t)
{
    if (t->nnodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap);
    }
    return t->nnodes++;
}

static int kd_build_rec(orc_kdtree *t, int b, int e)
{
    int id = kd_new_node(t);
    kd_node nd;
    nd.left = nd.right = -1;
    nd.begin = b;
    nd.end = e;
    nd.dim = 0;
    nd.lo = nd.hi = 0.0f;
    if (e - b > LEAF_MAX) {
        float mn[3] = {INFINITY, INFINITY, INFINITY};
        float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = b; i < e; ++i) {
            const float *p = t->pts + 3 * (size_t)t->perm[i];
            for (int d = 0; d < 3; ++d) {
                if (p[d] < mn[d]) mn[d] = p[d];
                if (p[d] > mx[d]) mx[d] = p[d];
            }
        }
        int dim = 0;
        for (int d = 1; d < 3; ++d)
            if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
        if (mx[dim] > mn[dim]) {
            float split = 0.5f * (mn[dim] + mx[dim]);
            /* partition: < split left, >= split right */
            int i = b, j = e - 1;
            while (i <= j) {
                while (i <= j && t->pts[3 * (size_t)t->perm[i] + dim] < split) ++i;
                while (i <= j && t->pts[3 * (size_t)t->perm[j] + dim] >= split) --j;
                if (i < j) {
                    int tmp = t->perm[i];
                    t->perm[i] = t->perm[j];
                    t->perm[j] = tmp;
                    ++i;
                    --j;
                }
            }
            int mid = i;
            if (mid == b || mid == e) mid = (b + e) / 2; /* cannot happen */
            float lmax = -INFINITY, rmin = INFINITY;
            for (int r = b; r < mid; ++r) {
                float v = t->pts[3 * (size_t)t->perm[r] + dim];
                if (v > lmax) lmax = v;
            }
            for (int r = mid; r < e; ++r) {
                float v = t->pts[3 * (size_t)t->perm[r] + dim];
                if (v < rmin) rmin = v;
            }
            nd.dim = dim;
            nd.lo = lmax;
            nd.hi = rmin;
            t->nodes[id] = nd;
            int l = kd_build_rec(t, b, mid);
            int r = kd_build_rec(t, mid, e);
            t->nodes[id].left = l;
            t->nodes[id].right = r;
            return id;
        }
    }
    t->nodes[id] = nd;
    return id;
}

orc_kdtree *orc_kdtree_build(const float *tgt, int m)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof *t);
    t->pts = tgt;
    t->m = m;
    t->perm = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    /* non-finite points are never indexed nor returned (what setInputCloud's IndicesPtr of a cloud that went
     * through removeNaNFromPointCloud amounts to, scanRegistration.cpp:260-263); a NaN would also send the
     * median split below into an endless loop */
    int mf = 0;
    for (int i = 0; i < m; ++i)
        if (isfinite(tgt[3 * (size_t)i]) && isfinite(tgt[3 * (size_t)i + 1]) && isfinite(tgt[3 * (size_t)i + 2]))
            t->perm[mf++] = i;
    t->m = m = mf;
    for (int d = 0; d < 3; ++d) {
        t->bbmin[d] = INFINITY;
        t->bbmax[d] = -INFINITY;
    }
    for (int i = 0; i < m; ++i)
        for (int d = 0; d < 3; ++d) {
            float v = tgt[3 * (size_t)t->perm[i] + d];
            if (v < t->bbmin[d]) t->bbmin[d] = v;
            if (v > t->bbmax[d]) t->bbmax[d] = v;
        }
    if (m > 0) kd_build_rec(t, 0, m);
    t->reord = (float *)malloc(sizeof(float) * 3 * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i)
        memcpy(t->reord + 3 * (size_t)i, tgt + 3 * (size_t)t->perm[i], 3 * sizeof(float));
    return t;
}

void orc_kdtree_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->perm);
    free(t->reord);
    free(t->nodes);
    free(t);
}

typedef struct {
    const orc_kdtree *t;
    const float *q;
    int k, cnt;
    int32_t *idx;
    float *sqd;
} kd_query;

/* The float32 fmaf chain can round a squared distance slightly below the real
 * one, so prune on the exact (double) box bound only beyond this slack; then
 * the result equals brute force bit for bit, ties included. */
#define PRUNE_SLACK (1.0 + 1e-6)

static void kd_search_rec(kd_query *Q, int id, double mind, double off[3])
{
    const kd_node *nd = &Q->t->nodes[id];
    if (nd->left < 0) {
        for (int i = nd->begin; i < nd->end; ++i)
            klist_insert(Q->k, Q->idx, Q->sqd, &Q->cnt, Q->t->perm[i],
                         orc_sqdist(Q->q, Q->t->reord + 3 * (size_t)i));
        return;
    }
    /* FLANN KDTreeSingleIndex::searchLevel: descend into the nearer child,
     * then the other one if the accumulated per-axis box bound allows. */
    int d = nd->dim;
    double v = Q->q[d];
    double diff1 = v - nd->lo; /* lo = max of the left subtree along d  */
    double diff2 = v - nd->hi; /* hi = min of the right subtree along d */
    int first, second;
    double cut;
    if (diff1 + diff2 < 0.0) {
        first = nd->left;
        second = nd->right;
        cut = diff2 * diff2;
    } else {
        first = nd->right;
        second = nd->left;
        cut = diff1 * diff1;
    }
    kd_search_rec(Q, first, mind, off);
    double old = off[d];
    double nmind = mind + cut - old;
    double worst = Q->cnt < Q->k ? INFINITY : (double)Q->sqd[Q->k - 1];
    if (nmind <= worst * PRUNE_SLACK) {
        off[d] = cut;
        kd_search_rec(Q, second, nmind, off);
        off[d] = old;
    }
}

int orc_kdtree_search(const orc_kdtree *t, const float *q, int n, int k,
                      int32_t *idx, float *sqd)
{
    if (!t || k <= 0) return -1;
    for (int i = 0; i < n; ++i) {
        kd_query Q;
        Q.t = t;
        Q.q = q + 3 * (size_t)i;
        Q.k = k;
        Q.cnt = 0;
        Q.idx = idx + (size_t)i * k;
        Q.sqd = sqd + (size_t)i * k;
        if (t->m > 0) {
            double off[3] = {0, 0, 0}, mind = 0.0;
            for (int d = 0; d < 3; ++d) {
                double v = Q.q[d];
                if (v < t->bbmin[d]) off[d] = (t->bbmin[d] - v) * (t->bbmin[d] - v);
                if (v > t->bbmax[d]) off[d] = (v - t->bbmax[d]) * (v - t->bbmax[d]);
                mind += off[d];
            }
            kd_search_rec(&Q, 0, mind, off);
        }
        for (int r = Q.cnt; r < k; ++r) {
            Q.idx[r] = -1;
            Q.sqd[r] = INFINITY;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------- ICP  */

static void t32_from_T(const double T[16], float r[9], float tr[3])
{
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r[3 * i + j] = (float)T[4 * i + j];
        tr[i] = (float)T[4 * i + 3];
    }
}

static void xform1(const float r[9], const float tr[3], const float *s, float *p)
{
    for (int i = 0; i < 3; ++i)
        p[i] = fmaf(r[3 * i + 0], s[0],
                    fmaf(r[3 * i + 1], s[1], fmaf(r[3 * i + 2], s[2], tr[i])));
}

void orc_transform_f32(const double T[16], const float *src, int n, float *dst)
{
    float r[9], tr[3];
    t32_from_T(T, r, tr);
    for (int i = 0; i < n; ++i) xform1(r, tr, src + 3 * (size_t)i, dst + 3 * (size_t)i);
}

int orc_icp_iterate(const orc_kdtree *t, const float *src, int n,
                    const double *w, const double T_in[16], double T_out[16],
                    double *mean_err, int32_t *idx_out, float *sqd_out)
{
    if (!t || n <= 0 || t->m <= 0) return -1;
    float r[9], tr[3];
    t32_from_T(T_in, r, tr);
    float *P = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    int32_t *idx = idx_out ? idx_out : (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    float *sqd = sqd_out ? sqd_out : (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) xform1(r, tr, src + 3 * (size_t)i, P + 3 * (size_t)i);
    orc_kdtree_search(t, P, n, 1, idx, sqd);

    double sw = 0, sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, serr = 0;
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        const float *p = P + 3 * (size_t)i;
        const float *qq = t->pts + 3 * (size_t)idx[i];
        for (int d = 0; d < 3; ++d) {
            sp[d] += (double)p[d] * wi;
            sq[d] += (double)qq[d] * wi;
        }
        sw += wi;
        serr += sqrt((double)sqd[i]);
    }
    double cp[3], cq[3];
    for (int d = 0; d < 3; ++d) {
        cp[d] = sp[d] / sw;
        cq[d] = sq[d] / sw;
    }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        const float *p = P + 3 * (size_t)i;
        const float *qq = t->pts + 3 * (size_t)idx[i];
        double a[3], b[3];
        for (int d = 0; d < 3; ++d) {
            a[d] = ((double)p[d] - cp[d]) * wi;
            b[d] = ((double)qq[d] - cq[d]) * wi;
        }
        for (int rr = 0; rr < 3; ++rr)
            for (int c = 0; c < 3; ++c) H[3 * rr + c] += a[rr] * b[c];
    }
    double R[9];
    orc_kabsch_from_H(H, R);
    double dT[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) {
        double rc = 0;
        for (int k = 0; k < 3; ++k) rc += R[3 * i + k] * cp[k];
        for (int j = 0; j < 3; ++j) dT[4 * i + j] = R[3 * i + j];
        dT[4 * i + 3] = cq[i] - rc;
    }
    double To[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += dT[4 * i + k] * T_in[4 * k + j];
            To[4 * i + j] = acc;
        }
    memcpy(T_out, To, sizeof To);
    if (mean_err) *mean_err = serr / (double)n;
    free(P);
    if (!idx_out) free(idx);
    if (!sqd_out) free(sqd);
    return 0;
}

int orc_icp_run(const orc_kdtree *t, const float *src, int n, const double *w,
                int iters, const double T0[16], double T_out[16],
                double *hist)
{
    double T[16];
    memcpy(T, T0, sizeof T);
    for (int it = 0; it < iters; ++it) {
        double e = 0, Tn[16];
        int rc = orc_icp_iterate(t, src, n, w, T, Tn, &e, NULL, NULL);
        if (rc) return rc;
        memcpy(T, Tn, sizeof T);
        if (hist) hist[it] = e;
    }
    memcpy(T_out, T, sizeof T);
    return 0;
}
