/*
 * lisreg_oracle.c — CPU restatement of LIS-SLAM's LOAM-style scan-to-submap registration.
 * TEST INFRASTRUCTURE ONLY (see lisreg_oracle.h for the rules and the "parity unpinned" statement).
 *
 * Follows, operation for operation, /root/reference/src/node/odomEstimationNode.cpp:596-1006 (copy #1) and the
 * label-weighted copies src/node/subMapOptmizationNode.cpp:1509-2001 / :4485-4977; pose algebra from
 * src/core/common.cpp:54-57,285-291.  All arithmetic is float unless the reference's C++ promotes to double
 * (the 0.1 / 0.9 / 2.0 literals, pow(), tf's double quaternions) — those promotions are reproduced.
 * Compile WITHOUT -ffast-math and with -ffp-contract=off (x86-64 reference build has no FMA contraction).
 */
#include "lisreg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------- */
/* pose algebra — pcl::getTransformation via trans2Affine3f (common.cpp:54-57)                               */
/* ------------------------------------------------------------------------------------------------------- */
void orc_pose_to_matrix(const float T[6], float M[12])
{
    /* PCL common/impl/eigen.hpp getTransformation(x,y,z,roll,pitch,yaw): Scalar = float */
    float A = cosf(T[2]), B = sinf(T[2]);   /* yaw   */
    float C = cosf(T[1]), D = sinf(T[1]);   /* pitch */
    float E = cosf(T[0]), F = sinf(T[0]);   /* roll  */
    float DE = D * E, DF = D * F;
    M[0] = A * C;  M[1] = A * DF - B * E;  M[2]  = B * F + A * DE;  M[3]  = T[3];
    M[4] = B * C;  M[5] = A * E + B * DF;  M[6]  = B * DE - A * F;  M[7]  = T[4];
    M[8] = -D;     M[9] = C * F;           M[10] = C * E;           M[11] = T[5];
}

/* pointAssociateToMap (odomEstimationNode.cpp:243-258) */
static void transform_point(const float M[12], const float p[3], float o[3])
{
    o[0] = M[0] * p[0] + M[1] * p[1] + M[2]  * p[2] + M[3];
    o[1] = M[4] * p[0] + M[5] * p[1] + M[6]  * p[2] + M[7];
    o[2] = M[8] * p[0] + M[9] * p[1] + M[10] * p[2] + M[11];
}

/* ------------------------------------------------------------------------------------------------------- */
/* exact k-NN: kd-tree in the shape of FLANN's KDTreeSingleIndex (leaf 15, reordered points, box pruning)   */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct orc_node {
    int   left, right;      /* child node ids, -1 for leaf                 */
    int   lo, hi;           /* leaf: point range [lo,hi) in reordered array */
    int   dim;
    float divlow, divhigh;
} orc_node;

struct orc_kdtree {
    int       n, leaf;
    float*    pts;          /* reordered xyz */
    int*      perm;         /* reordered -> original index */
    orc_node* nodes;
    int       n_nodes, cap_nodes;
    float     bbmin[3], bbmax[3];
};

static void swap_int(int* a, int* b) { int t = *a; *a = *b; *b = t; }

/* quickselect on indices by coordinate `dim` so that ind[lo..mid) <= ind[mid] <= ind(mid..hi) */
static void select_nth(const float* xyz, int* ind, int lo, int hi, int mid, int dim)
{
    while (hi - lo > 1) {
        int a = lo, c = hi - 1, b = lo + (hi - lo) / 2;
        /* median of three -> pivot value */
        float va = xyz[3 * ind[a] + dim], vb = xyz[3 * ind[b] + dim], vc = xyz[3 * ind[c] + dim];
        float pv = (va < vb) ? ((vb < vc) ? vb : (va < vc ? vc : va)) : ((va < vc) ? va : (vb < vc ? vc : vb));
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (xyz[3 * ind[i] + dim] < pv) i++;
            while (xyz[3 * ind[j] + dim] > pv) j--;
            if (i <= j) { swap_int(&ind[i], &ind[j]); i++; j--; }
        }
        if (mid <= j) hi = j + 1;
        else if (mid >= i) lo = i;
        else return;
    }
}

static int build_rec(orc_kdtree* t, const float* xyz, int* ind, int lo, int hi)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes * 2 + 16;
        t->nodes = (orc_node*)realloc(t->nodes, sizeof(orc_node) * (size_t)t->cap_nodes);
    }
    int id = t->n_nodes++;
    orc_node nd; memset(&nd, 0, sizeof nd);
    nd.left = nd.right = -1; nd.lo = lo; nd.hi = hi;
    if (hi - lo <= t->leaf) { t->nodes[id] = nd; return id; }
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = lo; i < hi; ++i)
        for (int d = 0; d < 3; ++d) {
            float v = xyz[3 * ind[i] + d];
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    int dim = 0; float ext = mx[0] - mn[0];
    for (int d = 1; d < 3; ++d) if (mx[d] - mn[d] > ext) { ext = mx[d] - mn[d]; dim = d; }
    int mid = lo + (hi - lo) / 2;
    select_nth(xyz, ind, lo, hi, mid, dim);
    float dl = -FLT_MAX, dh = FLT_MAX;
    for (int i = lo; i < mid; ++i) { float v = xyz[3 * ind[i] + dim]; if (v > dl) dl = v; }
    for (int i = mid; i < hi; ++i) { float v = xyz[3 * ind[i] + dim]; if (v < dh) dh = v; }
    nd.dim = dim; nd.divlow = dl; nd.divhigh = dh;
    t->nodes[id] = nd;
    int l = build_rec(t, xyz, ind, lo, mid);
    int r = build_rec(t, xyz, ind, mid, hi);
    t->nodes[id].left = l; t->nodes[id].right = r;
    return id;
}

orc_kdtree* orc_kdtree_build(const float* xyz, int n, int leaf_size)
{
    orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof *t);
    t->n = n; t->leaf = leaf_size > 0 ? leaf_size : 15;
    t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    t->pts = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) t->perm[i] = i;
    for (int d = 0; d < 3; ++d) { t->bbmin[d] = FLT_MAX; t->bbmax[d] = -FLT_MAX; }
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            float v = xyz[3 * i + d];
            if (v < t->bbmin[d]) t->bbmin[d] = v;
            if (v > t->bbmax[d]) t->bbmax[d] = v;
        }
    if (n > 0) build_rec(t, xyz, t->perm, 0, n);
    for (int i = 0; i < n; ++i) memcpy(&t->pts[3 * i], &xyz[3 * t->perm[i]], 3 * sizeof(float));
    return t;
}

void orc_kdtree_free(orc_kdtree* t)
{
    if (!t) return;
    free(t->pts); free(t->perm); free(t->nodes); free(t);
}

typedef struct { int k, cnt; int* idx; float* sqd; } knn_set;

static inline float knn_worst(const knn_set* s) { return s->cnt < s->k ? FLT_MAX : s->sqd[s->k - 1]; }

static inline void knn_add(knn_set* s, float d, int id)
{
    /* sorted insertion, ascending by (distance, index).  FLANN keeps equal distances in traversal order, i.e. which of two exactly
     * equidistant points is the fifth neighbour is an accident of the tree; the restatement fixes it to the smaller index so that
     * the kd-tree search, the brute-force search and every GPU front-end agree on ties (about one query in 10^5) */
    int i;
    if (s->cnt < s->k) i = s->cnt++;
    else {
        const float w = s->sqd[s->k - 1];
        if (!(d < w || (d == w && id < s->idx[s->k - 1]))) return;
        i = s->k - 1;
    }
    while (i > 0 && (s->sqd[i - 1] > d || (s->sqd[i - 1] == d && s->idx[i - 1] > id))) { s->sqd[i] = s->sqd[i - 1]; s->idx[i] = s->idx[i - 1]; --i; }
    s->sqd[i] = d; s->idx[i] = id;
}

/* squared L2 as flann::L2_Simple<float>: sequential float accumulation over x,y,z */
static inline float sqdist3(const float* a, const float* b)
{
    float r = 0.f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

static void search_rec(const orc_kdtree* t, int id, const float q[3], float mindist, float dists[3], knn_set* s)
{
    const orc_node* nd = &t->nodes[id];
    if (nd->left < 0) {
        for (int i = nd->lo; i < nd->hi; ++i) {
            float d = sqdist3(q, &t->pts[3 * i]);
            if (d <= knn_worst(s) || s->cnt < s->k) knn_add(s, d, t->perm[i]);
        }
        return;
    }
    int dim = nd->dim;
    float val = q[dim];
    float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other; float cut;
    if (diff1 + diff2 < 0) { best = nd->left;  other = nd->right; cut = diff2 * diff2; }
    else                   { best = nd->right; other = nd->left;  cut = diff1 * diff1; }
    search_rec(t, best, q, mindist, dists, s);
    float dst = dists[dim];
    mindist = mindist + cut - dst;
    dists[dim] = cut;
    if (mindist <= knn_worst(s)) search_rec(t, other, q, mindist, dists, s);
    dists[dim] = dst;
}

int orc_kdtree_knn(const orc_kdtree* t, const float q[3], int k, int* idx, float* sqd)
{
    knn_set s = { k, 0, idx, sqd };
    if (t->n == 0) return 0;
    float dists[3] = { 0, 0, 0 }, mind = 0;
    for (int d = 0; d < 3; ++d) {
        if (q[d] < t->bbmin[d]) { dists[d] = (q[d] - t->bbmin[d]) * (q[d] - t->bbmin[d]); mind += dists[d]; }
        if (q[d] > t->bbmax[d]) { dists[d] = (q[d] - t->bbmax[d]) * (q[d] - t->bbmax[d]); mind += dists[d]; }
    }
    search_rec(t, 0, q, mind, dists, &s);
    return s.cnt;
}

int orc_bruteforce_knn(const float* xyz, int n, const float q[3], int k, int* idx, float* sqd)
{
    knn_set s = { k, 0, idx, sqd };
    for (int i = 0; i < n; ++i) {
        float d = sqdist3(q, &xyz[3 * i]);
        if (s.cnt < k || d <= knn_worst(&s)) knn_add(&s, d, i);
    }
    return s.cnt;
}

/* ------------------------------------------------------------------------------------------------------- */
/* small dense algebra                                                                                       */
/* ------------------------------------------------------------------------------------------------------- */
static inline float hypot_f(float a, float b)
{
    a = fabsf(a); b = fabsf(b);
    if (a > b) { b /= a; return a * sqrtf(1 + b * b); }
    if (b > 0) { a /= b; return b * sqrtf(1 + a * a); }
    return 0.f;
}

/* cv::eigen for symmetric CV_32F: classical Jacobi with largest-off-diagonal pivoting, then a descending
 * selection sort that swaps eigenvector ROWS (call sites odomEstimationNode.cpp:690, :928). */
void orc_eigen_sym(const float* Ain, int n, float* W, float* V)
{
    float A[36];
    int indR[6], indC[6];
    memcpy(A, Ain, sizeof(float) * (size_t)(n * n));
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.f : 0.f;
    for (int k = 0; k < n; ++k) {
        W[k] = A[k * n + k];
        if (k < n - 1) {
            int m = k + 1; float mv = fabsf(A[k * n + m]);
            for (int i = k + 2; i < n; ++i) { float v = fabsf(A[k * n + i]); if (mv < v) { mv = v; m = i; } }
            indR[k] = m;
        }
        if (k > 0) {
            int m = 0; float mv = fabsf(A[k]);
            for (int i = 1; i < k; ++i) { float v = fabsf(A[i * n + k]); if (mv < v) { mv = v; m = i; } }
            indC[k] = m;
        }
    }
    if (n > 1) {
        int maxIters = n * n * 30;
        for (int it = 0; it < maxIters; ++it) {
            int k = 0, l; float mv = fabsf(A[indR[0]]);
            for (int i = 1; i < n - 1; ++i) { float v = fabsf(A[i * n + indR[i]]); if (mv < v) { mv = v; k = i; } }
            l = indR[k];
            for (int i = 1; i < n; ++i) { float v = fabsf(A[indC[i] * n + i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
            float p = A[k * n + l];
            if (fabsf(p) <= FLT_EPSILON) break;
            float y = (W[l] - W[k]) * 0.5f;
            float t = fabsf(y) + hypot_f(p, y);
            float s = hypot_f(p, t);
            float c = t / s;
            s = p / s; t = (p / t) * p;
            if (y < 0) { s = -s; t = -t; }
            A[k * n + l] = 0;
            W[k] -= t; W[l] += t;
#define ORC_ROT(v0, v1) do { float a0_ = (v0), b0_ = (v1); (v0) = a0_ * c - b0_ * s; (v1) = a0_ * s + b0_ * c; } while (0)
            for (int i = 0; i < k; ++i)      ORC_ROT(A[i * n + k], A[i * n + l]);
            for (int i = k + 1; i < l; ++i)  ORC_ROT(A[k * n + i], A[i * n + l]);
            for (int i = l + 1; i < n; ++i)  ORC_ROT(A[k * n + i], A[l * n + i]);
            for (int i = 0; i < n; ++i)      ORC_ROT(V[k * n + i], V[l * n + i]);
#undef ORC_ROT
            for (int j = 0; j < 2; ++j) {
                int idx = j == 0 ? k : l;
                if (idx < n - 1) {
                    int m = idx + 1; float mv2 = fabsf(A[idx * n + m]);
                    for (int i = idx + 2; i < n; ++i) { float v = fabsf(A[idx * n + i]); if (mv2 < v) { mv2 = v; m = i; } }
                    indR[idx] = m;
                }
                if (idx > 0) {
                    int m = 0; float mv2 = fabsf(A[idx]);
                    for (int i = 1; i < idx; ++i) { float v = fabsf(A[i * n + idx]); if (mv2 < v) { mv2 = v; m = i; } }
                    indC[idx] = m;
                }
            }
        }
    }
    for (int k = 0; k < n - 1; ++k) {
        int m = k;
        for (int i = k + 1; i < n; ++i) if (W[m] < W[i]) m = i;
        if (m != k) {
            float tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (int i = 0; i < n; ++i) { float tv = V[m * n + i]; V[m * n + i] = V[k * n + i]; V[k * n + i] = tv; }
        }
    }
}

/* Eigen::Matrix<float,5,3>::colPivHouseholderQr().solve(b) (odomEstimationNode.cpp:783) */
void orc_lstsq5x3(const float Ain[15], const float bin[5], float x[3])
{
    enum { R = 5, C = 3 };
    float A[R][C], c[R];
    int perm[C] = { 0, 1, 2 };
    for (int i = 0; i < R; ++i) { c[i] = bin[i]; for (int j = 0; j < C; ++j) A[i][j] = Ain[i * C + j]; }
    float maxn = 0.f;
    for (int j = 0; j < C; ++j) {
        float s = 0.f; for (int i = 0; i < R; ++i) s += A[i][j] * A[i][j];
        s = sqrtf(s); if (s > maxn) maxn = s;
    }
    float thr = (maxn * FLT_EPSILON) * (maxn * FLT_EPSILON) / (float)R;
    int rank = C;
    for (int k = 0; k < C; ++k) {
        int bj = k; float bn = -1.f;
        for (int j = k; j < C; ++j) {
            float s = 0.f; for (int i = k; i < R; ++i) s += A[i][j] * A[i][j];
            if (s > bn) { bn = s; bj = j; }
        }
        if (bn < thr * (float)(R - k)) { rank = k; break; }
        if (bj != k) {
            for (int i = 0; i < R; ++i) { float t = A[i][k]; A[i][k] = A[i][bj]; A[i][bj] = t; }
            int tp = perm[k]; perm[k] = perm[bj]; perm[bj] = tp;
        }
        /* Householder reflector for column k, rows k..R-1 (Eigen makeHouseholder) */
        float c0 = A[k][k], tail = 0.f;
        for (int i = k + 1; i < R; ++i) tail += A[i][k] * A[i][k];
        float beta, tau, v[R];
        if (tail <= FLT_MIN) { tau = 0.f; beta = c0; for (int i = k + 1; i < R; ++i) v[i] = 0.f; }
        else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
            for (int i = k + 1; i < R; ++i) v[i] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        for (int i = k + 1; i < R; ++i) A[i][k] = 0.f;
        if (tau != 0.f) {
            for (int j = k + 1; j < C; ++j) {
                float dot = A[k][j];
                for (int i = k + 1; i < R; ++i) dot += v[i] * A[i][j];
                dot *= tau;
                A[k][j] -= dot;
                for (int i = k + 1; i < R; ++i) A[i][j] -= dot * v[i];
            }
            float dot = c[k];
            for (int i = k + 1; i < R; ++i) dot += v[i] * c[i];
            dot *= tau;
            c[k] -= dot;
            for (int i = k + 1; i < R; ++i) c[i] -= dot * v[i];
        }
    }
    float y[C] = { 0, 0, 0 };
    for (int i = rank - 1; i >= 0; --i) {
        float s = c[i];
        for (int j = i + 1; j < rank; ++j) s -= A[i][j] * y[j];
        y[i] = s / A[i][i];
    }
    x[0] = x[1] = x[2] = 0.f;
    for (int i = 0; i < rank; ++i) x[perm[i]] = y[i];
}

/* cv::solve(AtA, AtB, X, DECOMP_QR) for 6x6 float (odomEstimationNode.cpp:921): Householder QR, no pivoting */
int orc_solve6(const float Ain[36], const float bin[6], float x[6])
{
    enum { N = 6 };
    float A[N][N], c[N];
    for (int i = 0; i < N; ++i) { c[i] = bin[i]; for (int j = 0; j < N; ++j) A[i][j] = Ain[i * N + j]; }
    for (int k = 0; k < N; ++k) {
        float c0 = A[k][k], tail = 0.f, v[N];
        for (int i = k + 1; i < N; ++i) tail += A[i][k] * A[i][k];
        float beta, tau;
        if (tail <= FLT_MIN) { tau = 0.f; beta = c0; for (int i = k + 1; i < N; ++i) v[i] = 0.f; }
        else {
            beta = sqrtf(c0 * c0 + tail);
            if (c0 >= 0.f) beta = -beta;
            for (int i = k + 1; i < N; ++i) v[i] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        if (tau != 0.f) {
            for (int j = k + 1; j < N; ++j) {
                float dot = A[k][j];
                for (int i = k + 1; i < N; ++i) dot += v[i] * A[i][j];
                dot *= tau;
                A[k][j] -= dot;
                for (int i = k + 1; i < N; ++i) A[i][j] -= dot * v[i];
            }
            float dot = c[k];
            for (int i = k + 1; i < N; ++i) dot += v[i] * c[i];
            dot *= tau;
            c[k] -= dot;
            for (int i = k + 1; i < N; ++i) c[i] -= dot * v[i];
        }
    }
    for (int i = 0; i < N; ++i) if (fabsf(A[i][i]) <= FLT_MIN) { memset(x, 0, sizeof(float) * N); return 0; }
    for (int i = N - 1; i >= 0; --i) {
        float s = c[i];
        for (int j = i + 1; j < N; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return 1;
}

/* cv::Mat::inv() default DECOMP_LU for 6x6 float (odomEstimationNode.cpp:945) */
int orc_inv6(const float Ain[36], float out[36])
{
    enum { N = 6 };
    float A[N][N], B[N][N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { A[i][j] = Ain[i * N + j]; B[i][j] = (i == j) ? 1.f : 0.f; }
    for (int i = 0; i < N; ++i) {
        int k = i;
        for (int j = i + 1; j < N; ++j) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < FLT_EPSILON * 100) { memset(out, 0, sizeof(float) * 36); return 0; }
        if (k != i) for (int j = 0; j < N; ++j) {
            float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t;
            t = B[i][j]; B[i][j] = B[k][j]; B[k][j] = t;
        }
        float d = -1.f / A[i][i];
        for (int j = i + 1; j < N; ++j) {
            float alpha = A[j][i] * d;
            for (int m = i + 1; m < N; ++m) A[j][m] += alpha * A[i][m];
            for (int m = 0; m < N; ++m) B[j][m] += alpha * B[i][m];
        }
    }
    for (int i = N - 1; i >= 0; --i)
        for (int j = 0; j < N; ++j) {
            float s = B[i][j];
            for (int k = i + 1; k < N; ++k) s -= A[i][k] * B[k][j];
            B[i][j] = s / A[i][i];
        }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) out[i * N + j] = B[i][j];
    return 1;
}

/* ------------------------------------------------------------------------------------------------------- */
/* per-point residual models                                                                                 */
/* ------------------------------------------------------------------------------------------------------- */
int orc_corner_coeff(const float nb[15], const float ps[3], float w, const lisreg_params* p, float coeff[4])
{
    /* odomEstimationNode.cpp:658-678 centroid and covariance of the 5 neighbours (float, sequential) */
    float cx = 0, cy = 0, cz = 0;
    for (int j = 0; j < 5; ++j) { cx += nb[3 * j]; cy += nb[3 * j + 1]; cz += nb[3 * j + 2]; }
    cx /= 5; cy /= 5; cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
    for (int j = 0; j < 5; ++j) {
        float ax = nb[3 * j] - cx, ay = nb[3 * j + 1] - cy, az = nb[3 * j + 2] - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az; a22 += ay * ay; a23 += ay * az; a33 += az * az;
    }
    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
    float A[9] = { a11, a12, a13, a12, a22, a23, a13, a23, a33 }, D[3], V[9];
    orc_eigen_sym(A, 3, D, V);                                            /* :690 */
    if (!(D[0] > p->line_ratio * D[1])) return 0;                         /* :692 */
    float x0 = ps[0], y0 = ps[1], z0 = ps[2];
    /* :697-702 — `0.1 * float` is double arithmetic, truncated on assignment */
    float x1 = (float)((double)cx + 0.1 * (double)V[0]);
    float y1 = (float)((double)cy + 0.1 * (double)V[1]);
    float z1 = (float)((double)cz + 0.1 * (double)V[2]);
    float x2 = (float)((double)cx - 0.1 * (double)V[0]);
    float y2 = (float)((double)cy - 0.1 * (double)V[1]);
    float z2 = (float)((double)cz - 0.1 * (double)V[2]);
    float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
    float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
    float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
    float a012 = sqrtf(m11 * m11 + m22 * m22 + m33 * m33);               /* :704-709 */
    float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
    float la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;         /* :713-715 */
    float lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;        /* :717-719 */
    float lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;        /* :721-723 */
    float ld2 = a012 / l12;
    float s = (float)(1.0 - 0.9 * (double)fabsf(ld2));                   /* :727 */
    coeff[0] = w * s * la; coeff[1] = w * s * lb; coeff[2] = w * s * lc; coeff[3] = w * s * ld2;
    return s > p->accept_s;                                              /* :734 (uses s, not w*s) */
}

int orc_surf_coeff(const float nb[15], const float ps[3], float w, const lisreg_params* p, float coeff[4])
{
    float b[5] = { -1, -1, -1, -1, -1 }, X[3];
    orc_lstsq5x3(nb, b, X);                                               /* :783 */
    float pa = X[0], pb = X[1], pc = X[2], pd = 1;
    float psn = sqrtf(pa * pa + pb * pb + pc * pc);
    pa /= psn; pb /= psn; pc /= psn; pd /= psn;                           /* :790-791 */
    for (int j = 0; j < 5; ++j)                                           /* :794-802 */
        if (fabsf(pa * nb[3 * j] + pb * nb[3 * j + 1] + pc * nb[3 * j + 2] + pd) > p->plane_tol) return 0;
    float pd2 = pa * ps[0] + pb * ps[1] + pc * ps[2] + pd;
    float rng = sqrtf(sqrtf(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2]));
    float s = (float)(1.0 - 0.9 * (double)fabsf(pd2) / (double)rng);      /* :807 */
    coeff[0] = w * s * pa; coeff[1] = w * s * pb; coeff[2] = w * s * pc; coeff[3] = w * s * pd2;
    return s > p->accept_s;                                              /* :814 */
}

void orc_jacobian_row(const float T[6], const float ori[3], const float cf[4], float row[6], float* b)
{
    /* LMOptimization :862-867: lidar -> camera angle naming */
    float srx = sinf(T[1]), crx = cosf(T[1]);
    float sry = sinf(T[2]), cry = cosf(T[2]);
    float srz = sinf(T[0]), crz = cosf(T[0]);
    float px = ori[1], py = ori[2], pz = ori[0];            /* :889-891 */
    float cx = cf[1], cy = cf[2], cz = cf[0];               /* :893-895 */
    float arx = (crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * cx +
                (-srx * srz * px - crz * srx * py - crx * pz) * cy +
                (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * cz;
    float ary = ((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * cx +
                ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * cz;
    float arz = ((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * cx +
                (crx * crz * px - crx * srz * py) * cy +
                ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * cz;
    row[0] = arz; row[1] = arx; row[2] = ary; row[3] = cz; row[4] = cx; row[5] = cy;   /* :909-914 */
    *b = -cf[3];                                                                         /* :915 */
}

/* ------------------------------------------------------------------------------------------------------- */
/* transformUpdate — tf quaternion slerp restated in double (odomEstimationNode.cpp:976-1006)                */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct { double x, y, z, w; } quat;

static quat q_from_rpy(double roll, double pitch, double yaw)
{
    double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    quat q = { sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
               cr * cp * cy + sr * sp * sy };
    return q;
}
static double q_dot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

static quat q_slerp(quat a, quat b, double t)
{
    double s = sqrt(q_dot(a, a) * q_dot(b, b));
    double d = q_dot(a, b);
    double theta = (d < 0 ? acos(-d / s) * 2.0 : acos(d / s) * 2.0) / 2.0;   /* angleShortestPath / 2 */
    if (theta != 0.0) {
        double dd = 1.0 / sin(theta), s0 = sin((1.0 - t) * theta), s1 = sin(t * theta);
        quat r;
        if (d < 0) { r.x = (a.x * s0 + -b.x * s1) * dd; r.y = (a.y * s0 + -b.y * s1) * dd;
                     r.z = (a.z * s0 + -b.z * s1) * dd; r.w = (a.w * s0 + -b.w * s1) * dd; }
        else       { r.x = (a.x * s0 + b.x * s1) * dd;  r.y = (a.y * s0 + b.y * s1) * dd;
                     r.z = (a.z * s0 + b.z * s1) * dd;  r.w = (a.w * s0 + b.w * s1) * dd; }
        return r;
    }
    return a;
}

static void q_get_rpy(quat q, double* roll, double* pitch, double* yaw)
{
    /* tf::Matrix3x3::setRotation + getEulerYPR (solution 1) */
    double d = q_dot(q, q), s = 2.0 / d;
    double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    double m00 = 1.0 - (yy + zz), m01 = xy - wz;
    double m10 = xy + wz;
    double m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (fabs(m20) >= 1) {
        *yaw = 0;
        double delta = atan2(m21, m22);   /* tf: atan2(m[2].y, m[2].z) */
        if (m20 < 0) { *pitch = M_PI / 2.0; *roll = delta; }
        else         { *pitch = -M_PI / 2.0; *roll = delta; }
        (void)m01;
    } else {
        *pitch = -asin(m20);
        double cp = cos(*pitch);
        *roll = atan2(m21 / cp, m22 / cp);
        *yaw = atan2(m10 / cp, m00 / cp);
    }
}

static float clampf(float v, float lim)   /* constraintTransformation, common.cpp:285-291 */
{
    if (v < -lim) v = -lim;
    if (v > lim) v = lim;
    return v;
}

void orc_transform_update(const lisreg_params* p, const lisreg_imu* imu, float T[6])
{
    if (p->use_imu_blend && imu && imu->imu_available) {
        if (fabsf(imu->imu_pitch_init) < 1.4f) {                 /* std::abs(float) < 1.4 */
            double w = (double)p->imu_rpy_weight, r, pi, y;
            quat tq = q_from_rpy((double)T[0], 0, 0), iq = q_from_rpy((double)imu->imu_roll_init, 0, 0);
            q_get_rpy(q_slerp(tq, iq, w), &r, &pi, &y);
            T[0] = (float)r;
            tq = q_from_rpy(0, (double)T[1], 0); iq = q_from_rpy(0, (double)imu->imu_pitch_init, 0);
            q_get_rpy(q_slerp(tq, iq, w), &r, &pi, &y);
            T[1] = (float)pi;
        }
    }
    T[0] = clampf(T[0], p->rotation_tol);
    T[1] = clampf(T[1], p->rotation_tol);
    T[5] = clampf(T[5], p->z_tol);
}

/* ------------------------------------------------------------------------------------------------------- */
/* cloud access                                                                                               */
/* ------------------------------------------------------------------------------------------------------- */
static void unpack_cloud(const void* cloud, int n, int stride, int fmt, float* xyz, unsigned short* label)
{
    const unsigned char* b = (const unsigned char*)cloud;
    for (int i = 0; i < n; ++i) {
        const unsigned char* r = b + (size_t)i * (size_t)stride;
        memcpy(&xyz[3 * i], r, 12);
        unsigned short l = 0;
        if (fmt == LISREG_FMT_XYZIL) memcpy(&l, r + 20, 2);
        if (label) label[i] = l;
    }
}

static float label_weight(const lisreg_params* p, unsigned short label)
{
    if (!p->use_label_weight) return 1.f;
    /* subMapOptmizationNode.cpp:1671: LabelSorce is a std::map<uint16_t,float> read with operator[] — a label outside
     * config/label.yaml:214-234 default-constructs 0.f, so its weight is 2.0 (label_score[20..31] are 0 by default) */
    return label < 32 ? (float)(2.0 - (double)p->label_score[label]) : 2.0f;
}

typedef struct {
    int n; float* xyz; unsigned short* label; orc_kdtree* tree;
} cloud_t;

static void cloud_init(cloud_t* c, const void* data, int n, int stride, int fmt)
{
    c->n = n;
    c->xyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    c->label = (unsigned short*)malloc(sizeof(unsigned short) * (size_t)(n > 0 ? n : 1));
    c->tree = NULL;
    if (n > 0) unpack_cloud(data, n, stride, fmt, c->xyz, c->label);
}
static void cloud_free(cloud_t* c) { free(c->xyz); free(c->label); orc_kdtree_free(c->tree); }

static int knn5(const cloud_t* tgt, int use_tree, const float q[3], int idx[5], float sqd[5])
{
    if (use_tree && tgt->tree) return orc_kdtree_knn(tgt->tree, q, 5, idx, sqd);
    return orc_bruteforce_knn(tgt->xyz, tgt->n, q, 5, idx, sqd);
}

/* one stage (cornerOptimization :633-747 or surfOptimization :749-827) over all source points */
static void run_stage(int kind, const cloud_t* tgt, const cloud_t* src, const lisreg_params* p,
                      const float M[12], int use_tree, int n_threads, unsigned char* flags, float* coeffs)
{
    (void)n_threads;
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int i = 0; i < src->n; ++i) {
        float psel[3], nb[15], sqd[5], cf[4];
        int idx[5];
        flags[i] = 0;
        transform_point(M, &src->xyz[3 * i], psel);
        int found = knn5(tgt, use_tree, psel, idx, sqd);
        /* copy #1 reads pointSearchSqDis[4] unguarded (UB with < 5 target points); #2/#3 test size()==5.
         * The restatement defines the <5 case as "no correspondence" for all variants. */
        if (found < 5 || !(sqd[4] < p->knn_sq_thresh)) continue;
        for (int j = 0; j < 5; ++j) memcpy(&nb[3 * j], &tgt->xyz[3 * idx[j]], 12);
        float w = label_weight(p, src->label[i]);
        int ok = kind == 0 ? orc_corner_coeff(nb, psel, w, p, cf) : orc_surf_coeff(nb, psel, w, p, cf);
        if (ok) { flags[i] = 1; memcpy(&coeffs[4 * i], cf, 16); }
    }
}

void orc_stage_coeffs(int kind, const void* tgt, int n_t, const void* src, int n_s, int stride, int fmt,
                      const lisreg_params* params, const float T[6], unsigned char* flags, float* coeffs)
{
    cloud_t t, s; float M[12];
    cloud_init(&t, tgt, n_t, stride, fmt);
    cloud_init(&s, src, n_s, stride, fmt);
    t.tree = orc_kdtree_build(t.xyz, t.n, 15);
    orc_pose_to_matrix(T, M);
    memset(coeffs, 0, sizeof(float) * 4 * (size_t)n_s);
    run_stage(kind, &t, &s, params, M, 1, 1, flags, coeffs);
    cloud_free(&t); cloud_free(&s);
}

/* ------------------------------------------------------------------------------------------------------- */
/* scan2SubMapOptimization (odomEstimationNode.cpp:596-626) and LMOptimization (:852-974)                    */
/* ------------------------------------------------------------------------------------------------------- */
int orc_align(const void* tgt_corner, int n_tc, const void* tgt_surf, int n_ts,
              const void* src_corner, int n_sc, const void* src_surf, int n_ss,
              int stride, int fmt, const lisreg_params* p, const lisreg_imu* imu,
              float T[6], int* degenerate, lisreg_stats* stats,
              float* trace, int max_trace, int n_threads, int use_kdtree)
{
    lisreg_stats st; memset(&st, 0, sizeof st);
    st.deltaR = 100.f; st.deltaT = 100.f;                     /* member initialisers :70-71 */
    int isDeg = degenerate ? *degenerate : 0;
    st.degenerate = isDeg;
    if (!(n_sc > p->edge_min && n_ss > p->surf_min)) {        /* :598 */
        st.status = LISREG_NOT_ENOUGH_FEATURES;
        if (stats) *stats = st;
        return st.status;
    }
    if (n_threads < 1) n_threads = 1;
    cloud_t tc, ts, sc, ss;
    cloud_init(&tc, tgt_corner, n_tc, stride, fmt);
    cloud_init(&ts, tgt_surf, n_ts, stride, fmt);
    cloud_init(&sc, src_corner, n_sc, stride, fmt);
    cloud_init(&ss, src_surf, n_ss, stride, fmt);
    if (use_kdtree) {                                          /* :602-603: two builds per registration */
        tc.tree = orc_kdtree_build(tc.xyz, tc.n, 15);
        ts.tree = orc_kdtree_build(ts.xyz, ts.n, 15);
    }
    int n_src = n_sc + n_ss;
    unsigned char* flag_c = (unsigned char*)calloc((size_t)(n_sc > 0 ? n_sc : 1), 1);
    unsigned char* flag_s = (unsigned char*)calloc((size_t)(n_ss > 0 ? n_ss : 1), 1);
    float* coef_c = (float*)calloc((size_t)(n_sc > 0 ? n_sc : 1) * 4, sizeof(float));
    float* coef_s = (float*)calloc((size_t)(n_ss > 0 ? n_ss : 1) * 4, sizeof(float));
    float* ori = (float*)malloc(sizeof(float) * 3 * (size_t)(n_src > 0 ? n_src : 1));   /* laserCloudOri */
    float* sel = (float*)malloc(sizeof(float) * 4 * (size_t)(n_src > 0 ? n_src : 1));   /* coeffSel      */
    float* A = (float*)malloc(sizeof(float) * 6 * (size_t)(n_src > 0 ? n_src : 1));
    float* B = (float*)malloc(sizeof(float) * (size_t)(n_src > 0 ? n_src : 1));
    float P[36]; memset(P, 0, sizeof P);

    int bound = p->fixed_iters > 0 ? p->fixed_iters : p->max_iters;
    int iter = 0, any_solved = 0;
    for (; iter < bound; ++iter) {
        float M[12];
        orc_pose_to_matrix(T, M);                              /* updatePointAssociateToSubMap :628-631 */
        int do_corner = !(p->skip_empty_target && n_tc == 0);  /* :4505-4509 */
        int do_surf = !(p->skip_empty_target && n_ts == 0);
        if (do_corner) run_stage(0, &tc, &sc, p, M, use_kdtree, n_threads, flag_c, coef_c);
        else memset(flag_c, 0, (size_t)(n_sc > 0 ? n_sc : 1));
        if (do_surf) run_stage(1, &ts, &ss, p, M, use_kdtree, n_threads, flag_s, coef_s);
        else memset(flag_s, 0, (size_t)(n_ss > 0 ? n_ss : 1));
        /* combineOptimizationCoeffs :829-850 — corners in index order, then surfs */
        int n_sel = 0;
        for (int i = 0; i < n_sc; ++i) if (flag_c[i]) {
            memcpy(&ori[3 * n_sel], &sc.xyz[3 * i], 12); memcpy(&sel[4 * n_sel], &coef_c[4 * i], 16); ++n_sel; }
        for (int i = 0; i < n_ss; ++i) if (flag_s[i]) {
            memcpy(&ori[3 * n_sel], &ss.xyz[3 * i], 12); memcpy(&sel[4 * n_sel], &coef_s[4 * i], 16); ++n_sel; }
        st.n_corr_last = n_sel;
        float* tr = (trace && iter < max_trace) ? &trace[(size_t)iter * LISREG_TRACE_STRIDE] : NULL;
        if (tr) { memset(tr, 0, sizeof(float) * LISREG_TRACE_STRIDE); tr[0] = (float)n_sel; memcpy(&tr[49], T, 24); }
        if (n_sel < p->min_corr) continue;                     /* :870-872 return false, pose untouched */
        any_solved = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static)
        for (int i = 0; i < n_sel; ++i) orc_jacobian_row(T, &ori[3 * i], &sel[4 * i], &A[6 * i], &B[i]);
        /* :918-920 matAtA = At*A, matAtB = At*B — cv GEMM on CV_32F accumulates in double */
        float AtA[36], AtB[6], X[6];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) {
                double s = 0; for (int i = 0; i < n_sel; ++i) s += (double)A[6 * i + r] * (double)A[6 * i + c];
                AtA[6 * r + c] = (float)s;
            }
            double s = 0; for (int i = 0; i < n_sel; ++i) s += (double)A[6 * i + r] * (double)B[i];
            AtB[r] = (float)s;
        }
        orc_solve6(AtA, AtB, X);                               /* :921 */
        if (iter == 0) {                                       /* :923-946 */
            float E[6], V[36], V2[36], Vi[36];
            orc_eigen_sym(AtA, 6, E, V);
            memcpy(V2, V, sizeof V2);
            isDeg = 0;
            for (int i = 5; i >= 0; --i) {
                if (E[i] < p->eig_thresh) { for (int j = 0; j < 6; ++j) V2[6 * i + j] = 0; isDeg = 1; }
                else break;
            }
            orc_inv6(V, Vi);
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
                double s = 0; for (int k = 0; k < 6; ++k) s += (double)Vi[6 * r + k] * (double)V2[6 * k + c];
                P[6 * r + c] = (float)s;
            }
        } else if (p->emulate_matp_shadow) {
            memset(P, 0, sizeof P);                            /* local cv::Mat matP zero-initialised :880 */
        }
        if (isDeg) {                                           /* :948-953 */
            float X2[6]; memcpy(X2, X, sizeof X2);
            for (int r = 0; r < 6; ++r) {
                double s = 0; for (int k = 0; k < 6; ++k) s += (double)P[6 * r + k] * (double)X2[k];
                X[r] = (float)s;
            }
        }
        for (int k = 0; k < 6; ++k) T[k] += X[k];              /* :955-960 */
        /* :962-967 — pcl::rad2deg(float) is float; pow(float,int) and the sum are double */
        double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
        double t0 = (double)(X[3] * 100), t1 = (double)(X[4] * 100), t2 = (double)(X[5] * 100);
        st.deltaR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
        st.deltaT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);
        int conv = (st.deltaR < p->conv_deg && st.deltaT < p->conv_cm);
        if (tr) {
            memcpy(&tr[1], AtA, sizeof AtA); memcpy(&tr[37], AtB, sizeof AtB); memcpy(&tr[43], X, sizeof X);
            memcpy(&tr[49], T, 24); tr[55] = 1.f;
        }
        if (conv && p->fixed_iters <= 0) break;                /* :969-972, :617 */
    }
    st.iters = iter;
    orc_transform_update(p, imu, T);                           /* :622 */
    st.degenerate = isDeg;
    st.status = any_solved ? LISREG_OK : LISREG_TOO_FEW_CORRESPONDENCES;
    if (degenerate) *degenerate = isDeg;
    if (stats) *stats = st;
    free(flag_c); free(flag_s); free(coef_c); free(coef_s); free(ori); free(sel); free(A); free(B);
    cloud_free(&tc); cloud_free(&ts); cloud_free(&sc); cloud_free(&ss);
    return st.status;
}

/* ------------------------------------------------------------------------------------------------------- */
/* §8 f-1: pcl::VoxelGrid (default settings) and transformPointCloud                                         */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct { unsigned int idx; int pt; } vox_pair;

static int vox_cmp(const void* a, const void* b)
{
    const vox_pair* x = (const vox_pair*)a; const vox_pair* y = (const vox_pair*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);       /* fixed choice inside PCL's unstable order */
}

int orc_voxel_grid(const void* in, int n, int stride, int fmt, float leaf, void* out, int* n_out)
{
    const unsigned char* src = (const unsigned char*)in;
    unsigned char* dst = (unsigned char*)out;
    *n_out = 0;
    if (n <= 0) return 0;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int i = 0; i < n; ++i) {                             /* getMinMax3D */
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
    }
    const float inv = 1.0f / leaf;                            /* inverse_leaf_size_ = Ones() / leaf_size_ */
    long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
              dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > 2147483647LL) {                        /* "Leaf size is too small for the input dataset" */
        memcpy(dst, src, (size_t)n * (size_t)stride);
        *n_out = n;
        return 3;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = (int)floorf(mn[d] * inv);
        max_b[d] = (int)floorf(mx[d] * inv);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    vox_pair* iv = (vox_pair*)malloc(sizeof(vox_pair) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        int ijk0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int ijk1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int ijk2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        iv[i].idx = (unsigned int)(ijk0 + ijk1 * mul1 + ijk2 * mul2);
        iv[i].pt = i;
    }
    qsort(iv, (size_t)n, sizeof(vox_pair), vox_cmp);
    int no = 0;
    for (int a = 0; a < n;) {
        int b = a + 1;
        while (b < n && iv[b].idx == iv[a].idx) ++b;
        /* CentroidPoint: AccumulatorXYZ (Vector3f sum), AccumulatorIntensity (float sum), AccumulatorLabel (map) */
        float sx = 0, sy = 0, sz = 0, si = 0;
        for (int k = a; k < b; ++k) {
            const unsigned char* r = src + (size_t)iv[k].pt * (size_t)stride;
            float p[3], it = 0; memcpy(p, r, 12);
            if (stride >= 20) memcpy(&it, r + 16, 4);
            sx += p[0]; sy += p[1]; sz += p[2]; si += it;
        }
        const float cnt = (float)(size_t)(b - a);
        unsigned char* o = dst + (size_t)no * (size_t)stride;
        memset(o, 0, (size_t)stride);
        float c[3] = { sx / cnt, sy / cnt, sz / cnt }, ci = si / cnt;
        memcpy(o, c, 12);
        if (stride >= 20) memcpy(o + 16, &ci, 4);
        if (fmt == LISREG_FMT_XYZIL) {
            unsigned short best = 0; int bestc = 0;
            for (int k = a; k < b; ++k) {
                unsigned short lk; memcpy(&lk, src + (size_t)iv[k].pt * (size_t)stride + 20, 2);
                int cc = 0;
                for (int m = a; m < b; ++m) { unsigned short lm; memcpy(&lm, src + (size_t)iv[m].pt * (size_t)stride + 20, 2); cc += lm == lk; }
                if (cc > bestc || (cc == bestc && lk < best)) { bestc = cc; best = lk; }   /* smallest label on ties */
            }
            memcpy(o + 20, &best, 2);
        }
        ++no;
        a = b;
    }
    free(iv);
    *n_out = no;
    return 0;
}

void orc_transform_cloud(const void* in, int n, int stride, int fmt, const float T[6], void* out)
{
    (void)fmt;
    float M[12];
    orc_pose_to_matrix(T, M);                                 /* pcl::getTransformation (common.cpp:140-142) */
    const unsigned char* src = (const unsigned char*)in;
    unsigned char* dst = (unsigned char*)out;
    for (int i = 0; i < n; ++i) {
        float p[3], q[3];
        if (dst != src) memcpy(dst + (size_t)i * (size_t)stride, src + (size_t)i * (size_t)stride, (size_t)stride);
        memcpy(p, src + (size_t)i * (size_t)stride, 12);
        transform_point(M, p, q);
        memcpy(dst + (size_t)i * (size_t)stride, q, 12);
    }
}

/* ------------------------------------------------------------------------------------------------------- */
/* §8 f-2: LaserProcessing range-image projection + feature extraction (src/core/laserProcessing.cpp)        */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct { float value; int ind; } smooth_t;

static int smooth_cmp(const void* a, const void* b)
{
    const smooth_t* x = (const smooth_t*)a; const smooth_t* y = (const smooth_t*)b;
    if (x->value != y->value) return x->value < y->value ? -1 : 1;      /* by_value (laserProcessing.h:25-31) */
    return x->ind < y->ind ? -1 : (x->ind > y->ind ? 1 : 0);            /* std::sort is unstable: fix ties by index */
}

/* the two +-5 suppression loops of extractFeatures (:648-660, :681-694); index accesses bounds-checked */
static void suppress_neighbours(int ind, int cloudSize, const int* colInd, int* picked)
{
    for (int l = 1; l <= 5; l++) {
        if (ind + l >= cloudSize || ind + l - 1 < 0) break;
        if (abs(colInd[ind + l] - colInd[ind + l - 1]) > 10) break;
        picked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; l--) {
        if (ind + l < 0 || ind + l + 1 >= cloudSize) break;
        if (abs(colInd[ind + l] - colInd[ind + l + 1]) > 10) break;
        picked[ind + l] = 1;
    }
}

void orc_extract_features(const void* cloud, int n, int stride, const lisreg_feature_params* P,
                          int* deskewed, int* corner, int* surface, int* corner_sharp, int* surface_sharp, int counts[5])
{
    const unsigned char* src = (const unsigned char*)cloud;
    const int H = P->n_scan, W = P->horizon_scan, HW = H * W;
    float* rangeMat = (float*)malloc(sizeof(float) * (size_t)HW);
    int* owner = (int*)malloc(sizeof(int) * (size_t)HW);
    for (int i = 0; i < HW; ++i) { rangeMat[i] = FLT_MAX; owner[i] = -1; }          /* resetParameters :53 */
    /* projectPointCloud :467-510 */
    for (int i = 0; i < n; ++i) {
        const unsigned char* r = src + (size_t)i * (size_t)stride;
        float p[3]; unsigned short ring;
        memcpy(p, r, 12); memcpy(&ring, r + 20, 2);
        float range = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);                /* pointDistance, common.h:110-113 */
        if (range < P->min_range || range > P->max_range) continue;
        int rowIdn = ring;
        if (rowIdn < 0 || rowIdn >= H) continue;
        if (rowIdn % P->downsample_rate != 0) continue;
        /* :489 `atan2(x, y) * 180 / M_PI` — float atan2 there; libm float atan2 is not correctly rounded and differs
         * between implementations in the last ulp, which can move a point across a column boundary.  The restatement
         * defines it as the correctly rounded value: double atan2 rounded to float. */
        float at = (float)atan2((double)p[0], (double)p[1]);
        float horizonAngle = (float)((double)(at * 180) / M_PI);
        float ang_res_x = (float)(360.0 / (double)(float)W);
        int columnIdn = (int)(-round(((double)horizonAngle - 90.0) / (double)ang_res_x) + (double)(W / 2));
        if (columnIdn >= W) columnIdn -= W;
        if (columnIdn < 0 || columnIdn >= W) continue;
        if (rangeMat[rowIdn * W + columnIdn] != FLT_MAX) continue;                   /* first point wins */
        rangeMat[rowIdn * W + columnIdn] = range;
        owner[rowIdn * W + columnIdn] = i;
    }
    /* cloudExtraction :515-539 */
    int* startRing = (int*)malloc(sizeof(int) * (size_t)H); int* endRing = (int*)malloc(sizeof(int) * (size_t)H);
    int* colInd = (int*)calloc((size_t)HW + 16, sizeof(int));
    float* prange = (float*)calloc((size_t)HW + 16, sizeof(float));
    int count = 0;
    for (int i = 0; i < H; ++i) {
        startRing[i] = count - 1 + 5;
        for (int j = 0; j < W; ++j)
            if (rangeMat[i * W + j] != FLT_MAX) { colInd[count] = j; prange[count] = rangeMat[i * W + j]; deskewed[count] = owner[i * W + j]; ++count; }
        endRing[i] = count - 1 - 5;
    }
    const int cloudSize = count;
    /* per-frame arrays: zero-initialised here (the reference leaves [0,5) and [size-5,size) untouched) */
    float* curv = (float*)calloc((size_t)HW + 16, sizeof(float));
    int* picked = (int*)calloc((size_t)HW + 16, sizeof(int));
    int* label = (int*)calloc((size_t)HW + 16, sizeof(int));
    smooth_t* sm = (smooth_t*)malloc(sizeof(smooth_t) * ((size_t)HW + 16));
    for (int i = 0; i < HW + 16; ++i) { sm[i].value = 0.f; sm[i].ind = i; }
    /* calculateSmoothness :544-563 */
    for (int i = 5; i < cloudSize - 5; i++) {
        float diffRange = prange[i - 5] + prange[i - 4] + prange[i - 3] + prange[i - 2] + prange[i - 1] - prange[i] * 10 +
                          prange[i + 1] + prange[i + 2] + prange[i + 3] + prange[i + 4] + prange[i + 5];
        curv[i] = diffRange * diffRange;
        picked[i] = 0; label[i] = 0;
        sm[i].value = curv[i]; sm[i].ind = i;
    }
    /* markOccludedPoints :568-605 */
    for (int i = 5; i < cloudSize - 6; ++i) {
        float depth1 = prange[i], depth2 = prange[i + 1];
        int columnDiff = abs(colInd[i + 1] - colInd[i]);
        if (columnDiff < 10) {
            if (depth1 - depth2 > 0.3) { for (int l = -5; l <= 0; ++l) picked[i + l] = 1; }
            else if (depth2 - depth1 > 0.3) { for (int l = 1; l <= 6; ++l) picked[i + l] = 1; }
        }
        float diff1 = fabsf(prange[i - 1] - prange[i]), diff2 = fabsf(prange[i + 1] - prange[i]);
        if (diff1 > 0.02 * prange[i] && diff2 > 0.02 * prange[i]) picked[i] = 1;
    }
    /* extractFeatures :610-713 */
    int nc = 0, ns = 0, ncs = 0, nss = 0;
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < 6; j++) {
            int sp = (startRing[i] * (6 - j) + endRing[i] * j) / 6;
            int ep = (startRing[i] * (5 - j) + endRing[i] * (j + 1)) / 6 - 1;
            if (sp >= ep) continue;
            qsort(sm + sp, (size_t)(ep - sp), sizeof(smooth_t), smooth_cmp);        /* [sp, ep): ep itself stays put */
            int largestPickedNum = 0;
            for (int k = ep; k >= sp; k--) {
                int ind = sm[k].ind;
                if (picked[ind] == 0 && curv[ind] > P->edge_threshold) {
                    largestPickedNum++;
                    if (largestPickedNum <= 20) {
                        label[ind] = 1;
                        corner[nc++] = deskewed[ind];
                        if (largestPickedNum <= 4) corner_sharp[ncs++] = deskewed[ind];
                    } else break;
                    picked[ind] = 1;
                    suppress_neighbours(ind, cloudSize, colInd, picked);
                }
            }
            largestPickedNum = 0;
            for (int k = sp; k <= ep; k++) {
                int ind = sm[k].ind;
                if (picked[ind] == 0 && curv[ind] < P->surf_threshold) {
                    largestPickedNum++;
                    label[ind] = -1;
                    picked[ind] = 1;
                    if (largestPickedNum <= 10) surface_sharp[nss++] = deskewed[ind];
                    suppress_neighbours(ind, cloudSize, colInd, picked);
                }
            }
            for (int k = sp; k <= ep; k++) if (label[k] <= 0) surface[ns++] = deskewed[k];
        }
    }
    counts[0] = cloudSize; counts[1] = nc; counts[2] = ns; counts[3] = ncs; counts[4] = nss;
    free(rangeMat); free(owner); free(startRing); free(endRing); free(colInd); free(prange); free(curv); free(picked);
    free(label); free(sm);
}

/* categoryMapping (src/node/semanticFusionNode.cpp:173-189) */
void orc_semantic_classes(const void* cloud, int n, int stride, const unsigned int using_label[32], unsigned char* cls)
{
    const unsigned char* src = (const unsigned char*)cloud;
    for (int i = 0; i < n; ++i) {
        unsigned short label; memcpy(&label, src + (size_t)i * (size_t)stride + 20, 2);
        unsigned int u = using_label[label & 31];
        cls[i] = u == 10 ? 0 : (u == 40 ? 1 : (u == 50 ? 2 : (u == 81 ? 3 : 4)));
    }
}

/* ---- §8 f-3: local-map maintenance filters ------------------------------------------------------------------- */
/* get_cloud_bbx (src/include/subMap.h:131-163): float coordinates widened to double; an empty cloud leaves +-DBL_MAX */
void orc_cloud_bounds(const void* cloud, int n, int stride, double b[6])
{
    const unsigned char* src = (const unsigned char*)cloud;
    b[0] = b[1] = b[2] = DBL_MAX; b[3] = b[4] = b[5] = -DBL_MAX;
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        for (int d = 0; d < 3; ++d) {
            if (b[d] > p[d]) b[d] = p[d];
            if (b[3 + d] < p[d]) b[3 + d] = p[d];
        }
    }
}

/* bbx_filter (src/include/subMap.h:1124-1152): strict inequalities, float against double; keep[] = surviving input indices */
void orc_bbx_filter(const void* cloud, int n, int stride, const double b[6], int delete_box, int* keep, int* n_keep)
{
    const unsigned char* src = (const unsigned char*)cloud;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        int in = p[0] > b[0] && p[0] < b[3] && p[1] > b[1] && p[1] < b[4] && p[2] > b[2] && p[2] < b[5];
        if (in) { if (!delete_box) keep[m++] = i; }
        else    { if (delete_box) keep[m++] = i; }
    }
    *n_keep = m;
}

/* map_scan_feature_pts_distance_removal (src/include/subMap.h:1064-1100).  Returns 0 when the reference returns false
 * (<= 10 points: cloud untouched, keep[] = identity), else 1.  pcl::search::KdTree::nearestKSearch(k = 1) -> exact
 * nearest neighbour, squared L2 in float (FLANN L2_Simple).  An empty map is undefined behaviour in the reference
 * (distances_square[0] of an empty vector); here it keeps every point. */
int orc_dynamic_filter(const void* map, int n_map, const void* cloud, int n, int stride, float center_radius,
                       float dist_thre_min, float dist_thre_max, float near_dist_thre, int* keep, int* n_keep)
{
    const unsigned char* src = (const unsigned char*)cloud;
    if (n <= 10 || n_map <= 0) {
        for (int i = 0; i < n; ++i) keep[i] = i;
        *n_keep = n;
        return n > 10;
    }
    float* mxyz = (float*)malloc(sizeof(float) * 3 * (size_t)n_map);
    unpack_cloud(map, n_map, stride, LISREG_FMT_XYZI, mxyz, NULL);
    orc_kdtree* tree = orc_kdtree_build(mxyz, n_map, 15);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        if (p[0] * p[0] + p[1] * p[1] > center_radius * center_radius) keep[m++] = i;
        else {
            int idx; float d2;
            orc_kdtree_knn(tree, p, 1, &idx, &d2);
            if ((d2 > near_dist_thre * near_dist_thre && d2 < dist_thre_min * dist_thre_min) || d2 > dist_thre_max * dist_thre_max)
                keep[m++] = i;
        }
    }
    *n_keep = m;
    orc_kdtree_free(tree); free(mxyz);
    return 1;
}

/* k = 1 queries (test helper for the grid search): idx[i] = nearest map point (-1 if farther than max_dist), sqd[i] */
void orc_nearest(const void* map, int n_map, const void* query, int n, int stride, float max_dist, int* idx, float* sqd)
{
    float* mxyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n_map > 0 ? n_map : 1));
    unpack_cloud(map, n_map, stride, LISREG_FMT_XYZI, mxyz, NULL);
    orc_kdtree* tree = n_map > 0 ? orc_kdtree_build(mxyz, n_map, 15) : NULL;
    const unsigned char* src = (const unsigned char*)query;
    for (int i = 0; i < n; ++i) {
        float p[3]; memcpy(p, src + (size_t)i * (size_t)stride, 12);
        idx[i] = -1; sqd[i] = FLT_MAX;
        if (!tree) continue;
        int id; float d2;
        orc_kdtree_knn(tree, p, 1, &id, &d2);
        if (d2 <= max_dist * max_dist) { idx[i] = id; sqd[i] = d2; }
    }
    if (tree) orc_kdtree_free(tree);
    free(mxyz);
}

/* ---- §8 f-4: pcl::IterativeClosestPoint as the loop-closure code drives it -------------------------------------- */
/* 3x3 SVD by one-sided Jacobi in double (any accurate SVD yields the same R below when sigma has rank >= 2) */
static void svd3(const double Ain[9], double U[9], double S[3], double V[9])
{
    double A[9]; memcpy(A, Ain, sizeof A);
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int r = 0; r < 3; ++r) { a += A[3 * r + p] * A[3 * r + p]; b += A[3 * r + q] * A[3 * r + q]; c += A[3 * r + p] * A[3 * r + q]; }
                if (fabs(c) <= 1e-300 || fabs(c) <= 1e-15 * sqrt(a * b)) continue;
                off += fabs(c);
                double zeta = (b - a) / (2.0 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < 3; ++r) {
                    double x = A[3 * r + p], y = A[3 * r + q];
                    A[3 * r + p] = cs * x - sn * y; A[3 * r + q] = sn * x + cs * y;
                    x = V[3 * r + p]; y = V[3 * r + q];
                    V[3 * r + p] = cs * x - sn * y; V[3 * r + q] = sn * x + cs * y;
                }
            }
        if (off == 0) break;
    }
    int ord[3] = { 0, 1, 2 };
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (nrm[ord[j]] > nrm[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[9];
    for (int j = 0; j < 3; ++j) {
        S[j] = nrm[ord[j]];
        for (int r = 0; r < 3; ++r) { Vs[3 * r + j] = V[3 * r + ord[j]]; U[3 * r + j] = S[j] > 0 ? A[3 * r + ord[j]] / S[j] : 0.0; }
    }
    memcpy(V, Vs, sizeof Vs);
    /* complete U for (numerically) vanishing singular values so that it stays orthogonal */
    const double tiny = 1e-12 * (S[0] > 0 ? S[0] : 1.0);
    if (S[1] <= tiny) {                       /* rank <= 1: any orthonormal completion */
        double u0[3] = { U[0], U[3], U[6] };
        if (S[0] <= 0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
        int k = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
        double e[3] = { 0, 0, 0 }; e[k] = 1;
        double d = e[0] * u0[0] + e[1] * u0[1] + e[2] * u0[2];
        double u1[3] = { e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2] };
        double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        for (int r = 0; r < 3; ++r) U[3 * r + 1] = u1[r] / n1;
    }
    if (S[2] <= tiny) {
        U[2] = U[3] * U[7] - U[6] * U[4];      /* u2 = u0 x u1 */
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* pcl::registration::TransformationEstimationSVD::estimateRigidTransformation -> pcl::umeyama(src, tgt, false)
 * (PCL 1.8.1 registration/impl/transformation_estimation_svd.hpp, Eigen 3.3 Umeyama.h): means, demeaned covariance
 * sigma = (1/n) * dst_demean * src_demean^T, SVD, S = diag(1, 1, sign(det U * det V)), R = U S V^T, t = dst_mean - R src_mean.
 * Eigen evaluates all of it in float: the means by a sequential float sum (rowwise().sum() of a strided row), sigma by a
 * blocked GEMM whose summation order is unspecified.  float_sums = 1 restates that (means summed in float in input order);
 * float_sums = 0 accumulates the means in double.  A float running sum over 1e5 coordinates of ~10-50 m is only good to
 * ~1e-4..1e-3 m and depends on the summation ORDER, so it cannot be the parity target of a parallel reduction: the GPU path
 * is compared with float_sums = 0, and tests/test_icp.py measures how far the float_sums = 1 result sits from it (the
 * reference's own summation noise).  Either way: float demeaned coordinates, products accumulated in double, SVD in double,
 * result rounded to float. */
void orc_umeyama(const float* src, const float* dst, int n, int float_sums, float T[16])
{
    float ms[3] = { 0, 0, 0 }, md[3] = { 0, 0, 0 };
    const float inv = 1.0f / (float)n;
    if (float_sums) {
        for (int i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { ms[d] += src[3 * i + d]; md[d] += dst[3 * i + d]; }
        for (int d = 0; d < 3; ++d) { ms[d] *= inv; md[d] *= inv; }
    } else {
        double as[3] = { 0, 0, 0 }, ad[3] = { 0, 0, 0 };
        for (int i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { as[d] += src[3 * i + d]; ad[d] += dst[3 * i + d]; }
        for (int d = 0; d < 3; ++d) { ms[d] = (float)(as[d] / (double)n); md[d] = (float)(ad[d] / (double)n); }
    }
    double sg[9] = { 0 };
    for (int i = 0; i < n; ++i) {
        float a[3], b[3];
        for (int d = 0; d < 3; ++d) { a[d] = src[3 * i + d] - ms[d]; b[d] = dst[3 * i + d] - md[d]; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) sg[3 * r + c] += (double)b[r] * (double)a[c];
    }
    for (int k = 0; k < 9; ++k) sg[k] = (double)(float)(sg[k] * (double)inv);
    double U[9], S[3], V[9];
    svd3(sg, U, S, V);
    double s2 = det3(U) * det3(V) < 0 ? -1.0 : 1.0;
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        R[3 * r + c] = U[3 * r + 0] * V[3 * c + 0] + U[3 * r + 1] * V[3 * c + 1] + s2 * U[3 * r + 2] * V[3 * c + 2];
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[4 * r + c] = (float)R[3 * r + c];
        T[4 * r + 3] = md[r] - (T[4 * r + 0] * ms[0] + T[4 * r + 1] * ms[1] + T[4 * r + 2] * ms[2]);
    }
}

static void mat4_mul(const float A[16], const float B[16], float C[16])
{
    float R[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c)
        R[4 * r + c] = ((A[4 * r] * B[c] + A[4 * r + 1] * B[4 + c]) + A[4 * r + 2] * B[8 + c]) + A[4 * r + 3] * B[12 + c];
    memcpy(C, R, sizeof R);
}

static void mat4_apply(const float M[16], const float p[3], float o[3])
{
    for (int r = 0; r < 3; ++r) o[r] = ((M[4 * r] * p[0] + M[4 * r + 1] * p[1]) + M[4 * r + 2] * p[2]) + M[4 * r + 3];
}

/* pcl::IterativeClosestPoint::align + getFitnessScore + hasConverged as called at
 * src/node/subMapOptmizationNode.cpp:1444-1465, :2763-2833, :4400-4420 — restated from PCL 1.8.1
 * registration/impl/icp.hpp (computeTransformation), impl/correspondence_estimation.hpp (determineCorrespondences),
 * impl/default_convergence_criteria.hpp (hasConverged), impl/registration.hpp (getFitnessScore).
 * prm->prev_mse is DefaultConvergenceCriteria::correspondences_prev_mse_, which the reference's `static` ICP objects carry
 * from one align() to the next (DBL_MAX on a fresh object); res->prev_mse is its value afterwards. */
void orc_icp_align(const void* target, int n_t, const void* source, int n_s, int stride, const lisreg_icp_params* prm,
                   const float* guess, int float_sums, lisreg_icp_result* res)
{
    float* txyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n_t > 0 ? n_t : 1));
    float* sxyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n_s > 0 ? n_s : 1));
    float* cur  = (float*)malloc(sizeof(float) * 3 * (size_t)(n_s > 0 ? n_s : 1));
    float* ca   = (float*)malloc(sizeof(float) * 3 * (size_t)(n_s > 0 ? n_s : 1));
    float* cb   = (float*)malloc(sizeof(float) * 3 * (size_t)(n_s > 0 ? n_s : 1));
    unpack_cloud(target, n_t, stride, LISREG_FMT_XYZI, txyz, NULL);
    unpack_cloud(source, n_s, stride, LISREG_FMT_XYZI, sxyz, NULL);
    orc_kdtree* tree = n_t > 0 ? orc_kdtree_build(txyz, n_t, 15) : NULL;
    float F[16], Tm[16];
    for (int k = 0; k < 16; ++k) F[k] = guess ? guess[k] : ((k % 5 == 0) ? 1.0f : 0.0f);
    for (int i = 0; i < n_s; ++i) mat4_apply(F, sxyz + 3 * i, cur + 3 * i);        /* identity guess: exact copy */
    int iters = 0, converged = 0, state = LISREG_ICP_NOT_CONVERGED, n_corr = 0;
    double prev_mse = prm->prev_mse, cur_mse = DBL_MAX;
    const double max_d2 = (double)prm->max_corr_dist * (double)prm->max_corr_dist;
    for (;;) {
        int cnt = 0;
        double dsum = 0;
        for (int i = 0; i < n_s && tree; ++i) {
            int id; float d2;
            orc_kdtree_knn(tree, cur + 3 * i, 1, &id, &d2);
            if ((double)d2 > max_d2) continue;
            memcpy(ca + 3 * cnt, cur + 3 * i, 12); memcpy(cb + 3 * cnt, txyz + 3 * id, 12);
            dsum += d2; ++cnt;
        }
        n_corr = cnt;
        if (cnt < 3) { state = LISREG_ICP_NO_CORRESPONDENCES; converged = 0; break; }     /* min_number_correspondences_ = 3 */
        orc_umeyama(ca, cb, cnt, float_sums, Tm);
        for (int i = 0; i < n_s; ++i) { float o[3]; mat4_apply(Tm, cur + 3 * i, o); memcpy(cur + 3 * i, o, 12); }
        mat4_mul(Tm, F, F);
        ++iters;
        /* DefaultConvergenceCriteria::hasConverged */
        state = LISREG_ICP_NOT_CONVERGED;
        if (iters >= prm->max_iters) { state = LISREG_ICP_ITERATIONS; converged = 1; break; }
        double cos_angle = 0.5 * ((double)Tm[0] + (double)Tm[5] + (double)Tm[10] - 1.0);
        double tr2 = (double)Tm[3] * Tm[3] + (double)Tm[7] * Tm[7] + (double)Tm[11] * Tm[11];
        if (cos_angle >= 1.0 - prm->transformation_epsilon && tr2 <= prm->transformation_epsilon) { state = LISREG_ICP_TRANSFORM; converged = 1; break; }
        cur_mse = dsum / (double)cnt;
        if (fabs(cur_mse - prev_mse) < 1e-12) { state = LISREG_ICP_ABS_MSE; converged = 1; break; }
        if (fabs(cur_mse - prev_mse) / prev_mse < prm->euclidean_fitness_epsilon) { state = LISREG_ICP_REL_MSE; converged = 1; break; }
        prev_mse = cur_mse;
    }
    /* getFitnessScore(): unbounded k = 1 of the source moved by the final transformation */
    double fit = 0; int nr = 0;
    for (int i = 0; i < n_s && tree; ++i) {
        float p[3]; int id; float d2;
        mat4_apply(F, sxyz + 3 * i, p);
        orc_kdtree_knn(tree, p, 1, &id, &d2);
        fit += d2; ++nr;
    }
    memcpy(res->final_transform, F, sizeof F);
    res->converged = converged; res->iters = iters; res->state = state; res->n_corr_last = n_corr;
    res->fitness = nr > 0 ? fit / nr : DBL_MAX;
    res->prev_mse = prev_mse;
    if (tree) orc_kdtree_free(tree);
    free(txyz); free(sxyz); free(cur); free(ca); free(cb);
}

/* ---- §8 f-4 (second half): OptimizedICPGN (src/core/registration.cpp:19-115; never called in the reference) ------ */
/* 6x6 float LU with partial pivoting (Eigen::PartialPivLU, which Matrix<float,6,6>::determinant() / inverse() use):
 * returns the determinant; inv (may be NULL) = LU.solve(Identity). */
static float lu6_det_inv(const float Hin[36], float inv[36])
{
    float A[36]; int perm[6]; float sign = 1.f;
    memcpy(A, Hin, sizeof A);
    for (int i = 0; i < 6; ++i) perm[i] = i;
    for (int k = 0; k < 6; ++k) {
        int piv = k; float big = fabsf(A[6 * k + k]);
        for (int r = k + 1; r < 6; ++r) if (fabsf(A[6 * r + k]) > big) { big = fabsf(A[6 * r + k]); piv = r; }
        if (piv != k) {
            for (int c = 0; c < 6; ++c) { float t = A[6 * k + c]; A[6 * k + c] = A[6 * piv + c]; A[6 * piv + c] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t; sign = -sign;
        }
        if (A[6 * k + k] != 0.f)
            for (int r = k + 1; r < 6; ++r) {
                A[6 * r + k] /= A[6 * k + k];
                for (int c = k + 1; c < 6; ++c) A[6 * r + c] -= A[6 * r + k] * A[6 * k + c];
            }
    }
    float det = sign;
    for (int k = 0; k < 6; ++k) det *= A[6 * k + k];
    if (inv && det != 0.f)
        for (int col = 0; col < 6; ++col) {
            float y[6];
            for (int r = 0; r < 6; ++r) { float s = (perm[r] == col) ? 1.f : 0.f; for (int c = 0; c < r; ++c) s -= A[6 * r + c] * y[c]; y[r] = s; }
            for (int r = 5; r >= 0; --r) { float s = y[r]; for (int c = r + 1; c < 6; ++c) s -= A[6 * r + c] * inv[6 * c + col]; inv[6 * r + col] = s / A[6 * r + r]; }
        }
    return det;
}

/* Sophus::SO3f::exp(omega).matrix() — src/sophus/so3.hpp:279-312 (this vendored version's small-angle real_factor uses
 * 0.5 * theta^2 where the series has 1/8; restated as written), Constants<float>::epsilon = 1e-5 (sophus_common.hpp:146),
 * then Eigen's Quaternion::toRotationMatrix. */
static void so3_exp_matrix(const float w[3], float E[9])
{
    float theta_sq = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    float theta = sqrtf(theta_sq), half = 0.5f * theta, imag, real;
    if (theta < 1e-5f) {
        float po4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * po4;
        real = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * po4;
    } else { imag = sinf(half) / theta; real = cosf(half); }
    float qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
    float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
    float twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    E[0] = 1.f - (tyy + tzz); E[1] = txy - twz; E[2] = txz + twy;
    E[3] = txy + twz; E[4] = 1.f - (txx + tzz); E[5] = tyz - twx;
    E[6] = txz - twy; E[7] = tyz + twx; E[8] = 1.f - (txx + tyy);
}

/* OptimizedICPGN::Match + GetFitnessScore.  Quirk kept: the SQUARED k = 1 distance is compared with the un-squared
 * max_correspond_distance (:50).  No convergence test: max_iterations Gauss-Newton steps, skipping a step whose Hessian has
 * determinant exactly 0.  The reference accumulates Hessian / B / the fitness score in float in input order; as for
 * orc_umeyama, float_sums = 1 restates that and float_sums = 0 (the GPU's parity target) accumulates them in double. */
void orc_icp_gn(const void* target, int n_t, const void* source, int n_s, int stride, unsigned max_iterations,
                float max_correspond_distance, const float predict_pose[16], int float_sums, lisreg_icpgn_result* res)
{
    float* txyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n_t > 0 ? n_t : 1));
    float* sxyz = (float*)malloc(sizeof(float) * 3 * (size_t)(n_s > 0 ? n_s : 1));
    unpack_cloud(target, n_t, stride, LISREG_FMT_XYZI, txyz, NULL);
    unpack_cloud(source, n_s, stride, LISREG_FMT_XYZI, sxyz, NULL);
    orc_kdtree* tree = n_t > 0 ? orc_kdtree_build(txyz, n_t, 15) : NULL;
    float T[16];
    memcpy(T, predict_pose, sizeof T);
    int n_corr = 0, applied = 0;
    for (unsigned it = 0; it < max_iterations; ++it) {
        double Hd[36] = { 0 }, Bd[6] = { 0 };
        float Hf[36] = { 0 }, Bf[6] = { 0 };
        n_corr = 0;
        for (int j = 0; j < n_s && tree; ++j) {
            const float* p = sxyz + 3 * j;
            if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
            float tp[3]; int id; float d2;
            mat4_apply(T, p, tp);
            orc_kdtree_knn(tree, tp, 1, &id, &d2);
            if (d2 > max_correspond_distance) continue;
            const float* q = txyz + 3 * id;
            float e[3] = { tp[0] - q[0], tp[1] - q[1], tp[2] - q[2] };
            float J[18];                                       /* 3 x 6: [I | -R hat(p)] */
            for (int r = 0; r < 3; ++r) {
                const float R0 = T[4 * r], R1 = T[4 * r + 1], R2 = T[4 * r + 2];
                J[6 * r + 0] = r == 0; J[6 * r + 1] = r == 1; J[6 * r + 2] = r == 2;
                J[6 * r + 3] = -(R1 * p[2] - R2 * p[1]);
                J[6 * r + 4] = -(R2 * p[0] - R0 * p[2]);
                J[6 * r + 5] = -(R0 * p[1] - R1 * p[0]);
            }
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) {
                    float h = (J[a] * J[b] + J[6 + a] * J[6 + b]) + J[12 + a] * J[12 + b];
                    if (float_sums) Hf[6 * a + b] += h; else Hd[6 * a + b] += h;
                }
                float g = -((J[a] * e[0] + J[6 + a] * e[1]) + J[12 + a] * e[2]);
                if (float_sums) Bf[a] += g; else Bd[a] += g;
            }
            ++n_corr;
        }
        if (!float_sums) { for (int k = 0; k < 36; ++k) Hf[k] = (float)Hd[k]; for (int k = 0; k < 6; ++k) Bf[k] = (float)Bd[k]; }
        float inv[36];
        if (lu6_det_inv(Hf, inv) == 0.f) continue;
        float dx[6];
        for (int r = 0; r < 6; ++r) { float s = 0.f; for (int c = 0; c < 6; ++c) s += inv[6 * r + c] * Bf[c]; dx[r] = s; }
        T[3] += dx[0]; T[7] += dx[1]; T[11] += dx[2];
        float E[9], Rn[9];
        so3_exp_matrix(dx + 3, E);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            Rn[3 * r + c] = (T[4 * r] * E[c] + T[4 * r + 1] * E[3 + c]) + T[4 * r + 2] * E[6 + c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[4 * r + c] = Rn[3 * r + c];
        ++applied;
    }
    double fd = 0; float ff = 0.f; int nr = 0;
    for (int j = 0; j < n_s && tree; ++j) {
        float tp[3]; int id; float d2;
        mat4_apply(T, sxyz + 3 * j, tp);
        orc_kdtree_knn(tree, tp, 1, &id, &d2);
        if (float_sums) ff += d2; else fd += d2;
        ++nr;
    }
    memcpy(res->final_transform, T, sizeof T);
    res->n_corr_last = n_corr; res->steps_applied = applied;
    res->fitness = nr > 0 ? (float_sums ? ff / (float)nr : (float)(fd / nr)) : FLT_MAX;
    if (tree) orc_kdtree_free(tree);
    free(txyz); free(sxyz);
}

/* ---- §8 f-2, IMU de-skew: LaserProcessing::deskewPoint / findRotation (src/core/laserProcessing.cpp:368-399, 427-462) --- */
/* pcl::getTransformation(0, 0, 0, roll, pitch, yaw).linear() (PCL common/impl/eigen.hpp), float */
static void rot_from_rpy(float roll, float pitch, float yaw, float R[9])
{
    /* cos/sin of a float, defined here as the correctly rounded value (double evaluation rounded once): what glibc's cosf/sinf
     * deliver, and reproducible on the device, whose single-precision library differs from glibc in the last ulp */
    float A = (float)cos((double)yaw), B = (float)sin((double)yaw), C = (float)cos((double)pitch), D = (float)sin((double)pitch),
          E = (float)cos((double)roll), F = (float)sin((double)roll), DE = D * E, DF = D * F;
    R[0] = A * C; R[1] = A * DF - B * E; R[2] = B * F + A * DE;
    R[3] = B * C; R[4] = A * E + B * DF; R[5] = B * DE - A * F;
    R[6] = -D;    R[7] = C * F;          R[8] = C * E;
}

/* Eigen's 3x3 inverse (compute_inverse<Matrix3f>: cofactors and one division by the determinant), which
 * Eigen::Affine3f::inverse() applies to the linear part */
static void inv3_cofactor(const float m[9], float inv[9])
{
#define COF(i, j) (m[3 * (((i) + 1) % 3) + ((j) + 1) % 3] * m[3 * (((i) + 2) % 3) + ((j) + 2) % 3] - m[3 * (((i) + 1) % 3) + ((j) + 2) % 3] * m[3 * (((i) + 2) % 3) + ((j) + 1) % 3])
    const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    const float det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
    const float invdet = 1.0f / det;
    inv[0] = c00 * invdet; inv[1] = c10 * invdet; inv[2] = c20 * invdet;
    inv[3] = COF(0, 1) * invdet; inv[4] = COF(1, 1) * invdet; inv[5] = COF(2, 1) * invdet;
    inv[6] = COF(0, 2) * invdet; inv[7] = COF(1, 2) * invdet; inv[8] = COF(2, 2) * invdet;
#undef COF
}

static void find_rotation(const lisreg_deskew* dk, double pointTime, float rot[3])
{
    int front = 0;
    while (front < dk->imu_pointer_cur) { if (pointTime < dk->imu_time[front]) break; ++front; }
    if (pointTime > dk->imu_time[front] || front == 0) {
        rot[0] = (float)dk->imu_rot_x[front]; rot[1] = (float)dk->imu_rot_y[front]; rot[2] = (float)dk->imu_rot_z[front];
    } else {
        int back = front - 1;
        double rf = (pointTime - dk->imu_time[back]) / (dk->imu_time[front] - dk->imu_time[back]);
        double rb = (dk->imu_time[front] - pointTime) / (dk->imu_time[front] - dk->imu_time[back]);
        rot[0] = (float)(dk->imu_rot_x[front] * rf + dk->imu_rot_x[back] * rb);
        rot[1] = (float)(dk->imu_rot_y[front] * rf + dk->imu_rot_y[back] * rb);
        rot[2] = (float)(dk->imu_rot_z[front] * rf + dk->imu_rot_z[back] * rb);
    }
}

/* The de-skewed coordinates of the points that own a range-image pixel (idx[m] = their input indices, e.g. the `deskewed`
 * list of orc_extract_features).  projectPointCloud calls deskewPoint in input order for exactly these points, so the one
 * with the smallest index sets transStartInverse (firstPointFlag, :439-443).  findPosition returns zeros (:405-421).
 * xyz_out[m][3].  With dk->enabled == 0 the points come back unchanged (:429). */
void orc_deskew_points(const void* cloud, int stride, const lisreg_deskew* dk, const int* idx, int m, float* xyz_out)
{
    const unsigned char* src = (const unsigned char*)cloud;
    int first = -1;
    for (int k = 0; k < m; ++k) if (first < 0 || idx[k] < first) first = idx[k];
    float Rsi[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (dk->enabled && first >= 0) {
        float t, rot[3], R0[9];
        memcpy(&t, src + (size_t)first * (size_t)stride + 24, 4);
        find_rotation(dk, dk->time_scan_cur + (double)t, rot);
        rot_from_rpy(rot[0], rot[1], rot[2], R0);
        inv3_cofactor(R0, Rsi);
    }
    for (int k = 0; k < m; ++k) {
        const unsigned char* r = src + (size_t)idx[k] * (size_t)stride;
        float p[3], t;
        memcpy(p, r, 12); memcpy(&t, r + 24, 4);
        if (!dk->enabled) { memcpy(xyz_out + 3 * k, p, 12); continue; }
        float rot[3], Rf[9], Rb[9];
        find_rotation(dk, dk->time_scan_cur + (double)t, rot);
        rot_from_rpy(rot[0], rot[1], rot[2], Rf);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
            Rb[3 * a + b] = (Rsi[3 * a] * Rf[b] + Rsi[3 * a + 1] * Rf[3 + b]) + Rsi[3 * a + 2] * Rf[6 + b];
        for (int a = 0; a < 3; ++a)                           /* transBt(a,3) = 0: both translations are zero */
            xyz_out[3 * k + a] = ((Rb[3 * a] * p[0] + Rb[3 * a + 1] * p[1]) + Rb[3 * a + 2] * p[2]) + 0.0f;
    }
}
