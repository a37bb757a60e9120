/*
 * oracle/icp_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, double-precision CPU restatement of the reference's per-scan hot
 * path (DUBSON0/iterative-closest-point-avmi, utilities/icp.py and
 * utilities/mapping.py).  It is the checker the HIP path is compared with, and
 * the "port" CPU baseline timed by bench.py.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (libicpmi.so and the
 * Python host modules) never does.
 *
 * Pinned: every function below is checked against golden vectors produced by
 * RUNNING the reference in the build container (tests/golden/make_golden.py,
 * tests/test_oracle_golden.py).  Third-party arithmetic the reference leans on
 * (SciPy cKDTree 1.15.3, NumPy/LAPACK svd/solve/eigh) is restated from its
 * published algorithm: exact k-d tree search with direct-difference squared
 * distances, partial-pivot LU, closed-form 2x2 symmetric eigenvector, Jacobi SVD.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so
 * sums of squares round exactly as NumPy's do).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_CONVERGED 1
#define ORC_MAXITER 2
#define ORC_FEW_INLIERS 3

/* ------------------------------------------------------------------------- */
/* voxel_downsample — reference utilities/icp.py:117-129                      */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t k[3]; int idx; } vkey_t;
static __thread int g_vdim;   /* per thread: the bench times independent pairs on a thread pool */
static int vkey_cmp(const void* a, const void* b) {
    const vkey_t* x = (const vkey_t*)a; const vkey_t* y = (const vkey_t*)b;
    for (int d = 0; d < g_vdim; ++d) {
        if (x->k[d] < y->k[d]) return -1;
        if (x->k[d] > y->k[d]) return 1;
    }
    return (x->idx > y->idx) - (x->idx < y->idx);   /* stable: input order inside a voxel */
}

/* out must hold n*dim doubles; returns the number of voxels. icp.py:119 min_bound,
 * :120 floor((p-min)/v) keys, :121 lexicographic unique, :123-128 per-voxel mean
 * with sums taken sequentially in input order (np.bincount with weights). */
int orc_voxel_downsample(const double* pts, int n, int dim, double voxel, double* out) {
    if (n <= 0) return 0;
    double mn[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
        mn[d] = pts[d];
        for (int i = 1; i < n; ++i) if (pts[(size_t)i * dim + d] < mn[d]) mn[d] = pts[(size_t)i * dim + d];
    }
    vkey_t* keys = (vkey_t*)malloc(sizeof(vkey_t) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        keys[i].idx = i;
        keys[i].k[0] = keys[i].k[1] = keys[i].k[2] = 0;
        for (int d = 0; d < dim; ++d)
            keys[i].k[d] = (int64_t)floor((pts[(size_t)i * dim + d] - mn[d]) / voxel);
    }
    g_vdim = dim;
    qsort(keys, (size_t)n, sizeof(vkey_t), vkey_cmp);
    int nv = 0;
    int i = 0;
    while (i < n) {
        int j = i;
        double s[3] = {0, 0, 0};
        while (j < n && keys[j].k[0] == keys[i].k[0] && keys[j].k[1] == keys[i].k[1] &&
               keys[j].k[2] == keys[i].k[2]) {
            for (int d = 0; d < dim; ++d) s[d] += pts[(size_t)keys[j].idx * dim + d];
            ++j;
        }
        double cnt = (double)(j - i);
        for (int d = 0; d < dim; ++d) out[(size_t)nv * dim + d] = s[d] / cnt;
        ++nv;
        i = j;
    }
    free(keys);
    return nv;
}

/* ------------------------------------------------------------------------- */
/* nearest neighbour — reference icp.py:173,179 (scipy.spatial.KDTree.query)   */
/* ------------------------------------------------------------------------- */
static inline double sqdist(const double* a, const double* b, int dim) {
    double s = 0.0;                       /* s += d*d per axis, no FMA: cKDTree / NumPy order */
    for (int d = 0; d < dim; ++d) { double t = a[d] - b[d]; s += t * t; }
    return s;
}

/* Exhaustive search, lowest index wins ties. dist = sqrt(d2) like KDTree.query. */
void orc_nn_brute(const double* src, int n, const double* tgt, int m, int dim,
                  int32_t* idx, double* dist) {
    for (int i = 0; i < n; ++i) {
        double best = INFINITY; int bi = -1;
        for (int j = 0; j < m; ++j) {
            double s = sqdist(src + (size_t)i * dim, tgt + (size_t)j * dim, dim);
            if (s < best) { best = s; bi = j; }
        }
        idx[i] = bi; dist[i] = sqrt(best);
    }
}

/* A small exact k-d tree (median split on the widest axis, leaves of <= 12
 * points).  Same answers as the exhaustive search including lowest-index ties;
 * it exists so the CPU baseline is O(N log M) like the reference's. */
#define KD_LEAF 12
typedef struct {
    int dim, m, nnodes;
    const double* pts;
    int* perm;
    int *lo, *hi, *left, *right, *sdim;
    double* sval;
} kdt_t;

static void kd_select(kdt_t* t, int lo, int hi, int mid, int d) {
    /* quickselect on perm[lo:hi) by coordinate d (ties by index for determinism) */
    const double* p = t->pts; int dim = t->dim; int* a = t->perm;
    while (hi - lo > 1) {
        int piv = a[lo + (hi - lo) / 2];
        double pv = p[(size_t)piv * dim + d];
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (p[(size_t)a[i] * dim + d] < pv || (p[(size_t)a[i] * dim + d] == pv && a[i] < piv)) ++i;
            while (p[(size_t)a[j] * dim + d] > pv || (p[(size_t)a[j] * dim + d] == pv && a[j] > piv)) --j;
            if (i <= j) { int tmp = a[i]; a[i] = a[j]; a[j] = tmp; ++i; --j; }
        }
        if (mid <= j) hi = j + 1; else if (mid >= i) lo = i; else return;
    }
}

static int kd_build(kdt_t* t, int lo, int hi) {
    int id = t->nnodes++;
    t->lo[id] = lo; t->hi[id] = hi; t->left[id] = t->right[id] = -1;
    if (hi - lo <= KD_LEAF) return id;
    int dim = t->dim, bd = 0; double bs = -1.0;
    for (int d = 0; d < dim; ++d) {
        double mn = INFINITY, mx = -INFINITY;
        for (int i = lo; i < hi; ++i) {
            double v = t->pts[(size_t)t->perm[i] * dim + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        if (mx - mn > bs) { bs = mx - mn; bd = d; }
    }
    if (!(bs > 0.0)) return id;                 /* all points identical: keep as a leaf */
    int mid = lo + (hi - lo) / 2;
    kd_select(t, lo, hi, mid, bd);
    t->sdim[id] = bd;
    t->sval[id] = t->pts[(size_t)t->perm[mid] * dim + bd];
    int l = kd_build(t, lo, mid);
    int r = kd_build(t, mid, hi);
    t->left[id] = l; t->right[id] = r;
    return id;
}

kdt_t* orc_kd_create(const double* pts, int m, int dim) {
    kdt_t* t = (kdt_t*)calloc(1, sizeof(kdt_t));
    t->dim = dim; t->m = m; t->pts = pts;
    int cap = 2 * (m > 0 ? m : 1) + 2;
    t->perm = (int*)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    t->lo = (int*)malloc(sizeof(int) * cap); t->hi = (int*)malloc(sizeof(int) * cap);
    t->left = (int*)malloc(sizeof(int) * cap); t->right = (int*)malloc(sizeof(int) * cap);
    t->sdim = (int*)malloc(sizeof(int) * cap); t->sval = (double*)malloc(sizeof(double) * cap);
    for (int i = 0; i < m; ++i) t->perm[i] = i;
    if (m > 0) kd_build(t, 0, m);
    return t;
}

void orc_kd_destroy(kdt_t* t) {
    if (!t) return;
    free(t->perm); free(t->lo); free(t->hi); free(t->left); free(t->right); free(t->sdim); free(t->sval);
    free(t);
}

/* k best (d2, idx) kept sorted ascending, lexicographic on (d2, idx). */
static inline void kbest_push(double* bd, int* bi, int k, double s, int j) {
    if (!(s < bd[k - 1] || (s == bd[k - 1] && j < bi[k - 1]))) return;
    int p = k - 1;
    while (p > 0 && (s < bd[p - 1] || (s == bd[p - 1] && j < bi[p - 1]))) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
    bd[p] = s; bi[p] = j;
}

static void kd_search(const kdt_t* t, int node, const double* q, double* bd, int* bi, int k) {
    if (t->left[node] < 0) {
        for (int i = t->lo[node]; i < t->hi[node]; ++i) {
            int j = t->perm[i];
            kbest_push(bd, bi, k, sqdist(q, t->pts + (size_t)j * t->dim, t->dim), j);
        }
        return;
    }
    double diff = q[t->sdim[node]] - t->sval[node];
    int near = diff < 0 ? t->left[node] : t->right[node];
    int far = diff < 0 ? t->right[node] : t->left[node];
    kd_search(t, near, q, bd, bi, k);
    if (diff * diff <= bd[k - 1]) kd_search(t, far, q, bd, bi, k);
}

/* k nearest of every query; idx/d2 are n*k, ascending by (d2, index). */
void orc_kd_knn(const kdt_t* t, const double* q, int n, int k, int32_t* idx, double* d2) {
    double* bd = (double*)malloc(sizeof(double) * (size_t)k);
    int* bi = (int*)malloc(sizeof(int) * (size_t)k);
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < k; ++c) { bd[c] = INFINITY; bi[c] = 0x7fffffff; }
        if (t->m > 0) kd_search(t, 0, q + (size_t)i * t->dim, bd, bi, k);
        for (int c = 0; c < k; ++c) { idx[(size_t)i * k + c] = bi[c]; d2[(size_t)i * k + c] = bd[c]; }
    }
    free(bd); free(bi);
}

void orc_nn_kdtree(const double* src, int n, const double* tgt, int m, int dim,
                   int32_t* idx, double* dist) {
    kdt_t* t = orc_kd_create(tgt, m, dim);
    orc_kd_knn(t, src, n, 1, idx, dist);
    for (int i = 0; i < n; ++i) dist[i] = sqrt(dist[i]);
    orc_kd_destroy(t);
}

/* ------------------------------------------------------------------------- */
/* estimate_normals_2d — reference icp.py:51-76                               */
/* ------------------------------------------------------------------------- */
/* Smallest-eigenvalue eigenvector of [[a,b],[b,c]] (what eigh's column 0 is,
 * up to sign).  Pick the better conditioned of the two closed forms. */
static void smallest_evec_2x2(double a, double b, double c, double* vx, double* vy) {
    double h = 0.5 * (a - c);
    double r = sqrt(h * h + b * b);
    double lam = 0.5 * (a + c) - r;
    double x1 = b, y1 = lam - a;           /* (A - lam I) row 0 -> orthogonal vector */
    double x2 = lam - c, y2 = b;           /* row 1 */
    double n1 = x1 * x1 + y1 * y1, n2 = x2 * x2 + y2 * y2;
    double x, y, nn;
    if (n1 >= n2) { x = x1; y = y1; nn = n1; } else { x = x2; y = y2; nn = n2; }
    if (nn == 0.0) { *vx = 1.0; *vy = 0.0; return; }   /* isotropic: eigh returns e0 */
    nn = sqrt(nn);
    *vx = x / nn; *vy = y / nn;
}

void orc_normals_2d(const double* pts, int m, int k, double* normals) {
    if (m <= 0) return;
    if (k > m - 1) k = m - 1;                              /* icp.py:61 */
    int kk = k + 1;                                        /* self included, icp.py:66 */
    kdt_t* t = orc_kd_create(pts, m, 2);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)m * kk);
    double* d2 = (double*)malloc(sizeof(double) * (size_t)m * kk);
    orc_kd_knn(t, pts, m, kk, idx, d2);
    for (int i = 0; i < m; ++i) {
        double mx = 0, my = 0;
        for (int c = 0; c < kk; ++c) { mx += pts[2 * idx[(size_t)i * kk + c]]; my += pts[2 * idx[(size_t)i * kk + c] + 1]; }
        mx /= kk; my /= kk;
        double sxx = 0, sxy = 0, syy = 0;
        for (int c = 0; c < kk; ++c) {
            double dx = pts[2 * idx[(size_t)i * kk + c]] - mx, dy = pts[2 * idx[(size_t)i * kk + c] + 1] - my;
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
        double den = (double)(kk - 1);                     /* np.cov ddof=1 */
        double vx, vy;
        if (kk - 1 <= 0) { vx = 1.0; vy = 0.0; }           /* degenerate: one point */
        else smallest_evec_2x2(sxx / den, sxy / den, syy / den, &vx, &vy);
        double nn = sqrt(vx * vx + vy * vy);
        if (nn < 1e-10) nn = 1e-10;                        /* icp.py:74-75 */
        normals[2 * i] = vx / nn; normals[2 * i + 1] = vy / nn;
    }
    free(idx); free(d2); orc_kd_destroy(t);
}

/* ------------------------------------------------------------------------- */
/* _point_to_line_solve_2d — reference icp.py:79-115                          */
/* ------------------------------------------------------------------------- */
/* 3x3 solve by LU with partial pivoting (what LAPACK gesv does); returns 0 when
 * a pivot is exactly zero (np.linalg.solve raises LinAlgError then). */
static int solve3(double A[3][3], double b[3], double x[3]) {
    int p[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) {
        int best = c; double bv = fabs(A[p[c]][c]);
        for (int r = c + 1; r < 3; ++r) if (fabs(A[p[r]][c]) > bv) { bv = fabs(A[p[r]][c]); best = r; }
        int tmp = p[c]; p[c] = p[best]; p[best] = tmp;
        if (A[p[c]][c] == 0.0) return 0;
        for (int r = c + 1; r < 3; ++r) {
            double f = A[p[r]][c] / A[p[c]][c];
            A[p[r]][c] = f;
            for (int q = c + 1; q < 3; ++q) A[p[r]][q] -= f * A[p[c]][q];
        }
    }
    double y[3];
    for (int r = 0; r < 3; ++r) { y[r] = b[p[r]]; for (int q = 0; q < r; ++q) y[r] -= A[p[r]][q] * y[q]; }
    for (int r = 2; r >= 0; --r) { double s = y[r]; for (int q = r + 1; q < 3; ++q) s -= A[p[r]][q] * x[q]; x[r] = s / A[p[r]][r]; }
    return 1;
}

/* src: K points (already the inlier subset), idx: their matches into tgt/normals. */
void orc_p2l_solve_2d(const double* src, int K, const double* tgt, const double* normals,
                      const int32_t* idx, double R[4], double t[2]) {
    double A[3][3] = {{0}}, b[3] = {0, 0, 0};
    for (int i = 0; i < K; ++i) {
        int j = idx[i];
        double nx = normals[2 * j], ny = normals[2 * j + 1];
        double px = src[2 * i], py = src[2 * i + 1];
        double dx = px - tgt[2 * j], dy = py - tgt[2 * j + 1];
        double c = ny * px - nx * py;                       /* icp.py:97 */
        double bi = -(nx * dx + ny * dy);                   /* icp.py:101 */
        double row[3] = {c, nx, ny};
        for (int r = 0; r < 3; ++r) { for (int q = 0; q < 3; ++q) A[r][q] += row[r] * row[q]; b[r] += row[r] * bi; }
    }
    double x[3];
    if (!solve3(A, b, x)) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 1; t[0] = t[1] = 0;  /* icp.py:107-108 */
        return;
    }
    double ct = cos(x[0]), st = sin(x[0]);
    R[0] = ct; R[1] = -st; R[2] = st; R[3] = ct; t[0] = x[1]; t[1] = x[2];
}

/* ------------------------------------------------------------------------- */
/* point-to-point step — reference icp.py:196-207                             */
/* ------------------------------------------------------------------------- */
/* One-sided Jacobi SVD of a 3x3 (dim<=3) matrix W = U diag(s) V^T, s descending. */
static void svd_jacobi(int d, const double* W, double* U, double* S, double* V) {
    double A[9], Vm[9];
    for (int i = 0; i < d * d; ++i) A[i] = W[i];
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) Vm[i * d + j] = (i == j);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < d - 1; ++p) for (int q = p + 1; q < d; ++q) {
            double a = 0, b = 0, c = 0;
            for (int i = 0; i < d; ++i) { a += A[i * d + p] * A[i * d + p]; b += A[i * d + q] * A[i * d + q]; c += A[i * d + p] * A[i * d + q]; }
            if (c == 0.0) continue;
            off += fabs(c) / sqrt(a * b + 1e-300);
            double zeta = (b - a) / (2.0 * c);
            double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
            for (int i = 0; i < d; ++i) {
                double x = A[i * d + p], y = A[i * d + q];
                A[i * d + p] = cs * x - sn * y; A[i * d + q] = sn * x + cs * y;
                x = Vm[i * d + p]; y = Vm[i * d + q];
                Vm[i * d + p] = cs * x - sn * y; Vm[i * d + q] = sn * x + cs * y;
            }
        }
        if (off < 1e-15) break;
    }
    int ord[3] = {0, 1, 2}; double sv[3];
    for (int j = 0; j < d; ++j) { double s = 0; for (int i = 0; i < d; ++i) s += A[i * d + j] * A[i * d + j]; sv[j] = sqrt(s); }
    for (int i = 0; i < d; ++i) for (int j = i + 1; j < d; ++j) if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int jj = 0; jj < d; ++jj) {
        int j = ord[jj]; S[jj] = sv[j];
        for (int i = 0; i < d; ++i) { V[i * d + jj] = Vm[i * d + j]; U[i * d + jj] = sv[j] > 0 ? A[i * d + j] / sv[j] : 0.0; }
    }
    /* complete U to an orthonormal basis when W is rank deficient */
    if (d == 3) {
        int rank = (S[0] > 0) + (S[1] > 0) + (S[2] > 0);
        if (rank == 2) {
            U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
            U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
            U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
        }
    } else if (d == 2 && S[0] > 0 && !(S[1] > 0)) {
        U[0 * 2 + 1] = -U[1 * 2 + 0]; U[1 * 2 + 1] = U[0 * 2 + 0];
    }
}

static double det_nd(int d, const double* M) {
    if (d == 2) return M[0] * M[3] - M[1] * M[2];
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* P, Q: K matched pairs (inliers only). r = V U^T with the reflection fix on the
 * last right-singular vector (icp.py:202-206); t = mu_q - r mu_p (icp.py:207). */
void orc_p2p_step(const double* P, const double* Q, int K, int dim, double* r, double* t) {
    double mp[3] = {0, 0, 0}, mq[3] = {0, 0, 0};
    for (int i = 0; i < K; ++i) for (int d = 0; d < dim; ++d) { mp[d] += P[(size_t)i * dim + d]; mq[d] += Q[(size_t)i * dim + d]; }
    for (int d = 0; d < dim; ++d) { mp[d] /= K; mq[d] /= K; }
    double W[9] = {0};
    for (int i = 0; i < K; ++i)
        for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b)
            W[a * dim + b] += (P[(size_t)i * dim + a] - mp[a]) * (Q[(size_t)i * dim + b] - mq[b]);
    double U[9], S[3], V[9];
    svd_jacobi(dim, W, U, S, V);
    for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) {
        double s = 0; for (int c = 0; c < dim; ++c) s += V[a * dim + c] * U[b * dim + c];
        r[a * dim + b] = s;
    }
    if (det_nd(dim, r) < 0) {
        for (int a = 0; a < dim; ++a) V[a * dim + (dim - 1)] = -V[a * dim + (dim - 1)];
        for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) {
            double s = 0; for (int c = 0; c < dim; ++c) s += V[a * dim + c] * U[b * dim + c];
            r[a * dim + b] = s;
        }
    }
    for (int a = 0; a < dim; ++a) { double s = 0; for (int b = 0; b < dim; ++b) s += r[a * dim + b] * mp[b]; t[a] = mq[a] - s; }
}

/* ------------------------------------------------------------------------- */
/* ICP — reference icp.py:132-223                                             */
/* ------------------------------------------------------------------------- */
/* method: 0 = point_to_point, 1 = point_to_line.  max_corr_dist < 0 means None.
 * R_init/t_init: both non-NULL or the initial guess is ignored (icp.py:153).
 * out: R (dim*dim), t (dim), err, iters executed, status, last |prev-err|.
 * use_kdtree selects the O(N log M) search (identical answers). */
int orc_icp(const double* source, int ns, const double* target, int nt, int dim,
            double error_threshold, int max_iterations, double voxel_size,
            const double* R_init, const double* t_init, int method, int normal_k,
            double max_corr_dist, int use_kdtree,
            double* R_out, double* t_out, double* err_out, int* iters_out, double* delta_out,
            int* n_src_out, int* n_tgt_out) {
    double* src = (double*)malloc(sizeof(double) * (size_t)(ns > 0 ? ns : 1) * dim);
    double* tgt = (double*)malloc(sizeof(double) * (size_t)(nt > 0 ? nt : 1) * dim);
    int N = orc_voxel_downsample(source, ns, dim, voxel_size, src);     /* icp.py:150 */
    int M = orc_voxel_downsample(target, nt, dim, voxel_size, tgt);     /* icp.py:151 */
    if (n_src_out) *n_src_out = N;
    if (n_tgt_out) *n_tgt_out = M;
    double rt[9], tt[3];
    for (int a = 0; a < dim; ++a) { tt[a] = 0; for (int b = 0; b < dim; ++b) rt[a * dim + b] = (a == b); }
    double* cur = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1) * dim);
    if (R_init && t_init) {                                              /* icp.py:153-156 */
        for (int i = 0; i < N; ++i) for (int a = 0; a < dim; ++a) {
            double s = 0; for (int b = 0; b < dim; ++b) s += src[(size_t)i * dim + b] * R_init[a * dim + b];
            cur[(size_t)i * dim + a] = s + t_init[a];
        }
        memcpy(rt, R_init, sizeof(double) * dim * dim); memcpy(tt, t_init, sizeof(double) * dim);
    } else memcpy(cur, src, sizeof(double) * (size_t)N * dim);
    int use_p2l = (method == 1 && dim == 2);                             /* icp.py:162 */
    double* normals = NULL;
    if (use_p2l) { normals = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1) * 2); orc_normals_2d(tgt, M, normal_k, normals); }
    double max_corr_sq = max_corr_dist >= 0 ? max_corr_dist * max_corr_dist : -1.0;
    kdt_t* tree = use_kdtree ? orc_kd_create(tgt, M, dim) : NULL;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    double* dist = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    double* Pin = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1) * dim);
    double* Qin = (double*)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1) * dim);
    int32_t* iin = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    double prev = INFINITY, err = INFINITY, delta = INFINITY;
    int status = ORC_MAXITER, iters = 0;
    for (int it = 0; it < max_iterations; ++it) {
        if (tree) { orc_kd_knn(tree, cur, N, 1, idx, dist); for (int i = 0; i < N; ++i) dist[i] = sqrt(dist[i]); }
        else orc_nn_brute(cur, N, tgt, M, dim, idx, dist);               /* icp.py:179 */
        int K = 0;
        for (int i = 0; i < N; ++i) {
            int in = 1;
            if (max_corr_sq >= 0) in = (dist[i] * dist[i] < max_corr_sq); /* icp.py:184-185 */
            if (in) {
                for (int a = 0; a < dim; ++a) { Pin[(size_t)K * dim + a] = cur[(size_t)i * dim + a]; Qin[(size_t)K * dim + a] = tgt[(size_t)idx[i] * dim + a]; }
                iin[K] = idx[i]; ++K;
            }
        }
        if (max_corr_sq >= 0) {
            int need = N / 10 > 3 ? N / 10 : 3;                          /* icp.py:186 */
            if (K < need) { status = ORC_FEW_INLIERS; break; }
        }
        double r[9], t[3];
        if (use_p2l) orc_p2l_solve_2d(Pin, K, tgt, normals, iin, r, t);  /* icp.py:193 */
        else orc_p2p_step(Pin, Qin, K, dim, r, t);                       /* icp.py:197-207 */
        double nr[9], nt2[3];
        for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) { double s = 0; for (int c = 0; c < dim; ++c) s += r[a * dim + c] * rt[c * dim + b]; nr[a * dim + b] = s; }
        for (int a = 0; a < dim; ++a) { double s = 0; for (int b = 0; b < dim; ++b) s += tt[b] * r[a * dim + b]; nt2[a] = s + t[a]; }
        memcpy(rt, nr, sizeof(double) * dim * dim); memcpy(tt, nt2, sizeof(double) * dim);   /* icp.py:210-211 */
        double se = 0;
        for (int i = 0; i < N; ++i) {                                    /* icp.py:212, :215 over ALL points */
            double np_[3];
            for (int a = 0; a < dim; ++a) { double s = 0; for (int b = 0; b < dim; ++b) s += cur[(size_t)i * dim + b] * r[a * dim + b]; np_[a] = s + t[a]; }
            double e = 0;
            for (int a = 0; a < dim; ++a) { cur[(size_t)i * dim + a] = np_[a]; double dq = tgt[(size_t)idx[i] * dim + a] - np_[a]; e += dq * dq; }
            se += e;
        }
        err = se / (double)N;
        iters = it + 1;
        delta = fabs(prev - err);
        if (delta < error_threshold) { status = ORC_CONVERGED; break; }  /* icp.py:216-219 */
        prev = err;
    }
    memcpy(R_out, rt, sizeof(double) * dim * dim); memcpy(t_out, tt, sizeof(double) * dim);
    *err_out = err; *iters_out = iters; if (delta_out) *delta_out = delta;
    free(src); free(tgt); free(cur); free(normals); free(idx); free(dist); free(Pin); free(Qin); free(iin);
    orc_kd_destroy(tree);
    return status;
}

/* ------------------------------------------------------------------------- */
/* OccupancyGrid2D — reference utilities/mapping.py                           */
/* ------------------------------------------------------------------------- */
/* _bresenham, mapping.py:68-89: start cell emitted, end cell excluded. Returns
 * the number of cells; writes at most cap (x,y) pairs. */
int64_t orc_bresenham(int64_t x0, int64_t y0, int64_t x1, int64_t y1, int32_t* cells, int64_t cap) {
    int64_t dx = llabs(x1 - x0), dy = llabs(y1 - y0);
    int64_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int64_t err = dx - dy, x = x0, y = y0, n = 0;
    for (;;) {
        if (x == x1 && y == y1) break;
        if (n < cap) { cells[2 * n] = (int32_t)x; cells[2 * n + 1] = (int32_t)y; }
        ++n;
        int64_t e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x += sx; }
        if (e2 < dx) { err += dx; y += sy; }
    }
    return n;
}

/* _world_to_grid(_batch), mapping.py:57-60,94-98: floor((w - min) / res). */
void orc_world_to_grid(const double* w, int n, double mn, double res, int64_t* out) {
    for (int i = 0; i < n; ++i) out[i] = (int64_t)floor((w[i] - mn) / res);
}

/* update_scan, mapping.py:103-141.  log_odds is float32 (ny, nx); every add is
 * float32(float64(old) + l) as NumPy evaluates `f32 += np.float64`.  Returns the
 * number of cell updates performed (in-bounds hit adds + in-bounds free adds). */
int64_t orc_grid_update_scan(float* log_odds, int ny, int nx, double min_x, double min_y, double res,
                             double ox_w, double oy_w, const double* hits, int nb,
                             double l_hit, double l_miss, double lo, double hi) {
    if (nb <= 0) return 0;                                    /* mapping.py:113 */
    int64_t ox = (int64_t)floor((ox_w - min_x) / res), oy = (int64_t)floor((oy_w - min_y) / res);
    int64_t* hx = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t* hy = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t updates = 0;
    for (int i = 0; i < nb; ++i) {
        hx[i] = (int64_t)floor((hits[2 * i] - min_x) / res);
        hy[i] = (int64_t)floor((hits[2 * i + 1] - min_y) / res);
    }
    for (int i = 0; i < nb; ++i)                              /* np.add.at, mapping.py:124-129 */
        if (hx[i] >= 0 && hx[i] < nx && hy[i] >= 0 && hy[i] < ny) {
            float* c = &log_odds[(size_t)hy[i] * nx + hx[i]];
            *c = (float)((double)*c + l_hit);
            ++updates;
        }
    for (int i = 0; i < nb; ++i) {                            /* mapping.py:135-139 */
        int64_t x0 = ox, y0 = oy, x1 = hx[i], y1 = hy[i];
        int64_t dx = llabs(x1 - x0), dy = llabs(y1 - y0);
        int64_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
        int64_t err = dx - dy, x = x0, y = y0;
        for (;;) {
            if (x == x1 && y == y1) break;
            if (x >= 0 && x < nx && y >= 0 && y < ny) {
                float* c = &log_odds[(size_t)y * nx + x];
                *c = (float)((double)*c + l_miss);
                ++updates;
            }
            int64_t e2 = 2 * err;
            if (e2 > -dy) { err -= dy; x += sx; }
            if (e2 < dx) { err += dx; y += sy; }
        }
    }
    float lof = (float)lo, hif = (float)hi;                   /* np.clip on a float32 array, mapping.py:141 */
    size_t tot = (size_t)ny * nx;
    for (size_t i = 0; i < tot; ++i) { float v = log_odds[i]; if (v < lof) v = lof; if (v > hif) v = hif; log_odds[i] = v; }
    free(hx); free(hy);
    return updates;
}

/* ------------------------------------------------------------------------- */
/* rotation_search scoring — reference utilities/features.py:213-218          */
/* ------------------------------------------------------------------------- */
/* score[a] = mean over rows of (NN distance)^2 of (src_c @ R(a).T + shift) in tgt, with
 * R(a) = [[c,-s],[s,c]] given as cs[2a], cs[2a+1]; distance = sqrt(d2) squared again,
 * as the reference's `np.mean(dists ** 2)` on KDTree distances. */
void orc_rotation_scores(const double* src_c, int n, const double* tgt, int m, const double* cs, int n_angles,
                         double shift_x, double shift_y, double* scores) {
    kdt_t* t = orc_kd_create(tgt, m, 2);
    double* rot = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * 2);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    double* d2 = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int a = 0; a < n_angles; ++a) {
        const double c = cs[2 * a], s = cs[2 * a + 1];
        for (int i = 0; i < n; ++i) {
            const double x = src_c[2 * i], y = src_c[2 * i + 1];
            rot[2 * i] = (x * c + y * -s) + shift_x;
            rot[2 * i + 1] = (x * s + y * c) + shift_y;
        }
        orc_kd_knn(t, rot, n, 1, idx, d2);
        double acc = 0.0;
        for (int i = 0; i < n; ++i) { const double d = sqrt(d2[i]); acc += d * d; }
        scores[a] = acc / (double)n;
    }
    free(rot); free(idx); free(d2); orc_kd_destroy(t);
}
