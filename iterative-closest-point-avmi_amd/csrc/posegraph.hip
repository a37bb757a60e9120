// posegraph.hip — PoseGraph2D.optimize (reference utilities/pose_graph.py:83-134) in one launch.
//
// The reference builds the dense 3n x 3n normal matrix H edge by edge in Python
// and solves it with LAPACK on every Gauss-Newton iteration.  The graphs it
// builds (slam.py:543-549, 593) are an odometry chain 0-1-2-...-(n-1) plus a few
// loop-closure edges, so here
//     H = T + W C W^T
// with T block tridiagonal (3x3 blocks: the chain edges and the anchor), W the
// 3n x 3k Jacobian columns of the k other edges and C = blockdiag(Omega_e):
//     T^-1 by block cyclic reduction along the chain (log2 n levels, each parallel
//     over nodes and over all 1 + 3k right-hand sides), then the 3k x 3k system
//     (I + C W^T T^-1 W) y = C W^T T^-1 r,  dx = T^-1 r - T^-1 W y.
// O(n k^2) work instead of O(n^3), and the whole optimisation — every
// iteration's linearisation, solve, update and convergence test — runs inside
// ONE persistent workgroup: no launches or host round trips between iterations.
// A graph with a gap in the chain (or whose nodes are not numbered along it)
// takes the general path: the same dense matrix as the reference, eliminated
// with partial pivoting by the workgroup in global memory.
//
// Every sum has a fixed order (per-node gathers in edge order, fixed reduction
// trees): results are reproducible run to run.
#include "common.hpp"
#include <vector>

namespace icpmi {

constexpr int PG_THREADS = 512;                    // one workgroup, 2 waves per SIMD
constexpr double PG_PI = 3.141592653589793;        // np.pi
constexpr double PG_2PI = 6.283185307179586;       // 2 * np.pi

struct PgArgs {
    double* nodes;                 // [n][3] x, y, theta — updated in place
    const int32_t* ei;             // [m]
    const int32_t* ej;             // [m]
    const double* z;               // [m][3]
    const double* omega;           // [m][9]
    const int32_t* csr_ptr;        // [n+1] incident edges of every node ...
    const int32_t* csr_edge;       // ... in edge order
    const int32_t* loop_slot;      // [m] index among the non-chain edges, -1 for a chain edge
    const int32_t* loop_edge;      // [k] edge ids of the non-chain edges
    int n, m, k, dense, n_iter, fix;
    double eps;
    double *D, *U, *L, *b;         // block tridiagonal T per node: D[v] = H[v][v], U[v] = H[v][v+s], L[v] = H[v][v-s] (s = level stride)
    double *Dinv, *G;              // per even node of a level: L[e] * inv(D[e-s]) and U[e] * inv(D[e+s])
    double* X;                     // [3n][1 + 3k] right-hand sides -> solutions (column 0 = r = -b, then W)
    double* AB;                    // [k][18] Jacobians of the non-chain edges
    double *M, *g, *y;             // [3k][3k], [3k], [3k]
    double* dx;                    // [3n]
    double* Hd;                    // [3n][3n], dense path only
    double* info;                  // [0] iterations run, [1] status, [2] last step norm
};

enum { PG_NOTHING = 0, PG_CONVERGED = 1, PG_MAXITER = 2, PG_SINGULAR = 3 };

// ── 3x3 helpers (row-major, fully unrolled) ─────────────────────────────────
struct M3 { double a[9]; };
__device__ __forceinline__ M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.a[i] = 0.0; return r; }
__device__ __forceinline__ M3 m3_load(const double* p) { M3 r; for (int i = 0; i < 9; ++i) r.a[i] = p[i]; return r; }
__device__ __forceinline__ void m3_store(double* p, const M3& m) { for (int i = 0; i < 9; ++i) p[i] = m.a[i]; }
__device__ __forceinline__ M3 m3_mul(const M3& x, const M3& y) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.a[3 * i + j] = (x.a[3 * i] * y.a[j] + x.a[3 * i + 1] * y.a[3 + j]) + x.a[3 * i + 2] * y.a[6 + j];
    return r;
}
__device__ __forceinline__ M3 m3_tmul(const M3& x, const M3& y) {      // x^T y
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.a[3 * i + j] = (x.a[i] * y.a[j] + x.a[3 + i] * y.a[3 + j]) + x.a[6 + i] * y.a[6 + j];
    return r;
}
__device__ __forceinline__ void m3_add(M3& x, const M3& y) { for (int i = 0; i < 9; ++i) x.a[i] += y.a[i]; }
__device__ __forceinline__ void m3_sub(M3& x, const M3& y) { for (int i = 0; i < 9; ++i) x.a[i] -= y.a[i]; }
__device__ __forceinline__ void m3_vec(const M3& m, const double* v, double* out) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = (m.a[3 * i] * v[0] + m.a[3 * i + 1] * v[1]) + m.a[3 * i + 2] * v[2];
}
__device__ __forceinline__ void m3_tvec(const M3& m, const double* v, double* out) {   // m^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = (m.a[i] * v[0] + m.a[3 + i] * v[1]) + m.a[6 + i] * v[2];
}
// inverse by cofactors; false when the determinant is exactly zero
__device__ __forceinline__ bool m3_inv(const M3& m, M3& r) {
    const double* a = m.a;
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = (a[0] * c0 + a[1] * c1) + a[2] * c2;
    if (det == 0.0 || !(fabs(det) < 1e308)) return false;
    const double id = 1.0 / det;
    r.a[0] = c0 * id; r.a[1] = (a[2] * a[7] - a[1] * a[8]) * id; r.a[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r.a[3] = c1 * id; r.a[4] = (a[0] * a[8] - a[2] * a[6]) * id; r.a[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r.a[6] = c2 * id; r.a[7] = (a[1] * a[6] - a[0] * a[7]) * id; r.a[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}

// normalize_angle, pose_graph.py:15-17, with Python's float modulo (result takes the sign of the divisor)
__device__ __forceinline__ double pg_wrap(double a) {
    double r = fmod(a + PG_PI, PG_2PI);
    if (r < 0.0) r += PG_2PI;
    return r - PG_PI;
}

// _error_and_jacobians, pose_graph.py:138-182
__device__ __forceinline__ void pg_linearise(const double* xi, const double* xj, const double* z, double* e, M3& A, M3& B) {
    const double ci = cos(xi[2]), si = sin(xi[2]);
    const double dx = xj[0] - xi[0], dy = xj[1] - xi[1];
    e[0] = (ci * dx + si * dy) - z[0];
    e[1] = (-si * dx + ci * dy) - z[1];
    e[2] = pg_wrap(pg_wrap(xj[2] - xi[2]) - z[2]);
    A = m3_zero(); B = m3_zero();
    A.a[0] = -ci; A.a[1] = -si; A.a[3] = si; A.a[4] = -ci;
    A.a[2] = -si * dx + ci * dy;
    A.a[5] = -ci * dx + -si * dy;
    A.a[8] = -1.0;
    B.a[0] = ci; B.a[1] = si; B.a[3] = -si; B.a[4] = ci; B.a[8] = 1.0;
}

// ── workgroup-wide helpers ───────────────────────────────────────────────────
// argmax of |v| with the lowest index among equals; every thread gets the result
__device__ __forceinline__ void block_argmax(double v, int idx, double* s_val, int* s_idx, double& out_v, int& out_i) {
    s_val[threadIdx.x] = v; s_idx[threadIdx.x] = idx;
    __syncthreads();
    for (int o = PG_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double a = s_val[threadIdx.x], b = s_val[threadIdx.x + o];
            const int ia = s_idx[threadIdx.x], ib = s_idx[threadIdx.x + o];
            if (b > a || (b == a && ib < ia)) { s_val[threadIdx.x] = b; s_idx[threadIdx.x] = ib; }
        }
        __syncthreads();
    }
    out_v = s_val[0]; out_i = s_idx[0];
    __syncthreads();
}
__device__ __forceinline__ double block_total(double v, double* s_val) {
    s_val[threadIdx.x] = v;
    __syncthreads();
    for (int o = PG_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_val[threadIdx.x] += s_val[threadIdx.x + o];
        __syncthreads();
    }
    const double r = s_val[0];
    __syncthreads();
    return r;
}

// Solve a x = rhs (a: N x N row-major in global memory, destroyed; x may alias rhs) by Gaussian elimination
// with partial pivoting, the whole workgroup on one system.  false = a pivot is exactly zero (LAPACK's
// "singular matrix", the LinAlgError of pose_graph.py:117-119).
__device__ bool block_lu_solve(double* a, int N, double* rhs, double* s_val, int* s_idx) {
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;                       // update tile: 32 columns x (PG_THREADS / 32) rows
    for (int p = 0; p < N; ++p) {
        if (wave_id() == 0) {                                     // pivot search: one wave, no workgroup barriers
            double best = -1.0; int bi = 0x7fffffff;
            for (int r = p + lane_id(); r < N; r += ICPMI_WAVE) {
                const double v = fabs(a[(size_t)r * N + p]);
                if (v > best) { best = v; bi = r; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, ICPMI_WAVE);
                const int oi = __shfl_xor(bi, o, ICPMI_WAVE);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane_id() == 0) { s_val[0] = best; s_idx[0] = bi; }
        }
        __syncthreads();
        const double pv = s_val[0];
        const int pr = s_idx[0];
        if (!(pv > 0.0)) return false;
        if (pr != p) {
            for (int c = p + tid; c < N; c += PG_THREADS) {       // columns left of p are never read again
                const double t = a[(size_t)p * N + c];
                a[(size_t)p * N + c] = a[(size_t)pr * N + c];
                a[(size_t)pr * N + c] = t;
            }
            if (tid == 0) { const double t = rhs[p]; rhs[p] = rhs[pr]; rhs[pr] = t; }
        }
        __syncthreads();
        // eliminate column p from the rows below; column p itself is only read here, so every thread takes its
        // row's multiplier from it directly (column index N stands for the right-hand side)
        const double piv = a[(size_t)p * N + p];
        for (int r = p + 1 + ty; r < N; r += PG_THREADS / 32) {
            const double l = a[(size_t)r * N + p] / piv;
            for (int c = p + 1 + tx; c <= N; c += 32) {
                if (c == N) rhs[r] -= l * rhs[p];
                else a[(size_t)r * N + c] -= l * a[(size_t)p * N + c];
            }
        }
        __syncthreads();
    }
    for (int r = N - 1; r >= 0; --r) {                                                     // back substitution
        if (tid == 0) rhs[r] = rhs[r] / a[(size_t)r * N + r];
        __syncthreads();
        const double xr = rhs[r];
        for (int q = tid; q < r; q += PG_THREADS) rhs[q] -= a[(size_t)q * N + r] * xr;
        __syncthreads();
    }
    return true;
}

__global__ __launch_bounds__(PG_THREADS) void pose_graph_kernel(PgArgs g) {
    __shared__ double s_val[PG_THREADS];
    __shared__ int s_idx[PG_THREADS];
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    const int n = g.n, k = g.k, K = 3 * g.k, ncols = 1 + 3 * g.k, N3 = 3 * g.n, f = g.fix;
    int status = PG_MAXITER, iters = g.n_iter;
    double step = 0.0;
    // phase clocks (thread 0, 100 MHz wall clock): assemble, chain LU + W, sweeps, closure system, its solve, dx, update
    long long t_acc[7] = {0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
#define PG_MARK(i)                                                                      \
    do {                                                                                \
        if (tid == 0) { const long long now = wall_clock64(); t_acc[i] += now - t_last; t_last = now; } \
    } while (0)

    for (int it = 0; it < g.n_iter; ++it) {
        if (tid == 0) s_flag = 0;
        if (g.dense)
            for (size_t t = tid; t < (size_t)N3 * N3; t += PG_THREADS) g.Hd[t] = 0.0;
        __syncthreads();
        // ── linearise every edge and gather per node (pose_graph.py:96-105) ──────
        for (int v = tid; v < n; v += PG_THREADS) {
            M3 D = m3_zero(), U = m3_zero(), L = m3_zero();
            double b[3] = {0.0, 0.0, 0.0};
            for (int t = g.csr_ptr[v]; t < g.csr_ptr[v + 1]; ++t) {
                const int q = g.csr_edge[t];
                const int i = g.ei[q], j = g.ej[q];
                double e[3], tmp[3], Oe[3];
                M3 A, B;
                pg_linearise(g.nodes + 3 * i, g.nodes + 3 * j, g.z + 3 * q, e, A, B);
                const M3 Om = m3_load(g.omega + 9 * (size_t)q);
                const M3 OA = m3_mul(Om, A), OB = m3_mul(Om, B);
                m3_vec(Om, e, Oe);
                const bool in_T = g.dense || g.loop_slot[q] < 0;
                if (i == v) {
                    m3_tvec(A, Oe, tmp); b[0] += tmp[0]; b[1] += tmp[1]; b[2] += tmp[2];
                    if (in_T) m3_add(D, m3_tmul(A, OA));
                }
                if (j == v) {
                    m3_tvec(B, Oe, tmp); b[0] += tmp[0]; b[1] += tmp[1]; b[2] += tmp[2];
                    if (in_T) m3_add(D, m3_tmul(B, OB));
                }
                if (g.dense) {
                    if (i == v && j != f && v != f) {                    // H[i][j] += A^T O B (row block owned by this thread)
                        const M3 c = m3_tmul(A, OB);
                        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) g.Hd[(size_t)(3 * v + r) * N3 + 3 * j + cc] += c.a[3 * r + cc];
                    }
                    if (j == v && i != f && v != f) {                    // H[j][i] += B^T O A
                        const M3 c = m3_tmul(B, OA);
                        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) g.Hd[(size_t)(3 * v + r) * N3 + 3 * i + cc] += c.a[3 * r + cc];
                    }
                } else if (in_T && v == min(i, j) && i != j) {
                    if (i == v) { m3_add(U, m3_tmul(A, OB)); m3_add(L, m3_tmul(B, OA)); }
                    else { m3_add(U, m3_tmul(B, OA)); m3_add(L, m3_tmul(A, OB)); }
                }
            }
            if (v == f) {                                                // anchor, pose_graph.py:107-112
                D = m3_zero(); D.a[0] = D.a[4] = D.a[8] = 1e10;
                U = m3_zero(); L = m3_zero();
                b[0] = b[1] = b[2] = 0.0;
            }
            if (v + 1 == f) { U = m3_zero(); L = m3_zero(); }
            if (g.dense) {
                if (v == f) { for (int r = 0; r < 3; ++r) g.Hd[(size_t)(3 * v + r) * N3 + 3 * v + r] = 1e10; }
                else for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) g.Hd[(size_t)(3 * v + r) * N3 + 3 * v + cc] += D.a[3 * r + cc];
                for (int c = 0; c < 3; ++c) g.dx[3 * v + c] = -b[c];
            } else {
                // per node: D[v] = H[v][v], U[v] = H[v][v+1] (zero for the last node), L[v] = H[v][v-1] (zero for node 0)
                m3_store(g.D + 9 * (size_t)v, D); m3_store(g.U + 9 * (size_t)v, U);
                if (v + 1 < n) m3_store(g.L + 9 * (size_t)(v + 1), L);
                if (v == 0) m3_store(g.L, m3_zero());
                for (int c = 0; c < 3; ++c) g.X[(size_t)(3 * v + c) * ncols] = -b[c];           // column 0: r = -b
            }
        }
        __syncthreads();
        PG_MARK(0);

        bool singular = false;
        if (g.dense) {
            singular = !block_lu_solve(g.Hd, N3, g.dx, s_val, s_idx);
        } else {
            // ── right-hand sides 1..3k: the Jacobian columns W of the non-chain edges ──
            for (size_t t = tid; t < (size_t)N3 * (size_t)K; t += PG_THREADS) g.X[(t / K) * ncols + 1 + (t % K)] = 0.0;
            __syncthreads();
            for (int q = tid; q < k; q += PG_THREADS) {
                const int eq = g.loop_edge[q];
                const int i = g.ei[eq], j = g.ej[eq];
                double e[3];
                M3 A, B;
                pg_linearise(g.nodes + 3 * i, g.nodes + 3 * j, g.z + 3 * eq, e, A, B);
                m3_store(g.AB + 18 * (size_t)q, A); m3_store(g.AB + 18 * (size_t)q + 9, B);
                for (int a = 0; a < 3; ++a)
                    for (int c = 0; c < 3; ++c) {                        // W_e = [A^T ; B^T]: rows of node i / j, column 3q + a
                        if (i != f) g.X[(size_t)(3 * i + c) * ncols + 1 + 3 * q + a] += A.a[3 * a + c];
                        if (j != f) g.X[(size_t)(3 * j + c) * ncols + 1 + 3 * q + a] += B.a[3 * a + c];
                    }
            }
            __syncthreads();
            PG_MARK(1);
            // ── T^-1 on every column by block cyclic reduction along the chain ───────
            // Level with stride s: the nodes that are odd multiples of s are eliminated into their neighbours at
            // distance s (which then couple at distance 2s); log2(n) levels, every level parallel over nodes and
            // columns.  An eliminated node keeps its inverted diagonal block and its two couplings for the way back.
            int s = 1;
            for (; s < n; s <<= 1) {
                for (int idx = tid; s * (2 * idx + 1) < n; idx += PG_THREADS) {          // odd nodes: invert the diagonal block
                    const int i = s * (2 * idx + 1);
                    M3 inv;
                    if (!m3_inv(m3_load(g.D + 9 * (size_t)i), inv)) s_flag = 1;
                    m3_store(g.D + 9 * (size_t)i, inv);
                }
                __syncthreads();
                for (int idx = tid; 2 * s * idx < n; idx += PG_THREADS) {                // even nodes: absorb both odd neighbours
                    const int e = 2 * s * idx, i1 = e - s, i2 = e + s;
                    M3 P = m3_zero(), Q = m3_zero(), Dn = m3_load(g.D + 9 * (size_t)e), Ln = m3_zero(), Un = m3_zero();
                    if (i1 >= 0) {
                        P = m3_mul(m3_load(g.L + 9 * (size_t)e), m3_load(g.D + 9 * (size_t)i1));
                        m3_sub(Dn, m3_mul(P, m3_load(g.U + 9 * (size_t)i1)));
                        Ln = m3_mul(P, m3_load(g.L + 9 * (size_t)i1));
                        for (int c = 0; c < 9; ++c) Ln.a[c] = -Ln.a[c];
                    }
                    if (i2 < n) {
                        Q = m3_mul(m3_load(g.U + 9 * (size_t)e), m3_load(g.D + 9 * (size_t)i2));
                        m3_sub(Dn, m3_mul(Q, m3_load(g.L + 9 * (size_t)i2)));
                        Un = m3_mul(Q, m3_load(g.U + 9 * (size_t)i2));
                        for (int c = 0; c < 9; ++c) Un.a[c] = -Un.a[c];
                    }
                    m3_store(g.Dinv + 9 * (size_t)e, P); m3_store(g.G + 9 * (size_t)e, Q);
                    m3_store(g.D + 9 * (size_t)e, Dn); m3_store(g.L + 9 * (size_t)e, Ln); m3_store(g.U + 9 * (size_t)e, Un);
                }
                __syncthreads();
                const int n_even = (n + 2 * s - 1) / (2 * s);
                for (int t = tid; t < n_even * ncols; t += PG_THREADS) {                 // right-hand sides of the even nodes
                    const int e = 2 * s * (t / ncols), col = t % ncols, i1 = e - s, i2 = e + s;
                    double r[3], x1[3] = {0.0, 0.0, 0.0}, x2[3] = {0.0, 0.0, 0.0}, t1[3], t2[3];
                    for (int c = 0; c < 3; ++c) r[c] = g.X[(size_t)(3 * e + c) * ncols + col];
                    if (i1 >= 0) for (int c = 0; c < 3; ++c) x1[c] = g.X[(size_t)(3 * i1 + c) * ncols + col];
                    if (i2 < n) for (int c = 0; c < 3; ++c) x2[c] = g.X[(size_t)(3 * i2 + c) * ncols + col];
                    m3_vec(m3_load(g.Dinv + 9 * (size_t)e), x1, t1);
                    m3_vec(m3_load(g.G + 9 * (size_t)e), x2, t2);
                    for (int c = 0; c < 3; ++c) g.X[(size_t)(3 * e + c) * ncols + col] = (r[c] - t1[c]) - t2[c];
                }
                __syncthreads();
            }
            if (tid == 0) {                                                              // the last node standing: node 0
                M3 inv;
                if (!m3_inv(m3_load(g.D), inv)) s_flag = 1;
                m3_store(g.D, inv);
            }
            __syncthreads();
            singular = s_flag != 0;
            if (!singular) {
                for (int col = tid; col < ncols; col += PG_THREADS) {
                    double r[3], x[3];
                    for (int c = 0; c < 3; ++c) r[c] = g.X[(size_t)c * ncols + col];
                    m3_vec(m3_load(g.D), r, x);
                    for (int c = 0; c < 3; ++c) g.X[(size_t)c * ncols + col] = x[c];
                }
                __syncthreads();
                for (s >>= 1; s >= 1; s >>= 1) {                                         // back: the odd nodes of every level
                    const int n_odd = (n - s + 2 * s - 1) / (2 * s);                     // nodes s, 3s, 5s, ... below n
                    for (int t = tid; t < n_odd * ncols; t += PG_THREADS) {
                        const int i = s * (2 * (t / ncols) + 1), col = t % ncols, a = i - s, b = i + s;
                        double r[3], xa[3], xb[3] = {0.0, 0.0, 0.0}, t1[3], t2[3], w[3], x[3];
                        for (int c = 0; c < 3; ++c) { r[c] = g.X[(size_t)(3 * i + c) * ncols + col]; xa[c] = g.X[(size_t)(3 * a + c) * ncols + col]; }
                        if (b < n) for (int c = 0; c < 3; ++c) xb[c] = g.X[(size_t)(3 * b + c) * ncols + col];
                        m3_vec(m3_load(g.L + 9 * (size_t)i), xa, t1);
                        m3_vec(m3_load(g.U + 9 * (size_t)i), xb, t2);
                        for (int c = 0; c < 3; ++c) w[c] = (r[c] - t1[c]) - t2[c];
                        m3_vec(m3_load(g.D + 9 * (size_t)i), w, x);
                        for (int c = 0; c < 3; ++c) g.X[(size_t)(3 * i + c) * ncols + col] = x[c];
                    }
                    __syncthreads();
                }
                PG_MARK(2);
                if (k > 0) {
                    // ── (I + C W^T Y) y = C W^T T^-1 r ───────────────────────────────────
                    for (int t = tid; t < K * (K + 1); t += PG_THREADS) {
                        const int r = t / (K + 1), c = t % (K + 1);                  // c == K: the right-hand side (column 0 of X)
                        const int q = r / 3, a = r % 3;
                        const int eq = g.loop_edge[q];
                        const int i = g.ei[eq], j = g.ej[eq];
                        const int col = c == K ? 0 : 1 + c;
                        const double* AB = g.AB + 18 * (size_t)q;
                        const double* Om = g.omega + 9 * (size_t)eq;
                        double acc = 0.0;
                        for (int bb = 0; bb < 3; ++bb) {
                            double s = 0.0;
                            for (int cc = 0; cc < 3; ++cc) {
                                if (i != f) s += AB[3 * bb + cc] * g.X[(size_t)(3 * i + cc) * ncols + col];
                                if (j != f) s += AB[9 + 3 * bb + cc] * g.X[(size_t)(3 * j + cc) * ncols + col];
                            }
                            acc += Om[3 * a + bb] * s;
                        }
                        if (c == K) g.g[r] = acc;
                        else g.M[(size_t)r * K + c] = acc + (r == c ? 1.0 : 0.0);
                    }
                    __syncthreads();
                    PG_MARK(3);
                    singular = !block_lu_solve(g.M, K, g.g, s_val, s_idx);
                    PG_MARK(4);
                }
                if (!singular) {
                    // dx = Y[:, 0] - Y[:, 1:] y: one wave per row, lanes along the row (coalesced), fixed-tree sum
                    const int l16 = tid & 15;
                    for (int t0 = 0; t0 < N3; t0 += PG_THREADS / 16) {                   // 16 lanes per row, uniform trip count
                        const int t = min(t0 + (tid >> 4), N3 - 1);
                        double part = 0.0;
                        for (int c = l16; c < K; c += 16) part += g.X[(size_t)t * ncols + 1 + c] * g.g[c];
                        const double tot = row_sum(part);
                        if (l16 == 0 && t0 + (tid >> 4) < N3) g.dx[t] = g.X[(size_t)t * ncols] - tot;
                    }
                }
            }
        }
        __syncthreads();
        PG_MARK(5);
        if (singular) { status = PG_SINGULAR; iters = it; break; }           // pose_graph.py:117-119: nodes keep their values
        // ── apply the update, pose_graph.py:121-129 ──────────────────────────────
        double ss = 0.0;
        for (int v = tid; v < n; v += PG_THREADS) {
            const double d0 = g.dx[3 * v], d1 = g.dx[3 * v + 1], d2 = g.dx[3 * v + 2];
            g.nodes[3 * v] += d0;
            g.nodes[3 * v + 1] += d1;
            g.nodes[3 * v + 2] = pg_wrap(g.nodes[3 * v + 2] + d2);
            ss += (d0 * d0 + d1 * d1) + d2 * d2;
        }
        step = sqrt(block_total(ss, s_val));
        PG_MARK(6);
        if (step < g.eps) { status = PG_CONVERGED; iters = it + 1; break; }
    }
    if (tid == 0) {
        g.info[0] = (double)iters; g.info[1] = (double)status; g.info[2] = step;
        for (int i = 0; i < 7; ++i) g.info[3 + i] = (double)t_acc[i] * 0.01;      // microseconds per phase, all iterations
    }
#undef PG_MARK
}

// total_error, pose_graph.py:189-194: e^T Omega e summed in edge order
__global__ __launch_bounds__(PG_THREADS) void pose_graph_error_kernel(const double* nodes, const int32_t* ei, const int32_t* ej,
                                                                      const double* z, const double* omega, int m,
                                                                      double* per_edge, double* out) {
    for (int q = threadIdx.x; q < m; q += PG_THREADS) {
        double e[3], Oe[3];
        M3 A, B;
        pg_linearise(nodes + 3 * ei[q], nodes + 3 * ej[q], z + 3 * q, e, A, B);
        m3_vec(m3_load(omega + 9 * (size_t)q), e, Oe);
        per_edge[q] = (e[0] * Oe[0] + e[1] * Oe[1]) + e[2] * Oe[2];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < m; ++q) t += per_edge[q];
        out[0] = t;
    }
}

struct PgPlan {
    int k = 0, dense = 0;
    std::vector<int32_t> csr_ptr, csr_edge, loop_slot, loop_edge;
};

// Classify the edges (host side: the edge list is a few KB of integers the caller holds on the host anyway).
static bool pg_plan(const int32_t* ij, int n, int m, PgPlan& p) {
    p.csr_ptr.assign(n + 1, 0);
    p.loop_slot.assign(m, -1);
    std::vector<char> linked(n > 1 ? n - 1 : 0, 0);
    for (int q = 0; q < m; ++q) {
        const int i = ij[2 * q], j = ij[2 * q + 1];
        if (i < 0 || i >= n || j < 0 || j >= n) return false;
        ++p.csr_ptr[i + 1];
        if (j != i) ++p.csr_ptr[j + 1];
        if (i - j == 1 || j - i == 1) linked[i < j ? i : j] = 1;
        else { p.loop_slot[q] = p.k++; p.loop_edge.push_back(q); }
    }
    for (int v = 0; v < n; ++v) p.csr_ptr[v + 1] += p.csr_ptr[v];
    p.csr_edge.resize(p.csr_ptr[n]);
    std::vector<int32_t> fill(p.csr_ptr.begin(), p.csr_ptr.end() - 1);
    for (int q = 0; q < m; ++q) {
        const int i = ij[2 * q], j = ij[2 * q + 1];
        p.csr_edge[fill[i]++] = q;
        if (j != i) p.csr_edge[fill[j]++] = q;
    }
    for (char c : linked) if (!c) p.dense = 1;               // a gap in the chain: T would be singular
    if (p.dense) { p.k = 0; p.loop_edge.clear(); std::fill(p.loop_slot.begin(), p.loop_slot.end(), -1); }
    return true;
}

static size_t pg_align(size_t b) { return (b + 255) / 256 * 256; }

static size_t pg_bytes(int n, int m, int k, int dense) {
    const size_t N3 = 3 * (size_t)n, K = 3 * (size_t)k;
    size_t b = 0;
    b += pg_align((size_t)m * 2 * 4);                        // ei, ej
    b += pg_align(((size_t)n + 1) * 4) + pg_align(((size_t)2 * m + 1) * 4) + pg_align((size_t)(m + 1) * 4) + pg_align((size_t)(k + 1) * 4);
    b += 6 * pg_align(9 * (size_t)n * 8);                    // D U L Dinv G + b (oversized)
    b += pg_align(N3 * (1 + K) * 8);                         // X
    b += pg_align(18 * (size_t)(k + 1) * 8);                 // AB
    b += pg_align((K * K + 1) * 8) + 2 * pg_align((K + 1) * 8);
    b += pg_align(N3 * 8);                                   // dx
    b += pg_align((size_t)(m + 1) * 8);                      // per-edge errors
    if (dense) b += pg_align(N3 * N3 * 8);
    return b + 256;
}

}  // namespace icpmi

extern "C" size_t icpmi_pose_graph_workspace_bytes(const int32_t* edges_ij_host, int32_t n_nodes, int32_t n_edges) {
    if (n_nodes < 0 || n_edges < 0 || (n_edges > 0 && !edges_ij_host)) return 0;
    icpmi::PgPlan p;
    if (!icpmi::pg_plan(edges_ij_host, n_nodes, n_edges, p)) return 0;
    return icpmi::pg_bytes(n_nodes, n_edges, p.k, p.dense);
}

extern "C" int icpmi_pose_graph_optimize(double* nodes, const int32_t* edges_ij_host, const double* edges_z,
                                         const double* edges_omega, int32_t n_nodes, int32_t n_edges,
                                         int32_t n_iterations, int32_t fix_node, double convergence_eps,
                                         double* info, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace icpmi;
    if (!nodes || !info || n_nodes < 0 || n_edges < 0 || n_iterations < 0) return ICPMI_ERR_ARG;
    if (n_edges > 0 && (!edges_ij_host || !edges_z || !edges_omega)) return ICPMI_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n_nodes < 2 || n_edges == 0 || n_iterations == 0) {                  // pose_graph.py:89-91: nothing to do
        const double none[10] = {0.0, n_iterations == 0 && n_nodes >= 2 && n_edges > 0 ? (double)PG_MAXITER : (double)PG_NOTHING, 0.0};
        if (hipMemcpyAsync(info, none, sizeof(none), hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return ICPMI_ERR_HIP;    // `none` is on this stack frame
        return ICPMI_OK;
    }
    if (fix_node < 0 || fix_node >= n_nodes) return ICPMI_ERR_ARG;
    PgPlan p;
    if (!pg_plan(edges_ij_host, n_nodes, n_edges, p)) return ICPMI_ERR_ARG;
    if (!workspace || workspace_bytes < pg_bytes(n_nodes, n_edges, p.k, p.dense)) return ICPMI_ERR_WORKSPACE;
    const int n = n_nodes, m = n_edges, k = p.k;
    const size_t N3 = 3 * (size_t)n, K = 3 * (size_t)k;
    unsigned char* w = (unsigned char*)workspace;
    auto take = [&](size_t bytes) { unsigned char* r = w; w += pg_align(bytes); return r; };
    int32_t* d_ei = (int32_t*)take((size_t)m * 2 * 4);
    int32_t* d_ej = d_ei + m;
    int32_t* d_ptr = (int32_t*)take(((size_t)n + 1) * 4);
    int32_t* d_edge = (int32_t*)take(((size_t)2 * m + 1) * 4);
    int32_t* d_slot = (int32_t*)take((size_t)(m + 1) * 4);
    int32_t* d_loop = (int32_t*)take((size_t)(k + 1) * 4);
    PgArgs a{};
    a.D = (double*)take(9 * (size_t)n * 8); a.U = (double*)take(9 * (size_t)n * 8); a.L = (double*)take(9 * (size_t)n * 8);
    a.Dinv = (double*)take(9 * (size_t)n * 8); a.G = (double*)take(9 * (size_t)n * 8); a.b = (double*)take(9 * (size_t)n * 8);
    a.X = (double*)take(N3 * (1 + K) * 8);
    a.AB = (double*)take(18 * (size_t)(k + 1) * 8);
    a.M = (double*)take((K * K + 1) * 8); a.g = (double*)take((K + 1) * 8); a.y = (double*)take((K + 1) * 8);
    a.dx = (double*)take(N3 * 8);
    take((size_t)(m + 1) * 8);
    a.Hd = p.dense ? (double*)take(N3 * N3 * 8) : nullptr;
    std::vector<int32_t> sep(2 * (size_t)m);
    for (int q = 0; q < m; ++q) { sep[q] = edges_ij_host[2 * q]; sep[m + q] = edges_ij_host[2 * q + 1]; }
    // pageable host memory: the runtime stages these copies before returning, so the vectors may go out of scope
    if (hipMemcpyAsync(d_ei, sep.data(), sep.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpyAsync(d_ptr, p.csr_ptr.data(), p.csr_ptr.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
    if (!p.csr_edge.empty() && hipMemcpyAsync(d_edge, p.csr_edge.data(), p.csr_edge.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipMemcpyAsync(d_slot, p.loop_slot.data(), p.loop_slot.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
    if (k > 0 && hipMemcpyAsync(d_loop, p.loop_edge.data(), (size_t)k * 4, hipMemcpyHostToDevice, st) != hipSuccess) return ICPMI_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return ICPMI_ERR_HIP;        // the staging vectors die with this call
    a.nodes = nodes; a.ei = d_ei; a.ej = d_ej; a.z = edges_z; a.omega = edges_omega;
    a.csr_ptr = d_ptr; a.csr_edge = d_edge; a.loop_slot = d_slot; a.loop_edge = d_loop;
    a.n = n; a.m = m; a.k = k; a.dense = p.dense; a.n_iter = n_iterations; a.fix = fix_node; a.eps = convergence_eps;
    a.info = info;
    pose_graph_kernel<<<1, PG_THREADS, 0, st>>>(a);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

extern "C" int icpmi_pose_graph_error(const double* nodes, const int32_t* edges_i, const int32_t* edges_j, const double* edges_z,
                                      const double* edges_omega, int32_t n_edges, double* scratch, double* out, void* stream) {
    if (!out || n_edges < 0) return ICPMI_ERR_ARG;
    if (n_edges > 0 && (!nodes || !edges_i || !edges_j || !edges_z || !edges_omega || !scratch)) return ICPMI_ERR_ARG;
    icpmi::pose_graph_error_kernel<<<1, icpmi::PG_THREADS, 0, (hipStream_t)stream>>>(nodes, edges_i, edges_j, edges_z, edges_omega,
                                                                                  n_edges, scratch, out);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}
