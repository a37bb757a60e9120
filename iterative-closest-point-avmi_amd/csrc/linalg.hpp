// linalg.hpp — the tiny dense solves of the ICP step, evaluated redundantly by
// every lane on wave-uniform data (no divergence, no broadcast needed).
#pragma once
#include "common.hpp"

namespace icpmi {

// 3x3 solve by LU with partial pivoting — what np.linalg.solve (LAPACK gesv)
// does at reference utilities/icp.py:106.  Returns false when a pivot is exactly
// zero (NumPy raises LinAlgError there and the caller falls back to identity,
// icp.py:107-108).  Rows are swapped by value so nothing is indexed dynamically.
__device__ __forceinline__ bool solve3(double (&A)[3][3], double (&b)[3], double (&x)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        double bv = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < 3; ++r)
            if (fabs(A[r][c]) > bv) { bv = fabs(A[r][c]); piv = r; }
#pragma unroll
        for (int r = c + 1; r < 3; ++r)
            if (piv == r) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double t = A[c][q]; A[c][q] = A[r][q]; A[r][q] = t; }
                const double t = b[c]; b[c] = b[r]; b[r] = t;
            }
        if (A[c][c] == 0.0) return false;
#pragma unroll
        for (int r = c + 1; r < 3; ++r) {
            const double f = A[r][c] / A[c][c];
#pragma unroll
            for (int q = c + 1; q < 3; ++q) A[r][q] -= f * A[c][q];
            b[r] -= f * b[c];
        }
    }
    x[2] = b[2] / A[2][2];
    x[1] = (b[1] - A[1][2] * x[2]) / A[1][1];
    x[0] = ((b[0] - A[0][1] * x[1]) - A[0][2] * x[2]) / A[0][0];
    return true;
}

// sin and cos of the point-to-line angle (icp.py:110-111).  |theta| is small in
// any converging registration: the Taylor series to x^19 / x^18 is exact to
// double rounding below 0.25 rad and costs ~25 multiply-adds instead of the
// library's argument reduction; larger angles take the library path.
__device__ __forceinline__ void sincos_step(double x, double& s, double& c) {
    if (fabs(x) < 0.25) {
        const double z = x * x;
        double ps = -1.0 / 121645100408832000.0;                 // -1/19!
        ps = ps * z + 1.0 / 355687428096000.0;                   //  1/17!
        ps = ps * z - 1.0 / 1307674368000.0;                     // -1/15!
        ps = ps * z + 1.0 / 6227020800.0;                        //  1/13!
        ps = ps * z - 1.0 / 39916800.0;                          // -1/11!
        ps = ps * z + 1.0 / 362880.0;                            //  1/9!
        ps = ps * z - 1.0 / 5040.0;                              // -1/7!
        ps = ps * z + 1.0 / 120.0;                               //  1/5!
        ps = ps * z - 1.0 / 6.0;                                 // -1/3!
        s = x + x * (z * ps);
        double pc = 1.0 / 6402373705728000.0;                    //  1/18!
        pc = pc * z - 1.0 / 20922789888000.0;                    // -1/16!
        pc = pc * z + 1.0 / 87178291200.0;                       //  1/14!
        pc = pc * z - 1.0 / 479001600.0;                         // -1/12!
        pc = pc * z + 1.0 / 3628800.0;                           //  1/10!
        pc = pc * z - 1.0 / 40320.0;                             // -1/8!
        pc = pc * z + 1.0 / 720.0;                               //  1/6!
        pc = pc * z - 1.0 / 24.0;                                // -1/4!
        pc = pc * z + 0.5;                                       //  1/2!
        c = 1.0 - z * pc;
    } else {
        sincos(x, &s, &c);
    }
}

// Optimal proper rotation for the 2x2 cross-covariance W = sum pc qc^T: what
// r = V U^T with the det<0 fix evaluates to (reference icp.py:202-206).
// tr(R W) = c (W00+W11) + s (W01-W10) is maximal at (c, s) parallel to those.
__device__ __forceinline__ void kabsch2(const double (&W)[4], double (&r)[4]) {
    const double a = W[0] + W[3], b = W[1] - W[2];
    const double n = sqrt(a * a + b * b);
    if (n > 0.0) {
        const double c = a / n, s = b / n;
        r[0] = c; r[1] = -s; r[2] = s; r[3] = c;
    } else {
        r[0] = 1.0; r[1] = 0.0; r[2] = 0.0; r[3] = 1.0;
    }
}

// 3x3 case: one-sided Jacobi SVD W = U S V^T (columns sorted by descending
// singular value), r = V U^T, reflection fixed on the last column of V.
__device__ inline void kabsch3(const double (&W)[9], double (&r)[9]) {
    double A[9], V[9];
    for (int i = 0; i < 9; ++i) { A[i] = W[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double offn = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; ++i) { a += A[i * 3 + p] * A[i * 3 + p]; b += A[i * 3 + q] * A[i * 3 + q]; c += A[i * 3 + p] * A[i * 3 + q]; }
                if (c == 0.0) continue;
                offn += fabs(c) / sqrt(a * b + 1e-300);
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    double x = A[i * 3 + p], y = A[i * 3 + q];
                    A[i * 3 + p] = cs * x - sn * y; A[i * 3 + q] = sn * x + cs * y;
                    x = V[i * 3 + p]; y = V[i * 3 + q];
                    V[i * 3 + p] = cs * x - sn * y; V[i * 3 + q] = sn * x + cs * y;
                }
            }
        if (offn < 1e-15) break;
    }
    double sv[3];
    int ord[3] = {0, 1, 2};
    for (int j = 0; j < 3; ++j) { double s = 0; for (int i = 0; i < 3; ++i) s += A[i * 3 + j] * A[i * 3 + j]; sv[j] = sqrt(s); }
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (sv[ord[j]] > sv[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double U[9], Vs[9], S[3];
    for (int jj = 0; jj < 3; ++jj) {
        const int j = ord[jj];
        S[jj] = sv[j];
        for (int i = 0; i < 3; ++i) { Vs[i * 3 + jj] = V[i * 3 + j]; U[i * 3 + jj] = sv[j] > 0 ? A[i * 3 + j] / sv[j] : 0.0; }
    }
    if (S[0] > 0 && S[1] > 0 && !(S[2] > 0)) {       // rank 2: complete U with the cross product
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { double s = 0; for (int c = 0; c < 3; ++c) s += Vs[a * 3 + c] * U[b * 3 + c]; r[a * 3 + b] = s; }
    const double det = r[0] * (r[4] * r[8] - r[5] * r[7]) - r[1] * (r[3] * r[8] - r[5] * r[6]) + r[2] * (r[3] * r[7] - r[4] * r[6]);
    if (det < 0) {
        for (int a = 0; a < 3; ++a) Vs[a * 3 + 2] = -Vs[a * 3 + 2];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { double s = 0; for (int c = 0; c < 3; ++c) s += Vs[a * 3 + c] * U[b * 3 + c]; r[a * 3 + b] = s; }
    }
}

// Unit eigenvector of the smaller eigenvalue of [[a,b],[b,c]] — column 0 of
// np.linalg.eigh at reference icp.py:72-73, up to sign.
__device__ __forceinline__ void smallest_evec_2x2(double a, double b, double c, double& vx, double& vy) {
    const double h = 0.5 * (a - c);
    const double rad = sqrt(h * h + b * b);
    const double lam = 0.5 * (a + c) - rad;
    const double x1 = b, y1 = lam - a;
    const double x2 = lam - c, y2 = b;
    const double n1 = x1 * x1 + y1 * y1, n2 = x2 * x2 + y2 * y2;
    double x = x2, y = y2, nn = n2;
    if (n1 >= n2) { x = x1; y = y1; nn = n1; }
    if (nn == 0.0) { vx = 1.0; vy = 0.0; return; }
    nn = sqrt(nn);
    vx = x / nn; vy = y / nn;
}

}  // namespace icpmi
