// misc.hip — version / error strings and the stand-alone point-to-line step.
#include "linalg.hpp"

namespace icpmi {

constexpr int P2L_THREADS = 512;

// _point_to_line_solve_2d, reference utilities/icp.py:79-115, over K given
// correspondences: wavefront reduction of the 6 + 3 normal-equation sums, then
// the 3x3 solve, evaluated by every lane on the reduced (uniform) sums.
__global__ __launch_bounds__(P2L_THREADS) void p2l_solve_kernel(
    const double* __restrict__ src, int K, const double* __restrict__ tgt, const double* __restrict__ nrm,
    const int32_t* __restrict__ idx, double* __restrict__ out) {
    __shared__ double red[block_sum_doubles<9>()];
    block_sum_init(red, block_sum_doubles<9>());
    __syncthreads();
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < K; i += P2L_THREADS) {
        const int j = idx[i];
        const double px = src[2 * i], py = src[2 * i + 1];
        const double nx = nrm[2 * j], ny = nrm[2 * j + 1];
        const double dx = px - tgt[2 * j], dy = py - tgt[2 * j + 1];
        const double c = ny * px - nx * py;           // icp.py:97
        const double bi = -(nx * dx + ny * dy);       // icp.py:101
        acc[0] += c * c;  acc[1] += c * nx;  acc[2] += c * ny;
        acc[3] += nx * nx; acc[4] += nx * ny; acc[5] += ny * ny;
        acc[6] += c * bi; acc[7] += nx * bi; acc[8] += ny * bi;
    }
    block_sum<9, P2L_THREADS / ICPMI_WAVE>(acc, red);
    if (threadIdx.x == 0) {
        double A[3][3] = {{acc[0], acc[1], acc[2]}, {acc[1], acc[3], acc[4]}, {acc[2], acc[4], acc[5]}};
        double rhs[3] = {acc[6], acc[7], acc[8]}, x[3];
        if (solve3(A, rhs, x)) {
            double st, ct;
            sincos_step(x[0], st, ct);
            out[0] = ct; out[1] = -st; out[2] = st; out[3] = ct; out[4] = x[1]; out[5] = x[2];
        } else {                                      // icp.py:107-108
            out[0] = 1.0; out[1] = 0.0; out[2] = 0.0; out[3] = 1.0; out[4] = 0.0; out[5] = 0.0;
        }
    }
}

}  // namespace icpmi

extern "C" int icpmi_p2l_solve_2d(const double* src, int32_t n_src, const double* tgt, const double* normals,
                                  const int32_t* nn_idx, double* out_Rt, void* stream) {
    if (!out_Rt || n_src < 0 || (n_src > 0 && (!src || !tgt || !normals || !nn_idx))) return ICPMI_ERR_ARG;
    icpmi::p2l_solve_kernel<<<1, icpmi::P2L_THREADS, 0, (hipStream_t)stream>>>(src, n_src, tgt, normals, nn_idx, out_Rt);
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

extern "C" const char* icpmi_version(void) { return "icpmi 0.1 (gfx950)"; }

extern "C" const char* icpmi_strerror(int code) {
    switch (code) {
        case ICPMI_OK: return "ok";
        case ICPMI_ERR_ARG: return "bad argument";
        case ICPMI_ERR_WORKSPACE: return "workspace missing or too small";
        case ICPMI_ERR_HIP: return "HIP runtime error";
        case ICPMI_ERR_UNSUPPORTED: return "unsupported size or parameter";
        default: return "unknown error";
    }
}
