// prep_big.hip — prepare a target cloud that does not fit the LDS path of prep.hip
// (more than 4096 points: the rolling submap of slam.py:103-108,217 is ~10 k).
//
// Same result as prep_targets_kernel — search axis, copy sorted along it, row map,
// optional normals (reference utilities/icp.py:51-76) — as four steps on global
// memory: axis choice + sort keys (one workgroup), rocPRIM radix sort of
// (key, row), gather, and the k-NN sweep of prep.hip reading the sorted copy
// through L2 instead of LDS.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "prep_common.hpp"

namespace icpmi {

constexpr int BIG_THREADS = 1024;
constexpr int BIG_MAXW = BIG_THREADS / ICPMI_WAVE;

__global__ __launch_bounds__(BIG_THREADS) void big_axis_keys_kernel(
    const double* __restrict__ P, const int32_t* __restrict__ cnt_c, int n_cap, int32_t* __restrict__ dir_out,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ rows) {
    __shared__ double dsc[8 * BIG_MAXW];
    __shared__ int hist[4 * PREP_BINS];
    const int M = cnt_c ? min(*cnt_c, n_cap) : n_cap;
    const int dir = choose_axis<BIG_THREADS>(P, M, dsc, hist);
    if (threadIdx.x == 0) *dir_out = M > 0 ? dir : -1;
    for (int i = threadIdx.x; i < n_cap; i += BIG_THREADS) {
        keys[i] = i < M ? f64_sortable(proj(dir, P[2 * i], P[2 * i + 1])) : ~0ull;   // padding sorts last
        rows[i] = (uint32_t)i;
    }
}

__global__ void big_gather_kernel(const double* __restrict__ P, const int32_t* __restrict__ cnt_c, int n_cap,
                                  const uint32_t* __restrict__ rows, double2* __restrict__ o_sxy,
                                  int32_t* __restrict__ o_sorig) {
    const int M = cnt_c ? min(*cnt_c, n_cap) : n_cap;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int row = (int)rows[i];
    o_sxy[i] = make_double2(P[2 * row], P[2 * row + 1]);
    o_sorig[i] = row;
}

template <int KK>
__global__ __launch_bounds__(256) void big_normals_kernel(
    const int32_t* __restrict__ cnt_c, int n_cap, const int32_t* __restrict__ dir_c, int k,
    const double2* __restrict__ sxy, const int32_t* __restrict__ sorig, double2* __restrict__ o_snrm,
    double* __restrict__ o_rows) {
    const int M = cnt_c ? min(*cnt_c, n_cap) : n_cap;
    const int s0 = blockIdx.x * 256;
    if (s0 >= M) return;
    const int kk = min(k, M - 1) + 1;                            // icp.py:61,66
    prep_normals<KK>(sxy, sorig, M, s0, min(M, s0 + 256), *dir_c, kk, o_snrm, o_rows);
}

static size_t align256b(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t big_radix_temp(int n) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0, false);
    return bytes;
}

size_t prep_big_scratch_bytes(int n) {
    return 2 * align256b((size_t)n * 8) + 2 * align256b((size_t)n * 4) + align256b(big_radix_temp(n)) + 256;
}

// one cloud; P = its first row, cnt_c = its device-side row count (or null), outputs already offset to the cloud
int prep_big_cloud(const double* P, const int32_t* cnt_c, int n_cap, int normal_k, double2* o_sxy, double2* o_snrm,
                   int32_t* o_sorig, int32_t* dir_c, double* o_rows, void* scratch, size_t scratch_bytes, hipStream_t st) {
    if (scratch_bytes < prep_big_scratch_bytes(n_cap)) return ICPMI_ERR_WORKSPACE;
    unsigned char* b = (unsigned char*)scratch;
    size_t o = 0;
    uint64_t* k0 = (uint64_t*)(b + o); o += align256b((size_t)n_cap * 8);
    uint64_t* k1 = (uint64_t*)(b + o); o += align256b((size_t)n_cap * 8);
    uint32_t* r0 = (uint32_t*)(b + o); o += align256b((size_t)n_cap * 4);
    uint32_t* r1 = (uint32_t*)(b + o); o += align256b((size_t)n_cap * 4);
    size_t tb = big_radix_temp(n_cap);
    big_axis_keys_kernel<<<1, BIG_THREADS, 0, st>>>(P, cnt_c, n_cap, dir_c, k0, r0);
    if (rocprim::radix_sort_pairs(b + o, tb, k0, k1, r0, r1, (size_t)n_cap, 0, 64, st, false) != hipSuccess) return ICPMI_ERR_HIP;
    const int blocks = (n_cap + 255) / 256;
    big_gather_kernel<<<blocks, 256, 0, st>>>(P, cnt_c, n_cap, r1, o_sxy, o_sorig);
    if (normal_k >= 0) {
        if (normal_k + 1 <= 8) big_normals_kernel<8><<<blocks, 256, 0, st>>>(cnt_c, n_cap, dir_c, normal_k, o_sxy, o_sorig, o_snrm, o_rows);
        else if (normal_k + 1 <= 13) big_normals_kernel<13><<<blocks, 256, 0, st>>>(cnt_c, n_cap, dir_c, normal_k, o_sxy, o_sorig, o_snrm, o_rows);
        else if (normal_k + 1 <= 16) big_normals_kernel<16><<<blocks, 256, 0, st>>>(cnt_c, n_cap, dir_c, normal_k, o_sxy, o_sorig, o_snrm, o_rows);
        else big_normals_kernel<32><<<blocks, 256, 0, st>>>(cnt_c, n_cap, dir_c, normal_k, o_sxy, o_sorig, o_snrm, o_rows);
    }
    ICPMI_LAUNCH_CHECK();
    return ICPMI_OK;
}

}  // namespace icpmi
