// Micro-benchmark (diagnostic): sustained FP64 VALU rate of the K1 instruction mix
// (v_add_f64 / v_mul_f64 / v_min_f64, no FMA) with everything in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1.0000001, d = 0.999;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
        if (MODE == 1) { a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c; }
        if (MODE == 2) { a0 = fmin(a0 + c, a1); a1 = fmin(a1 * d, a2); a2 = fmin(a2 + c, a3); a3 = fmin(a3 * d, a0);
                         a4 = fmin(a4 + c, a5); a5 = fmin(a5 * d, a6); a6 = fmin(a6 + c, a7); a7 = fmin(a7 * d, a4); }
        if (MODE == 3) { a0 = __builtin_fma(a0, c, d); a1 = __builtin_fma(a1, c, d); a2 = __builtin_fma(a2, c, d); a3 = __builtin_fma(a3, c, d);
                         a4 = __builtin_fma(a4, c, d); a5 = __builtin_fma(a5, c, d); a6 = __builtin_fma(a6, c, d); a7 = __builtin_fma(a7, c, d); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char* name, int ops_per_iter, int waves_per_simd) {
    double* out; hipMalloc(&out, 8 * 256 * 4096);
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256 CUs x 4 SIMDs; 4 waves per block
    k<MODE><<<blocks, 256>>>(out, 100, 1.0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * ops_per_iter;
    printf("%-28s waves/SIMD=%d  %.2f T lane-ops/s (%.3f ms)\n", name, waves_per_simd, ops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", 8, w); run<1>("v_mul_f64", 8, w); run<2>("add/mul + min (16 ops)", 16, w); run<3>("v_fma_f64 (1 op each)", 8, w);
    }
    return 0;
}
