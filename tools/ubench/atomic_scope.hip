// Micro-benchmark (diagnostic, not shipped): rate of scattered 32-bit atomic adds on a counter region of the size the
// ray-cast kernel uses, by memory scope.  Device (agent) scope must be coherent across the 8 XCDs of an MI355X, whose
// L2s are private; workgroup scope may be served by the issuing XCD's L2.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/atomic_scope.hip -o tools/ubench/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE, bool RET>
__global__ void scatter(uint32_t* counts, uint32_t mask, int per_thread, uint32_t* sink) {
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t idx = (h >> 8) & mask;
        if (RET) acc += __hip_atomic_fetch_add(&counts[idx], 1u, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(&counts[idx], 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == 0xffffffffu) *sink = acc;
}

__global__ void linear(uint32_t* counts, uint32_t mask, int per_thread) {
    uint32_t h = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t idx = (((h >> 8) & mask) & ~63u) + (threadIdx.x & 63);
        atomicAdd(&counts[idx], 1u);
    }
}
__global__ void plain_rmw(uint32_t* counts, uint32_t mask, int per_thread) {
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t idx = (h >> 8) & mask;
        counts[idx] = counts[idx] + 1u;
    }
}

// one XCD's share: workgroups read their XCC id and touch only the slice of the region that belongs to it
__global__ void scatter_xcd(uint32_t* counts, uint32_t mask, int per_thread, uint32_t* xcc_seen) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0) atomicAdd(&xcc_seen[xcc], 1u);
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t slice = (mask + 1) >> 3;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t idx = xcc * slice + ((h >> 8) & (slice - 1));
        __hip_atomic_fetch_add(&counts[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <typename F>
static float timed(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const uint32_t cells = 1u << 23;            // 32 MB of counters
    uint32_t *counts, *sink;
    hipMalloc(&counts, (size_t)cells * 4);
    hipMalloc(&sink, 64);
    hipMemset(sink, 0, 64);
    const int blocks = 4096, threads = 256, per = 64;
    const double n = (double)blocks * threads * per;
    hipMemset(counts, 0, (size_t)cells * 4);
    float t;
    t = timed([&] { scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<blocks, threads>>>(counts, cells - 1, per, sink); }, 5);
    printf("agent scope, no return    : %.3f ms  %.1f G atomics/s\n", t, n / t * 1e-6);
    t = timed([&] { scatter<__HIP_MEMORY_SCOPE_AGENT, true><<<blocks, threads>>>(counts, cells - 1, per, sink); }, 5);
    printf("agent scope, returning    : %.3f ms  %.1f G atomics/s\n", t, n / t * 1e-6);
    t = timed([&] { scatter<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<blocks, threads>>>(counts, cells - 1, per, sink); }, 5);
    printf("workgroup scope, no return: %.3f ms  %.1f G atomics/s (NOT coherent across XCDs: rate only)\n", t, n / t * 1e-6);
    t = timed([&] { scatter<__HIP_MEMORY_SCOPE_WAVEFRONT, false><<<blocks, threads>>>(counts, cells - 1, per, sink); }, 5);
    printf("wavefront scope, no return: %.3f ms  %.1f G atomics/s (rate only)\n", t, n / t * 1e-6);
    for (uint32_t c : {1u << 16, 1u << 18, 1u << 20, 1u << 21, 1u << 23}) {
        t = timed([&] { scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<blocks, threads>>>(counts, c - 1, per, sink); }, 5);
        printf("agent scope, region %6u KB: %.3f ms  %.1f G atomics/s\n", c * 4 / 1024, t, n / t * 1e-6);
    }
    t = timed([&] { linear<<<blocks, threads>>>(counts, cells - 1, per); }, 5);
    printf("agent scope, lanes on consecutive words (one 256-B run per wave-instruction): %.3f ms  %.1f G atomics/s\n", t, n / t * 1e-6);
    t = timed([&] { plain_rmw<<<blocks, threads>>>(counts, cells - 1, per); }, 5);
    printf("non-atomic scattered load+store (rate reference, racy): %.3f ms  %.1f G updates/s\n", t, n / t * 1e-6);
    // correctness of XCD-sliced workgroup-scope counting: totals must equal the number of adds
    hipMemset(counts, 0, (size_t)cells * 4);
    hipMemset(sink, 0, 64);
    hipDeviceSynchronize();
    scatter_xcd<<<blocks, threads>>>(counts, cells - 1, per, sink);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(cells);
    hipMemcpy(h.data(), counts, (size_t)cells * 4, hipMemcpyDeviceToHost);
    uint32_t seen[16];
    hipMemcpy(seen, sink, 64, hipMemcpyDeviceToHost);
    unsigned long long total = 0;
    for (uint32_t v : h) total += v;
    printf("xcd-sliced workgroup scope: sum %llu of %.0f adds; workgroups per XCC id:", total, n);
    for (int i = 0; i < 16; ++i) printf(" %u", seen[i]);
    printf("\n");
    hipMemset(sink, 0, 64);
    t = timed([&] { scatter_xcd<<<blocks, threads>>>(counts, cells - 1, per, sink); }, 5);
    printf("xcd-sliced workgroup scope: %.3f ms  %.1f G atomics/s\n", t, n / t * 1e-6);
    return 0;
}
