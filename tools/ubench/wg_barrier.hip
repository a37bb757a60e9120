// Micro-benchmark (diagnostic, not shipped): what a barrier between the workgroups of a cluster costs on an MI355X —
// the price of spreading ONE scan pair of the fused ICP loop over several CUs (two such barriers per iteration:
// partial sums in, transform out).  Clusters of C workgroups (one per CU, launched together, far fewer than CUs)
// meet at a counter in global memory: arrive = device-scope atomic add, wait = spin on an atomic load.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/wg_barrier.hip -o tools/ubench/wg_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void cluster_barriers(uint32_t* counters, int cluster, int rounds, long long* cycles) {
    const int c = blockIdx.x / cluster;
    uint32_t* ctr = counters + 32 * c;                       // one 128-byte line per cluster
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = (uint32_t)r * (uint32_t)cluster;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x % cluster == 0) cycles[c] = wall_clock64() - t0;
}

int main() {
    int rate_khz = 0;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    uint32_t* counters; long long* cycles;
    (void)hipMalloc(&counters, 64 * 128); (void)hipMalloc(&cycles, 64 * 8);
    const int rounds = 2000;
    for (int cluster : {1, 2, 4, 8}) {
        for (int n_clusters : {1, 16}) {
            (void)hipMemset(counters, 0, 64 * 128);
            cluster_barriers<<<cluster * n_clusters, 1024>>>(counters, cluster, rounds, cycles);
            (void)hipDeviceSynchronize();
            long long h[64];
            (void)hipMemcpy(h, cycles, n_clusters * 8, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int i = 0; i < n_clusters; ++i) worst = h[i] > worst ? (double)h[i] : worst;
            printf("cluster of %d workgroups x 1024 threads, %2d clusters: %.2f us per barrier\n", cluster, n_clusters,
                   worst / rounds / (rate_khz * 1e-3));
        }
    }
    return 0;
}
