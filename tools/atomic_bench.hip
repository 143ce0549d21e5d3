// Microbenchmark: throughput of scattered fp32 atomic adds on a 4 MiB table (one hashed grid level),
// by memory scope.  hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o tools/_exp/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int SCOPE, int C>
__global__ __launch_bounds__(256) void k_scatter(float *tab, uint32_t rows, uint32_t n, uint32_t xcd_part) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t h = i * 2654435761u;
    h ^= h >> 15; h *= 805459861u; h ^= h >> 13;
    uint32_t row = h % rows;
    if (xcd_part) row = (row & ~7u) | (blockIdx.x & 7u);       // rows owned by "this" XCD (round-robin guess)
    float *p = tab + (size_t)row * C;
#pragma unroll
    for (int c = 0; c < C; c++) {
        if (SCOPE == 0) atomicAdd(p + c, 1.0f);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else p[c] += 1.0f;                                       // plain RMW (racy): the non-atomic ceiling
    }
}

template <int SCOPE, int C>
static void run(const char *name, float *tab, uint32_t rows, uint32_t n, uint32_t part) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(tab, 0, (size_t)rows * C * 4);
    hipLaunchKernelGGL((k_scatter<SCOPE, C>), dim3((n + 255) / 256), dim3(256), 0, 0, tab, rows, n, part);
    hipDeviceSynchronize();
    hipMemset(tab, 0, (size_t)rows * C * 4);
    hipEventRecord(e0);
    for (int it = 0; it < 5; it++)
        hipLaunchKernelGGL((k_scatter<SCOPE, C>), dim3((n + 255) / 256), dim3(256), 0, 0, tab, rows, n, part);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // checksum: total must equal 5 * n * C if no update was lost
    float *h = (float *)malloc((size_t)rows * C * 4);
    hipMemcpy(h, tab, (size_t)rows * C * 4, hipMemcpyDeviceToHost);
    double s = 0; for (size_t k = 0; k < (size_t)rows * C; k++) s += h[k];
    free(h);
    printf("%-34s C=%d part=%u: %7.3f ms  %7.1f G row-updates/s  sum/expected = %.6f\n", name, C, part, ms, n / ms / 1e6,
           s / (5.0 * n * C));
}

int main() {
    const uint32_t rows = 524288, n = 50331648;   // 8192 rays x 128 samples x 48 corners
    float *tab; hipMalloc(&tab, (size_t)rows * 2 * 4);
    run<0, 2>("agent scope (atomicAdd)", tab, rows, n, 0);
    run<1, 2>("workgroup scope", tab, rows, n, 0);
    run<2, 2>("wavefront scope", tab, rows, n, 0);
    run<3, 2>("plain RMW (racy)", tab, rows, n, 0);
    run<0, 2>("agent scope, XCD-partitioned rows", tab, rows, n, 1);
    run<1, 2>("workgroup scope, XCD-partitioned", tab, rows, n, 1);
    run<0, 1>("agent scope (atomicAdd)", tab, rows, n, 0);
    run<1, 1>("workgroup scope", tab, rows, n, 0);
    return 0;
}
