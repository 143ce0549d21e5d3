// LDS-DMA (global_load_lds_dwordx4) fill rate per CU: every workgroup streams `span` bytes of a buffer into its LDS over and over.
// span = 16 KiB: the source stays in the CU's L1; span = 512 KiB: it comes from L2 (every workgroup reads the SAME bytes, like the
// MLP's weight stream); offset per workgroup: distinct L2 lines.   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_bench.hip -o /tmp/d
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float *src, uint32_t span_kb, uint32_t wg_stride_kb, int iters, float *out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)lds + wave * 1024u;
    const char *base = (const char *)src + (size_t)blockIdx.x * wg_stride_kb * 1024u;
    const uint32_t voff = lane * 16u;
    uint32_t kb = wave;                                       // this wave's next KiB of the span
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const char *g = base + (size_t)kb * 1024u;
            const uint32_t l = lbase + (uint32_t)((p & 7) * WAVES) * 1024u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(l), "v"(voff), "s"(g) : "memory");
            kb += WAVES;
            if (kb >= span_kb) kb -= span_kb;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
int main() {
    float *src, *out;
    hipMalloc(&src, 600u << 20);
    hipMemset(src, 0, 600u << 20);
    hipMalloc(&out, 4096 * 4);
    for (int waves : {4, 8}) for (int wgs_per_cu : {1, 2}) for (uint32_t span : {16u, 64u, 512u}) for (uint32_t stride : {0u, 1024u}) {
        const int iters = 2000, grid = 256 * wgs_per_cu;
        if (stride * grid > (500u << 10)) continue;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int n) {
            if (waves == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 8 * 4 * 1024, 0, src, span, stride, n, out);
            else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 8 * 8 * 1024, 0, src, span, stride, n, out);
        };
        launch(100); hipDeviceSynchronize();
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes_per_cu = (double)iters * 8 * waves * 1024 * wgs_per_cu;
        printf("%d waves x %d WG/CU, span %3u KiB, per-WG offset %4u KiB: %.3f ms  %.1f GB/s per CU = %.1f B/clk at 2.4 GHz, %.2f TB/s aggregate\n",
               waves, wgs_per_cu, span, stride, ms, bytes_per_cu / ms / 1e6, bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * 256 / ms / 1e9);
    }
    return 0;
}
