// Does VALU work of the SAME wave hide under its MFMAs?  One wave per SIMD issues [1 MFMA, K VALU] x many and the time
// per MFMA is compared with the MFMA-only stream (32 cycles).  VALU flavours: fma on VGPRs, the f32 -> f16 hi/lo split
// of the MLP kernels, v_accvgpr_read of accumulators no MFMA in flight touches.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_bench.hip -o /tmp/b && /tmp/b
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int K>
__global__ __launch_bounds__(512) void k(float *out, int iters, h8 a, h8 b, float seed) {
    f16v acc0, acc1, other;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; other[r] = seed * r; }
    float v[8];
    for (int e = 0; e < 8; e++) v[e] = seed + e;
    h8 hi = a, lo = b;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 6; u++) {
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, lo, acc0, 0, 0, 0);
            if constexpr (MODE == 1) {                 // independent fma chains
#pragma unroll
                for (int q = 0; q < K; q++) v[q % 8] = fmaf(v[q % 8], 1.0001f, 0.5f);
            } else if constexpr (MODE == 2) {          // hi/lo split of 2 values per 7 VALU (cvt, cvt back, sub, cvt, pack)
#pragma unroll
                for (int q = 0; q < K / 4; q++) {
                    const float x = v[q % 8];
                    const _Float16 h = (_Float16)x;
                    const _Float16 l = (_Float16)(x - (float)h);
                    hi[q % 8] = h; lo[q % 8] = l;
                    v[q % 8] = x * 1.0001f;
                }
            } else if constexpr (MODE == 3) {          // read accumulators that are not being written
#pragma unroll
                for (int q = 0; q < K; q++) v[q % 8] += other[(q + u) % 16];
            }
            if constexpr (MODE != 0) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K + 2, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + other[r];
    for (int e = 0; e < 8; e++) s += v[e] + (float)hi[e] + (float)lo[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 6);
}
static int g_threads = 256;          // 256: one wave per SIMD; 512: two
template <int MODE, int K>
void run(float *d, const char *name) {
    h8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)0.5f; b[e] = (_Float16)0.25f; }
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(g_threads), 0, 0, d, 2000, a, b, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(g_threads), 0, 0, d, 20000, a, b, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms, cyc;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD  %-24s K=%2d: %.2f ns per MFMA of one wave = %.2f ns per MFMA of the SIMD; %.1f s_memtime ticks per MFMA of one wave -> %.2f GHz if a tick is a shader cycle\n",
           g_threads / 256, name, K, ms * 1e6 / (20000.0 * 6), ms * 1e6 / (20000.0 * 6) / (g_threads / 256), cyc, cyc / (ms * 1e6 / (20000.0 * 6)));
}
int main() {
    float *d;
    hipMalloc(&d, 256 * 512 * 4);
    for (g_threads = 256; g_threads <= 512; g_threads += 256) {
    run<0, 0>(d, "MFMA only");
    run<1, 4>(d, "fma"); run<1, 7>(d, "fma"); run<1, 12>(d, "fma");
    run<2, 8>(d, "hi/lo split (cvt)"); run<2, 16>(d, "hi/lo split (cvt)");
    run<3, 4>(d, "accumulator reads"); run<3, 7>(d, "accumulator reads"); run<3, 12>(d, "accumulator reads");
    }
    return 0;
}
