// Throughput of v_mfma_f32_32x32x16_f16 issued by ONE wave per SIMD with N independent accumulator chains:
// MFMAs on the same accumulator are dependent, so N chains give the pipe N instructions per result latency.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_bench.hip -o /tmp/mfma_chain_bench && /tmp/mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int N>
__global__ __launch_bounds__(256) void k(float *out, int iters, h8 a0, h8 b0) {
    f16v acc[N];
    h8 a = a0, b = b0;
    for (int n = 0; n < N; n++)
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 12 / N; u++)
#pragma unroll
            for (int n = 0; n < N; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int n = 0; n < N; n++)
        for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 12);
}
template <int N>
void run(float *d) {
    h8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)0.5f; b[e] = (_Float16)0.25f; }
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, d, 2000, a, b);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<N>, dim3(256), dim3(256), 0, 0, d, 20000, a, b);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms, cyc;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    const double mf = 20000.0 * 12;
    printf("chains %d: %.1f counter ticks per MFMA (s_memtime), %.2f ns per MFMA per wave -> %.0f TF on 1024 SIMDs\n", N, cyc,
           ms * 1e6 / mf, 1024 * 32768.0 / (ms * 1e6 / mf) / 1e3);
}
int main() {
    float *d;
    hipMalloc(&d, 256 * 256 * 4);
    run<1>(d); run<2>(d); run<3>(d); run<4>(d); run<6>(d);
    return 0;
}
