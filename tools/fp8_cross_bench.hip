// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, E8M0 block scales) for the split product's cross terms:
//   (1) operand layout: hypothesis lane l holds row / column l % 32 and k = 32 (l / 32) + byte (8 VGPRs in order);
//   (2) scale semantics: lane l's scale byte applies to its row's k-block l / 32;
//   (3) rate against v_mfma_f32_32x32x16_f16 (cycles per instruction, one wave per SIMD and two).
//   hipcc --offload-arch=gfx950 -O3 tools/fp8_cross_bench.hip -o /tmp/f8 && /tmp/f8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// OCP e4m3fn decode (host)
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

__global__ void k_probe(const uint8_t *A /*[32][64] row-major bytes*/, const uint8_t *B /*[64][32]*/, const uint8_t *sa /*[32][2]*/,
                        const uint8_t *sb /*[2][32]*/, float *C /*[32][32]*/, int layout) {
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    i8v a, b;
    for (int v = 0; v < 8; v++) {
        uint32_t wa = 0, wb = 0;
        for (int q = 0; q < 4; q++) {
            const int byte = 4 * v + q;
            const int k = layout == 0 ? 32 * h + byte : (16 * h + (byte % 16) + 32 * (byte / 16));
            wa |= (uint32_t)A[j * 64 + k] << (8 * q);
            wb |= (uint32_t)B[k * 32 + j] << (8 * q);
        }
        a[v] = (int)wa; b[v] = (int)wb;
    }
    f16v c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    const int scale_a = sa[j * 2 + h], scale_b = sb[h * 32 + j];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = c[r];
}

template <int MODE>
__global__ __launch_bounds__(512) void k_rate(float *out, int iters, i8v a8, i8v b8, h8 ah, h8 bh) {
    f16v acc0, acc1;
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (MODE == 0) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
            } else if constexpr (MODE == 1) {
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc0, 0, 0, 0, 127, 0, 116);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc1, 0, 0, 0, 127, 0, 116);
            } else {                  // the mix the split product would issue: 2 f16 + 1 fp8 per input tile
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc0, 0, 0, 0, 127, 0, 116);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc1, 0, 0, 0, 127, 0, 116);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (float)(t1 - t0) / (float)iters;
}

template <int MODE>
static void rate(float *d, int threads, const char *name, int per_iter) {
    i8v a8, b8; h8 ah, bh;
    for (int e = 0; e < 8; e++) { a8[e] = 0x38383838; b8[e] = 0x30303030; ah[e] = (_Float16)0.5f; bh[e] = (_Float16)0.25f; }
    hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(threads), 0, 0, d, 4000, a8, b8, ah, bh);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(threads), 0, 0, d, 4000, a8, b8, ah, bh);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms, ticks;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&ticks, d, 4, hipMemcpyDeviceToHost);
    printf("%-44s %d waves/SIMD: %.3f ms, %.1f ns per MFMA per wave (%d per iteration)\n", name, threads / 256, ms,
           ms * 1e6 / 4000.0 / per_iter, per_iter);
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(64 * 32), sa(64), sb(64);
    srand(3);
    auto rnd = [] { uint8_t v; do { v = (uint8_t)(rand() & 0xff); } while (((v >> 3) & 15) == 15 || ((v >> 3) & 15) < 3); return v; };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (int i = 0; i < 64; i++) { sa[i] = 127 - (i % 3) - 4 * (i & 1); sb[i] = 127 + (i % 2) - 3 * (i / 32); }
    uint8_t *dA, *dB, *dsa, *dsb; float *dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 4096 + 512 * 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    for (int layout = 0; layout < 2; layout++) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, layout);
        std::vector<float> C(1024);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        double worst = 0, worst_noscale = 0, mag = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                double ref = 0, ref1 = 0;
                for (int k = 0; k < 64; k++) {
                    const double p = (double)e4m3(A[i * 64 + k]) * (double)e4m3(B[k * 32 + j]);
                    ref += p * ldexp(1.0, (int)sa[i * 2 + k / 32] - 127) * ldexp(1.0, (int)sb[(k / 32) * 32 + j] - 127);
                    ref1 += p;
                }
                worst = fmax(worst, fabs(C[i * 32 + j] - ref));
                worst_noscale = fmax(worst_noscale, fabs(C[i * 32 + j] - ref1));
                mag = fmax(mag, fabs(ref));
            }
        printf("layout %d: max |C - ref(scaled)| = %.3e, vs unscaled ref %.3e (|ref| up to %.3e)\n", layout, worst, worst_noscale, mag);
    }
    for (int t = 256; t <= 512; t += 256) {
        rate<0>(dC, t, "v_mfma_f32_32x32x16_f16", 8);
        rate<1>(dC, t, "v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3)", 8);
        rate<2>(dC, t, "mix 2 f16 + 1 fp8 per input tile", 24);
    }
    return 0;
}
