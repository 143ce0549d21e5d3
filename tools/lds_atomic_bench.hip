// Microbenchmark: LDS ds_add_f32 cost model on one 128 KiB accumulator block per workgroup (the row-block
// backward of march_features.hip).  hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/_exp/lds_atomic_bench
//   pattern 0: lane-consecutive words        1: random words          2: one word for the whole wave
//   pattern 3: random, 1 lane in 4 active    4: random row, 2 channels (two ds_add per lane, like C = 2)
//   pattern 5: like 4 but ONE 8-byte read-modify-write under "this wave owns the row" (non-atomic ceiling)
//   pattern 6: like 4 but ONE 8-byte compare-and-swap loop (exact fp32 adds of both channels)
//   pattern 7: like 6, 1 lane in 4 active      8: two ds_add_u64 per row (fixed-point alternative; 8192 rows)
//   pattern 9 / 10: one ds_pk_add_f16 / ds_pk_add_bf16 per 4-byte row (both channels of a C = 2 row as a 16-bit pair)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr uint32_t kWords = 32768;          // 128 KiB
constexpr int kIter = 2048;

template <int PATTERN, bool FLOATS>
__global__ __launch_bounds__(1024) void k_lds(float *out) {
    extern __shared__ float s[];
    for (uint32_t i = threadIdx.x; i < kWords; i += 1024u) s[i] = 0.0f;
    __syncthreads();
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63u;
    for (int it = 0; it < kIter; it++) {
        h = h * 1664525u + 1013904223u;
        uint32_t w;
        bool act = true;
        if (PATTERN == 0) w = (threadIdx.x + it * 1024u) & (kWords - 1u);
        else if (PATTERN == 2) w = ((threadIdx.x >> 6) * 97u + it) & (kWords - 1u);
        else w = (h >> 9) & (kWords - 1u);
        if (PATTERN == 3 || PATTERN == 7) act = ((h >> 5) & 3u) == 0u;
        if (PATTERN >= 4) w &= ~1u;
        if (PATTERN == 8) w &= ~3u;
        if (!act) continue;
        if (PATTERN == 9 || PATTERN == 10) {                      // one packed 2 x 16-bit float add per 4-byte row
            const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)(s + w);
            const uint32_t one2 = PATTERN == 9 ? 0x3C003C00u : 0x3F803F80u;     // (1, 1) as half2 / bf16x2
            if (PATTERN == 9) asm volatile("ds_pk_add_f16 %0, %1" ::"v"(addr), "v"(one2) : "memory");
            else asm volatile("ds_pk_add_bf16 %0, %1" ::"v"(addr), "v"(one2) : "memory");
        } else if (PATTERN == 5) {
            float2 *p = reinterpret_cast<float2 *>(s + w);
            float2 v = *p; v.x += 1.0f; v.y += 1.0f; *p = v;
        } else if (PATTERN == 6 || PATTERN == 7) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(s + w);
            unsigned long long old = *p, seen;
            do {
                seen = old;
                float2 v = __builtin_bit_cast(float2, seen);
                v.x += 1.0f; v.y += 1.0f;
                old = atomicCAS(p, seen, __builtin_bit_cast(unsigned long long, v));
            } while (old != seen);
        } else if (PATTERN == 8) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(s + w);
            atomicAdd(p, 1ull);
            atomicAdd(p + 1, 1ull);
        } else if (FLOATS) {
            atomicAdd(s + w, 1.0f);
            if (PATTERN == 4) atomicAdd(s + w + 1, 1.0f);
        } else {
            atomicAdd(reinterpret_cast<uint32_t *>(s) + w, 1u);
            if (PATTERN == 4) atomicAdd(reinterpret_cast<uint32_t *>(s) + w + 1, 1u);
        }
    }
    __syncthreads();
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < kWords; i += 1024u) acc += FLOATS ? s[i] : (float)reinterpret_cast<uint32_t *>(s)[i];
    if (acc == -1.0f) out[0] = acc;
    (void)lane;
}

template <int PATTERN, bool FLOATS>
static void run(const char *name) {
    float *out; hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds<PATTERN, FLOATS>), hipFuncAttributeMaxDynamicSharedMemorySize, kWords * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_lds<PATTERN, FLOATS>), dim3(256), dim3(1024), kWords * 4, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_lds<PATTERN, FLOATS>), dim3(256), dim3(1024), kWords * 4, 0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu_instr = 16.0 * kIter * ((PATTERN == 4 || PATTERN == 8) ? 2 : 1);            // wave-level ds ops per CU
    const double lanes = per_cu_instr * 64.0 * ((PATTERN == 3 || PATTERN == 7) ? 0.25 : 1.0);
    printf("%-46s %s: %7.3f ms  %6.1f clk / wave-instr  %6.3f lane-ops / clk / CU  (%7.1f G lane-ops/s chip)\n", name,
           FLOATS ? "f32" : "u32", ms, ms * 2.4e6 / per_cu_instr, lanes / (ms * 2.4e6), lanes * 256 / ms / 1e6);
    hipFree(out);
}

int main() {
    run<0, true>("consecutive words");
    run<0, false>("consecutive words");
    run<1, true>("random words");
    run<1, false>("random words");
    run<2, true>("one word per wave");
    run<3, true>("random words, 1/4 lanes active");
    run<4, true>("random rows, 2 channels");
    run<5, true>("random rows, float2 RMW (no atomic, racy)");
    run<6, true>("random rows, float2 CAS loop");
    run<7, true>("random rows, float2 CAS loop, 1/4 lanes");
    run<8, true>("random rows, 2 x ds_add_u64");
    run<9, true>("random rows of 4 B, ds_pk_add_f16");
    run<10, true>("random rows of 4 B, ds_pk_add_bf16");
    return 0;
}
