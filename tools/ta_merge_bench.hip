// Does the texture-address path merge two lanes of ONE load instruction that hit the same 64-byte line?
// The fine hashed levels of k_march_features are bound by L1 line requests (~36 per sample and level: every corner pair its own line,
// 64 distinct lines per wave instruction).  The corners (x0, y, z) and (x0 + 1, y, z) hash to rows r and r ^ d with d < 8 in 7 cases
// of 8 -- the same aligned group of eight 8-byte rows, i.e. the SAME 64-byte line -- but 8 .. 56 bytes apart, so no single <= 16-byte
// load covers both.  Variants, same rows fetched in each:
//   0: lane = point; two 8-byte load instructions (row r, then row r ^ d)                                  [what the kernel does]
//   1: lane pair (2 i, 2 i + 1) = point i: ONE instruction fetches r (even lane) and r ^ d (odd lane); 2 instructions per 64 points
//   2: like 1 but the partner lanes are 32 apart (lane i and lane i + 32)
//   3: lane = point; one 16-byte load of the aligned pair (d = 1 only: the kernel's even-x0 case), for scale
//   4-7: variant 1 with cache-policy bits on the load (nt / sc0 / sc1 / all three); 8: variant 1 through the same inline asm (waitcnt per load: the control)
// hipcc --offload-arch=gfx950 -O3 tools/ta_merge_bench.hip -o tools/_exp/ta_merge_bench && tools/_exp/ta_merge_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr uint32_t kRows = 1u << 19;       // one level slice: 2^19 rows x 2 floats = 4 MiB

template <int V>
__global__ __launch_bounds__(256) void k(const float2 *__restrict__ tab, float *__restrict__ out, uint32_t iters, uint32_t dmask) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t h = gid * 2654435761u + 977u;
    float a = 0.f, b = 0.f;
    for (uint32_t it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t r = (h >> 8) & (kRows - 1u);
        uint32_t d = ((h >> 3) & dmask) | 1u;                   // odd xor distance within the 8-row group (1, 3, 5, 7) -> same 64 B line
        if (V == 0) {
            const float2 p = tab[r], q = tab[r ^ d];
            a += p.x + q.x; b += p.y + q.y;
        } else if (V == 1 || V == 2) {
            // two rounds of 32 points: round 0 serves the points of lanes 0..31 (V 2) / even pairs, round 1 the others
#pragma unroll
            for (uint32_t round = 0; round < 2; round++) {
                uint32_t src, second;
                if (V == 1) { src = (lane >> 1) + 32u * round; second = lane & 1u; }
                else { src = (lane & 31u) + 32u * round; second = lane >> 5; }
                const uint32_t rr = __shfl(r, src, 64), dd = __shfl(d, src, 64);
                const float2 p = tab[second ? rr ^ dd : rr];
                a += p.x; b += p.y;
            }
        } else if (V == 3) {
            const float4 p = reinterpret_cast<const float4 *>(tab)[r >> 1];
            a += p.x + p.z; b += p.y + p.w;
        } else {
            // variant 1's lane pairs with a cache-policy modifier on the load: 4 = nt, 5 = sc0, 6 = sc1, 7 = sc0 sc1 nt
#pragma unroll
            for (uint32_t round = 0; round < 2; round++) {
                const uint32_t src = (lane >> 1) + 32u * round, second = lane & 1u;
                const uint32_t rr = __shfl(r, src, 64), dd = __shfl(d, src, 64);
                const float2 *ptr = tab + (second ? rr ^ dd : rr);
                float2 p;
                if (V == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(ptr) : "memory");
                if (V == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(ptr) : "memory");
                if (V == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(ptr) : "memory");
                if (V == 7) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(ptr) : "memory");
                if (V == 8) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(p) : "v"(ptr) : "memory");
                a += p.x; b += p.y;
            }
        }
    }
    out[gid] = a + b;
}

int main() {
    float2 *tab; float *out;
    hipMalloc(&tab, kRows * sizeof(float2));
    hipMemset(tab, 0, kRows * sizeof(float2));
    const uint32_t blocks = 256 * 8, iters = 2048;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t dmask = 1; dmask <= 7; dmask += 6) {
        for (int v = 0; v < 9; v++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                if (v == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, tab, out, iters, dmask);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double rows = 2.0 * blocks * 256.0 * iters;
            printf("xor distance mask %u  variant %d: %.3f ms  %.1f G rows/s\n", dmask, v, best, rows / best / 1e6);
        }
    }
    return 0;
}
