// Dense layers of the fp32 TRAINING route on hand-written kernels (r04): tall-skinny fp32 GEMMs on v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulation -- gfx950 has no xf32; 157 TFLOP/s dense peak).
//
// The reference's shipped launch trains WITHOUT mixed precision (scripts/train_waymo.sh:3, train.py:165), i.e. every
// nn.Linear of the NeRF field (models.py:438-483, 581-674), the sky NeRF (models.py:743-820) and the colour-correction head
// (extrinsic_optimizer.py:4-48) is an fp32 GEMM over M = rays x samples ~ 1e6 rows against a weight of <= 256 x 544.  Until r03
// that route ran on hipBLASLt.  Three shapes cover a layer's forward and backward:
//   ucn_gemm_f32   Y[M, N] (+)= X[M, K] W[N, K]^T (+ bias) (ReLU)     forward; and d X = d Y W as the same kernel on W^T
//   ucn_wgrad_f32  GW[N, K] = GY[M, N]^T X[M, K], gb[N] = column sums of GY          reduction over the M samples
// Operands are plain row-major fp32 with a leading dimension, so column slices of wider buffers (the reference's concatenated
// layer inputs, models.py:620-640) are passed as views and never copied.
//
// k_gemm_f32: a workgroup = 4 waves x 32 rows of X against NT x 32 output columns.  The weight is streamed through LDS in
// chunks of 32 k (double buffered, quad-major [k / 4][column] float4 so that the B operands of four consecutive MFMAs are
// one conflict-free ds_read_b128; coalesced 128-byte global reads, registers -> LDS behind the MFMAs of the previous chunk);
// a lane's A operands for a chunk are four float4 of its own row, requested one chunk ahead.  8 tiles x 4 MFMAs per
// ds_read / global quad: the kernel is MFMA-bound for N >= 128.
// k_wgrad_f32: 32-row slabs of GY and X are staged in LDS as they lie in memory (coalesced), both MFMA operands are
// ds_read_b32 of a slab row (lanes = consecutive columns); a workgroup (8 waves) owns a 256 x 256 block of GW for one chunk
// of the samples, partial blocks are summed in a fixed order by k_reduce_f32 (deterministic).
#include "ucn_common.h"
#include "wave_dpp.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4e(const float4 &v, uint32_t s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

constexpr uint32_t kGemmAccum = 1u, kGemmRelu = 2u;

template <uint32_t NT>
__global__ __launch_bounds__(256) void k_gemm_f32(const float *__restrict__ X, uint32_t ldx, const float *__restrict__ W, uint32_t ldw,
                                                  const float *__restrict__ bias, uint32_t M, uint32_t N, uint32_t K, uint32_t flags,
                                                  float *__restrict__ Y, uint32_t ldy) {
    constexpr uint32_t NC = NT * 32u, QS = NC + 1u;                // columns per pass; float4 stride between the k quads (+1: the
    //                                                                staging writes of one row's 8 quads fall on 8 bank groups)
    constexpr uint32_t WPT = NC * 8u / 256u;                       // float4 of a weight chunk per thread (= NT)
    extern __shared__ float4 s_w[];                                // [2 buffers][8 quads][QS]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, kk = lane >> 5;
    const uint32_t m0 = blockIdx.x * 128u + wave * 32u, n0 = blockIdx.y * NC;
    const uint32_t row = m0 + i < M ? m0 + i : M - 1u;             // rows past the end: clamped loads, no stores
    const float *xrow = X + (size_t)row * ldx;
    f32x16 acc[NT];
#pragma unroll
    for (uint32_t t = 0; t < NT; t++) {
        const uint32_t col = n0 + 32u * t + i;
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) {
            const uint32_t ro = m0 + (r & 3u) + 8u * (r >> 2) + 4u * kk;
            acc[t][r] = ((flags & kGemmAccum) && col < N && ro < M) ? Y[(size_t)ro * ldy + col] : 0.0f;
        }
    }
    const uint32_t nchunks = (K + 31u) / 32u;
    float4 wreg[WPT], a_cur[4], a_nxt[4];
    auto load_w = [&](uint32_t c) {
#pragma unroll
        for (uint32_t u = 0; u < WPT; u++) {
            const uint32_t idx = threadIdx.x + u * 256u, n = idx >> 3, q = idx & 7u, k = c * 32u + 4u * q;
            wreg[u] = (n0 + n < N && k < K) ? *reinterpret_cast<const float4 *>(W + (size_t)(n0 + n) * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_w = [&](uint32_t buf) {
#pragma unroll
        for (uint32_t u = 0; u < WPT; u++) {
            const uint32_t idx = threadIdx.x + u * 256u, n = idx >> 3, q = idx & 7u;
            s_w[(buf * 8u + q) * QS + n] = wreg[u];
        }
    };
    auto load_a = [&](uint32_t c, float4 (&a)[4]) {
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const uint32_t k = c * 32u + 4u * (2u * p + kk);
            a[p] = k < K ? *reinterpret_cast<const float4 *>(xrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_w(0);
    load_a(0, a_cur);
    store_w(0);
    __syncthreads();
    for (uint32_t c = 0; c < nchunks; c++) {
        const bool more = c + 1u < nchunks;
        if (more) { load_w(c + 1u); load_a(c + 1u, a_nxt); }
        const float4 *buf = s_w + (c & 1u) * 8u * QS;
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            float4 b[NT];
#pragma unroll
            for (uint32_t t = 0; t < NT; t++) b[t] = buf[(2u * p + kk) * QS + 32u * t + i];
#pragma unroll
            for (uint32_t s = 0; s < 4; s++)
#pragma unroll
                for (uint32_t t = 0; t < NT; t++) acc[t] = mfma32x2(f4e(a_cur[p], s), f4e(b[t], s), acc[t]);
        }
        if (more) store_w((c + 1u) & 1u);               // the other buffer: its last readers passed the barrier of chunk c - 1
        __syncthreads();
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) a_cur[p] = a_nxt[p];
    }
#pragma unroll
    for (uint32_t t = 0; t < NT; t++) {
        const uint32_t col = n0 + 32u * t + i;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) {
            const uint32_t ro = m0 + (r & 3u) + 8u * (r >> 2) + 4u * kk;
            if (ro < M) {
                float v = acc[t][r] + bv;
                if (flags & kGemmRelu) v = fmaxf(v, 0.0f);
                Y[(size_t)ro * ldy + col] = v;
            }
        }
    }
}

// GW block [256 rows n][256 columns k] of one chunk of the samples.  Wave w: n tiles 2 (w & 3), + 1; k tiles 4 (w >> 2) ... + 3.
constexpr uint32_t kWgSlab = 32u, kWgCols = 512u;                 // slab row = [GY block (256) | X block (256)] floats
__global__ __launch_bounds__(512) void k_wgrad_f32(const float *__restrict__ GY, uint32_t ldg, const float *__restrict__ X, uint32_t ldx,
                                                   uint32_t M, uint32_t N, uint32_t K, uint32_t MC, float *__restrict__ ws,
                                                   float *__restrict__ wsb) {
    extern __shared__ float s_slab[];                              // [2][32][512]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, kk = lane >> 5;
    const uint32_t chunk = blockIdx.x, k0 = blockIdx.y * 256u, n0 = blockIdx.z * 256u;
    const uint32_t nt0 = 2u * (wave & 3u), kt0 = 4u * (wave >> 2);
    const uint32_t m_lo = chunk * MC, m_hi = m_lo + MC < M ? m_lo + MC : M;
    f32x16 acc[2][4];
#pragma unroll
    for (uint32_t a = 0; a < 2; a++)
#pragma unroll
        for (uint32_t t = 0; t < 4; t++)
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) acc[a][t][r] = 0.0f;
    float bsum[2] = {0.0f, 0.0f};
    // staging: 32 rows x 128 float4 (64 of GY, 64 of X) = 4096 float4, 8 per thread; a thread's float4 index keeps its column
    float4 reg[8];
    auto load_slab = [&](uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t idx = threadIdx.x + u * 512u, r = idx >> 7, c4 = idx & 127u, m = ms + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_hi) {
                if (c4 < 64u) { const uint32_t n = n0 + 4u * c4; if (n < N) v = *reinterpret_cast<const float4 *>(GY + (size_t)m * ldg + n); }
                else { const uint32_t k = k0 + 4u * (c4 - 64u); if (k < K) v = *reinterpret_cast<const float4 *>(X + (size_t)m * ldx + k); }
            }
            reg[u] = v;
        }
    };
    auto store_slab = [&](uint32_t buf) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t idx = threadIdx.x + u * 512u;
            reinterpret_cast<float4 *>(s_slab)[buf * (kWgSlab * kWgCols / 4u) + idx] = reg[u];
        }
    };
    const uint32_t nslabs = (m_hi - m_lo + kWgSlab - 1u) / kWgSlab;
    if (nslabs) { load_slab(m_lo); store_slab(0); }
    __syncthreads();
    for (uint32_t sidx = 0; sidx < nslabs; sidx++) {
        const bool more = sidx + 1u < nslabs;
        if (more) load_slab(m_lo + (sidx + 1u) * kWgSlab);
        const float *sl = s_slab + (sidx & 1u) * (kWgSlab * kWgCols);
#pragma unroll 4
        for (uint32_t q = 0; q < 16; q++) {
            const float *rowp = sl + (2u * q + kk) * kWgCols;
            const float a0 = rowp[32u * nt0 + i], a1 = rowp[32u * (nt0 + 1u) + i];
            float b[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) b[t] = rowp[256u + 32u * (kt0 + t) + i];
            bsum[0] += a0;
            bsum[1] += a1;
#pragma unroll
            for (uint32_t t = 0; t < 4; t++) {
                acc[0][t] = mfma32x2(a0, b[t], acc[0][t]);
                acc[1][t] = mfma32x2(a1, b[t], acc[1][t]);
            }
        }
        if (more) store_slab((sidx + 1u) & 1u);
        __syncthreads();
    }
    float *out = ws + (size_t)chunk * N * K;
#pragma unroll
    for (uint32_t a = 0; a < 2; a++)
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t k = k0 + 32u * (kt0 + t) + i;
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) {
                const uint32_t n = n0 + 32u * (nt0 + a) + (r & 3u) + 8u * (r >> 2) + 4u * kk;
                if (n < N && k < K) out[(size_t)n * K + k] = acc[a][t][r];
            }
        }
    if (wsb && blockIdx.y == 0 && kt0 == 0u) {
#pragma unroll
        for (uint32_t a = 0; a < 2; a++) {
            const float tot = xor32_sum(bsum[a]);
            const uint32_t n = n0 + 32u * (nt0 + a) + i;
            if (kk == 0u && n < N) wsb[(size_t)chunk * N + n] = tot;
        }
    }
}

// out[e] = sum over the chunks of ws[c][e], in chunk order (deterministic); a second region for the bias partials
__global__ __launch_bounds__(256) void k_reduce_f32(const float *__restrict__ ws, uint32_t chunks, size_t stride, size_t n, float *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    float s = 0.0f;
    for (uint32_t c = 0; c < chunks; c++) s += ws[(size_t)c * stride + e];
    out[e] = s;
}

uint32_t wgrad_chunk_rows(uint32_t M) {
    // ~2 rounds of workgroups over 256 CUs per (n block, k block); a chunk is a whole number of 32-row slabs
    uint32_t mc = (M + 511u) / 512u;
    mc = (mc + 31u) / 32u * 32u;
    return mc < 512u ? 512u : mc;
}

}  // namespace

extern "C" int ucn_gemm_f32(const float *X, uint32_t ldx, const float *W, uint32_t ldw, const float *bias, uint32_t M, uint32_t N,
                            uint32_t K, int flags, float *Y, uint32_t ldy, ucn_stream_t stream) {
    UCN_REQUIRE(X && W && Y, "gemm_f32: null pointer argument");
    UCN_REQUIRE(K % 4u == 0u && ldx % 4u == 0u && ldw % 4u == 0u && ldx >= K && ldw >= K && ldy >= N,
                "gemm_f32: K, ldx, ldw must be multiples of 4 (16-byte operand loads) and cover the operands (K %u ldx %u ldw %u N %u ldy %u)",
                K, ldx, ldw, N, ldy);
    UCN_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15u) == 0u, "gemm_f32: X and W must be 16-byte aligned");
    UCN_REQUIRE((flags & ~3) == 0, "gemm_f32: flags = UCN_GEMM_ACCUMULATE | UCN_GEMM_RELU");
    if (M == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nt = N <= 32u ? 1u : N <= 64u ? 2u : N <= 128u ? 4u : 8u;
    const dim3 grid(ucn_div_up(M, 128), ucn_div_up(N, nt * 32u));
#define UCN_G(NT)                                                                                                         \
    hipLaunchKernelGGL((k_gemm_f32<NT>), grid, dim3(256), 2u * 8u * (NT * 32u + 1u) * sizeof(float4), st, X, ldx, W, ldw, \
                       bias, M, N, K, (uint32_t)flags, Y, ldy)
    switch (nt) {
        case 1: UCN_G(1); break;
        case 2: UCN_G(2); break;
        case 4: UCN_G(4); break;
        default: UCN_G(8); break;
    }
#undef UCN_G
    UCN_LAUNCH_CHECK("gemm_f32");
    return 0;
}

extern "C" uint64_t ucn_wgrad_f32_ws_floats(uint32_t N, uint32_t K, uint64_t M) {
    if (M == 0 || M > 0xFFFFFFFFull) return 0;
    const uint32_t mc = wgrad_chunk_rows((uint32_t)M);
    const uint64_t chunks = (M + mc - 1u) / mc;
    return chunks * ((uint64_t)N * K + N);
}

extern "C" int ucn_wgrad_f32(const float *GY, uint32_t ldg, const float *X, uint32_t ldx, uint32_t M, uint32_t N, uint32_t K,
                             float *ws, float *GW, float *gb, ucn_stream_t stream) {
    UCN_REQUIRE(GY && X && ws && GW, "wgrad_f32: null pointer argument");
    UCN_REQUIRE(N % 4u == 0u && K % 4u == 0u && ldg % 4u == 0u && ldx % 4u == 0u && ldg >= N && ldx >= K,
                "wgrad_f32: N, K, ldg, ldx must be multiples of 4 and cover the operands (N %u K %u ldg %u ldx %u)", N, K, ldg, ldx);
    UCN_REQUIRE((((uintptr_t)GY | (uintptr_t)X) & 15u) == 0u, "wgrad_f32: GY and X must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (hipMemsetAsync(GW, 0, (size_t)N * K * sizeof(float), st) != hipSuccess) return ucn_fail("wgrad_f32: hipMemsetAsync failed");
        if (gb && hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), st) != hipSuccess) return ucn_fail("wgrad_f32: hipMemsetAsync failed");
        return 0;
    }
    const uint32_t mc = wgrad_chunk_rows(M), chunks = ucn_div_up(M, mc);
    float *wsb = ws + (size_t)chunks * N * K;
    hipLaunchKernelGGL(k_wgrad_f32, dim3(chunks, ucn_div_up(K, 256), ucn_div_up(N, 256)), dim3(512), 2u * kWgSlab * kWgCols * sizeof(float), st,
                       GY, ldg, X, ldx, M, N, K, mc, ws, gb ? wsb : nullptr);
    const size_t nk = (size_t)N * K;
    hipLaunchKernelGGL(k_reduce_f32, dim3(ucn_div_up(nk, 256)), dim3(256), 0, st, ws, chunks, nk, nk, GW);
    if (gb) hipLaunchKernelGGL(k_reduce_f32, dim3(ucn_div_up(N, 256)), dim3(256), 0, st, wsb, chunks, (size_t)N, (size_t)N, gb);
    UCN_LAUNCH_CHECK("wgrad_f32");
    return 0;
}
