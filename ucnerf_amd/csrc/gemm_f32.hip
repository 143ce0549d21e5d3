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

constexpr uint32_t kGemmAccum = 1u, kGemmRelu = 2u, kGemmVec = 4u;

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
    // The product is formed TRANSPOSED, D[feature][sample] = W_tile X_tile^T (A operand = weight rows from LDS, B operand = this lane's
    // sample row): accumulator register r of lane (sample j, half kk) is then feature 32 t + (r & 3) + 8 (r >> 2) + 4 kk -- four
    // consecutive output columns per register quad, i.e. ONE 16-byte store per quad instead of four 4-byte stores spread over four rows
    // (a 1 M x 256 output is 4 M wave-level dword stores otherwise: the K = 64 layer ran at 1.4 TB/s of stores).
    const bool vec = (flags & kGemmVec) != 0;                      // N, ldy multiples of 4 and Y 16-byte aligned (host-checked)
    const uint32_t orow = m0 + i;                                  // this lane's output row
    f32x16 acc[NT];
#pragma unroll
    for (uint32_t t = 0; t < NT; t++) {
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t col = n0 + 32u * t + 8u * g + 4u * kk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((flags & kGemmAccum) && orow < M) {
                if (vec) { if (col < N) v = *reinterpret_cast<const float4 *>(Y + (size_t)orow * ldy + col); }
                else {
                    if (col < N) v.x = Y[(size_t)orow * ldy + col];
                    if (col + 1u < N) v.y = Y[(size_t)orow * ldy + col + 1u];
                    if (col + 2u < N) v.z = Y[(size_t)orow * ldy + col + 2u];
                    if (col + 3u < N) v.w = Y[(size_t)orow * ldy + col + 3u];
                }
            }
            acc[t][4u * g] = v.x; acc[t][4u * g + 1u] = v.y; acc[t][4u * g + 2u] = v.z; acc[t][4u * g + 3u] = v.w;
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
                for (uint32_t t = 0; t < NT; t++) acc[t] = mfma32x2(f4e(b[t], s), f4e(a_cur[p], s), acc[t]);
        }
        if (more) store_w((c + 1u) & 1u);               // the other buffer: its last readers passed the barrier of chunk c - 1
        __syncthreads();
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) a_cur[p] = a_nxt[p];
    }
    // Epilogue.  Full tiles with 16-byte-aligned rows go through LDS (the weight buffers are free after the last barrier): a lane
    // holds 4 consecutive columns of ITS row, so a direct store instruction is 32 rows x two 16-byte pieces = 64 partial-line
    // requests; written to LDS as [row][column] and read back row-contiguous, a store instruction covers whole 128-byte lines
    // (64 lanes x 16 B = 1 KiB of one or two rows) -- the K = 64 layer went from 1.4 to ~3 TB/s of stores.  Two passes of 16
    // accumulator quads' worth per wave (16 rows x NC floats x 2 = the wave's quarter of the 64 KiB).
    if constexpr (NT >= 4u) {
        if (vec && n0 + NC <= N) {
            constexpr uint32_t RS = NC + 4u;                           // row stride in floats (+4: the 32 rows of a write fall on 8 bank groups)
            float *tile = reinterpret_cast<float *>(s_w) + wave * 16u * RS;       // this wave's 16 rows x NC columns (the launch reserves max(weight buffers, 4 such tiles))
#pragma unroll
            for (uint32_t half = 0; half < 2; half++) {
                // rows 16 half ... 16 half + 15 of the wave's 32: the lanes whose row is in this half write their quads
                if ((i >> 4) == half) {
#pragma unroll
                    for (uint32_t t = 0; t < NT; t++)
#pragma unroll
                        for (uint32_t g = 0; g < 4; g++) {
                            float4 v = make_float4(acc[t][4u * g], acc[t][4u * g + 1u], acc[t][4u * g + 2u], acc[t][4u * g + 3u]);
                            *reinterpret_cast<float4 *>(tile + (i & 15u) * RS + 32u * t + 8u * g + 4u * kk) = v;
                        }
                }
                wave_lds_handoff();
                // read back: NC / 4 float4 per row; the wave's 64 lanes take consecutive float4 of consecutive rows
#pragma unroll
                for (uint32_t u = 0; u < 16u * (NC / 4u) / 64u; u++) {
                    const uint32_t f = lane + 64u * u, r = f / (NC / 4u), c4 = f % (NC / 4u);
                    const uint32_t ro = m0 + 16u * half + r, col = n0 + 4u * c4;
                    float4 v = *reinterpret_cast<const float4 *>(tile + r * RS + 4u * c4);
                    if (bias) { const float4 bv = *reinterpret_cast<const float4 *>(bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                    if (flags & kGemmRelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (ro < M) *reinterpret_cast<float4 *>(Y + (size_t)ro * ldy + col) = v;
                }
                wave_lds_handoff();
            }
            return;
        }
    }
    if (orow < M) {
#pragma unroll
        for (uint32_t t = 0; t < NT; t++) {
#pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                const uint32_t col = n0 + 32u * t + 8u * g + 4u * kk;
                if (col >= N) continue;
                float4 v = make_float4(acc[t][4u * g], acc[t][4u * g + 1u], acc[t][4u * g + 2u], acc[t][4u * g + 3u]);
                if (bias) {
                    if (vec) { const float4 bv = *reinterpret_cast<const float4 *>(bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
                    else {
                        v.x += bias[col];
                        if (col + 1u < N) v.y += bias[col + 1u];
                        if (col + 2u < N) v.z += bias[col + 2u];
                        if (col + 3u < N) v.w += bias[col + 3u];
                    }
                }
                if (flags & kGemmRelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                float *yo = Y + (size_t)orow * ldy + col;
                if (vec) *reinterpret_cast<float4 *>(yo) = v;
                else {
                    yo[0] = v.x;
                    if (col + 1u < N) yo[1] = v.y;
                    if (col + 2u < N) yo[2] = v.z;
                    if (col + 3u < N) yo[3] = v.w;
                }
            }
        }
    }
}

// GW block [256 rows n][256 columns k] of one chunk of the samples.  Wave w: n tiles 2 (w & 3), + 1; k tiles 4 (w >> 2) ... + 3.
constexpr uint32_t kWgSlab = 32u, kWgCols = 512u;                 // slab row = [GY block (256) | X block (256)] floats
// NB = n tiles of the block: 8 (N > 64: wave w owns n tiles 2 (w & 3), + 1 x k tiles 4 (w >> 2) ... + 3), or 1 / 2 (narrow outputs -- the
// rgb, density and colour-head layers: wave w owns all NB n tiles x k tile w; a 3 x 256 gradient must not cost a 256 x 256 block of MFMAs)
template <uint32_t NB>
__global__ __launch_bounds__(512) void k_wgrad_f32(const float *__restrict__ GY, uint32_t ldg, const float *__restrict__ X, uint32_t ldx,
                                                   uint32_t M, uint32_t N, uint32_t K, uint32_t MC, float *__restrict__ ws,
                                                   float *__restrict__ wsb) {
    extern __shared__ float s_slab[];                              // [2][32][512]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, kk = lane >> 5;
    constexpr uint32_t NA = NB == 8u ? 2u : NB, NK = NB == 8u ? 4u : 1u;       // n tiles x k tiles per wave
    const uint32_t chunk = blockIdx.x, k0 = blockIdx.y * 256u, n0 = blockIdx.z * (NB * 32u);
    const uint32_t nt0 = NB == 8u ? 2u * (wave & 3u) : 0u, kt0 = NB == 8u ? 4u * (wave >> 2) : wave;
    const uint32_t m_lo = chunk * MC, m_hi = m_lo + MC < M ? m_lo + MC : M;
    f32x16 acc[NA][NK];
#pragma unroll
    for (uint32_t a = 0; a < NA; a++)
#pragma unroll
        for (uint32_t t = 0; t < NK; t++)
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) acc[a][t][r] = 0.0f;
    float bsum[NA];
#pragma unroll
    for (uint32_t a = 0; a < NA; a++) bsum[a] = 0.0f;
    // staging: 32 rows x 128 float4 (64 of GY, 64 of X) = 4096 float4, 8 per thread; a thread's float4 index keeps its column
    float4 reg[8];
    auto load_slab = [&](uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t idx = threadIdx.x + u * 512u, r = idx >> 7, c4 = idx & 127u, m = ms + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_hi) {
                if (c4 < 64u) { const uint32_t n = n0 + 4u * c4; if (n < N && c4 < NB * 8u) v = *reinterpret_cast<const float4 *>(GY + (size_t)m * ldg + n); }
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
            float av[NA], b[NK];
#pragma unroll
            for (uint32_t a = 0; a < NA; a++) av[a] = rowp[32u * (nt0 + a) + i];
#pragma unroll
            for (uint32_t t = 0; t < NK; t++) b[t] = rowp[256u + 32u * (kt0 + t) + i];
#pragma unroll
            for (uint32_t a = 0; a < NA; a++) bsum[a] += av[a];
#pragma unroll
            for (uint32_t t = 0; t < NK; t++)
#pragma unroll
                for (uint32_t a = 0; a < NA; a++) acc[a][t] = mfma32x2(av[a], b[t], acc[a][t]);
        }
        if (more) store_slab((sidx + 1u) & 1u);
        __syncthreads();
    }
    float *out = ws + (size_t)chunk * N * K;
#pragma unroll
    for (uint32_t a = 0; a < NA; a++)
#pragma unroll
        for (uint32_t t = 0; t < NK; t++) {
            const uint32_t k = k0 + 32u * (kt0 + t) + i;
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) {
                const uint32_t n = n0 + 32u * (nt0 + a) + (r & 3u) + 8u * (r >> 2) + 4u * kk;
                if (n < N && k < K) out[(size_t)n * K + k] = acc[a][t][r];
            }
        }
    if (wsb && blockIdx.y == 0 && kt0 == 0u) {
#pragma unroll
        for (uint32_t a = 0; a < NA; a++) {
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
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           // 8 loads in flight; the order of the additions is fixed by (chunks)
    uint32_t c = 0;
    for (; c + 8u <= chunks; c += 8u) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) p[u] += ws[(size_t)(c + u) * stride + e];
    }
    for (uint32_t u = 0; c < chunks; c++, u++) p[u] += ws[(size_t)c * stride + e];
    out[e] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

uint32_t wgrad_chunk_rows(uint32_t M) {
    // 256 chunks = one round of workgroups over 256 CUs per (n block, k block); a chunk is a whole number of 32-row slabs
    uint32_t mc = (M + 255u) / 256u;
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
    const bool vec = N % 4u == 0u && ldy % 4u == 0u && ((uintptr_t)Y & 15u) == 0u && (!bias || ((uintptr_t)bias & 15u) == 0u);
    const uint32_t kflags = (uint32_t)flags | (vec ? kGemmVec : 0u);
    const dim3 grid(ucn_div_up(M, 128), ucn_div_up(N, nt * 32u));
#define UCN_G(NT)                                                                                                         \
    hipLaunchKernelGGL((k_gemm_f32<NT>), grid, dim3(256),                                                                 \
                       (2u * 8u * (NT * 32u + 1u) * 16u > 4u * 16u * (NT * 32u + 4u) * 4u ? 2u * 8u * (NT * 32u + 1u) * 16u          \
                                                                                       : 4u * 16u * (NT * 32u + 4u) * 4u),         \
                       st, X, ldx, W, ldw,                                                                                \
                       bias, M, N, K, kflags, Y, ldy)
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
#define UCN_WG(NB)                                                                                                          \
    hipLaunchKernelGGL((k_wgrad_f32<NB>), dim3(chunks, ucn_div_up(K, 256), ucn_div_up(N, NB * 32u)), dim3(512),               \
                       2u * kWgSlab * kWgCols * sizeof(float), st, GY, ldg, X, ldx, M, N, K, mc, ws, gb ? wsb : nullptr)
    if (N <= 32u) UCN_WG(1);
    else if (N <= 64u) UCN_WG(2);
    else UCN_WG(8);
#undef UCN_WG
    const size_t nk = (size_t)N * K;
    hipLaunchKernelGGL(k_reduce_f32, dim3(ucn_div_up(nk, 256)), dim3(256), 0, st, ws, chunks, nk, nk, GW);
    if (gb) hipLaunchKernelGGL(k_reduce_f32, dim3(ucn_div_up(N, 256)), dim3(256), 0, st, wsb, chunks, (size_t)N, (size_t)N, gb);
    UCN_LAUNCH_CHECK("wgrad_f32");
    return 0;
}
