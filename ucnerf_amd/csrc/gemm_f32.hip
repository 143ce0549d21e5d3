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
#include "gemm_epilogue.h"

namespace {

__device__ __forceinline__ f32x16 mfma32x2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// the same instruction with the accumulator PINNED to the accumulation registers ("a" class): the 64-row shape of k_gemm_f32 holds 256
// accumulator registers, exactly the AGPR half of the file -- left to the allocator they were split over both halves and ~1500
// v_accvgpr_read / _write / _mov per chunk shuffled them around the MFMAs (r05 first build)
__device__ __forceinline__ void mfma32x2_acc(float a, float b, f32x16 &c) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ float f4e(const float4 &v, uint32_t s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// ---- k_gemm_f32: the weight streams through LDS in chunks of 4 CQ k (any N, K) ------------------------------------------------------
// CQ = k quads per chunk (8: 32 k), OCC = workgroups per CU the register budget is cut for, RT = 32-row tiles per wave.
//   <NT, 8, 2, 1>: 4 waves x 32 rows, two workgroups per CU (<= 256 registers): one workgroup's first-touch reads, stores and chunk
//                  barriers sit behind the other one's MFMAs.
//   <8, 8, 1, 2>:  4 waves x 64 rows x 256 columns, 256 accumulator registers, one workgroup per CU (the shape of the vendor library's
//                  256 x 256 macro tile): every weight operand read from LDS and every staged weight quad feeds TWO MFMAs, a barrier
//                  per 16 K MFMA cycles instead of 8 K.
template <uint32_t NT, uint32_t CQ, uint32_t OCC, uint32_t RT>
__global__ __launch_bounds__(256, OCC) void k_gemm_f32(const float *__restrict__ X, uint32_t ldx, const float *__restrict__ W, uint32_t ldw,
                                                       uint32_t K, GemmOut o) {
    constexpr uint32_t NC = NT * 32u, QS = NC + 1u;                // columns per pass; float4 stride between the k quads (+1: the
    //                                                                staging writes of one row's quads fall on different bank groups)
    constexpr uint32_t CK = 4u * CQ, NP = CQ / 2u;                 // k per chunk; quad PAIRS per chunk (one per lane half kk)
    constexpr uint32_t WPT = NC * CQ / 256u;                       // float4 of a weight chunk per thread
    constexpr uint32_t WPP = WPT >= 4u ? 4u : WPT, PARTS = WPT / WPP;    // staged through registers in pieces of <= 4 float4
    static_assert(PARTS * 2u <= NP || PARTS == 1u, "a piece is requested before one quad pair and written behind the next");
    static_assert(RT == 1u || RT == 2u, "one or two 32-row tiles per wave");
    extern __shared__ float4 s_w[];                                // [2 buffers][CQ quads][QS]
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, kk = lane >> 5;
    const uint32_t m0 = blockIdx.x * (128u * RT) + wave * (32u * RT), n0 = blockIdx.y * NC;
    const float *xrow[RT];
#pragma unroll
    for (uint32_t r = 0; r < RT; r++) {
        const uint32_t row = m0 + 32u * r + i < o.M ? m0 + 32u * r + i : o.M - 1u;      // rows past the end: clamped loads, no stores
        xrow[r] = X + (size_t)row * ldx;
    }
    f32x16 acc[RT][NT];
#pragma unroll
    for (uint32_t r = 0; r < RT; r++) gemm_init_acc<NT>(acc[r], o, m0 + 32u * r + i, n0, kk);
    const uint32_t nchunks = (K + CK - 1u) / CK;
    float4 wreg[WPP], a_cur[RT][NP], a_nxt[RT][NP];
    // Every load of the loop is UNCONDITIONAL (clamped address, value zeroed by a select where it is consumed): a predicated load
    // compiles to a branch around it, and behind a branch the compiler's wait-count pass no longer knows how many loads are in flight --
    // it put `s_waitcnt vmcnt(0)` in front of the first MFMA of every chunk, i.e. the just-requested weights' full L2 latency.
    auto load_w = [&](uint32_t c, uint32_t part) {
#pragma unroll
        for (uint32_t u = 0; u < WPP; u++) {
            const uint32_t idx = threadIdx.x + (part * WPP + u) * 256u, n = idx / CQ, q = idx % CQ, k = c * CK + 4u * q;
            const uint32_t nc = n0 + n < o.N ? n0 + n : o.N - 1u, kc = k < K ? k : 0u;
            wreg[u] = *reinterpret_cast<const float4 *>(W + (size_t)nc * ldw + kc);
        }
    };
    auto store_w = [&](uint32_t c, uint32_t part) {
#pragma unroll
        for (uint32_t u = 0; u < WPP; u++) {
            const uint32_t idx = threadIdx.x + (part * WPP + u) * 256u, n = idx / CQ, q = idx % CQ, k = c * CK + 4u * q;
            const bool live = n0 + n < o.N && k < K;
            float4 v = wreg[u];
            v.x = live ? v.x : 0.f; v.y = live ? v.y : 0.f; v.z = live ? v.z : 0.f; v.w = live ? v.w : 0.f;
            s_w[((c & 1u) * CQ + q) * QS + n] = v;
        }
    };
    auto load_a = [&](uint32_t c, float4 (&a)[RT][NP]) {
#pragma unroll
        for (uint32_t r = 0; r < RT; r++)
#pragma unroll
            for (uint32_t p = 0; p < NP; p++) {
                const uint32_t k = c * CK + 4u * (2u * p + kk);
                a[r][p] = *reinterpret_cast<const float4 *>(xrow[r] + (k < K ? k : 0u));      // k >= K: zeroed below
            }
    };
    auto zero_tail = [&](uint32_t c, float4 (&a)[RT][NP]) {                         // (0 x inf = NaN: both operands of a padding quad are zeroed)
        if (c * CK + CK <= K) return;                                               // uniform: whole chunks skip the selects
#pragma unroll
        for (uint32_t r = 0; r < RT; r++)
#pragma unroll
            for (uint32_t p = 0; p < NP; p++) {
                const bool live = c * CK + 4u * (2u * p + kk) < K;
                float4 &v = a[r][p];
                v.x = live ? v.x : 0.f; v.y = live ? v.y : 0.f; v.z = live ? v.z : 0.f; v.w = live ? v.w : 0.f;
            }
    };
#pragma unroll
    for (uint32_t part = 0; part < PARTS; part++) { load_w(0, part); store_w(0, part); }
    load_a(0, a_cur);
    zero_tail(0, a_cur);
    __syncthreads();
    for (uint32_t c = 0; c < nchunks; c++) {
        // (the loads of the last chunk's iteration re-read chunk 0: harmless, unconditional -- see above)
        const uint32_t cn = c + 1u < nchunks ? c + 1u : 0u;
        load_a(cn, a_nxt);
        const float4 *buf = s_w + (c & 1u) * CQ * QS;
        float4 b[NT];
#pragma unroll
        for (uint32_t p = 0; p < NP; p++) {
            // the next chunk's weights: piece j requested before quad pair 2 j and written behind pair 2 j + 1 -- into the OTHER
            // buffer, whose last readers passed the barrier of chunk c - 1
            if ((p & 1u) == 0u && p / 2u < PARTS) load_w(cn, p / 2u);
            if (p == 0u) {
#pragma unroll
                for (uint32_t t = 0; t < NT; t++) b[t] = buf[kk * QS + 32u * t + i];
            }
            // the weight operands of the NEXT quad pair are requested tile by tile, each right behind the last MFMA that reads the
            // register quad it lands in (s = 3): 7 RT MFMAs (>= 450 cycles) of cover per ds_read_b128 and no second set of registers
#pragma unroll
            for (uint32_t s = 0; s < 4; s++)
#pragma unroll
                for (uint32_t t = 0; t < NT; t++) {
#pragma unroll
                    for (uint32_t r = 0; r < RT; r++) {
                        if constexpr (RT == 2u) mfma32x2_acc(f4e(b[t], s), f4e(a_cur[r][p], s), acc[r][t]);
                        else acc[r][t] = mfma32x2(f4e(b[t], s), f4e(a_cur[r][p], s), acc[r][t]);
                    }
                    if (s == 3u && p + 1u < NP) {
                        b[t] = buf[(2u * (p + 1u) + kk) * QS + 32u * t + i];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            if (((p & 1u) == 1u || NP == 1u) && p / 2u < PARTS && c + 1u < nchunks) store_w(c + 1u, p / 2u);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < RT; r++)
#pragma unroll
            for (uint32_t p = 0; p < NP; p++) a_cur[r][p] = a_nxt[r][p];
        zero_tail(c + 1u, a_cur);
    }
    // Full tiles with 16-byte-aligned rows go through LDS (the weight buffers are free after the last barrier; a wave's staging tile is
    // 8.5 KiB, the launch reserves max(weight buffers, 4 such tiles)).
    // (the epilogue's bounds are made opaque HERE: its ~70 loop-invariant column / row predicates were hoisted above the chunk loop and
    // held in scalar registers across it -- 210 of them spilled into vector lanes, which the 64-row shape does not have to spare)
    GemmOut oe = o;
    asm volatile("" : "+s"(oe.N), "+s"(oe.M), "+s"(oe.flags));
    if constexpr (RT == 2u) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // inline-asm MFMAs: their result latency (16 passes) is not known to the hazard recogniser
    const bool staged = NT >= 2u && (oe.flags & kGemmVec) && n0 + NC <= oe.N;
    float *tile = reinterpret_cast<float *>(s_w) + wave * kStageFloats;
    if constexpr (NT >= 2u) {
        if (staged) {
            gemm_store_staged<NT>(acc[0], oe, tile, m0, n0, lane);
            if constexpr (RT == 2u) gemm_store_staged<NT>(acc[1], oe, tile, m0 + 32u, n0, lane);
            return;
        }
    }
    gemm_store_direct<NT>(acc[0], oe, m0 + i, n0, kk);
    if constexpr (RT == 2u) gemm_store_direct<NT>(acc[1], oe, m0 + 32u + i, n0, kk);
}

// ---- k_gemm_f32_res: the WHOLE weight resident in LDS, persistent waves (r05) -------------------------------------------------------
// For the layers whose [N, K] weight fits beside the output staging (N K <= ~16 K floats: 256 x 64 -- the composed colour layers --,
// 64 x 256 -- their input gradients --, the narrow heads): one workgroup of 8 waves per CU loads the weight ONCE (the chunked kernel
// re-streams it per 128 rows: as many L2 -> LDS bytes as the activations themselves), then every wave walks its own 32-row tiles with no
// barrier at all: X rows requested PD chunks ahead (across tile boundaries), products from the resident weight, output through the wave's
// own 8-row staging tile.  The waves drift apart, so one wave's stores and first-touch reads sit behind the other wave's MFMAs on each SIMD.
template <uint32_t NT, uint32_t PD>
__global__ __launch_bounds__(512) void k_gemm_f32_res(const float *__restrict__ X, uint32_t ldx, const float *__restrict__ W, uint32_t ldw,
                                                      uint32_t K, GemmOut o) {
    constexpr uint32_t NC = NT * 32u, QS = NC + 1u;
    extern __shared__ float4 s_w[];                                // [KQ quads][QS], then 8 staging tiles of kStageFloats
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, kk = lane >> 5;
    const uint32_t nchunks = (K + 31u) / 32u, KQ = nchunks * 8u;
    for (uint32_t idx = threadIdx.x; idx < NC * KQ; idx += 512u) {
        const uint32_t n = idx / KQ, q = idx - n * KQ;             // consecutive threads: consecutive 16-byte pieces of a weight row
        s_w[q * QS + n] = (n < o.N && 4u * q < K) ? *reinterpret_cast<const float4 *>(W + (size_t)n * ldw + 4u * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    float *tile = reinterpret_cast<float *>(s_w + KQ * QS) + wave * kStageFloats;
    const uint32_t ntiles = (o.M + 31u) / 32u, gw = blockIdx.x * 8u + wave, GW = gridDim.x * 8u;
    float4 a[PD + 1u][4];
    uint32_t pt = gw, pc = 0;                                      // prefetch cursor: (tile, chunk)
    // unconditional loads (clamped row / column; see k_gemm_f32): rows past the end re-read row M - 1 and are never stored, the k quads
    // past K are zero in the resident weight and zeroed here before use (tail chunk only)
    auto fetch = [&](float4 (&dst)[4]) {
        const uint64_t r = (uint64_t)pt * 32u + i;
        const uint32_t row = r < o.M ? (uint32_t)r : o.M - 1u;
        const float *xr = X + (size_t)row * ldx;
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const uint32_t k = pc * 32u + 4u * (2u * p + kk);
            dst[p] = *reinterpret_cast<const float4 *>(xr + (k < K ? k : 0u));
        }
        if (++pc == nchunks) { pc = 0; pt += GW; }
    };
    auto zero_tail = [&](uint32_t c, float4 (&v)[4]) {
        if (c * 32u + 32u <= K) return;
#pragma unroll
        for (uint32_t p = 0; p < 4; p++) {
            const bool live = c * 32u + 4u * (2u * p + kk) < K;
            v[p].x = live ? v[p].x : 0.f; v[p].y = live ? v[p].y : 0.f; v[p].z = live ? v[p].z : 0.f; v[p].w = live ? v[p].w : 0.f;
        }
    };
#pragma unroll
    for (uint32_t d = 0; d < PD; d++) fetch(a[d]);
    for (uint32_t t0 = gw; t0 < ntiles; t0 += GW) {
        const uint32_t m0 = t0 * 32u;
        f32x16 acc[NT];
        gemm_init_acc<NT>(acc, o, m0 + i, 0u, kk);
        for (uint32_t c = 0; c < nchunks; c++) {
            fetch(a[PD]);
            zero_tail(c, a[0]);
            const float4 *buf = s_w + c * 8u * QS;
            float4 b[NT];
#pragma unroll
            for (uint32_t p = 0; p < 4; p++) {
                if (p == 0u) {
#pragma unroll
                    for (uint32_t t = 0; t < NT; t++) b[t] = buf[kk * QS + 32u * t + i];
                }
#pragma unroll
                for (uint32_t s = 0; s < 4; s++)
#pragma unroll
                    for (uint32_t t = 0; t < NT; t++) {
                        acc[t] = mfma32x2(f4e(b[t], s), f4e(a[0][p], s), acc[t]);
                        if (s == 3u && p + 1u < 4u) {              // the next quad pair's operands, behind their registers' last reader
                            b[t] = buf[(2u * (p + 1u) + kk) * QS + 32u * t + i];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
#pragma unroll
            for (uint32_t d = 0; d < PD; d++)
#pragma unroll
                for (uint32_t p = 0; p < 4; p++) a[d][p] = a[d + 1u][p];
        }
        if constexpr (NT >= 2u) {
            if ((o.flags & kGemmVec) && NC <= o.N) { gemm_store_staged<NT>(acc, o, tile, m0, 0u, lane); continue; }
        }
        gemm_store_direct<NT>(acc, o, m0 + i, 0u, kk);
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
    // unconditional loads (clamped addresses, zeroed by selects at the LDS write): predicated loads become branches, and behind
    // branches the wait-count pass gives up counting and drains every load in flight (see k_gemm_f32)
    auto load_slab = [&](uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t idx = threadIdx.x + u * 512u, r = idx >> 7, c4 = idx & 127u;
            const uint32_t m = ms + r < m_hi ? ms + r : m_hi - 1u;
            const bool gy = c4 < 64u;
            const uint32_t n = n0 + 4u * c4, k = k0 + 4u * (c4 - 64u);
            const uint32_t col = gy ? (n < N ? n : 0u) : (k < K ? k : 0u);
            const float *src = gy ? GY + (size_t)m * ldg : X + (size_t)m * ldx;
            reg[u] = *reinterpret_cast<const float4 *>(src + col);
        }
    };
    auto store_slab = [&](uint32_t buf, uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t idx = threadIdx.x + u * 512u, r = idx >> 7, c4 = idx & 127u;
            const bool live = ms + r < m_hi && (c4 < 64u ? (n0 + 4u * c4 < N && c4 < NB * 8u) : (k0 + 4u * (c4 - 64u) < K));
            float4 v = reg[u];
            v.x = live ? v.x : 0.f; v.y = live ? v.y : 0.f; v.z = live ? v.z : 0.f; v.w = live ? v.w : 0.f;
            reinterpret_cast<float4 *>(s_slab)[buf * (kWgSlab * kWgCols / 4u) + idx] = v;
        }
    };
    const uint32_t nslabs = (m_hi - m_lo + kWgSlab - 1u) / kWgSlab;
    if (nslabs) { load_slab(m_lo); store_slab(0, m_lo); }
    __syncthreads();
    for (uint32_t sidx = 0; sidx < nslabs; sidx++) {
        const bool more = sidx + 1u < nslabs;
        load_slab(more ? m_lo + (sidx + 1u) * kWgSlab : m_lo);        // (last slab: an unconditional re-read of the first, unused)
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
        if (more) store_slab((sidx + 1u) & 1u, m_lo + (sidx + 1u) * kWgSlab);
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

// CU count and LDS per workgroup of the CURRENT device (cached per device id: a process may drive several GPUs)
static void gemm_device_limits(int &cus, size_t &lds) {
    static int s_cus[16] = {0};
    static size_t s_lds[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { cus = 256; lds = 64u * 1024u; return; }
    if (s_cus[dev] == 0) {
        int v = 0, l = 0;
        s_cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        s_lds[dev] = (hipDeviceGetAttribute(&l, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && l > 0) ? (size_t)l : 64u * 1024u;
    }
    cus = s_cus[dev];
    lds = s_lds[dev];
}

extern "C" int ucn_gemm_f32_ex(const float *X, uint32_t ldx, const float *W, uint32_t ldw, const float *bias, uint32_t M, uint32_t N,
                               uint32_t K, int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm, const float *rowbias,
                               uint32_t ldr, uint32_t rgroup, ucn_stream_t stream) {
    UCN_REQUIRE(X && W && Y, "gemm_f32: null pointer argument");
    UCN_REQUIRE(K % 4u == 0u && ldx % 4u == 0u && ldw % 4u == 0u && ldx >= K && ldw >= K && ldy >= N,
                "gemm_f32: K, ldx, ldw must be multiples of 4 (16-byte operand loads) and cover the operands (K %u ldx %u ldw %u N %u ldy %u)",
                K, ldx, ldw, N, ldy);
    UCN_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15u) == 0u, "gemm_f32: X and W must be 16-byte aligned");
    UCN_REQUIRE((flags & ~7) == 0, "gemm_f32: flags = UCN_GEMM_ACCUMULATE | UCN_GEMM_RELU | UCN_GEMM_MASK");
    UCN_REQUIRE(!(flags & (int)kGemmMask) || (mask && ldm >= N), "gemm_f32: UCN_GEMM_MASK needs a mask [M, N] (ldm %u N %u)", ldm, N);
    UCN_REQUIRE(!rowbias || (rgroup > 0 && ldr >= N), "gemm_f32: a row-group bias needs rgroup > 0 and ldr >= N (rgroup %u ldr %u N %u)", rgroup, ldr, N);
    if (M == 0 || N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nt = N <= 32u ? 1u : N <= 64u ? 2u : N <= 128u ? 4u : 8u;
    const bool vec = N % 4u == 0u && ldy % 4u == 0u && ((uintptr_t)Y & 15u) == 0u && (!bias || ((uintptr_t)bias & 15u) == 0u) &&
                     (!(flags & (int)kGemmMask) || (ldm % 4u == 0u && ((uintptr_t)mask & 15u) == 0u)) &&
                     (!rowbias || (ldr % 4u == 0u && ((uintptr_t)rowbias & 15u) == 0u));
    GemmOut o{bias, (flags & (int)kGemmMask) ? mask : nullptr, rowbias, Y, ldy, ldm, ldr, rgroup ? rgroup : 1u, M, N, (uint32_t)flags | (vec ? kGemmVec : 0u)};
    // whole weight resident (persistent waves) when it fits beside the staging tiles and there are tiles for every wave
    const uint32_t kq = ucn_div_up(K, 32) * 8u, nc = nt * 32u;
    const size_t res_lds = (size_t)kq * (nc + 1u) * 16u + (nt >= 2u ? 8u * kStageFloats * 4u : 0u);
    const uint32_t ntiles = ucn_div_up(M, 32);
    static const bool no_res = getenv("UCN_GEMM_NO_RESIDENT") != nullptr;            // A/B switch (tools/gemm_f32_bench.py)
    int num_cus = 256;
    size_t max_lds = 0;
    gemm_device_limits(num_cus, max_lds);
    // (a device whose workgroups cannot hold the resident weight takes the weight-streaming kernel below)
    if (N <= nc && res_lds <= 150u * 1024u && res_lds <= max_lds && ntiles >= 64u && !no_res) {
        const uint32_t wgs = ucn_div_up(ntiles, 8) < (uint32_t)num_cus ? ucn_div_up(ntiles, 8) : (uint32_t)num_cus;
#define UCN_GR(NT) hipLaunchKernelGGL((k_gemm_f32_res<NT, 2u>), dim3(wgs), dim3(512), res_lds, st, X, ldx, W, ldw, K, o)
        switch (nt) {
            case 1: UCN_GR(1); break;
            case 2: UCN_GR(2); break;
            case 4: UCN_GR(4); break;
            default: UCN_GR(8); break;
        }
#undef UCN_GR
        UCN_LAUNCH_CHECK("gemm_f32 (resident)");
        return 0;
    }
    static const int variant = getenv("UCN_GEMM_VARIANT") ? atoi(getenv("UCN_GEMM_VARIANT")) : 0;       // A/B switch (tools/gemm_f32_bench.py)
    // UCN_GEMM_VARIANT=2: 64-row waves, one workgroup per CU (the library's macro-tile shape).  MEASURED SLOWER (80.8 / 99.6 TF at
    // K = 256 / 544 against 93.9 / 120.5 for two 32-row workgroups per CU, profiles/r05/gemm_f32_variants.txt): with one wave per SIMD
    // every LDS / barrier / first-touch bubble is exposed, which the vendor kernel avoids by hand-scheduled assembly.  Kept as the A/B.
    const uint32_t rt = (nt == 8u && variant == 2) ? 2u : 1u;
    const dim3 grid(ucn_div_up(M, 128u * rt), ucn_div_up(N, nt * 32u));
#define UCN_LDS(NT, CQ) (2u * CQ * (NT * 32u + 1u) * 16u > 4u * kStageFloats * 4u ? 2u * CQ * (NT * 32u + 1u) * 16u : 4u * kStageFloats * 4u)
#define UCN_G(NT, CQ, OCC, RT) hipLaunchKernelGGL((k_gemm_f32<NT, CQ, OCC, RT>), grid, dim3(256), UCN_LDS(NT, CQ), st, X, ldx, W, ldw, K, o)
    switch (nt) {
        case 1: UCN_G(1, 8, 2, 1); break;
        case 2: UCN_G(2, 8, 2, 1); break;
        case 4: UCN_G(4, 8, 2, 1); break;
        default:
            if (rt == 2u) UCN_G(8, 8, 1, 2);
            else UCN_G(8, 8, 2, 1);
            break;
    }
#undef UCN_G
#undef UCN_LDS
    UCN_LAUNCH_CHECK("gemm_f32");
    return 0;
}

extern "C" int ucn_gemm_f32(const float *X, uint32_t ldx, const float *W, uint32_t ldw, const float *bias, uint32_t M, uint32_t N,
                            uint32_t K, int flags, float *Y, uint32_t ldy, ucn_stream_t stream) {
    UCN_REQUIRE((flags & ~3) == 0, "gemm_f32: flags = UCN_GEMM_ACCUMULATE | UCN_GEMM_RELU");
    return ucn_gemm_f32_ex(X, ldx, W, ldw, bias, M, N, K, flags, Y, ldy, nullptr, 0, nullptr, 0, 0, stream);
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
