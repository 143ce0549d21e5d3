// Dense layers of the fp32 TRAINING route on the split-f16 MFMA engine (r06): "fp32-class" products at 16/3 of the fp32 MFMA rate.
//
// The reference's shipped launch trains WITHOUT mixed precision (scripts/train_waymo.sh:3; train.py:165's autocast is a no-op then): every
// nn.Linear of the NeRF field (models.py:438-483, 581-674), the sky NeRF (models.py:743-820) and the colour-correction head
// (extrinsic_optimizer.py:4-48) is an fp32 GEMM forward and two backward (grid.py:68-89 is the table side; train.py:166-221 the step).
// gemm_f32.hip runs them as exact fp32 products on v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense peak: 64 cycles per 32x32x2).  The
// rendering path has shown what the same network costs as split-f16 products (field_mlp_h.hip: 7.6x faster at 2.4e-7 pixel difference):
//
//     x = x_hi + x_lo (+ <= 2^-22 |x|),   x_hi = f16(x),  x_lo = f16(x - x_hi);      x w ~ x_hi w_hi + x_hi w_lo + x_lo w_hi
//
// three v_mfma_f32_32x32x16_f16 (32 cycles each) per 16 k against eight fp32 MFMAs (64 cycles each), fp32 accumulation in both.  The lo
// halves need the operands inside f16's normal range: every operand tensor carries a power-of-two scale 2^e chosen from its absolute
// maximum (amax 2^e in [2^14, 2^15); exact, undone on the accumulators with one v_ldexp) -- activations and gradients from the maximum
// the PRODUCING kernel's epilogue recorded (a device float: no host round trip), weights when they are packed.  An element 2^-15 below
// its tensor's maximum still has a normal lo half (relative error 2^-22); below that the error is absolute, 2^-37 of the maximum.
//
//   ucn_pack_h3    W[N, K] fp32 (or its transpose) -> the A-operand fragment stream [k-step][tile][hi | lo][lane][8 halfs], scaled
//   ucn_gemm_h3    Y[M, N] (+)= X[M, K] W^T (+ bias) (+ row-group bias) (ReLU) (mask), N <= 256; records max |Y|
//   ucn_wgrad_h3   GW[N, K] = GY[M, N]^T X[M, K], gb[N] = column sums of GY (exact fp32 sums), fixed-order split-K partials
//   ucn_amax_f32   max |X| of a strided [M, K] operand no kernel of this file produced (atomic max on the bit pattern)
//
// k_gemm_h3: a workgroup = 4 waves x 32 rows against all NT x 32 output columns, two workgroups per CU; the weight stream passes
// through LDS one k-step (16 k: NT x 2 KiB) at a time, double buffered, requested two steps ahead into registers and written one step
// ahead; a wave's activations arrive in chunks of 64 k (coalesced 256-byte row pieces), parked in its private LDS tile, read back as
// operands, scaled and split in registers (8 v_ldexp + 20 VALU per 3 NT MFMAs).  2 KiB of HBM traffic per row against 96 MFMA cycles
// at N = K = 256.
// k_wgrad_h3: 32-row slabs of GY and X are scaled and split as they arrive (coalesced float4 loads), stored as [hi | lo] f16 images in
// the row-major order they have in memory and read back as MFMA fragments with ds_read_b64_tr_b16 (the transposing LDS read of gfx950).
#include <utility>

#include "gemm_epilogue.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));     // register-resident 16-byte units (arrays of HIP_vector_type structs
typedef float f4v __attribute__((ext_vector_type(4)));        // behind lambda reference parameters were left in scratch memory)

// (UCN_H3_EXP_*: timing-only experiment builds, tools/build_variant.sh -- results are garbage)
#ifdef UCN_H3_EXP_NOMFMA
__device__ __forceinline__ f32x16 mfma_h3(h8 a, h8 b, f32x16 c) { c[0] += (float)a[0] * (float)b[0]; return c; }
#else
__device__ __forceinline__ f32x16 mfma_h3(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#endif

// e with amax 2^e in [2^14, 2^15): hi halves stay below f16's maximum (65504), lo halves of everything within 2^-15 of the maximum stay
// normal.  Zero, infinite or NaN maxima: no scaling (e = 0); a subnormal maximum: 2^140.
__host__ __device__ __forceinline__ int h3_exponent(float amax) {
    uint32_t b;
    memcpy(&b, &amax, 4);
    const uint32_t ex = (b >> 23) & 0xffu;
    if (ex == 255u) return 0;
    if (ex == 0u) return (b & 0x7fffffu) ? 140 : 0;
    return 141 - (int)ex;
}

// hi = f16(v), lo = f16(v - hi).  Written so that the COMPILER emits v_fma_mix{lo,hi}_f16 (f16 source hi, f32 multiplier, f32 source v: the
// exact difference rounded once): `neg1` is -1.0f made opaque (h3_neg1) -- with the literal the fma is folded into a subtraction and the
// lo halves cost 6 instructions per pair instead of 2.  NOT inline assembly (the form mlp_ring.h uses), on purpose: the register
// allocator gave the asm's outputs the registers of the A operand of the MFMA issued just before (k_gemm_h3<4, 16>), and for an inline
// asm the hazard recogniser inserts no wait states -- the VALU write landed while the matrix core was still reading the operand and
// the first output tile of some rows lost its lo x hi term (errors of 2^-12, found by the N = 128 test).
__device__ __forceinline__ float h3_neg1() {
    float v = -1.0f;
    asm("" : "+s"(v));
    return v;
}
__device__ __forceinline__ void h3_split_pair(float a, float b, float neg1, uint32_t &hw, uint32_t &lw) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 hp, lp;
    hp[0] = (_Float16)a;
    hp[1] = (_Float16)b;
    lp[0] = (_Float16)__builtin_fmaf((float)hp[0], neg1, a);
    lp[1] = (_Float16)__builtin_fmaf((float)hp[1], neg1, b);
    hw = __builtin_bit_cast(uint32_t, hp);
    lw = __builtin_bit_cast(uint32_t, lp);
}
__device__ __forceinline__ void h3_split8(const float (&v)[8], float neg1, h8 &hi, h8 &lo) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; p++) h3_split_pair(v[2 * p], v[2 * p + 1], neg1, hw[p], lw[p]);
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}

template <uint32_t... Is, class F>
__device__ __forceinline__ void h3_static_for_impl(std::integer_sequence<uint32_t, Is...>, F &&f) {
    (f(std::integral_constant<uint32_t, Is>{}), ...);
}
template <uint32_t N, class F>
__device__ __forceinline__ void h3_static_for(F &&f) {
    h3_static_for_impl(std::make_integer_sequence<uint32_t, N>{}, f);
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}
// *slot = max(*slot, max over the workgroup of m) on the bit pattern (non-negative floats order like uints): ONE atomic per workgroup,
// its result unused (the wave does not wait for it).  (first r06 build: one atomicMax per wave -- 30 720 read-modify-writes of one
// address per GEMM, serialised at ~8 ns each: 0.25 ms of every call, more than the narrow shapes' whole data movement; a read of the
// slot in front, to skip the atomic, cost the workgroup's tail a memory round trip instead: 7 % of its time.)  `red`: >= blockDim.x / 64
// floats of LDS nobody else touches any more; every thread of the workgroup calls.
__device__ __forceinline__ void block_amax_to_slot(float m, float *red, uint32_t *slot) {
    m = wave_max(m);
    const uint32_t nw = blockDim.x >> 6;
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (uint32_t w = 1; w < nw; w++) m = fmaxf(m, red[w]);
        if (m > 0.0f) atomicMax(slot, __float_as_uint(m));          // no return value used: the wave does not wait for it
    }
}

// ---- ucn_amax_f32 --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_amax2d(const float *__restrict__ X, uint32_t ldx, uint64_t M, uint32_t K, uint32_t *__restrict__ out_bits) {
    float m = 0.0f;
    const bool vec = K % 4u == 0u && ldx % 4u == 0u && ((uintptr_t)X & 15u) == 0u;
    if (vec) {
        const uint32_t kq = K / 4u;
        const uint64_t n = M * kq;
        for (uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * 256u) {
            const uint64_t r = idx / kq;
            const uint32_t q = (uint32_t)(idx - r * kq);
            const float4 v = *reinterpret_cast<const float4 *>(X + r * ldx + 4u * q);
            m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
        }
    } else {
        const uint64_t n = M * K;
        for (uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * 256u) {
            const uint64_t r = idx / K;
            m = fmaxf(m, fabsf(X[r * ldx + (idx - r * K)]));
        }
    }
    __shared__ float s_red[4];                                    // (fmaxf drops NaNs: a NaN entry poisons the product, not the scale)
    block_amax_to_slot(m, s_red, out_bits);
}

// ---- ucn_pack_h3 ---------------------------------------------------------------------------------------------------------------------
// dst (16-byte units) [((s * nt + t) * 2 + part) * 64 + lane] = part(2^e V[32 t + (lane & 31)][16 s + 8 (lane >> 5) + 0..7]),
// V[n][k] = W[n * ldw + k] (transposed: W[k * ldw + n]) inside [N, K], 0 outside; part 0 = hi, 1 = lo; e from max |W| (every block
// forms the maximum itself: the weight is at most 256 x 544 and lives in L2; no second launch, no atomics).
__global__ __launch_bounds__(256) void k_pack_h3(const float *__restrict__ W, uint32_t ldw, uint32_t N, uint32_t K, int transposed,
                                                 uint32_t nt, uint32_t ksteps, uint4 *__restrict__ dst, float *__restrict__ wmax_out) {
    __shared__ float s_m[4];
    const uint32_t rows = transposed ? K : N, cols = transposed ? N : K;
    float m = 0.0f;
    if (ldw == cols && (rows * cols) % 4u == 0u && ((uintptr_t)W & 15u) == 0u) {     // one contiguous block (the usual case): no index arithmetic
        const uint32_t n4 = rows * cols / 4u;
        for (uint32_t idx = threadIdx.x; idx < n4; idx += 256u) {
            const float4 v = reinterpret_cast<const float4 *>(W)[idx];
            m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < rows * cols; idx += 256u) {
            const uint32_t r = idx / cols;
            m = fmaxf(m, fabsf(W[(size_t)r * ldw + (idx - r * cols)]));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63u) == 0u) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) *wmax_out = m;
    const int e = h3_exponent(m);
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= ksteps * nt * 64u) return;
    const uint32_t lane = gid & 63u, t = (gid >> 6) % nt, s = (gid >> 6) / nt;
    const uint32_t n = 32u * t + (lane & 31u), k0 = 16u * s + 8u * (lane >> 5);
    float v[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
        const uint32_t k = k0 + q;
        const float w = (n < N && k < K) ? (transposed ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k]) : 0.0f;
        v[q] = ldexpf(w, e);
    }
    h8 hi, lo;
    h3_split8(v, h3_neg1(), hi, lo);
    const size_t base = ((size_t)(s * nt + t) * 2u) * 64u + lane;
    dst[base] = __builtin_bit_cast(uint4, hi);
    dst[base + 64u] = __builtin_bit_cast(uint4, lo);
}

// ---- k_gemm_h3 -----------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kH3Waves = 4u, kH3Threads = 64u * kH3Waves;     // 4 waves x 32 rows per workgroup, TWO workgroups per CU (see below)
// The thread's share of the weight stream is requested PX k-steps ahead, one register set per step in flight.  A wave's activations
// arrive in CHUNKS of PX k-steps (64 k): eight coalesced loads of 4 rows x 256 bytes each, requested one chunk ahead, parked in the
// wave's private LDS tile as they lie in memory ([32 rows][64 + 4 floats]: the tile the epilogue stages through later) and read back
// as MFMA operands (a lane's 8 consecutive k of its own row: two conflict-free ds_read_b128 per step).
// (first r06 build: every lane fetched its own 2 x 16 bytes per step straight from HBM -- at any moment a CU had 64-byte pieces of
// 256 different rows in flight, and whatever the shape or the prefetch depth the kernel moved 2.4-3.0 TB/s: the DRAM page locality of
// scattered 64-byte visits, where the coalesced slab loads of k_wgrad_h3 reach 3.9 and a plain copy 4.9.)
// vmcnt retires in issue order, so both streams have the same depth on purpose: a wait for a weight piece requested d steps ago also
// waits for every activation piece requested before it.
// Workgroup shape (r06, second build): 4 waves, two workgroups per CU (<= 256 registers each), instead of one 8-wave workgroup.  Per
// workgroup of the 8-wave form, s_memtime (tools/h3_clock.py, N = K = 256): prologue 16 %, k loop 60 %, epilogue 17 %, maximum 7 % -- with
// one workgroup per CU nothing ran beside the 40 % that is not the loop.  Two independent workgroups put one's first-touch latency, its
// 128 KiB of stores and its tail behind the other's MFMAs; the weight stream is fetched per 128 rows instead of per 256 (L2 -> LDS,
// 13 B per clock and CU: nothing), its registers are the same (4 pieces per thread and step, two steps in flight).
// Where the 256 x 256 shape stands (profiles/r06/gemm_h3_notes.txt): 0.67 ms per 2^20 rows = 3.2 TB/s of X + Y; the same kernel without its
// MFMAs (and their fragment reads) 0.46 ms, without its stores 0.48 ms: the memory time and the matrix time ADD.  A start-up stagger of the
// first dispatch round (to rule out chip-wide lockstep of the load / multiply / store phases) changed nothing; neither did the prefetch
// depth, the activation access pattern (per-lane 16-byte pieces, coalesced 256-byte row pieces) or pairing the output tiles.  Counters:
// MFMA busy 34 % at the 1.7 GHz the kernel runs at, LDS 27 %, 28 % of the wave-cycles in s_waitcnt -- with one activation chunk (8 KiB)
// per wave in flight the chip holds 16 MB of requests, one loaded-latency's worth; a second chunk has no registers (256 of 256).
constexpr uint32_t kH3PX = 4u;                                    // k-steps per activation chunk
constexpr uint32_t kH3PW = 2u;                                    // the weight stream's depth in k-steps (register sets per thread)
constexpr uint32_t h3_slot_units(uint32_t nt) { return nt * 128u < kH3Threads ? kH3Threads : nt * 128u; }     // 16-byte units per LDS slot
constexpr size_t h3_lds_bytes(uint32_t nt) { return 2u * h3_slot_units(nt) * 16u + kH3Waves * kStageFloats * 4u; }   // weight ring + wave tiles

#ifdef UCN_H3_CLOCK
__device__ unsigned long long g_h3_clk[8];        // experiment build: summed s_memtime cycles per phase over the workgroups (thread 0)
#define H3_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_h3_clk[k], now_ - clk_last_); clk_last_ = now_; } } while (0)
#else
#define H3_STAMP(k) do { } while (0)
#endif

struct H3Scales {
    const float *xmax, *wmax;      // max |X|, max |W| (device): the operand scales
    uint32_t *ymax;                // atomic max of |Y| on the bit pattern (may be null)
    uint32_t halves;               // 1: a workgroup owns all NT tiles of the packed stream.  2: the stream holds 2 NT tiles and workgroup
    //                                b owns column half (b >> 3) & 1 of row tile (b >> 4) * 8 + (b & 7) -- the two halves of a row tile
    //                                are 8 blocks apart: the same XCD (blocks go round robin over the 8 XCDs), dispatched back to back,
    //                                so the second read of the activation rows is an L2 hit, not HBM traffic
};

// KS > 0: the k loop fully unrolled for K = 16 KS exactly (the layer widths of this model: 64, 128, 256).  Not a nicety: across a loop's
// back edge hipcc's wait-count pass does not carry the ORDER of the loads in flight -- a register requested in the previous trip is
// waited for as if every load of that trip had to land first (vmcnt(6) where 14 were allowed: the requests' real depth fell from four
// steps to one and a half).  In straight-line code its counts are exact.  KS = 0: the loop, any K.
#ifndef UCN_H3_OCC_NARROW
#define UCN_H3_OCC_NARROW 2          // experiment: workgroups per CU asked for the NT <= 4 instantiations (3: <= 168 registers)
#endif
template <uint32_t NT, uint32_t KS>
__global__ __launch_bounds__(kH3Threads, NT <= 4u ? UCN_H3_OCC_NARROW : 2) void k_gemm_h3(const float *__restrict__ X, uint32_t ldx, const u4v *__restrict__ Wp, uint32_t K_rt,
                                                          uint32_t ksteps_rt, H3Scales sc, GemmOut o) {
    constexpr uint32_t PX = kH3PX, PW = kH3PW;
    static_assert(PX % PW == 0u, "the weight sets rotate inside a chunk");
    const uint32_t K = KS ? 16u * KS : K_rt, ksteps = KS ? KS : ksteps_rt;
    constexpr uint32_t CH = NT * 128u, SLOT = h3_slot_units(NT);   // 16-byte units per k-step of the stream / per LDS slot
    constexpr uint32_t WPT = (CH + kH3Threads - 1u) / kH3Threads;   // units per thread and k-step
    constexpr uint32_t TP = NT >= 2u ? 2u : 1u, NP = NT / TP;      // output tiles go in pairs
    constexpr uint32_t RS = 68u;                                   // row stride of the wave tile in floats (= gemm_store_staged's)
    static_assert(kStageFloats == 32u * RS, "the activation tile is the epilogue's staging tile");
    extern __shared__ u4v s_ring[];                                // [2][SLOT] weight ring, then 8 wave tiles of kStageFloats floats
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, i = lane & 31u, g = lane >> 5;
    const uint32_t half = sc.halves == 2u ? (blockIdx.x >> 3) & 1u : 0u;
    const uint32_t rtile = sc.halves == 2u ? (blockIdx.x >> 4) * 8u + (blockIdx.x & 7u) : blockIdx.x;
    const uint32_t m0 = rtile * (32u * kH3Waves) + wave * 32u, n0 = half * (NT * 32u);
    if (rtile * (32u * kH3Waves) >= o.M) return;                   // (halves == 2: the grid is rounded up to whole groups of 16 blocks)
#ifdef UCN_H3_CLOCK
    unsigned long long clk_last_ = __builtin_amdgcn_s_memtime();
#endif
    float *tile = reinterpret_cast<float *>(s_ring + 2u * SLOT) + wave * kStageFloats;
    const int ex = h3_exponent(*sc.xmax), ew = h3_exponent(*sc.wmax);
    const float neg1 = h3_neg1();
    f32x16 acc[NT];
    gemm_init_acc<NT>(acc, o, m0 + i, n0, g);
    if (o.flags & kGemmAccum) {
#pragma unroll
        for (uint32_t t = 0; t < NT; t++)
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) acc[t][r] = ldexpf(acc[t][r], ex + ew);
    }
    f4v xs[8];                                                     // one chunk in flight: 32 rows x 64 k, 8 pieces of 4 rows x 256 bytes
    u4v wreg[PW][WPT];
    // a chunk piece u: row 4 u + lane / 16, columns 4 (lane % 16) .. + 3 of the chunk.  Every load of the loop is UNCONDITIONAL (clamped
    // address, value zeroed where it is consumed): see k_gemm_f32 -- behind a predicated load's branch the wait-count pass drains
    // everything in flight.
    const uint32_t prow = lane >> 4, pcol = 4u * (lane & 15u);
    auto load_chunk = [&](uint32_t c) {
        const uint32_t k = 64u * c + pcol;
        const float *xk = X + (k < K ? k : 0u);                     // past K: the row's first piece again (zeroed where it is consumed)
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t r = m0 + 4u * u + prow;
            xs[u] = *reinterpret_cast<const f4v *>(xk + (size_t)(r < o.M ? r : o.M - 1u) * ldx);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) *reinterpret_cast<f4v *>(tile + (4u * u + prow) * RS + pcol) = xs[u];
    };
    auto load_w = [&](uint32_t s, u4v (&d)[WPT]) {
        const u4v *src = Wp + ((size_t)(s < ksteps ? s : 0u) * sc.halves + half) * CH;
#pragma unroll
        for (uint32_t u = 0; u < WPT; u++) {
            const uint32_t idx = threadIdx.x + u * kH3Threads;
            d[u] = src[idx < CH ? idx : 0u];
        }
    };
    auto store_w = [&](uint32_t s, const u4v (&d)[WPT]) {
#pragma unroll
        for (uint32_t u = 0; u < WPT; u++) s_ring[(s & 1u) * SLOT + threadIdx.x + u * kH3Threads] = d[u];
    };
    auto step = [&](auto pc, uint32_t s) {
        constexpr uint32_t P = decltype(pc)::value;                 // s % PX = the step inside its chunk
        // 1. requests: at a chunk's first step the NEXT chunk of activations (its registers were parked in LDS at the end of the
        //    previous step), then the weight piece PX steps ahead into the set stored one step ago.  Past the end: the loop form
        //    re-reads the first chunk / step (unused), the unrolled form drops the requests.
#ifndef UCN_H3_EXP_NOW
        if (KS == 0u || s + PW < KS) load_w(s + PW, wreg[P % PW]);       // (before the chunk: the wait for it at the end of step s + 1
#endif                                                               //  must not also wait for the chunk)
#ifndef UCN_H3_EXP_NOX
        if (P == 0u && (KS == 0u || s + PX < KS)) load_chunk(s / PX + 1u);
#endif
        // 2. this step's activations from the wave tile: zero past K, scale, split
        const float *xr = tile + i * RS + 16u * P + 8u * g;
        const f4v x0 = *reinterpret_cast<const f4v *>(xr), x1 = *reinterpret_cast<const f4v *>(xr + 4u);
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        if (KS == 0u && 16u * s + 16u > K) {                        // uniform: whole steps skip the selects (0 x inf = NaN otherwise)
            const uint32_t k = 16u * s + 8u * g;
#pragma unroll
            for (uint32_t q = 0; q < 8; q++) v[q] = k + (q & 4u) < K ? v[q] : 0.0f;
        }
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) v[q] = ldexpf(v[q], ex);
        h8 bhi, blo;
        h3_split8(v, neg1, bhi, blo);
        // 3. products from slot s & 1.  Output tiles in PAIRS (two independent accumulation chains interleaved: three dependent MFMAs in a
        //    row on one accumulator leave the matrix core waiting on itself); a pair's four fragments are refilled IN PLACE, each right
        //    behind the last MFMA that reads it (the lo halves after the second product, the hi halves after the third), so the next
        //    pair's LDS latency sits behind the rest of this pair and no second set of fragment registers exists.
        const u4v *slot = s_ring + (s & 1u) * SLOT + lane;
        u4v ah[TP], al[TP];
#pragma unroll
        for (uint32_t q = 0; q < TP; q++) { ah[q] = slot[(q * 2u) * 64u]; al[q] = slot[(q * 2u + 1u) * 64u]; }
#pragma unroll
        for (uint32_t p = 0; p < NP; p++) {
#pragma unroll
            for (uint32_t q = 0; q < TP; q++) acc[TP * p + q] = mfma_h3(__builtin_bit_cast(h8, ah[q]), bhi, acc[TP * p + q]);
#pragma unroll
            for (uint32_t q = 0; q < TP; q++) acc[TP * p + q] = mfma_h3(__builtin_bit_cast(h8, al[q]), bhi, acc[TP * p + q]);
            if (p + 1u < NP) {
#pragma unroll
                for (uint32_t q = 0; q < TP; q++) al[q] = slot[((TP * (p + 1u) + q) * 2u + 1u) * 64u];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (uint32_t q = 0; q < TP; q++) acc[TP * p + q] = mfma_h3(__builtin_bit_cast(h8, ah[q]), blo, acc[TP * p + q]);
            if (p + 1u < NP) {
#pragma unroll
                for (uint32_t q = 0; q < TP; q++) ah[q] = slot[((TP * (p + 1u) + q) * 2u) * 64u];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // 4. the next step's weights (requested PX - 1 steps ago) into the other slot, whose readers passed the barrier of step s - 1;
        //    at a chunk's last step the next chunk of activations over this one (wave-private: the wave's LDS operations stay in order)
        if (KS == 0u || s + 1u < KS) store_w(s + 1u, wreg[(P + 1u) % PW]);
        if (P == PX - 1u && (KS == 0u || s + 1u < KS)) store_chunk();
#ifndef UCN_H3_EXP_NOBAR
        __syncthreads();
#endif
    };
    // prologue: W(0) and the first chunk go to LDS at once; W(1 .. PX - 1) fill the register sets
    load_w(0u, wreg[0]);
    load_chunk(0u);
#pragma unroll
    for (uint32_t p = 1; p < PW; p++) load_w(p, wreg[p]);
    store_w(0u, wreg[0]);
    store_chunk();
    __syncthreads();
    H3_STAMP(0);
    if constexpr (KS > 0u) {
        static_assert(KS % PX == 0u, "whole chunks");
        h3_static_for<KS>([&](auto sc_) { step(std::integral_constant<uint32_t, decltype(sc_)::value % PX>{}, decltype(sc_)::value); });
    } else {
        for (uint32_t s = 0; s < ksteps; s += PX) {                // ksteps is a multiple of PX (the packed stream is padded with zeros)
            h3_static_for<PX>([&](auto pc) { step(pc, s + decltype(pc)::value); });
        }
    }
    H3_STAMP(1);
    // undo the operand scales (exact), then the shared epilogue through the wave tile
#pragma unroll
    for (uint32_t t = 0; t < NT; t++)
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) acc[t][r] = ldexpf(acc[t][r], -(ex + ew));
    GemmOut oe = o;
    asm volatile("" : "+s"(oe.N), "+s"(oe.M), "+s"(oe.flags));      // (see k_gemm_f32: keeps the epilogue's predicates out of the loop's registers)
    float mx = 0.0f;                                                // always tracked (a run-time choice would put it in scratch memory)
    bool done = false;
#ifdef UCN_H3_EXP_NOSTORE
    if (acc[0][0] != 123.456f) return;
#endif
    if constexpr (NT >= 2u) {
        if ((oe.flags & kGemmVec) && n0 + NT * 32u <= oe.N) {
            // the epilogue's optional pieces are template flags of the store (run-time tests inside it cost the exact engine's resident
            // kernels their registers): second operand pair (256-wide outputs: the host checks), ReLU bits written / read (NT >= 4)
            if constexpr (NT >= 4u) {
                if (oe.x2) {
                    if (oe.bits_in) mx = gemm_store_staged<NT, true, true, 2>(acc, oe, tile, m0, n0, lane);
                    else if (oe.bits_out) mx = gemm_store_staged<NT, true, true, 1>(acc, oe, tile, m0, n0, lane);
                    else mx = gemm_store_staged<NT, true, true>(acc, oe, tile, m0, n0, lane);
                } else {
                    if (oe.bits_in) mx = gemm_store_staged<NT, true, false, 2>(acc, oe, tile, m0, n0, lane);
                    else if (oe.bits_out) mx = gemm_store_staged<NT, true, false, 1>(acc, oe, tile, m0, n0, lane);
                    else mx = gemm_store_staged<NT, true>(acc, oe, tile, m0, n0, lane);
                }
            } else {
                mx = gemm_store_staged<NT, true>(acc, oe, tile, m0, n0, lane);
            }
            done = true;
        }
    }
    if (!done) mx = gemm_store_direct<NT, true>(acc, oe, m0 + i, n0, g);
    H3_STAMP(2);
    if (sc.ymax) {                                                  // uniform
        __syncthreads();                                            // every wave is done with its tile: the weight ring is scratch now
        block_amax_to_slot(mx, reinterpret_cast<float *>(s_ring), sc.ymax);
    }
    H3_STAMP(3);
}

// ---- k_wgrad_h3 ----------------------------------------------------------------------------------------------------------------------
// GW block [NB x 32 rows n][256 columns k] of one chunk of the samples.  LDS: two buffers of four f16 images
// {GY hi, GY lo, X hi, X lo}[32 rows][256 columns], row stride 576 bytes (= 64 mod 256: the four rows of a 16-lane group of the
// transposing read and the two groups of a 32-lane half fall on disjoint banks; wgrad.hip).
constexpr uint32_t kW3Stride = 576u, kW3Img = 32u * kW3Stride, kW3Buf = 4u * kW3Img;      // 18 KiB, 72 KiB

__device__ __forceinline__ h8 tr_frag(const uint8_t *img_lane) {
    // ds_read_b64_tr_b16 x 2: lane (j, g) addressing row 8 g + (j & 15) / 4 (+ 4), column chunk 4 (j & 3) + 16 ((j >> 4) & 1) receives
    // rows 8 g .. 8 g + 7 of column j -- the 8 consecutive k of an MFMA operand -- with no shuffles (tools/tr_probe.hip prints the layout)
    typedef __attribute__((address_space(3))) s4v *lp;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img_lane));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img_lane + 4u * kW3Stride));
    typedef short s8v __attribute__((__vector_size__(8 * sizeof(short))));
    const s8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h8, v);
}

template <uint32_t NB>
__global__ __launch_bounds__(512, 1) void k_wgrad_h3(const float *__restrict__ GY, uint32_t ldg, const float *__restrict__ X, uint32_t ldx,
                                                     uint32_t M, uint32_t N, uint32_t K, uint32_t MC, const float *__restrict__ gmax,
                                                     const float *__restrict__ xmax, float *__restrict__ ws, float *__restrict__ wsb) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_img[];   // [2][kW3Buf]
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, j = lane & 31u, g = lane >> 5;
    constexpr uint32_t NA = NB == 8u ? 2u : NB, NK = NB == 8u ? 4u : 1u;       // n tiles x k tiles per wave
    const uint32_t chunk = blockIdx.x, k0 = blockIdx.y * 256u, n0 = blockIdx.z * (NB * 32u);
    const uint32_t nt0 = NB == 8u ? 2u * (wave & 3u) : 0u, kt0 = NB == 8u ? 4u * (wave >> 2) : wave;
    const uint32_t m_lo = chunk * MC, m_hi = m_lo + MC < M ? m_lo + MC : M;
    const int eg = h3_exponent(*gmax), ex = h3_exponent(*xmax);
    const float neg1 = h3_neg1();
    f32x16 acc[NA][NK];
#pragma unroll
    for (uint32_t a = 0; a < NA; a++)
#pragma unroll
        for (uint32_t t = 0; t < NK; t++)
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) acc[a][t][r] = 0.0f;
    // staging: 32 rows x 128 float4 (64 of GY, 64 of X) per slab, 8 per thread; a thread keeps its column: c4 = tid & 127, rows tid / 128 + 4 u
    const uint32_t c4 = tid & 127u, r0 = tid >> 7;
    const bool is_gy = c4 < 64u;
    const uint32_t cc = c4 & 63u;
    const uint32_t col = is_gy ? n0 + 4u * cc : k0 + 4u * cc;
    const bool col_live = is_gy ? (col < N && cc < NB * 8u) : col < K;
    const float *src = is_gy ? GY + (col_live ? col : 0u) : X + (col_live ? col : 0u);
    const uint32_t ld = is_gy ? ldg : ldx;
    const int esc = is_gy ? eg : ex;
    const uint32_t img_off = (is_gy ? 0u : 2u * kW3Img) + 8u * cc;        // hi image; lo = + kW3Img
    float4 reg[8];
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);                    // column sums of this thread's GY quad over its rows (exact fp32)
    auto load_slab = [&](uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t m = ms + r0 + 4u * u;
            reg[u] = *reinterpret_cast<const float4 *>(src + (size_t)(m < m_hi ? m : m_hi - 1u) * ld);
        }
    };
    auto store_slab = [&](uint32_t buf, uint32_t ms) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t r = r0 + 4u * u;
            const bool live = col_live && ms + r < m_hi;
            float4 v = reg[u];
            v.x = live ? v.x : 0.f; v.y = live ? v.y : 0.f; v.z = live ? v.z : 0.f; v.w = live ? v.w : 0.f;
            if (is_gy) { bs.x += v.x; bs.y += v.y; bs.z += v.z; bs.w += v.w; }
            uint32_t h0, h1, l0, l1;
            h3_split_pair(ldexpf(v.x, esc), ldexpf(v.y, esc), neg1, h0, l0);
            h3_split_pair(ldexpf(v.z, esc), ldexpf(v.w, esc), neg1, h1, l1);
            uint8_t *p = s_img + buf * kW3Buf + img_off + r * kW3Stride;
            *reinterpret_cast<uint2 *>(p) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(p + kW3Img) = make_uint2(l0, l1);
        }
    };
    const uint32_t lrow = 8u * g + ((j & 15u) >> 2), lcol = (4u * (j & 3u) + 16u * ((j >> 4) & 1u)) * 2u;
    const uint32_t lane_off = lrow * kW3Stride + lcol;
    const uint32_t nslabs = (m_hi - m_lo + 31u) / 32u;
    if (nslabs) { load_slab(m_lo); store_slab(0u, m_lo); }
    __syncthreads();
    for (uint32_t sidx = 0; sidx < nslabs; sidx++) {
        const bool more = sidx + 1u < nslabs;
        load_slab(more ? m_lo + (sidx + 1u) * 32u : m_lo);           // (last slab: an unconditional re-read of the first, unused)
        const uint8_t *buf = s_img + (sidx & 1u) * kW3Buf + lane_off;
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ks++) {
            const uint8_t *b = buf + ks * 16u * kW3Stride;
            h8 ah[NA], al[NA], bh[NK], bl[NK];
#pragma unroll
            for (uint32_t a = 0; a < NA; a++) {
                ah[a] = tr_frag(b + 64u * (nt0 + a));
                al[a] = tr_frag(b + kW3Img + 64u * (nt0 + a));
            }
#pragma unroll
            for (uint32_t t = 0; t < NK; t++) {
                bh[t] = tr_frag(b + 2u * kW3Img + 64u * (kt0 + t));
                bl[t] = tr_frag(b + 3u * kW3Img + 64u * (kt0 + t));
            }
#pragma unroll
            for (uint32_t t = 0; t < NK; t++)
#pragma unroll
                for (uint32_t a = 0; a < NA; a++) acc[a][t] = mfma_h3(ah[a], bh[t], acc[a][t]);
#pragma unroll
            for (uint32_t t = 0; t < NK; t++)
#pragma unroll
                for (uint32_t a = 0; a < NA; a++) acc[a][t] = mfma_h3(al[a], bh[t], acc[a][t]);
#pragma unroll
            for (uint32_t t = 0; t < NK; t++)
#pragma unroll
                for (uint32_t a = 0; a < NA; a++) acc[a][t] = mfma_h3(ah[a], bl[t], acc[a][t]);
        }
        if (more) store_slab((sidx + 1u) & 1u, m_lo + (sidx + 1u) * 32u);
        __syncthreads();
    }
    float *out = ws + (size_t)chunk * N * K;
#pragma unroll
    for (uint32_t a = 0; a < NA; a++)
#pragma unroll
        for (uint32_t t = 0; t < NK; t++) {
            const uint32_t k = k0 + 32u * (kt0 + t) + j;
#pragma unroll
            for (uint32_t r = 0; r < 16; r++) {
                const uint32_t n = n0 + 32u * (nt0 + a) + (r & 3u) + 8u * (r >> 2) + 4u * g;
                if (n < N && k < K) out[(size_t)n * K + k] = ldexpf(acc[a][t][r], -(eg + ex));
            }
        }
    if (wsb && blockIdx.y == 0) {                                   // uniform
        float4 *red = reinterpret_cast<float4 *>(s_img);            // [4 row groups][64 quads]: the images are free after the last barrier
        if (is_gy) red[r0 * 64u + cc] = bs;
        __syncthreads();
        if (tid < 64u) {
            const float4 p0 = red[tid], p1 = red[64u + tid], p2 = red[128u + tid], p3 = red[192u + tid];
            const float e[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t n = n0 + 4u * tid + c;
                if (n < N && 4u * tid + c < NB * 32u) wsb[(size_t)chunk * N + n] = e[c];
            }
        }
    }
}

// out[e] = sum over the chunks of ws[c][e], in chunk order (deterministic)
__global__ __launch_bounds__(256) void k_reduce_h3(const float *__restrict__ ws, uint32_t chunks, size_t stride, size_t n, float *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t c = 0;
    for (; c + 8u <= chunks; c += 8u) {
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) p[u] += ws[(size_t)(c + u) * stride + e];
    }
    for (uint32_t u = 0; c < chunks; c++, u++) p[u] += ws[(size_t)c * stride + e];
    out[e] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

uint32_t h3_chunk_rows(uint64_t M) {
    // 256 chunks = one workgroup per CU per (n block, k block); a chunk is a whole number of 32-row slabs
    uint64_t mc = (M + 255u) / 256u;
    mc = (mc + 31u) / 32u * 32u;
    return (uint32_t)(mc < 512u ? 512u : mc);
}
uint32_t h3_tiles(uint32_t N) { return N <= 32u ? 1u : N <= 64u ? 2u : N <= 128u ? 4u : 8u; }
uint32_t h3_ksteps(uint32_t K) { return (ucn_div_up(K, 16) + kH3PX - 1u) / kH3PX * kH3PX; }

}  // namespace

extern "C" int ucn_amax_f32(const float *X, uint32_t ldx, uint64_t M, uint32_t K, float *slot, ucn_stream_t stream) {
    UCN_REQUIRE(X && slot, "amax_f32: null pointer argument");
    UCN_REQUIRE(ldx >= K, "amax_f32: ldx %u < K %u", ldx, K);
    if (M == 0 || K == 0) return 0;
    const uint64_t n = M * K / 4u + 1u;
    const uint32_t blocks = (uint32_t)(n / 256u + 1u < 2048u ? n / 256u + 1u : 2048u);
    hipLaunchKernelGGL(k_amax2d, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, M, K, reinterpret_cast<uint32_t *>(slot));
    UCN_LAUNCH_CHECK("amax_f32");
    return 0;
}

#ifdef UCN_H3_CLOCK
extern "C" int ucn_h3_clock_read(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_h3_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_h3_clk), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" uint64_t ucn_relu_bits_words(uint64_t M, uint32_t N) { return gemm_bits_words(M, N); }

extern "C" uint64_t ucn_pack_h3_bytes(uint32_t N, uint32_t K) { return (uint64_t)h3_ksteps(K) * h3_tiles(N) * 2048u; }

extern "C" int ucn_pack_h3(const float *W, uint32_t ldw, uint32_t N, uint32_t K, int transposed, void *packed, float *wmax_out,
                           ucn_stream_t stream) {
    UCN_REQUIRE(W && packed && wmax_out, "pack_h3: null pointer argument");
    UCN_REQUIRE(N >= 1 && N <= 256 && K >= 1, "pack_h3: N = %u (1 .. 256), K = %u", N, K);
    UCN_REQUIRE(ldw >= (transposed ? N : K), "pack_h3: ldw %u does not cover a row", ldw);
    UCN_REQUIRE(((uintptr_t)packed & 15u) == 0u, "pack_h3: the packed stream must be 16-byte aligned");
    const uint32_t nt = h3_tiles(N), ks = h3_ksteps(K);
    hipLaunchKernelGGL(k_pack_h3, dim3(ucn_div_up((uint64_t)ks * nt * 64u, 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, transposed,
                       nt, ks, reinterpret_cast<uint4 *>(packed), wmax_out);
    UCN_LAUNCH_CHECK("pack_h3");
    return 0;
}

extern "C" int ucn_gemm_h3(const float *X, uint32_t ldx, const void *packed, const float *xmax, const float *wmax, const float *bias,
                           uint32_t M, uint32_t N, uint32_t K, int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm,
                           const float *rowbias, uint32_t ldr, uint32_t rgroup, float *ymax, ucn_stream_t stream) {
    return ucn_gemm_h3_x2(X, ldx, packed, xmax, wmax, bias, M, N, K, flags, Y, ldy, mask, ldm, rowbias, ldr, rgroup, nullptr, 0, nullptr, 0, nullptr,
                          nullptr, ymax, stream);
}

extern "C" int ucn_gemm_h3_x2(const float *X, uint32_t ldx, const void *packed, const float *xmax, const float *wmax, const float *bias,
                              uint32_t M, uint32_t N, uint32_t K, int flags, float *Y, uint32_t ldy, const float *mask, uint32_t ldm,
                              const float *rowbias, uint32_t ldr, uint32_t rgroup, const float *X2, uint32_t ldx2, const float *W2, uint32_t ldw2,
                              uint64_t *relu_bits_out, const uint64_t *mask_bits, float *ymax, ucn_stream_t stream) {
    UCN_REQUIRE(!X2 == !W2, "gemm_h3: the 4-wide second operand pair needs both X2 [M, 4] and W2 [N, 4]");
    UCN_REQUIRE(!X2 || (ldx2 >= 4 && ldw2 >= 4 && ldx2 % 4u == 0u && ldw2 % 4u == 0u && (((uintptr_t)X2 | (uintptr_t)W2) & 15u) == 0u),
                "gemm_h3: X2 / W2 rows are 4 floats, 16-byte aligned (ldx2 %u ldw2 %u)", ldx2, ldw2);
    UCN_REQUIRE(X && packed && Y && xmax && wmax, "gemm_h3: null pointer argument");
    UCN_REQUIRE(N >= 1 && N <= 256, "gemm_h3: N = %u (1 .. 256: one column block)", N);
    UCN_REQUIRE(K % 4u == 0u && ldx % 4u == 0u && ldx >= K && ldy >= N,
                "gemm_h3: K, ldx must be multiples of 4 (16-byte operand loads) and cover the operands (K %u ldx %u N %u ldy %u)", K, ldx, N, ldy);
    UCN_REQUIRE((((uintptr_t)X | (uintptr_t)packed) & 15u) == 0u, "gemm_h3: X and the packed weight must be 16-byte aligned");
    UCN_REQUIRE((flags & ~7) == 0, "gemm_h3: flags = UCN_GEMM_ACCUMULATE | UCN_GEMM_RELU | UCN_GEMM_MASK");
    UCN_REQUIRE(!(flags & (int)kGemmMask) || (mask && ldm >= N), "gemm_h3: UCN_GEMM_MASK needs a mask [M, N] (ldm %u N %u)", ldm, N);
    UCN_REQUIRE(!rowbias || (rgroup > 0 && ldr >= N), "gemm_h3: a row-group bias needs rgroup > 0 and ldr >= N (rgroup %u ldr %u N %u)", rgroup, ldr, N);
    if (M == 0) return 0;
    const bool vec = N % 4u == 0u && ldy % 4u == 0u && ((uintptr_t)Y & 15u) == 0u && (!bias || ((uintptr_t)bias & 15u) == 0u) &&
                     (!(flags & (int)kGemmMask) || (ldm % 4u == 0u && ((uintptr_t)mask & 15u) == 0u)) &&
                     (!rowbias || (ldr % 4u == 0u && ((uintptr_t)rowbias & 15u) == 0u));
    UCN_REQUIRE(!X2 || (vec && N == 256u), "gemm_h3: the second operand pair is built for 256-wide outputs with the vector epilogue (N %u)", N);
    GemmOut o{bias, (flags & (int)kGemmMask) ? mask : nullptr, rowbias, Y, ldy, ldm, ldr, rgroup ? rgroup : 1u, M, N, (uint32_t)flags | (vec ? kGemmVec : 0u)};
    o.x2 = X2; o.w2 = W2; o.ldx2 = ldx2; o.ldw2 = ldw2;
    UCN_REQUIRE(!(relu_bits_out || mask_bits) || (vec && N % 64u == 0u && N >= 128u && h3_tiles(N) * 32u == N),
                "gemm_h3: ReLU bit masks need the staged epilogue of a 128- or 256-wide output (N %u)", N);
    UCN_REQUIRE(!(mask_bits && (flags & (int)kGemmMask)), "gemm_h3: a float mask and a bit mask together");
    o.bits_out = reinterpret_cast<uint32_t *>(relu_bits_out);
    o.bits_in = reinterpret_cast<const uint32_t *>(mask_bits);
    const uint32_t ks = h3_ksteps(K);
    // UCN_H3_HALVES=2 (experiment, measured SLOWER: 0.755 against 0.656 ms at N = K = 256, profiles/r06/gemm_h3_notes.txt): 256-wide outputs as
    // two workgroups of 4 tiles per row tile instead of one of 8 -- the 4-tile kernel moves 4.3 TB/s on its own 128-wide shape, but here the
    // activation rows are fetched twice and the second fetch is not the L2 hit the block pairing was meant to make it
    static const bool two_wg = getenv("UCN_H3_HALVES") != nullptr && atoi(getenv("UCN_H3_HALVES")) == 2;
    const uint32_t halves = (h3_tiles(N) == 8u && two_wg) ? 2u : 1u;
    H3Scales sc{xmax, wmax, reinterpret_cast<uint32_t *>(ymax), halves};
    const uint32_t rtiles = ucn_div_up(M, 32u * kH3Waves);
    const dim3 grid(halves == 2u ? ucn_div_up(rtiles, 8) * 16u : rtiles);
    hipStream_t st = (hipStream_t)stream;
    static const bool no_unroll = getenv("UCN_H3_NO_UNROLL") != nullptr;              // A/B switch: every shape through the loop form
    const uint32_t ku = (K % 16u == 0u && !no_unroll) ? K / 16u : 0u;                 // unrolled forms: K = 64 (wide outputs only), 128, 256
#define UCN_H3K(NT, KS) hipLaunchKernelGGL((k_gemm_h3<NT, KS>), grid, dim3(kH3Threads), h3_lds_bytes(NT), st, X, ldx, reinterpret_cast<const u4v *>(packed), K, ks, sc, o)
#define UCN_H3(NT)                                         \
    do {                                                   \
        if (ku == 16u) UCN_H3K(NT, 16);                    \
        else if (ku == 8u) UCN_H3K(NT, 8);                 \
        else if (ku == 4u) UCN_H3K(NT, 4);                 \
        else UCN_H3K(NT, 0);                               \
    } while (0)
    switch (h3_tiles(N) / halves) {
        case 1: UCN_H3(1); break;
        case 2: UCN_H3(2); break;
        case 4: UCN_H3(4); break;
        default: UCN_H3(8); break;
    }
#undef UCN_H3
#undef UCN_H3K
    UCN_LAUNCH_CHECK("gemm_h3");
    return 0;
}

extern "C" uint64_t ucn_wgrad_h3_ws_floats(uint32_t N, uint32_t K, uint64_t M) {
    if (M == 0 || M > 0xFFFFFFFFull) return 0;
    const uint32_t mc = h3_chunk_rows(M);
    const uint64_t chunks = (M + mc - 1u) / mc;
    return chunks * ((uint64_t)N * K + N);
}

extern "C" int ucn_wgrad_h3(const float *GY, uint32_t ldg, const float *X, uint32_t ldx, const float *gmax, const float *xmax, uint32_t M,
                            uint32_t N, uint32_t K, float *ws, float *GW, float *gb, ucn_stream_t stream) {
    UCN_REQUIRE(GY && X && ws && GW && gmax && xmax, "wgrad_h3: null pointer argument");
    UCN_REQUIRE(N % 4u == 0u && K % 4u == 0u && ldg % 4u == 0u && ldx % 4u == 0u && ldg >= N && ldx >= K,
                "wgrad_h3: N, K, ldg, ldx must be multiples of 4 and cover the operands (N %u K %u ldg %u ldx %u)", N, K, ldg, ldx);
    UCN_REQUIRE((((uintptr_t)GY | (uintptr_t)X) & 15u) == 0u, "wgrad_h3: GY and X must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (hipMemsetAsync(GW, 0, (size_t)N * K * sizeof(float), st) != hipSuccess) return ucn_fail("wgrad_h3: hipMemsetAsync failed");
        if (gb && hipMemsetAsync(gb, 0, (size_t)N * sizeof(float), st) != hipSuccess) return ucn_fail("wgrad_h3: hipMemsetAsync failed");
        return 0;
    }
    const uint32_t mc = h3_chunk_rows(M), chunks = ucn_div_up(M, mc);
    float *wsb = ws + (size_t)chunks * N * K;
#define UCN_W3(NB)                                                                                                                 \
    hipLaunchKernelGGL((k_wgrad_h3<NB>), dim3(chunks, ucn_div_up(K, 256), ucn_div_up(N, NB * 32u)), dim3(512), 2u * kW3Buf, st, GY, ldg, X, \
                       ldx, M, N, K, mc, gmax, xmax, ws, gb ? wsb : nullptr)
    if (N <= 32u) UCN_W3(1);
    else if (N <= 64u) UCN_W3(2);
    else UCN_W3(8);
#undef UCN_W3
    const size_t nk = (size_t)N * K;
    hipLaunchKernelGGL(k_reduce_h3, dim3(ucn_div_up(nk, 256)), dim3(256), 0, st, ws, chunks, nk, nk, GW);
    if (gb) hipLaunchKernelGGL(k_reduce_h3, dim3(ucn_div_up(N, 256)), dim3(256), 0, st, wsb, chunks, (size_t)N, (size_t)N, gb);
    UCN_LAUNCH_CHECK("wgrad_h3");
    return 0;
}
