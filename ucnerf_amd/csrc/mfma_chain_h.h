// Split-f16 variant of the register-chained MFMA engine (mfma_chain.h): fp32-class accuracy at the
// f16 matrix rate.
//
// On CDNA4 the fp32-input MFMA runs at the fp32 VECTOR rate (157 TF), 1/16 of the f16/bf16 rate, so an
// fp32 MLP is capped at 2.7 M rays/s on this workload.  Here every fp32 operand v is split on the fly
// into two f16 numbers, hi = f16(v), lo = f16(v - hi) (|v - hi - lo| <= 2^-22 |v|), and each product
// is evaluated as  hi*hi + hi*lo + lo*hi  with v_mfma_f32_32x32x16_f16 (products of f16 are exact in
// fp32; accumulation is fp32; the dropped lo*lo term is <= 2^-22 relative).  Three MFMAs at 16x the
// rate = 5.3x the fp32-MFMA throughput, with a per-product error of ~3e-7 relative, i.e. ~5x fp32
// epsilon: measured effect on the rendered pixels is below the reference's own noise floor.
//
// Layouts.  C/D of 32x32x16 is the same as 32x32x2 (lane = column + 32*half g, register r = row
// (r&3)+8(r>>2)+4g).  A/B hold 8 consecutive-k halfs per lane, k = 8g+e.  A 32-row activation tile is
// consumed in two k-steps s = 0,1; step s takes accumulator registers r = 8s..8s+7 of each lane, so the
// 16 k's of a step are the rows (r&3)+8(r>>2)+4g -- a fixed permutation that the packers apply to
// the weight columns.  Stream: a PAIR of 1 KiB groups [hi][lo] per (out tile, in tile, s).
//
// With one wave per SIMD nothing but the wave's own instruction stream hides latency, and every
// instruction between two MFMAs costs issue time.  Hence:
//   * the steady state has NO global loads (biases ride in the weight stream: as broadcast bias tiles at
//     its head or as the weight column that multiplies a constant-1 input);
//   * A operands are requested from LDS kDepth pairs ahead of their MFMAs (APipe), across layer boundaries;
//   * stream positions are TEMPLATE arguments: one s_waitcnt lgkmcnt per double step with a compile-time
//     count (a hint on top of the compiler's own tracking of the reads), ring slots and chunk boundaries
//     resolved at compile time, and the L2 -> LDS DMA of the next chunk spread over the steps.
//     (Hand-written ds_read_b128 with immediate offsets were tried and returned wrong data; the compiler's
//     loads are just as fast here.)
#pragma once
#include <utility>

#include "mfma_chain.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct HTile {   // one 32-neuron activation tile as B operands: [k-step s]
    h8 hi[2], lo[2];
};
constexpr int kDepth = 6;   // A-operand pairs in flight: pair i + kDepth is requested right after step i issues
struct APipe {              // ring of A operands, slot = (pair index) % kDepth
    h8 hi[kDepth], lo[kDepth];
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>)
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ f32x16 mfma16h(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const float (&v)[8], h8 &hi, h8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)(v[e] - (float)h);
    }
}
__device__ __forceinline__ void split_tile(const f32x16 &a, HTile &t) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = a[8 * s + e];
        split8(v, t.hi[s], t.lo[s]);
    }
}
// ReLU + split of the 8 accumulator registers that form k-step s of a tile
__device__ __forceinline__ void relu_split_half(const f32x16 &a, const int s, HTile &t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = fmaxf(a[8 * s + e], 0.0f);
    split8(v, t.hi[s], t.lo[s]);
}

// lgkmcnt-only wait (vmcnt = 63, expcnt = 7 left alone)
template <int N>
__device__ __forceinline__ void wait_lds() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}

// Request pair G (even group index) into its ring slot.  The DMA of the NEXT chunk is spread over this
// chunk's requests, one piece per 4 groups: issued as a burst of 16 behind the barrier, the pieces cost
// the issuing (= computing) wave 100-185 cycles each while the address queue is full of them.
// (piece 0 of chunk 1 is the kernel's job: group 0 holds the biases and is never requested here)
// GEND = end of the pipelined segment; the stream carries kTailGroups more groups behind it, so chunk
// c exists iff c * kChunkGroups < GEND + kTailGroups.
constexpr int kTailGroups = 4;
template <int G, int GEND>
__device__ __forceinline__ void pipe_fetch(APipe &p, WeightStream &ws) {
    if constexpr (G % kChunkGroups == 0) ws.sync();
    if constexpr (G % 4 == 0 && (G / kChunkGroups + 1) * kChunkGroups < GEND + kTailGroups)
        ws.piece_unchecked(G / kChunkGroups + 1, (G % kChunkGroups) / 4);
    p.hi[(G / 2) % kDepth] = __builtin_bit_cast(h8, ws.group(G));
    p.lo[(G / 2) % kDepth] = __builtin_bit_cast(h8, ws.group(G + 1));
}
// reads still in flight that are YOUNGER than the four operands of the double step at G, once everything up to
// `fetched` (exclusive, group index) has been requested
template <int G, int FETCHED>
constexpr int younger_reads() { return FETCHED - (G + 4) > 0 ? FETCHED - (G + 4) : 0; }
constexpr int imin(int a, int b) { return a < b ? a : b; }

// start of a pipelined segment [G0, GEND): kDepth pairs in flight, the first double step's operands landed
template <int G0, int GEND>
__device__ __forceinline__ void pipe_prime(APipe &p, WeightStream &ws) {
    static_for<kDepth>([&](auto d) {
        if constexpr (G0 + 2 * d.value < GEND) pipe_fetch<G0 + 2 * d.value, GEND>(p, ws);
    });
    wait_lds<younger_reads<G0, imin(GEND, G0 + 2 * kDepth)>()>();
}

// One double step: two OUTPUT tiles (A pairs G and G+2) against the same 16 k's of the input,
//   acc0 += A(G) . B,  acc1 += A(G+2) . B,
// six MFMAs alternating between the two accumulators, then the two pairs kDepth ahead are requested
// into the slots just consumed, then ONE wait for the next double step's four operands.  The alternation
// matters: an MFMA that accumulates into the register block the previous MFMA is still writing does not
// issue back to back (MI355X_MICROARCH.md: "+43 cyc for the first extra state between two MFMAs on the
// SAME accumulator").  `valu_work` is independent VALU work (~40 instructions) that rides in the shadow
// of the six MFMAs: the sched_group_barriers ask for 1 MFMA, 7 VALU, 1 MFMA, 7 VALU ... so that each
// 32-cycle MFMA covers 28 cycles of VALU issue instead of the VALU phase idling the matrix pipe.
template <int G, int GEND, bool WITH_VALU, class F>
__device__ __forceinline__ void dstep_impl(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, APipe &p,
                                           WeightStream &ws, F &&valu_work) {
    constexpr int s0 = (G / 2) % kDepth, s1 = (G / 2 + 1) % kDepth;
    acc0 = mfma16h(p.hi[s0], bhi, acc0);
    acc1 = mfma16h(p.hi[s1], bhi, acc1);
    acc0 = mfma16h(p.hi[s0], blo, acc0);
    acc1 = mfma16h(p.hi[s1], blo, acc1);
    acc0 = mfma16h(p.lo[s0], bhi, acc0);
    acc1 = mfma16h(p.lo[s1], bhi, acc1);
    if constexpr (WITH_VALU) {
        valu_work();
#pragma unroll
        for (int i = 0; i < 6; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);   // VALU
        }
    }
    if constexpr (G + 2 * kDepth < GEND) pipe_fetch<G + 2 * kDepth, GEND>(p, ws);
    if constexpr (G + 2 * kDepth + 2 < GEND) pipe_fetch<G + 2 * kDepth + 2, GEND>(p, ws);
    if constexpr (G + 4 < GEND) wait_lds<younger_reads<G + 4, imin(GEND, G + 2 * kDepth + 4)>()>();
    __builtin_amdgcn_sched_barrier(0);   // keep each step's MFMAs and its requests together, in program order
}
template <int G, int GEND>
__device__ __forceinline__ void dstep_h(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, APipe &p, WeightStream &ws) {
    dstep_impl<G, GEND, false>(acc0, acc1, bhi, blo, p, ws, [] {});
}
template <int G, int GEND, class F>
__device__ __forceinline__ void dstep_h_with(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, APipe &p,
                                             WeightStream &ws, F &&valu_work) {
    dstep_impl<G, GEND, true>(acc0, acc1, bhi, blo, p, ws, valu_work);
}

// all NT_OUT x NT_IN tiles (NT_OUT even), stream order [ot pair][it][s][o2] from G0
template <int NT_OUT, int NT_IN, int G0, int GEND>
__device__ __forceinline__ void chain_h(f32x16 (&acc)[NT_OUT], const HTile (&in)[NT_IN], APipe &p, WeightStream &ws) {
    static_for<NT_OUT / 2>([&](auto otp) {
        static_for<NT_IN * 2>([&](auto i) {
            dstep_h<G0 + (otp.value * NT_IN * 2 + i.value) * 4, GEND>(acc[2 * otp.value], acc[2 * otp.value + 1],
                                                                      in[i.value / 2].hi[i.value % 2],
                                                                      in[i.value / 2].lo[i.value % 2], p, ws);
        });
    });
}

// Broadcast biases: groups 0..1 of the stream hold floats [tile < 16][h][16] = bias[acc_row(tile, r, h)]
// (valid while chunk 0 is resident, i.e. before the first pipe_fetch of chunk 2)
__device__ __forceinline__ void bias_tile_h(int tile, f32x16 &acc, int h, const WeightStream &ws) {
    const float4 *p = reinterpret_cast<const float4 *>(ws.group_ptr(0) + tile * 32 + h * 16);
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
        const float4 v = p[r4];
        acc[4 * r4 + 0] = v.x; acc[4 * r4 + 1] = v.y; acc[4 * r4 + 2] = v.z; acc[4 * r4 + 3] = v.w;
    }
}

// ---------------------------------------------------------------- packers
// Weight pairs: dst halfs
//   [(((otp*n_in + it)*2 + s)*2 + o2)*2 + part][lane][e] =
//        part(V[32(row_tile0 + 2otp + o2) + (lane&31)][col0 + 32it + perm(8s+e, lane>>5)])
// (output tiles go in PAIRS, the pair innermost: see dstep_impl; nt_out must be even)
// with perm(r, g) = (r&3) + 8(r>>2) + 4g, part 0 = f16(v), part 1 = f16(v - f16(v)), and
//   V[row][col] = W[row][col] for col < ld;  bias[row] for col == ld (if bias);  0 beyond
// (col == ld is the slot of the constant-1 input that follows the layer's real inputs).
static __global__ __launch_bounds__(256) void k_pack_chain_h(const float *__restrict__ W, uint32_t ld, uint32_t col0,
                                                             uint32_t row_tile0, uint32_t nt_out, uint32_t nt_in,
                                                             const float *__restrict__ bias, _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = nt_out * nt_in * 4u * 512u;
    if (i >= total) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, o2 = (grp >> 1) & 1u, s = (grp >> 2) & 1u, it = (grp >> 3) % nt_in;
    const uint32_t ot = 2u * ((grp >> 3) / nt_in) + o2;
    const uint32_t r = 8u * s + e;
    const uint32_t row = 32u * (row_tile0 + ot) + (lane & 31u);
    const uint32_t col = col0 + 32u * it + (r & 3u) + 8u * (r >> 2) + 4u * (lane >> 5);
    const float w = col < ld ? W[(size_t)row * ld + col] : (col == ld && bias ? bias[row] : 0.0f);
    const _Float16 h = (_Float16)w;
    dst[i] = part == 0u ? h : (_Float16)(w - (float)h);
}
// First layer (inputs in natural order k = 16s + 8g + e, zero-padded to KS k-steps):
//   dst[((s*2 + ot)*2 + part)][lane][e] = part(W[32ot + (lane&31)][16s + 8(lane>>5) + e])
static __global__ __launch_bounds__(256) void k_pack_first_h(const float *__restrict__ W, uint32_t F, uint32_t KS,
                                                             _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 2u * KS * 2u * 512u) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, ot = (grp >> 1) & 1u, s = grp >> 2;
    const uint32_t k = 16u * s + 8u * (lane >> 5) + e;
    const float w = k < F ? W[(size_t)(32u * ot + (lane & 31u)) * F + k] : 0.0f;
    const _Float16 h = (_Float16)w;
    dst[i] = part == 0u ? h : (_Float16)(w - (float)h);
}
// Bias tiles: dst[t*32 + h*16 + r] = bias[acc_row(t, r, h)], t < ntiles
static __global__ __launch_bounds__(256) void k_pack_bias_h(const float *__restrict__ bias, uint32_t ntiles,
                                                            float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= ntiles * 32u) return;
    const uint32_t r = i & 15u, h = (i >> 4) & 1u, t = i >> 5;
    dst[i] = bias[acc_row(t, r, h)];
}
