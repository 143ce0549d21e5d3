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
// With one wave per SIMD nothing but the wave's own instruction stream hides latency, so the steady
// state has NO global loads (biases ride in the weight stream: either as broadcast bias tiles at its
// head or as the weight column that multiplies a constant-1 input) and the A operands are requested
// from LDS kDepth steps ahead of their MFMAs (APipe), across layer boundaries as well.
#pragma once
#include "mfma_chain.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct HTile {   // one 32-neuron activation tile as B operands: [k-step s]
    h8 hi[2], lo[2];
};
constexpr int kDepth = 6;   // A-operand pairs in flight: pair i + kDepth is requested right after step i issues
struct APipe {              // ring of A operands, slot = (pair index) % kDepth
    h8 hi[kDepth], lo[kDepth];
};

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

__device__ __forceinline__ h8 group_h(const WeightStream &ws, int g) {
    const float4 v = ws.group(g);
    return __builtin_bit_cast(h8, v);
}
// request pair g (even group index) into its ring slot.  The DMA of the NEXT chunk is spread over this
// chunk's requests, one piece per 4 groups: issued as a burst of 16 behind the barrier, the pieces cost
// the issuing (= computing) wave 100-185 cycles each while the address queue is full of them.
// (piece 0 of chunk 1 is the kernel's job: group 0 holds the biases and is never requested here)
// GEND = end of the pipelined segment; the stream carries kTailGroups more groups behind it, so chunk
// c exists iff c * kChunkGroups < GEND + kTailGroups -- a compile-time fact at every call site.
constexpr int kTailGroups = 4;
__device__ __forceinline__ void pipe_fetch(const int g, const int GEND, APipe &p, WeightStream &ws) {
    if (g % kChunkGroups == 0) ws.sync();
    if (g % 4 == 0 && (g / kChunkGroups + 1) * kChunkGroups < GEND + kTailGroups)
        ws.piece_unchecked(g / kChunkGroups + 1, (g % kChunkGroups) / 4);
    p.hi[(g / 2) % kDepth] = group_h(ws, g);
    p.lo[(g / 2) % kDepth] = group_h(ws, g + 1);
}
// start of a pipelined segment [G0, GEND)
__device__ __forceinline__ void pipe_prime(const int G0, const int GEND, APipe &p, WeightStream &ws) {
#pragma unroll
    for (int d = 0; d < kDepth; d++)
        if (G0 + 2 * d < GEND) pipe_fetch(G0 + 2 * d, GEND, p, ws);
}
// One double step: two OUTPUT tiles (A pairs g and g+2) against the same 16 k's of the input,
//   acc0 += A(g) . B,  acc1 += A(g+2) . B,
// six MFMAs alternating between the two accumulators, then the two pairs kDepth ahead are requested
// into the slots just consumed.  The alternation is the point: an MFMA that accumulates into the
// register block the previous MFMA is still writing does not issue back to back -- a chain of 54
// dependent v_mfma_f32_32x32x16_f16 with ds_reads/waitcnts between them measured 67 cycles per MFMA
// instead of 32 (MI355X_MICROARCH.md: "+43 cyc for the first extra state between two MFMAs on the
// SAME accumulator").  With two accumulators every MFMA's predecessor on its own block is 64 cycles old.
__device__ __forceinline__ void dstep_h(const int g, const int GEND, f32x16 &acc0, f32x16 &acc1, const h8 bhi,
                                        const h8 blo, APipe &p, WeightStream &ws) {
    const int s0 = (g / 2) % kDepth, s1 = (g / 2 + 1) % kDepth;
    acc0 = mfma16h(p.hi[s0], bhi, acc0);
    acc1 = mfma16h(p.hi[s1], bhi, acc1);
    acc0 = mfma16h(p.hi[s0], blo, acc0);
    acc1 = mfma16h(p.hi[s1], blo, acc1);
    acc0 = mfma16h(p.lo[s0], bhi, acc0);
    acc1 = mfma16h(p.lo[s1], bhi, acc1);
    if (g + 2 * kDepth < GEND) pipe_fetch(g + 2 * kDepth, GEND, p, ws);
    if (g + 2 * kDepth + 2 < GEND) pipe_fetch(g + 2 * kDepth + 2, GEND, p, ws);
    // one wait for the NEXT step's four operands (the oldest of the 2*kDepth reads now in flight) instead
    // of the compiler's one-per-first-use: every instruction between two MFMAs costs issue time here
    __builtin_amdgcn_s_waitcnt(0xC07F | ((2 * kDepth - 4) << 8));
    __builtin_amdgcn_sched_barrier(0);   // keep each step's MFMAs and its requests together, in program order
}

// Two output tiles from NT_IN input tiles; pairs [it][s][o2] start at stream position G0.  Loops stay
// <= 32 iterations so that hipcc unrolls them completely (see mfma_chain.h).
template <int NT_IN>
__device__ __forceinline__ void chain_two_h(const int G0, const int GEND, f32x16 &acc0, f32x16 &acc1,
                                            const HTile (&in)[NT_IN], APipe &p, WeightStream &ws) {
#pragma unroll
    for (int i = 0; i < NT_IN * 2; i++) dstep_h(G0 + 4 * i, GEND, acc0, acc1, in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], p, ws);
}
template <int OTP, int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain_rec_h(const int G0, const int GEND, f32x16 (&acc)[NT_OUT], const HTile (&in)[NT_IN],
                                            APipe &p, WeightStream &ws) {
    chain_two_h<NT_IN>(G0 + OTP * NT_IN * 8, GEND, acc[2 * OTP], acc[2 * OTP + 1], in, p, ws);
    if constexpr (2 * OTP + 2 < NT_OUT) chain_rec_h<OTP + 1, NT_OUT, NT_IN>(G0, GEND, acc, in, p, ws);
}
// all NT_OUT x NT_IN tiles (NT_OUT even), order [ot pair][it][s][o2]
template <int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain_h(const int G0, const int GEND, f32x16 (&acc)[NT_OUT], const HTile (&in)[NT_IN],
                                        APipe &p, WeightStream &ws) {
    chain_rec_h<0, NT_OUT, NT_IN>(G0, GEND, acc, in, p, ws);
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
// (output tiles go in PAIRS, the pair innermost: see dstep_h; nt_out must be even)
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
