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
// 16 k's of a step are the rows (r&3)+8(r>>2)+4g -- a fixed permutation that ucn_field_pack applies to
// the weight columns.  Stream: 4 groups of 1 KiB per (out tile, in tile): [s0 hi][s0 lo][s1 hi][s1 lo].
#pragma once
#include "mfma_chain.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct HTile {   // one 32-neuron activation tile as B operands: [k-step s]
    h8 hi[2], lo[2];
};

__device__ __forceinline__ f32x16 mfma16h(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split_tile(const f32x16 &a, HTile &t) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = a[8 * s + e];
            const _Float16 h = (_Float16)v;
            t.hi[s][e] = h;
            t.lo[s][e] = (_Float16)(v - (float)h);
        }
}

__device__ __forceinline__ h8 group_h(const WeightStream &ws, int g) {
    const float4 v = ws.group(g);
    return __builtin_bit_cast(h8, v);
}

// acc += W . in for one output tile; groups [it][s][hi,lo] start at stream position G0
// (the A operands of step i+1 are read from LDS before the MFMAs of step i are issued: with one wave
//  per SIMD nothing else hides the ~128-cycle ds_read_b128 latency behind the 96 MFMA cycles of a step)
template <int NT_IN>
__device__ __forceinline__ void chain_one_h(const int G0, f32x16 &acc, const HTile (&in)[NT_IN], WeightStream &ws) {
    if (G0 % kChunkGroups == 0) ws.boundary(G0 / kChunkGroups);
    h8 a_hi = group_h(ws, G0), a_lo = group_h(ws, G0 + 1);
#pragma unroll
    for (int i = 0; i < NT_IN * 2; i++) {
        h8 n_hi = a_hi, n_lo = a_lo;
        if (i + 1 < NT_IN * 2) {
            const int g = G0 + 2 * (i + 1);
            if (g % kChunkGroups == 0) ws.boundary(g / kChunkGroups);
            n_hi = group_h(ws, g);
            n_lo = group_h(ws, g + 1);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this step's MFMAs
        const int it = i / 2, s = i % 2;
        acc = mfma16h(a_hi, in[it].hi[s], acc);
        acc = mfma16h(a_hi, in[it].lo[s], acc);
        acc = mfma16h(a_lo, in[it].hi[s], acc);
        a_hi = n_hi;
        a_lo = n_lo;
    }
}
// NT_OUT output tiles from one input tile; groups [ot][s][hi,lo]
template <int NT_OUT>
__device__ __forceinline__ void chain_from_one_h(const int G0, f32x16 (&acc)[NT_OUT], const HTile &in, WeightStream &ws) {
    if (G0 % kChunkGroups == 0) ws.boundary(G0 / kChunkGroups);
    h8 a_hi = group_h(ws, G0), a_lo = group_h(ws, G0 + 1);
#pragma unroll
    for (int i = 0; i < NT_OUT * 2; i++) {
        h8 n_hi = a_hi, n_lo = a_lo;
        if (i + 1 < NT_OUT * 2) {
            const int g = G0 + 2 * (i + 1);
            if (g % kChunkGroups == 0) ws.boundary(g / kChunkGroups);
            n_hi = group_h(ws, g);
            n_lo = group_h(ws, g + 1);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this step's MFMAs
        const int ot = i / 2, s = i % 2;
        acc[ot] = mfma16h(a_hi, in.hi[s], acc[ot]);
        acc[ot] = mfma16h(a_hi, in.lo[s], acc[ot]);
        acc[ot] = mfma16h(a_lo, in.hi[s], acc[ot]);
        a_hi = n_hi;
        a_lo = n_lo;
    }
}
template <int OT, int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain_rec_h(const int G0, f32x16 (&acc)[NT_OUT], const HTile (&in)[NT_IN], WeightStream &ws) {
    chain_one_h<NT_IN>(G0 + OT * NT_IN * 4, acc[OT], in, ws);
    if constexpr (OT + 1 < NT_OUT) chain_rec_h<OT + 1, NT_OUT, NT_IN>(G0, acc, in, ws);
}
template <int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain_h(const int G0, f32x16 (&acc)[NT_OUT], const HTile (&in)[NT_IN], WeightStream &ws) {
    chain_rec_h<0, NT_OUT, NT_IN>(G0, acc, in, ws);
}

// Weight packing for the split engine: dst halfs
//   [((ot*n_in + it)*2 + s)*2 + part][lane][e] = part(W[32(row_tile0+ot)+(lane&31)][col0 + 32it + perm(8s+e, lane>>5)])
// with perm(r, g) = (r&3) + 8(r>>2) + 4g, part 0 = f16(w), part 1 = f16(w - f16(w)).
static __global__ __launch_bounds__(256) void k_pack_chain_h(const float *__restrict__ W, uint32_t ld, uint32_t col0,
                                                             uint32_t row_tile0, uint32_t nt_out, uint32_t nt_in,
                                                             _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = nt_out * nt_in * 4u * 512u;
    if (i >= total) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, s = (grp >> 1) & 1u, it = (grp >> 2) % nt_in, ot = (grp >> 2) / nt_in;
    const uint32_t r = 8u * s + e;
    const uint32_t row = 32u * (row_tile0 + ot) + (lane & 31u);
    const uint32_t col = col0 + 32u * it + (r & 3u) + 8u * (r >> 2) + 4u * (lane >> 5);
    const float w = W[(size_t)row * ld + col];
    const _Float16 h = (_Float16)w;
    dst[i] = part == 0u ? h : (_Float16)(w - (float)h);
}
// First layer (inputs in natural order k = 16s + 8g + e, zero-padded to a multiple of 16):
//   dst[((ot*KS + s)*2 + part)][lane][e] = part(W[32ot + (lane&31)][16s + 8(lane>>5) + e])
static __global__ __launch_bounds__(256) void k_pack_first_h(const float *__restrict__ W, uint32_t F, uint32_t KS,
                                                             _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 2u * KS * 2u * 512u) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, s = (grp >> 1) % KS, ot = (grp >> 1) / KS;
    const uint32_t k = 16u * s + 8u * (lane >> 5) + e;
    const float w = k < F ? W[(size_t)(32u * ot + (lane & 31u)) * F + k] : 0.0f;
    const _Float16 h = (_Float16)w;
    dst[i] = part == 0u ? h : (_Float16)(w - (float)h);
}
