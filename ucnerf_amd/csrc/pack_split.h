// Packers of the split-f16 weight stream shared by the MLP kernels that run on mlp_ring.h (sky.hip; field_mlp_h.hip
// has scaled variants of its own): fragment order of v_mfma_f32_32x32x16_f16 A operands, [hi][lo] pairs of 1 KiB groups.
//
// Layouts.  C/D of 32x32x16 is the same as 32x32x2 (lane = column + 32*half g, register r = row
// (r&3)+8(r>>2)+4g).  A/B hold 8 consecutive-k halfs per lane, k = 8g+e.  A 32-row activation tile is
// consumed in two k-steps s = 0,1; step s takes accumulator registers r = 8s..8s+7 of each lane, so the
// 16 k's of a step are the rows (r&3)+8(r>>2)+4g -- a fixed permutation that the packers apply to
// the weight columns.  Stream: a PAIR of 1 KiB groups [hi][lo] per (out tile, in tile, s).
#pragma once
#include "mfma_chain.h"

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
