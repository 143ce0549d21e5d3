// Fused sample featurisation: cone cast -> contraction -> hash-grid gather -> erf damping -> mean of 6.
//
// Replaces, for one sampling level, the chain
//   render.cast_rays            (/root/reference/nerf/internal/render.py:94-152)
//   coord.track_linearize       (coord.py:60-116, 'contract')
//   GridEncoder.forward         (gridencoder/grid.py:158-174 -> gridencoder.cu:87-199)
//   erf down-weighting + mean   (models.py:494-496)
// The reference materialises [N*S*6,3] points, the [L,N*S*6,C] gather result, its permuted copy
// and the erf weights in HBM (~1.5 GB per 15000-ray chunk at 128 samples); here the six
// multisamples of a sample live in registers and only the [L][N*S][C] mean feature is written.
//
// Mapping (CDNA4): one thread = one sample x `levels_per_block` consecutive levels; blockIdx.y is
// the level group, so the grid is level-major in dispatch order and an XCD's L2 (4 MiB) sees one
// 4 MiB hashed level slice at a time (measured: levels_per_block 1 -> 16 costs 1.8x); lanes of a wave
// are consecutive samples of a ray -> the [L][B][C] store is a contiguous 64*C*4-byte run per wave.
//
// What bounds it (rocprofv3, r01b): with the tables L2/MALL-resident the kernel is VALU-bound
// (SQ_ACTIVE_INST_VALU = 21 % of wave-cycles at 4 waves/SIMD = 84 % of a SIMD), ~3000 VALU
// instructions per (sample, level).  Hence this file's shape:
//   * the level's addressing mode (xor-hash vs strided, pow2 mask vs modulo) is a TEMPLATE argument
//     chosen by a wave-uniform branch, not a per-corner select between two computed indices;
//   * y*P1 and z*P2 are multiplied once per point, the +1 corners add the prime (uint32 wrap keeps
//     (y+1)*P == y*P + P);
//   * quantities that only feed the erf damping (std) use fast reciprocals / exp2-log2 instead of
//     IEEE division and powf, and erf itself is the Abramowitz-Stegun 7.1.26 form (|err| <= 1.5e-7);
//     everything that feeds a COORDINATE keeps the reference's exact fp32 op sequence
//     (correctly-rounded div/sqrt, no contraction), so positions -- and the interpolated features
//     for given positions -- stay bit-identical to the oracle.
#include "ucn_common.h"
#include "wave_dpp.h"
#include <type_traits>

namespace {

struct HexPattern {
    float cs[2][6];   // cos of the deterministic angles for even / odd samples (render.py:126-131)
    float sn[2][6];
    float ang[6];     // pi/3 * [0,2,4,3,5,1]   (render.py:119)
    float cj[6];      // 3/sqrt(7) * (2j/5 - 1)  (render.py:116)
};

struct RayInputs {
    const float *sdist, *near_, *far_, *origins, *dirs, *basis, *radii, *flip, *spin;
};

constexpr uint32_t kP1 = 2654435761u, kP2 = 805459861u;   // gridencoder.cu:54

// erf(x), x >= 0: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (the damping multiplies O(1) features)
__device__ __forceinline__ float erf_pos(float x) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return fmaf(-(p * t), __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x), 1.0f);
}

// One multisample point in one level: lattice cell, fractions, the 8 corner rows.
// gridencoder.cu:146-159 (locate) and :66-84 (index) for D = 3, linear, align_corners = false.
template <bool HASHED, bool POW2>
__device__ __forceinline__ void corner_rows(const UcnLevel &lv, float px, float py, float pz, float &fx, float &fy,
                                            float &fz, uint32_t (&rows)[8]) {
    fx = fmaf(px, lv.scale, 0.5f); fy = fmaf(py, lv.scale, 0.5f); fz = fmaf(pz, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    uint32_t ya, yb, za, zb, xa, xb;
    if constexpr (HASHED) {
        xa = x0; xb = x0 + 1u;
        ya = y0 * kP1; yb = ya + kP1;
        za = z0 * kP2; zb = za + kP2;
    } else {
        xa = x0 * lv.stride[0]; xb = xa + lv.stride[0];
        ya = y0 * lv.stride[1]; yb = ya + lv.stride[1];
        za = z0 * lv.stride[2]; zb = za + lv.stride[2];
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t xv = (k & 1u) ? xb : xa, yv = (k & 2u) ? yb : ya, zv = (k & 4u) ? zb : za;
        uint32_t idx;
        if constexpr (HASHED) idx = xv ^ yv ^ zv;
        else idx = xv + yv + zv;
        if constexpr (POW2) rows[k] = idx & lv.mask;
        else rows[k] = idx < lv.rows ? idx : idx % lv.rows;
    }
}

// w_k = ((1*wx)*wy)*wz in the reference's multiplication order (gridencoder.cu:168-180)
__device__ __forceinline__ void corner_weights(float fx, float fy, float fz, float (&w)[8]) {
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;
    w[0] = w00 * gz; w[1] = w10 * gz; w[2] = w01 * gz; w[3] = w11 * gz;
    w[4] = w00 * fz; w[5] = w10 * fz; w[6] = w01 * fz; w[7] = w11 * fz;
}

__device__ __forceinline__ bool in_unit_cube(float px, float py, float pz) {
    return !(px < 0.0f || px > 1.0f || py < 0.0f || py > 1.0f || pz < 0.0f || pz > 1.0f);   // gridencoder.cu:110-135
}

// One table row -> C floats.  TT = float (the fp32 tables of rendering and of the fp32 training path) or _Float16: under
// autocast the reference gathers a HALF copy of the table (grid.py:41-44: `embeddings.to(torch.half)` whenever autocast is on
// and C is even) -- half the bytes per corner, and a 2 MiB level slice that fits an XCD's L2 beside the streaming traffic.
// The interpolation arithmetic stays fp32 here (the reference's is half: this side is the more exact one).
template <uint32_t C, typename TT>
__device__ __forceinline__ void load_row(const TT *__restrict__ tab, uint32_t row, float (&v)[C]) {
    const TT *r = tab + (size_t)row * C;
    if constexpr (sizeof(TT) == 4) {
        if constexpr (C == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(r);
            v[0] = t.x; v[1] = t.y;
        } else if constexpr (C == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(r);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) v[c] = (float)r[c];
        }
    } else {
        typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
        typedef _Float16 hx4 __attribute__((ext_vector_type(4)));
        if constexpr (C == 2) {
            const hx2 t = *reinterpret_cast<const hx2 *>(r);
            v[0] = (float)t[0]; v[1] = (float)t[1];
        } else if constexpr (C == 4) {
            const hx4 t = *reinterpret_cast<const hx4 *>(r);
            v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) v[c] = (float)r[c];
        }
    }
}
// rows r and r ^ 1 (an aligned pair, C = 2) in one request: out[0..1] = row (r & ~1), out[2..3] = row (r | 1)
template <typename TT>
__device__ __forceinline__ void load_row_pair(const TT *__restrict__ tab, uint32_t row_even, float (&o)[4]) {
    if constexpr (sizeof(TT) == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(tab + (size_t)row_even * 2);
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
        typedef _Float16 hx4 __attribute__((ext_vector_type(4)));
        const hx4 t = *reinterpret_cast<const hx4 *>(tab + (size_t)row_even * 2);
        o[0] = (float)t[0]; o[1] = (float)t[1]; o[2] = (float)t[2]; o[3] = (float)t[3];
    }
}

// sum_j damp_j * trilerp(point_j) for one level
template <uint32_t C, bool HASHED, bool POW2, typename TT>
__device__ __forceinline__ void level_accumulate(const UcnLevel &lv, const TT *__restrict__ tab,
                                                 const float (&u)[6][3], const float (&rs)[6], uint32_t G,
                                                 float (&acc)[C]) {
#pragma unroll
    for (uint32_t c = 0; c < C; c++) acc[c] = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (j < G && in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            float fx, fy, fz, w[8];
            uint32_t rows[8];
            corner_rows<HASHED, POW2>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows);
            float v[8][C];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) load_row<C, TT>(tab, rows[k], v[k]);
            corner_weights(fx, fy, fz, w);
            float f[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) f[c] = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
#pragma unroll
                for (uint32_t c = 0; c < C; c++) f[c] = fmaf(w[k], v[k][c], f[c]);
            // models.py:495: erf(1 / sqrt(8 std^2 gs^2)) = erf(rs_j * lv.inv_gs), gs^2 in wrapped int32
            const float damp = erf_pos(rs[j] * lv.inv_gs);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) acc[c] += f[c] * damp;
        }
    }
}

// Coarse levels (cells far larger than a sample's cone section): the six multisamples of a sample usually sit in ONE
// lattice cell.  Then its 8 corner rows are derived and fetched once instead of six times; every point still forms its
// own weights and accumulates in level_accumulate's order, so the result is bit-identical.  Lanes whose points straddle a
// cell boundary take the general path (the branch diverges; the caller enables this only where straddling is rare).
template <uint32_t C, bool HASHED, bool POW2, typename TT>
__device__ __forceinline__ void level_accumulate_shared(const UcnLevel &lv, const TT *__restrict__ tab,
                                                        const float (&u)[6][3], const float (&rs)[6], float (&acc)[C]) {
    float fx[6], fy[6], fz[6];
    uint32_t x0 = 0, y0 = 0, z0 = 0;
    bool same = true, any = false;
    uint32_t inside = 0;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            fx[j] = fmaf(u[j][0], lv.scale, 0.5f); fy[j] = fmaf(u[j][1], lv.scale, 0.5f); fz[j] = fmaf(u[j][2], lv.scale, 0.5f);
            const uint32_t xj = (uint32_t)floorf(fx[j]), yj = (uint32_t)floorf(fy[j]), zj = (uint32_t)floorf(fz[j]);
            fx[j] -= (float)xj; fy[j] -= (float)yj; fz[j] -= (float)zj;
            if (!any) { x0 = xj; y0 = yj; z0 = zj; any = true; }
            else same = same && xj == x0 && yj == y0 && zj == z0;
            inside |= 1u << j;
        }
    }
    if (!same) {
        level_accumulate<C, HASHED, POW2, TT>(lv, tab, u, rs, 6, acc);
        return;
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++) acc[c] = 0.0f;
    if (!any) return;
    uint32_t ya, yb, za, zb, xa, xb;
    if constexpr (HASHED) {
        xa = x0; xb = x0 + 1u;
        ya = y0 * kP1; yb = ya + kP1;
        za = z0 * kP2; zb = za + kP2;
    } else {
        xa = x0 * lv.stride[0]; xb = xa + lv.stride[0];
        ya = y0 * lv.stride[1]; yb = ya + lv.stride[1];
        za = z0 * lv.stride[2]; zb = za + lv.stride[2];
    }
    float v[8][C];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t xv = (k & 1u) ? xb : xa, yv = (k & 2u) ? yb : ya, zv = (k & 4u) ? zb : za;
        uint32_t idx;
        if constexpr (HASHED) idx = xv ^ yv ^ zv;
        else idx = xv + yv + zv;
        uint32_t row;
        if constexpr (POW2) row = idx & lv.mask;
        else row = idx < lv.rows ? idx : idx % lv.rows;
#ifdef UCN_EXP_NO_COARSE_LOADS            // experiment build (r04): the coarse levels WITHOUT their table reads -- the ceiling of any LDS staging
        (void)row;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) v[k][c] = __builtin_bit_cast(float, 0x3f000000u + k * 977u + c);
#else
        load_row<C, TT>(tab, row, v[k]);
#endif
    }
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (inside & (1u << j)) {
            float w[8];
            corner_weights(fx[j], fy[j], fz[j], w);
            float f[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) f[c] = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++)
#pragma unroll
                for (uint32_t c = 0; c < C; c++) f[c] = fmaf(w[k], v[k][c], f[c]);
            const float damp = erf_pos(rs[j] * lv.inv_gs);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) acc[c] += f[c] * damp;
        }
    }
}

// Fine hashed levels, C = 2, power-of-two table: the kernel is bound by the L2 request rate there (one
// request per gathered corner, ~16 per clock per XCD), so corners that are adjacent in memory are fetched
// together.  For an even lattice x the corners (x, y, z) and (x+1, y, z) hash to rows r and r^1 -- one
// aligned 16-byte pair; for an odd x they are unrelated and cost two requests.  6 requests per point on
// average instead of 8.  Same values, same fmaf order as level_accumulate.
template <typename TT>
__device__ __forceinline__ void level_accumulate_pairs(const UcnLevel &lv, const TT *__restrict__ tab,
                                                       const float (&u)[6][3], const float (&rs)[6], float (&acc)[2]) {
    acc[0] = acc[1] = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            float fx, fy, fz, w[8];
            uint32_t rows[8];
            corner_rows<true, true>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows);
            const bool even = ((rows[0] ^ rows[1]) == 1u);          // x0 even <=> the two rows differ in bit 0 only
            float v[8][2];
            if (even) {                                             // one divergent branch per point
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {                  // (y, z) choice; corners 2q (x0) and 2q+1 (x0+1)
                    const uint32_t r0 = rows[2 * q];
                    float t[4];
                    load_row_pair<TT>(tab, r0 & ~1u, t);
                    const bool hi = (r0 & 1u) != 0u;
                    v[2 * q][0] = hi ? t[2] : t[0]; v[2 * q][1] = hi ? t[3] : t[1];
                    v[2 * q + 1][0] = hi ? t[0] : t[2]; v[2 * q + 1][1] = hi ? t[1] : t[3];
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) load_row<2, TT>(tab, rows[k], v[k]);
            }
            corner_weights(fx, fy, fz, w);
            float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                f0 = fmaf(w[k], v[k][0], f0);
                f1 = fmaf(w[k], v[k][1], f1);
            }
            const float damp = erf_pos(rs[j] * lv.inv_gs);
            acc[0] += f0 * damp;
            acc[1] += f1 * damp;
        }
    }
}

// LANE-PAIRED fetch (r04), C = 2, power-of-two tables.  The corners (x0, y, z) and (x0 + 1, y, z) of a point hash to rows r and
// r ^ d (hashed levels; r, r + 1 on the strided ones): in 7 cases of 8 (x0 & 7 != 7) the same aligned group of eight 8-byte rows,
// i.e. the SAME 64-byte line -- but up to 56 bytes apart, so no single load of one lane covers both, and as two load
// instructions they are two line requests (the bound of the fine levels: L1 lines per clock).  The texture-address path DOES
// merge lanes of ONE instruction that hit the same line, wherever they sit in the wave (tools/ta_merge_bench.hip: two 8-byte
// loads per point 349 G rows/s; the same rows as ONE instruction over lane pairs 532 G rows/s -- the rate of an aligned 16-byte
// load).  So a wave fetches a (y, z) combination in two instructions that each serve 32 POINTS: lane i < 32 asks for the x0 row of
// point i while lane i + 32 asks for the x0 + 1 row of the same point (second instruction: the points of lanes 32 ... 63).  One
// v_permlane32_swap of (row_x0, row_x1) forms both address registers, one per channel sorts the values back:
//   swap(a, b) -> {a.lo, b.lo}, {a.hi, b.hi}.  4.5 line requests per point instead of 6 (pair fetch) or 8, no divergent branch.
// Points outside the unit cube fetch (masked, hence valid) dummy rows and are skipped at the accumulation: same values, same
// fmaf order as level_accumulate.  Needs every lane of the wave active (the caller checks).
template <bool HASHED, typename TT>
__device__ __forceinline__ void level_accumulate_lanepairs(const UcnLevel &lv, const TT *__restrict__ tab,
                                                           const float (&u)[6][3], const float (&rs)[6], float (&acc)[2]) {
    acc[0] = acc[1] = 0.0f;
#ifndef UCN_LANEPAIR_DEPTH
#define UCN_LANEPAIR_DEPTH 1
#endif
    constexpr uint32_t DEPTH = UCN_LANEPAIR_DEPTH;                  // points whose 8 loads are in flight together (experiment knob)
#pragma unroll
    for (uint32_t j0 = 0; j0 < 6; j0 += DEPTH) {
        float fx[DEPTH], fy[DEPTH], fz[DEPTH];
        uint32_t adr[DEPTH][4][2];
        bool valid[DEPTH];
#pragma unroll
        for (uint32_t d = 0; d < DEPTH; d++) {
            const uint32_t j = j0 + d;
            valid[d] = in_unit_cube(u[j][0], u[j][1], u[j][2]);
            uint32_t rows[8];
            corner_rows<HASHED, true>(lv, u[j][0], u[j][1], u[j][2], fx[d], fy[d], fz[d], rows);
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {                          // (y, z) choice; corners 2q (x0) and 2q + 1 (x0 + 1)
                const auto ad = __builtin_amdgcn_permlane32_swap(rows[2 * q], rows[2 * q + 1], false, false);
                adr[d][q][0] = ad[0]; adr[d][q][1] = ad[1];
            }
        }
        // every load of the group is issued before the first value is used
        uint32_t raw[DEPTH][4][2][sizeof(TT) == 4 ? 2 : 1];
#pragma unroll
        for (uint32_t d = 0; d < DEPTH; d++)
#pragma unroll
            for (uint32_t q = 0; q < 4; q++)
#pragma unroll
                for (uint32_t e = 0; e < 2; e++) {
                    if constexpr (sizeof(TT) == 4) {
                        const uint2 t = *reinterpret_cast<const uint2 *>(tab + (size_t)adr[d][q][e] * 2);
                        raw[d][q][e][0] = t.x; raw[d][q][e][1] = t.y;
                    } else {
                        raw[d][q][e][0] = *reinterpret_cast<const uint32_t *>(tab + (size_t)adr[d][q][e] * 2);   // a row = two halves = one word
                    }
                }
#pragma unroll
        for (uint32_t d = 0; d < DEPTH; d++) {
            float v[8][2], w[8];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                if constexpr (sizeof(TT) == 4) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(raw[d][q][0][0], raw[d][q][1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(raw[d][q][0][1], raw[d][q][1][1], false, false);
                    v[2 * q][0] = __builtin_bit_cast(float, (uint32_t)s0[0]); v[2 * q + 1][0] = __builtin_bit_cast(float, (uint32_t)s0[1]);
                    v[2 * q][1] = __builtin_bit_cast(float, (uint32_t)s1[0]); v[2 * q + 1][1] = __builtin_bit_cast(float, (uint32_t)s1[1]);
                } else {
                    typedef _Float16 hx2 __attribute__((ext_vector_type(2)));
                    const auto s0 = __builtin_amdgcn_permlane32_swap(raw[d][q][0][0], raw[d][q][1][0], false, false);
                    const hx2 h0 = __builtin_bit_cast(hx2, (uint32_t)s0[0]), h1 = __builtin_bit_cast(hx2, (uint32_t)s0[1]);
                    v[2 * q][0] = (float)h0[0]; v[2 * q][1] = (float)h0[1];
                    v[2 * q + 1][0] = (float)h1[0]; v[2 * q + 1][1] = (float)h1[1];
                }
            }
            corner_weights(fx[d], fy[d], fz[d], w);
            float f0 = 0.0f, f1 = 0.0f;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                f0 = fmaf(w[k], v[k][0], f0);
                f1 = fmaf(w[k], v[k][1], f1);
            }
            const float damp = erf_pos(rs[j0 + d] * lv.inv_gs);
            if (valid[d]) {
                acc[0] += f0 * damp;
                acc[1] += f1 * damp;
            }
        }
    }
}

// Backward of level_accumulate w.r.t. the table: grad_table[row_k] += w_k * damp_j * g
// (gridencoder.cu:304-339 composed with models.py:495-496).  When all multisamples of the sample share
// one lattice cell (coarse levels) their corner weights are summed first: 8*C atomics instead of 48*C
// -- atomics, unlike loads, do not coalesce across lanes.
template <uint32_t C, bool HASHED, bool POW2>
__device__ __forceinline__ void level_scatter(const UcnLevel &lv, float *__restrict__ gtab, const float (&u)[6][3],
                                              const float (&rs)[6], uint32_t G, const float (&gout)[C]) {
    uint32_t rows0[8];
    float wsum[8];
    bool have0 = false, shared = true;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { wsum[k] = 0.0f; rows0[k] = 0u; }
    // pass 1: is it one cell?  (compare the corner-0 row and the integer cell through rows[0], rows[7])
    float fxs[6], fys[6], fzs[6];
    uint32_t r0s[6], r7s[6];
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        fxs[j] = fys[j] = fzs[j] = 0.0f; r0s[j] = r7s[j] = 0u;
        if (j < G) {
            if (!in_unit_cube(u[j][0], u[j][1], u[j][2])) { shared = false; continue; }
            uint32_t rows[8];
            corner_rows<HASHED, POW2>(lv, u[j][0], u[j][1], u[j][2], fxs[j], fys[j], fzs[j], rows);
            r0s[j] = rows[0]; r7s[j] = rows[7];
            if (!have0) {
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) rows0[k] = rows[k];
                have0 = true;
            } else if (rows[0] != rows0[0] || rows[7] != rows0[7] || rows[1] != rows0[1] || rows[2] != rows0[2] ||
                       rows[4] != rows0[4]) {
                shared = false;
            }
        }
    }
    if (shared && have0) {
#pragma unroll
        for (uint32_t j = 0; j < 6; j++) {
            if (j < G) {
                float w[8];
                corner_weights(fxs[j], fys[j], fzs[j], w);
                const float damp = erf_pos(rs[j] * lv.inv_gs);
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) wsum[k] += w[k] * damp;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            float *r = gtab + (size_t)rows0[k] * C;
#pragma unroll
            for (uint32_t c = 0; c < C; c++) atomicAdd(r + c, wsum[k] * gout[c]);
        }
        return;
    }
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (j < G && in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            float fx, fy, fz, w[8];
            uint32_t rows[8];
            corner_rows<HASHED, POW2>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows);
            corner_weights(fx, fy, fz, w);
            const float damp = erf_pos(rs[j] * lv.inv_gs);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                float *r = gtab + (size_t)rows[k] * C;
#pragma unroll
                for (uint32_t c = 0; c < C; c++) atomicAdd(r + c, (w[k] * damp) * gout[c]);
            }
        }
    }
}

// acc[r][0..C) += v in the workgroup's LDS row block.  ds_add_f32 is serialised per LANE on gfx950 (measured with
// tools/lds_atomic_bench.hip: 193 clk per wave instruction, 3 clk per active lane, whatever the addresses -- 40x
// the integer ds_add_u32).  An 8-byte compare-and-swap updates two channels per ds_cmpst_rtn_b64 in 24-37 clk per
// wave when the lanes hit different rows, and is still an exact fp32 add per addend -- but a lane whose row was
// changed in between has to retry, and on the coarser levels neighbouring lanes DO share rows.  So:
//   CAS = false (coarse levels, run-merged updates): plain ds_add_f32, contention-proof;
//   CAS = true  (fine levels): one compare-and-swap attempt, the lanes that lose fall back to ds_add_f32.
// FIXED-POINT accumulation (the autocast training step; r04).  The row block may hold, instead of C floats per row, C / 2
// 64-bit words per row, each TWO int32 fixed-point channels: word = (b << 32) + sign_extend(a).  One fire-and-forget ds_add_u64
// per channel pair replaces the read + compare-and-swap round trip (53 % of this kernel's instruction waits were LDS,
// profiles/r03/pmc_table_train.txt): integer adds are associative, so the 64-bit total is exact mod 2^64 whatever the order, and
// it splits back uniquely into (a, b) as long as each channel's FINAL sum fits int32.  That is guaranteed, not hoped for: the
// addends of a sample are (w_k damp / 6) g_c with sum_k w_k = 1, damp <= 1, six points -- their absolute values sum to at most
// |g_c|, so any row's |sum| <= L1 = sum over the task's samples of |g| (accumulated by the mask pass); the task scales its
// gradients by the power of two that puts L1 at <= 2^30.  Rounding adds <= 1/2 per addend and a task has <= 48 B addends per row at
// the very worst (6 points x 8 corners of every one of its B <= 2^22 samples landing on ONE row): 2^30 + 24 x 2^22 < 2^30.1 < 2^31
// (the host entry asserts exactly this inequality).  A non-finite gradient among the task's samples poisons the task's WHOLE row
// block on that level with NaN (the float rows poison the touched rows only; after the reference's nan_to_num, train_utils.py:342,
// both lose that step's gradient for those rows -- a superset here, never a silently dropped NaN).  The
// accumulator type selects the mode: `float` rows (exact fp32 adds, the fp32 route and every test of it) or `FxLane` rows.
struct FxLane { float raw; };                                  // same size as float: row / channel pointer arithmetic is shared
template <typename A> constexpr bool kFixed = false;
template <> constexpr bool kFixed<FxLane> = true;
__device__ __forceinline__ unsigned long long fixed_pack(float a, float b) {
    const long long ia = (long long)__float2int_rn(a), ib = (long long)__float2int_rn(b);
    return (unsigned long long)((ib << 32) + ia);
}
__device__ __forceinline__ void fixed_unpack(unsigned long long w, float inv, float &a, float &b) {
    const int lo = (int)(uint32_t)w;
    const long long hi = ((long long)w - (long long)lo) >> 32;
    a = (float)lo * inv;
    b = (float)(int)hi * inv;
}
template <uint32_t C, bool CAS, typename A>
__device__ __forceinline__ void lds_row_add(A *acc_, uint32_t r, const float (&v)[C]) {
    if constexpr (kFixed<A>) {
        static_assert(C % 2u == 0u, "fixed-point rows pack channel pairs");
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) atomicAdd(reinterpret_cast<unsigned long long *>(acc_ + r * C + c), fixed_pack(v[c], v[c + 1]));
        return;
    }
    float *acc = reinterpret_cast<float *>(acc_);
    if constexpr (CAS && (C % 2u == 0u)) {
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            unsigned long long *p = reinterpret_cast<unsigned long long *>(acc + r * C + c);
            const unsigned long long seen = *p;
            float2 t = __builtin_bit_cast(float2, seen);
            t.x += v[c];
            t.y += v[c + 1];
            if (atomicCAS(p, seen, __builtin_bit_cast(unsigned long long, t)) != seen) {
                atomicAdd(acc + r * C + c, v[c]);
                atomicAdd(acc + r * C + c + 1, v[c + 1]);
            }
        }
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) atomicAdd(acc + r * C + c, v[c]);
    }
}

// Run merging for coarse levels: the six multisamples of a sample mostly sit in one lattice cell, and so do the
// neighbouring samples of the ray -- the LDS updates of such a level serialise on a handful of rows (level 0 of
// the benchmark grid took 16x a fine level).  Consecutive points with the same 8 rows are summed in registers
// first: 8 row updates per RUN instead of per point.  The state is carried by the caller, so a lane that walks
// several consecutive samples (the compacted kernel) keeps merging across them.
// (A rows-only pre-pass that drops samples without a corner in this block was tried: with 64 lanes per wave
//  some lane almost always stays, so the wave pays the pre-pass AND the full path -- 25 % slower.)
// C > 2 (r05; the reference's own waymo.gin grid is C = 4): the run keeps ONE summed corner weight per row instead of C values -- 16
// registers of state instead of 8 + 8 C -- and is flushed at the end of every sample with that sample's gradient (v = weight x g), so
// it no longer merges ACROSS samples.  With v[8][4] the kernel's 128-register budget (1024 threads) did not hold a run, the next item's
// prefetched geometry and a point's corner arithmetic together: the C = 4 instantiation spilled 405 registers, 500 of its 1561 scratch
// accesses in run_flush alone, inside the item loops of every coarse level (the whole proposal grid of waymo.gin).
template <uint32_t C>
struct RowRun {
    static constexpr bool kLean = C > 2;
    uint32_t cur[8];
    float v[8][kLean ? 1 : C];
    bool have;
};

template <uint32_t C, typename A>
__device__ __forceinline__ void run_flush(A *__restrict__ acc, uint32_t row_lo, uint32_t nrows, const RowRun<C> &run, const float (&gout)[C]) {
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t r = run.cur[k] - row_lo;
        if (r < nrows) {
            if constexpr (RowRun<C>::kLean) {
                float val[C];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) val[c] = run.v[k][0] * gout[c];
                lds_row_add<C, false>(acc, r, val);
            } else {
                lds_row_add<C, false>(acc, r, run.v[k]);
            }
        }
    }
}

template <uint32_t C, bool HASHED, bool POW2, typename A>
__device__ __forceinline__ void run_merge_sample(const UcnLevel &lv, A *__restrict__ acc, uint32_t row_lo, uint32_t nrows,
                                                 const float (&u)[6][3], const float (&rs)[6], const float (&gout)[C],
                                                 RowRun<C> &run) {
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            float fx, fy, fz, w[8];
            uint32_t rows[8];
            corner_rows<HASHED, POW2>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows);
            corner_weights(fx, fy, fz, w);
            const float damp = erf_pos(rs[j] * lv.inv_gs);
            bool same = run.have;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) same = same && rows[k] == run.cur[k];
            if (run.have && !same) run_flush<C>(acc, row_lo, nrows, run, gout);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const float wd = w[k] * damp;
                if constexpr (RowRun<C>::kLean) {
                    run.v[k][0] = same ? run.v[k][0] + wd : wd;
                } else {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) run.v[k][c] = same ? run.v[k][c] + wd * gout[c] : wd * gout[c];
                }
                run.cur[k] = rows[k];
            }
            run.have = true;
        }
    }
    if constexpr (RowRun<C>::kLean) {                                 // the run's weights belong to THIS sample's gradient
        if (run.have) run_flush<C>(acc, row_lo, nrows, run, gout);
        run.have = false;
    }
}

// Row-block variant of level_scatter: only corners whose row lies in [row_lo, row_lo + nrows) count, and
// they go to the workgroup's LDS copy of that row block (lds_row_add).
template <uint32_t C, bool HASHED, bool POW2, bool MERGE, typename A>
__device__ __forceinline__ void level_scatter_block(const UcnLevel &lv, A *__restrict__ acc, uint32_t row_lo,
                                                    uint32_t nrows, const float (&u)[6][3], const float (&rs)[6],
                                                    const float (&gout)[C]) {
    if constexpr (MERGE) {
        RowRun<C> run;
        run.have = false;
        run_merge_sample<C, HASHED, POW2>(lv, acc, row_lo, nrows, u, rs, gout, run);
        if (run.have) run_flush<C>(acc, row_lo, nrows, run, gout);
        return;
    }
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (in_unit_cube(u[j][0], u[j][1], u[j][2])) {
            float fx, fy, fz;
            uint32_t rows[8];
            corner_rows<HASHED, POW2>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows);
            bool any = false;
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                rows[k] -= row_lo;                      // uint32 wrap: rows below the block become huge
                any |= rows[k] < nrows;
            }
            if (any) {
                float w[8];
                corner_weights(fx, fy, fz, w);
                const float damp = erf_pos(rs[j] * lv.inv_gs);
#pragma unroll
                for (uint32_t k = 0; k < 8; k++)
                    if (rows[k] < nrows) {
                        float v[C];
#pragma unroll
                        for (uint32_t c = 0; c < C; c++) v[c] = (w[k] * damp) * gout[c];
                        lds_row_add<C, true>(acc, rows[k], v);
                    }
            }
        }
    }
}

#ifndef UCN_SHARED_CELL_MAX_RES
#define UCN_SHARED_CELL_MAX_RES 64
#endif
#ifndef UCN_LANEPAIR_MIN_RES
#define UCN_LANEPAIR_MIN_RES 2048u                                  // levels finer than this take the lane-paired fetch when a wave's rays are
//                                                                     neighbouring pixels (rendering: on the middle levels the lanes share lines anyway
//                                                                     and the three swaps per corner pair cost more than they save -- per-level times in
//                                                                     profiles/r04/level_times_*.txt); UCN_RAYS_INCOHERENT (random training rays, no
//                                                                     sharing on any hashed level) lowers it to kSharedCellMaxRes
#endif
constexpr uint32_t kSharedCellMaxRes = UCN_SHARED_CELL_MAX_RES;     // dense levels up to this resolution use level_accumulate_shared

// layout: 0 = [L][B][C] (b as given), 1 = [B][L*C]
template <uint32_t C, typename TT>
__device__ __forceinline__ void featurise(const UcnLevels &lvls, const TT *__restrict__ table, uint32_t lvl0,
                                          uint32_t lvl1, const float (&u)[6][3], const float (&rs)[6], uint32_t G,
                                          size_t B, size_t b, float *__restrict__ out, bool sample_major, bool out_bf16 = false,
                                          bool full_wave = false, uint32_t lp_min_res = UCN_LANEPAIR_MIN_RES) {
    const uint32_t F_out = lvls.L * C;
    for (uint32_t lvl = lvl0; lvl < lvl1; lvl++) {
        const UcnLevel lv = lvls.lv[lvl];
        const TT *tab = table + (size_t)lv.first_row * C;
        float acc[C];
        // wave-uniform dispatch on the level's addressing mode (lv lives in SGPRs)
        if (lv.hashed && G == 6 && lv.resolution <= kSharedCellMaxRes) {
            if (lv.mask) level_accumulate_shared<C, true, true>(lv, tab, u, rs, acc);
            else level_accumulate_shared<C, true, false>(lv, tab, u, rs, acc);
        } else if (lv.hashed) {
            if constexpr (C == 2) {
#ifndef UCN_NO_LANEPAIRS
                if (lv.mask && G == 6 && full_wave && lv.resolution > lp_min_res) level_accumulate_lanepairs<true, TT>(lv, tab, u, rs, acc);
                else
#endif
                if (lv.mask && lv.resolution > 2048u && G == 6) level_accumulate_pairs(lv, tab, u, rs, acc);
                else if (lv.mask) level_accumulate<C, true, true>(lv, tab, u, rs, G, acc);
                else level_accumulate<C, true, false>(lv, tab, u, rs, G, acc);
            } else {
                if (lv.mask) level_accumulate<C, true, true>(lv, tab, u, rs, G, acc);
                else level_accumulate<C, true, false>(lv, tab, u, rs, G, acc);
            }
        } else if (G == 6 && lv.resolution <= kSharedCellMaxRes) {
            if (lv.mask) level_accumulate_shared<C, false, true>(lv, tab, u, rs, acc);
            else level_accumulate_shared<C, false, false>(lv, tab, u, rs, acc);
        } else {
#ifndef UCN_NO_LANEPAIRS
            if constexpr (C == 2) {
                if (lv.mask && G == 6 && full_wave && lv.stride[0] == 1u && lv.resolution > lp_min_res) {
                    level_accumulate_lanepairs<false, TT>(lv, tab, u, rs, acc);
                } else {
                    if (lv.mask) level_accumulate<C, false, true>(lv, tab, u, rs, G, acc);
                    else level_accumulate<C, false, false>(lv, tab, u, rs, G, acc);
                }
            } else
#endif
            {
                if (lv.mask) level_accumulate<C, false, true>(lv, tab, u, rs, G, acc);
                else level_accumulate<C, false, false>(lv, tab, u, rs, G, acc);
            }
        }
        float *o = sample_major ? out + b * F_out + (size_t)lvl * C : out + ((size_t)lvl * B + b) * C;
        const float inv = (float)G;
        if constexpr (C == 2) {
            if (out_bf16) {            // [L][B] pairs of bf16 (round to nearest even): what the bf16 MLP would make of the floats
                typedef float f2v __attribute__((ext_vector_type(2)));
                typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
                const f2v t = {acc[0] / inv, acc[1] / inv};
                reinterpret_cast<uint32_t *>(out)[(size_t)lvl * B + b] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf2v));
                continue;
            }
            *reinterpret_cast<float2 *>(o) = make_float2(acc[0] / inv, acc[1] / inv);
        } else if constexpr (C == 4) {
            *reinterpret_cast<float4 *>(o) = make_float4(acc[0] / inv, acc[1] / inv, acc[2] / inv, acc[3] / inv);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) o[c] = acc[c] / inv;
        }
    }
}

template <uint32_t C>
__device__ __forceinline__ void featurise_bwd(const UcnLevels &lvls, float *__restrict__ grad_table, uint32_t lvl0,
                                              uint32_t lvl1, const float (&u)[6][3], const float (&rs)[6], uint32_t G,
                                              size_t B, size_t b, const float *__restrict__ grad, bool sample_major) {
    const uint32_t F = lvls.L * C;
    for (uint32_t lvl = lvl0; lvl < lvl1; lvl++) {
        const UcnLevel lv = lvls.lv[lvl];
        float *gtab = grad_table + (size_t)lv.first_row * C;
        const float *gp = sample_major ? grad + b * F + (size_t)lvl * C : grad + ((size_t)lvl * B + b) * C;
        float gout[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) gout[c] = gp[c] / (float)G;          // d(mean over G)
        if (lv.hashed) {
            if (lv.mask) level_scatter<C, true, true>(lv, gtab, u, rs, G, gout);
            else level_scatter<C, true, false>(lv, gtab, u, rs, G, gout);
        } else {
            if (lv.mask) level_scatter<C, false, true>(lv, gtab, u, rs, G, gout);
            else level_scatter<C, false, false>(lv, gtab, u, rs, G, gout);
        }
    }
}

// coord.py:60-72 followed by the /2 of models.py:491-493; returns the [0,1] grid coordinate (exact op
// sequence of the reference) and rs = 1/sqrt(8 std^2) of the contracted, halved std (fast math: it only
// feeds the erf damping).
// sd_out (ucn_cast_probe / ucn_contract_probe only): the contracted, halved std as the damping sees it.
__device__ __forceinline__ void contract_to_unit(float x, float y, float z, float sd, bool warp, float &u0, float &u1,
                                                 float &u2, float &rs, float &c0, float &c1, float &c2,
                                                 float *sd_out = nullptr) {
    if (warp) {
        const float m = fmaxf((x * x + y * y) + z * z, UCN_EPS);
        if (!(m <= 1.0f)) {
            const float root = sqrtf(m);
            const float k = (2.0f * root - 1.0f) / m;
            x = k * x; y = k * y; z = k * z;
            // ((2 root - 1)^(1/3) / root)^2 ; coord.py:69
            const float cb = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(2.0f * root - 1.0f) * 0.3333333432674408f);
            const float sh = cb * __builtin_amdgcn_rcpf(root);
            sd = (sh * sh) * sd;
        }
        x = x / 2.0f; y = y / 2.0f; z = z / 2.0f;
        sd = sd / 2.0f;
    }
    c0 = x; c1 = y; c2 = z;
    u0 = (x + 1.0f) / 2.0f; u1 = (y + 1.0f) / 2.0f; u2 = (z + 1.0f) / 2.0f;    // grid.py:162, bound = 1
    rs = __builtin_amdgcn_rsqf(8.0f * (sd * sd));
    if (sd_out) *sd_out = sd;
}

// The six multisample Gaussians of sample (ray, s): render.py:108-152 then contract_to_unit.
// Shared by the forward and the backward kernel (the backward recomputes it instead of reading back
// 6x4 floats per sample).
__device__ __forceinline__ void cast_sample(const RayInputs &in, const HexPattern &hx, float std_scale, uint32_t ray,
                                            uint32_t s, uint32_t S, float (&u)[6][3], float (&rs)[6],
                                            float (&csum)[3], float &tsum, float *probe = nullptr) {
    const float nr = in.near_[ray], fr = in.far_[ray];
    const float s0 = in.sdist[(size_t)ray * (S + 1) + s], s1 = in.sdist[(size_t)ray * (S + 1) + s + 1];
    const float t0 = s0 * fr + (1.0f - s0) * nr, t1 = s1 * fr + (1.0f - s1) * nr;
    const float rad = in.radii[ray];
    const float *bp = in.basis + (size_t)ray * 6;
    const float e1x = bp[0], e1y = bp[1], e1z = bp[2], e2x = bp[3], e2y = bp[4], e2z = bp[5];
    const float dx = in.dirs[ray * 3 + 0], dy = in.dirs[ray * 3 + 1], dz = in.dirs[ray * 3 + 2];
    const float ox = in.origins[ray * 3 + 0], oy = in.origins[ray * 3 + 1], oz = in.origins[ray * 3 + 2];
    // render.py:112-117
    const float t_m = (t0 + t1) / 2.0f, t_d = (t1 - t0) / 2.0f;
    const float td2 = t_d * t_d, tm2 = t_m * t_m;
    const float a_ = t_d / (td2 + 3.0f * tm2);
    const float inner = td2 - tm2;
    const float root = sqrtf(inner * inner + 4.0f * (tm2 * tm2));
    const float base = t1 * t1 + 2.0f * tm2;
    // angles: deterministic hexagon (rotated 30 deg + mirrored on odd samples) or random spin/flip
    const bool rnd = in.flip != nullptr;
    float spin2pi = 0.0f;
    bool keep = true;
    if (rnd) {
        keep = in.flip[(size_t)ray * S + s] > 0.5f;
        spin2pi = 6.2831854820251465f * in.spin[(size_t)ray * S + s];
    }
    const float sd_unit = (std_scale * rad) * 0.70710678118654752f;   // std only: multiply instead of IEEE divide
    const uint32_t odd = s & 1u;
    csum[0] = csum[1] = csum[2] = 0.0f;
    tsum = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        const float t = t0 + a_ * (base + hx.cj[j] * root);
        float cs, sn;
        if (rnd) {
            float ang = hx.ang[j] + spin2pi;
            if (!keep) ang = 5.235987663269043f - ang;
            // v_sin_f32 / v_cos_f32 (argument in revolutions, |error| ~1e-6 absolute): the angle only places a multisample
            // on its circle of radius ~r t / sqrt(2) (render.py:126-136), so the position moves by < 1e-9.  The precise
            // cosf / sinf were ~100 VALU instructions per point and level group; removing them did not change the training
            // forward's time (2.02 ms before and after: random rays leave it bound by the gather, not by VALU).
            const float rev = ang * 0.15915494309189535f;
            cs = __builtin_amdgcn_cosf(rev); sn = __builtin_amdgcn_sinf(rev);
        } else {
            cs = odd ? hx.cs[1][j] : hx.cs[0][j];
            sn = odd ? hx.sn[1][j] : hx.sn[0][j];
        }
        const float rt = rad * t;
        const float l0 = (rt * cs) / 1.4142135381698608f, l1 = (rt * sn) / 1.4142135381698608f;
        // math.matmul with basis^T (render.py:146-148): sum_k local_k * axis_k, then + origin
        const float wx = ((l0 * e1x + l1 * e2x) + t * dx) + ox;
        const float wy = ((l0 * e1y + l1 * e2y) + t * dy) + oy;
        const float wz = ((l0 * e1z + l1 * e2z) + t * dz) + oz;
        float c0, c1, c2;
        if (probe) {
            // ucn_cast_probe: what render.cast_rays returns (means, stds, t) and what the grid sees behind the contraction
            float *pr = probe + j * UCN_CAST_PROBE_FLOATS;
            float sdc;
            contract_to_unit(wx, wy, wz, sd_unit * t, true, u[j][0], u[j][1], u[j][2], rs[j], c0, c1, c2, &sdc);
            pr[0] = wx; pr[1] = wy; pr[2] = wz; pr[3] = sd_unit * t; pr[4] = t;
            pr[5] = c0; pr[6] = c1; pr[7] = c2; pr[8] = sdc; pr[9] = rs[j];
        } else {
            contract_to_unit(wx, wy, wz, sd_unit * t, true, u[j][0], u[j][1], u[j][2], rs[j], c0, c1, c2);
        }
        csum[0] += c0; csum[1] += c1; csum[2] += c2; tsum += t;
    }
}

// Introspection for the parity tests: the product's own cast_sample / contract_to_unit, results
// written out instead of consumed.
__global__ __launch_bounds__(256) void k_cast_probe(RayInputs in, HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                    float *__restrict__ out) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
    float u[6][3], rs[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum, out + b * 6 * UCN_CAST_PROBE_FLOATS);
}

__global__ __launch_bounds__(256) void k_contract_probe(const float *__restrict__ means, const float *__restrict__ stds,
                                                        uint32_t B, float *__restrict__ out_mean, float *__restrict__ out_std) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    float u0, u1, u2, rs, c0, c1, c2, sd;
    contract_to_unit(means[b * 3], means[b * 3 + 1], means[b * 3 + 2], stds[b], true, u0, u1, u2, rs, c0, c1, c2, &sd);
    out_mean[b * 3] = c0; out_mean[b * 3 + 1] = c1; out_mean[b * 3 + 2] = c2;
    out_std[b] = sd;
}

// Levels handled by one thread: group g = levels [lo[g], lo[g+1]).  A thread re-derives the sample's six
// contracted positions (~800 VALU instructions) once per GROUP, then pays ~400-500 per level.  Fine
// hashed levels stay alone (level-major dispatch keeps one 4 MiB slice per XCD L2); the coarse levels,
// which are VALU-bound and whose accesses are concentrated on a few cells, share the geometry.
struct LevelGroups {
    uint8_t lo[UCN_MAX_LEVELS + 1];
    uint32_t n;
};
static LevelGroups make_groups(const UcnLevels &lv, uint32_t levels_per_block) {
    LevelGroups g;
    g.n = 0;
    uint32_t l = 0;
    const uint32_t cres = 2048u, cgrp = 8u;   // measured: (512..8192) x (3..8) all within 3 %; this is the best
    while (l < lv.L) {
        g.lo[g.n++] = (uint8_t)l;
        uint32_t take = levels_per_block;
        if (levels_per_block == 0) take = lv.lv[l].resolution <= cres ? cgrp : 1u;          // auto
        uint32_t end = l + take < lv.L ? l + take : lv.L;
        if (levels_per_block == 0)                                                        // a group never mixes coarse and fine
            for (uint32_t k = l + 1; k < end; k++)
                if (lv.lv[k].resolution > cres) { end = k; break; }
        l = end;
    }
    g.lo[g.n] = (uint8_t)lv.L;
    return g;
}

// layout: 0 = [L][N*S][C] with b = ray*S+s; 1 = [N*S][L*C]; 2 = [L][S*N][C] with b = s*N+ray
// FEW_LEVELS is a NAME TAG only (same code): grids of <= 8 levels (the proposal fields: L = 6) and of more (the NeRF field:
// L = 16 / 10) get distinct kernel names, so that a rocprofv3 --stats summary separates the proposal-level from the
// NeRF-level launches without subtracting one from the other.
template <uint32_t C, uint32_t TPB, typename TT = float, bool FEW_LEVELS = false>
__global__ __launch_bounds__(TPB) void k_march_features(UcnLevels lvls, const TT *__restrict__ table, RayInputs in,
                                                        HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                        LevelGroups grp, int layout, float *__restrict__ features,
                                                        float *__restrict__ coord_out, float *__restrict__ tmean_out) {
    // co-resident shape (512 threads: launched beside an MLP workgroup on the same CU): a SIMD does not overlap VALU
    // with MFMA (tools/mfma_valu_bench.hip), so at equal priority every VALU instruction of this kernel would queue behind
    // one of the MLP wave's 32-cycle MFMAs.  With the higher priority the gather waves issue their bursts back to back and
    // the MFMAs fill the time in which all of them wait for memory.
    if constexpr (TPB == 512) __builtin_amdgcn_s_setprio(3);
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (b >= B) return;
    // layout 2 ("rays fastest"): the 64 lanes of a wave are NEIGHBOURING RAYS at one sample index.
    // On the dense coarse levels neighbouring pixels read the same few lattice cells, which the TA
    // coalesces: levels 0-7 drop to the VALU floor (-20 % on the kernel, r01b).  The model's default.
    uint32_t ray, s;
    if ((layout & 3) == 2) { s = (uint32_t)(b / N); ray = (uint32_t)(b - (size_t)s * N); }
    else { ray = (uint32_t)(b / S); s = (uint32_t)(b - (size_t)ray * S); }
    float u[6][3], rs[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum);
    const uint32_t lvl0 = grp.lo[blockIdx.y], lvl1 = grp.lo[blockIdx.y + 1];
    const bool full_wave = __ballot(true) == ~0ull;                  // the lane-paired fetch trades rows between lanes i and i + 32
    const uint32_t lp_min = (layout & 0x20) ? kSharedCellMaxRes : UCN_LANEPAIR_MIN_RES;     // 0x20: UCN_RAYS_INCOHERENT (private bit)
    if constexpr (sizeof(TT) == 2) featurise<C, TT>(lvls, table, lvl0, lvl1, u, rs, 6, B, b, features, (layout & 3) == 1, (layout & 0x10) != 0, full_wave, lp_min);
    else featurise<C, TT>(lvls, table, lvl0, lvl1, u, rs, 6, B, b, features, (layout & 3) == 1, false, full_wave, lp_min);
    if (blockIdx.y == 0) {
        const size_t o = (size_t)ray * S + s;                     // per-sample side outputs stay [N,S]
        if (coord_out) {
            coord_out[o * 3 + 0] = csum[0] / 6.0f; coord_out[o * 3 + 1] = csum[1] / 6.0f; coord_out[o * 3 + 2] = csum[2] / 6.0f;
        }
        if (tmean_out) tmean_out[o] = tsum / 6.0f;
    }
}

// d(loss)/d(table) of k_march_features.  (means/stds carry no gradient: coord.track_linearize is
// @torch.no_grad, coord.py:75, and sdist is detached, models.py:204-205.)  fp32 atomics in L2, like
// kernel_grid_backward (gridencoder.cu:336).
template <uint32_t C>
__global__ __launch_bounds__(256) void k_march_features_bwd(UcnLevels lvls, float *__restrict__ grad_table, RayInputs in,
                                                            HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                            uint32_t lpb, int layout,
                                                            const float *__restrict__ grad_features) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
    float u[6][3], rs[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum);
    const uint32_t lvl0 = blockIdx.y * lpb;
    const uint32_t lvl1 = lvl0 + lpb < lvls.L ? lvl0 + lpb : lvls.L;
    featurise_bwd<C>(lvls, grad_table, lvl0, lvl1, u, rs, 6, B, b, grad_features, layout == 1);
}

// Same gradient without global atomics.  Scattered fp32 atomics run at ~21 G/s on MI355X whatever their
// scope (tools/atomic_bench.hip: 50 M row updates x 2 channels = 4.8 ms, against 0.14 ms for the forward's
// gathers of the same rows), and a hashed level receives ~100 updates per row per 8192-ray batch.  So:
// one workgroup OWNS a block of `rpb` rows of one level (128 KiB of LDS), walks ALL samples, recomputes
// their corners (VALU is cheap here) and accumulates the ones that fall in its block with ds_add_f32;
// the block is then added to the table gradient with plain coalesced read-modify-writes -- no other
// workgroup touches those rows.  blockIdx.x enumerates (level, block) pairs, level-major.
// The six contracted multisample positions and damping arguments of every sample, as [N*S][6] float4
// {x, y, z, std argument} (one 16-byte load per point for the compacted kernel's scattered items): written once
// per backward call, read by every (level, row block) workgroup.
__global__ __launch_bounds__(256) void k_cast_cache(RayInputs in, HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                    float *__restrict__ geom) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
    float u[6][3], rs[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum);
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        reinterpret_cast<float4 *>(geom)[b * 6 + j] = make_float4(u[j][0], u[j][1], u[j][2], rs[j]);
    }
}

// ---- block masks: which row blocks of a level a sample touches.  Planes of [N*S] uint32 behind the geometry
// cache; a COARSE level (resolution <= 512: items are whole samples, run-merged; coarse = 2 up to resolution 64:
// a lane walks consecutive samples) has one plane = union over its 48 corners, bit p = some corner's row lies in
// block p (rows >> shift); a fine level one plane per GROUP of four blocks, bit 4 * j + (p & 3) of word p >> 2 =
// multisample j has a corner in block p -- either way a workgroup reads one word per sample.
// Where element (level, sample b, channel c) of the feature gradient lives: level * L + b * S + c * Cs floats.
//   layout 0 = [L][B][C]   1 = [B][L*C] (what autograd hands over)   3 = [L*C][B] (a transposed dgrad GEMM)
struct GradStrides {
    size_t level, sample, chan;
};
static GradStrides grad_strides(int layout, size_t B, uint32_t L, uint32_t C) {
    if (layout == 1) return {C, (size_t)L * C, 1};
    if (layout == 3) return {(size_t)C * B, 1, B};
    return {B * C, C, 1};
}

#ifndef UCN_KSCAN
#define UCN_KSCAN 4
#endif
constexpr uint32_t kScan = UCN_KSCAN;                          // samples per thread and scan step (8 measured: see DESIGN)
#ifndef UCN_MASKS_MIN_BLOCKS
#define UCN_MASKS_MIN_BLOCKS 1                           // experiment knob (r06): workgroups per CU the mask pass is compiled for.  8 = 64 registers (4 spilled), 8 waves per SIMD instead of 5 at 95: the autocast step's pass 0.419 -> 0.374 ms, but the fp32 step (row-major gradient copied and divided here) 14.52 -> 14.70 ms; 6 = 67 registers: 0.399 ms.  Not taken
#endif
#ifndef UCN_BWD_BYTE_MASKS_DEFAULT
#define UCN_BWD_BYTE_MASKS_DEFAULT 1
#endif
constexpr uint32_t kMaxMaskWords = 16;                  // sample-item levels: up to 512 row blocks (16 KiB of LDS in the mask pass)
struct MaskPlan {
    uint16_t plane[UCN_MAX_LEVELS];
    uint8_t coarse[UCN_MAX_LEVELS];
    uint32_t n_planes, shift;
    uint16_t split[UCN_MAX_LEVELS];        // workgroups per row block (cut along the samples): ~128 per level, 256 for the last
    uint32_t skip_fine;                    // 1: the fine levels are taken by the item-list kernels (k_bwd_list), not by bwd_cmp
    uint8_t order[UCN_MAX_LEVELS];         // levels in the order their workgroups are dispatched: longest workgroups first
    uint8_t fine_kind[UCN_MAX_LEVELS];     // point-item levels: 0 = ballot-ordered items + corner walk, 1 = lane-ordered items + corner walk, 2 = lane-ordered + (y, z) combinations, 3 = 2 on byte planes (cmp_block_bytes)
};
__host__ __device__ __forceinline__ uint32_t bwd_sample_split(uint32_t blocks_in_level, uint32_t target = 128u) {
    return blocks_in_level >= target ? 1u : target / blocks_in_level;     // ~`target` workgroups per level (default 128)
}
static bool make_mask_plan(const UcnLevels &lv, uint32_t rpb, size_t B, MaskPlan *mp, bool fixed_rows = false) {
    uint32_t shift = 0;
    while ((1u << shift) < rpb) shift++;
    if ((1u << shift) != rpb) return false;
    mp->shift = shift;
    mp->n_planes = 0;
    mp->skip_fine = 0u;
    for (uint32_t l = 0; l < lv.L; l++) {
        const uint32_t nb_l = (lv.lv[l].rows + rpb - 1) / rpb;
        if (nb_l > kMaxMaskWords * 32u) return false;                     // (> 512 blocks, e.g. 2^23 rows of C = 4: the atomic fallback)
        // measured per level on the benchmark grid (tools/level_times_bwd.py): sample items + run merging win up to
        // resolution 512, walking consecutive samples in one lane up to 64
        // r06, FIXED-POINT rows (the autocast step): a row update is one fire-and-forget ds_add_u64 there, so run merging buys nothing
        // and the ~8 row blocks a SAMPLE of the hashed levels 84 ... 446 touches each redo all of its 48 corners; as point items on byte
        // planes those levels cost 44-47 instead of 45-82 ms-CU each (tools/bwd_balance.py under UCN_TOOL_FX=1, whole call 3.54 ->
        // 3.20 ms; profiles/r06/bwd_coarse_res_fx.txt).  With float rows the same change LOSES (3.71 -> 5.3 ms: compare-and-swap
        // collisions; bwd_coarse_res_float.txt), so the threshold follows the row type.
        static const int coarse_res_env = getenv("UCN_BWD_COARSE_RES") ? atoi(getenv("UCN_BWD_COARSE_RES")) : -1;                 // experiment knob
        const uint32_t coarse_res = coarse_res_env >= 0 ? (uint32_t)coarse_res_env : (fixed_rows ? 64u : 512u);
        mp->coarse[l] = lv.lv[l].resolution <= coarse_res ? 1 : 0;
        // (experiment knob, r06: under fixed-point rows the plain sample item is cheaper than the walk on the dense levels too -- 17.1 / 19.4 /
        // 39.1 -> 12.7 / 16.3 / 32.3 ms-CU, call 3.205 -> 3.164 ms, step -0.05 ... -0.1 ms -- but merging runs ACROSS six samples also means six
        // times fewer ROUNDED addends on exactly the rows that collect the most samples: 2.4 x the fixed-point noise there (a fuzz case of
        // 640 000 random-sign gradients went from < 1e-3 to 1.24e-3 of the largest entry) for 1 % of the step.  Not taken: 64 for both row types)
        static const uint32_t runs_res = getenv("UCN_BWD_RUNS_RES") ? (uint32_t)atoi(getenv("UCN_BWD_RUNS_RES")) : 64u;
        if (mp->coarse[l] && !lv.lv[l].hashed && lv.lv[l].resolution <= runs_res) mp->coarse[l] = 2;
        // More than 32 row blocks per level (the reference's own waymo.gin grid: T = 2^21 rows of C = 4 -> 256 blocks of 8192 rows):
        // the per-point masks of the point-item shapes would need 6 bits x 256 blocks per sample and level, so such a level goes by
        // SAMPLE items as well -- one bit per (sample, block) in nb / 32 mask words -- with the unmerged per-point scatter (a point's
        // corners lie in ~4.5 of the 256 blocks: the `any corner in my block` test drops most points before weights and erf).
        // Before r03 these configurations fell back to the global-atomic kernel: 83.6 of the 88.5 ms of a waymo.gin training step.
        if (nb_l > 32u && mp->coarse[l] != 2) mp->coarse[l] = 3;         // (the run-merging dense levels keep their shape, with nb / 32 mask words)
        // experiment knob (r04): the hashed sample-item levels (resolution 128 ... 512) through the pair-item shape as well
        static const bool wide_mid = getenv("UCN_BWD_WIDE_MID") && atoi(getenv("UCN_BWD_WIDE_MID")) != 0;
        if (wide_mid && mp->coarse[l] == 1 && lv.lv[l].hashed && nb_l == 32u) mp->coarse[l] = 3;
        // Point-item levels, three shapes (workgroup clocks per level, tools/bwd_balance.py, ms-CU per level of the benchmark grid):
        //   0: items appended point by point (six ballots per step), all 8 corners walked          res 1024: 75, 2048: 62, finer: 60-62, strided: 78
        //   1: items appended lane by lane (popcount + one DPP prefix sum per step: the scan was a third of a fine level)   1024: 120 (!), 2048: 78, finer: 51-55
        //   2: 1 + the corners taken by (y, z) combinations (point_scatter_combos)                   1024: 120, 2048: 78, finer: 51-55 (shape 1 alone: 58-60), strided: 60
        // Lane order puts the six points of a sample next to each other in a batch: where they still share lattice cells
        // (resolution <= 2048 on these rays) their compare-and-swaps collide and fall back to ds_add_f32.
        mp->fine_kind[l] = 0;
        if (!mp->coarse[l]) mp->fine_kind[l] = (!lv.lv[l].hashed || lv.lv[l].resolution >= 4096u) ? 2 : 0;
        static const int force_kind = getenv("UCN_BWD_FINE_KIND") ? atoi(getenv("UCN_BWD_FINE_KIND")) : -1;      // experiment knob
        if (!mp->coarse[l] && force_kind >= 0 && force_kind <= 2) mp->fine_kind[l] = (uint8_t)force_kind;
        // 3 (r06, VERDICT r05 item 2 b): shape 2 with BYTE PLANES -- per row block one byte per sample (bit j = multisample j has a
        // corner in the block), so that a scanning lane reads FOUR samples of ITS block in one dword (popcount 3.4 on average)
        // where the nibble planes above give it one sample of four blocks (0.84 hits): one load, one popcount and one DPP prefix
        // sum per 4096-sample unit instead of four.  Same number of planes.  UCN_BWD_BYTE_MASKS=0: off; =2: every point-item level
        // (the lane order of its items makes the compare-and-swaps of resolution 1024 / 2048 collide, as with shape 1).
        static const int byte_masks_env = getenv("UCN_BWD_BYTE_MASKS") ? atoi(getenv("UCN_BWD_BYTE_MASKS")) : -1;
        const int byte_masks = byte_masks_env >= 0 ? byte_masks_env : (fixed_rows ? 2 : UCN_BWD_BYTE_MASKS_DEFAULT);   // fixed-point rows: every point-item level (no compare-and-swap to collide)
        if (!mp->coarse[l] && nb_l <= 32u && B % 4u == 0u && kScan == 4u &&
            ((byte_masks == 1 && mp->fine_kind[l] == 2) || byte_masks == 2 || (byte_masks == 3 && mp->fine_kind[l] == 2)))
            mp->fine_kind[l] = 3;
        // 4 (=3 only, experiment): the byte-plane scan with the CORNER WALK of shape 0 for the levels whose points still share cells
        // (resolution 1024 / 2048) -- only where the row updates are fire-and-forget fixed-point adds: with compare-and-swap rows the
        // lane order of the items makes neighbouring lanes fight for one row
        else if (!mp->coarse[l] && nb_l <= 32u && B % 4u == 0u && kScan == 4u && byte_masks == 3 && fixed_rows && mp->fine_kind[l] == 0)
            mp->fine_kind[l] = 4;
        mp->plane[l] = (uint16_t)mp->n_planes;
        if (mp->coarse[l] == 3) {
            // bit planes: [block][ceil(B / 64)] 64-bit words (one bit per sample) = nb x 2 x ceil(B / 64) 32-bit words, in units of B
            const uint64_t words = (uint64_t)nb_l * 2u * ((B + 63u) / 64u);
            mp->n_planes += (uint32_t)((words + B - 1u) / (B ? B : 1u));
        } else {
            mp->n_planes += mp->coarse[l] ? (nb_l + 31u) / 32u : (nb_l + 3u) / 4u;
        }
    }
    // Dispatch order = longest workgroups first, so that the chip drains on short ones (workgroup clocks of the benchmark
    // grid, tools/bwd_balance.py: in level order the last 1.3 ms of a 4.26 ms kernel ran at 50-85 % occupancy -- the two
    // finest levels started at 3.1 / 3.4 ms -- where the sum of the workgroup times is 3.70 ms per CU).  Classes by what was
    // measured per workgroup: unhashed levels of more than two row blocks load their blocks unevenly (the strided fine levels
    // of the uint32-wrap quirk: 150 ... 1010 us; the dense 65^3 level: 45 ... 890 us) and go first; then the hashed
    // sample-item levels, finest first (690 / 510 / 420 us); then the hashed point-item levels, coarsest first (600 ... 460 us);
    // the small dense levels (140 us) fill the tail.
    uint32_t key[UCN_MAX_LEVELS];
    for (uint32_t l = 0; l < lv.L; l++) {
        const uint32_t nb = (lv.lv[l].rows + rpb - 1) / rpb;
        uint32_t cls, sub;
        if (mp->coarse[l] == 3) { cls = 0u; sub = l; }                         // many short workgroups: they fill the tail
        else if (!lv.lv[l].hashed && nb > 2u) { cls = 3u; sub = l; }
        else if (lv.lv[l].hashed && mp->coarse[l]) { cls = 2u; sub = l; }
        else if (lv.lv[l].hashed) { cls = 1u; sub = UCN_MAX_LEVELS - 1u - l; }
        else { cls = 0u; sub = l; }
        key[l] = cls * 256u + sub;
        mp->order[l] = (uint8_t)l;
    }
    for (uint32_t i = 1; i < lv.L; i++)                                    // insertion sort, descending key
        for (uint32_t j = i; j > 0 && key[mp->order[j]] > key[mp->order[j - 1]]; j--) {
            const uint8_t t = mp->order[j]; mp->order[j] = mp->order[j - 1]; mp->order[j - 1] = t;
        }
    if (getenv("UCN_BWD_LEVEL_ORDER"))                                     // experiment knob: dispatch in level order
        for (uint32_t l = 0; l < lv.L; l++) mp->order[l] = (uint8_t)l;
    // ~128 workgroups per level (flat from 128 up, measured); the LAST long level of the order is cut twice as fine: the
    // chip drains on its workgroups (the small dense levels behind it are 35 ms-CU in all), and half-length ones halve that
    static const uint32_t wg_target = getenv("UCN_BWD_WGS") && atoi(getenv("UCN_BWD_WGS")) > 0 ? (uint32_t)atoi(getenv("UCN_BWD_WGS")) : 128u;
    static const bool fine_tail = !getenv("UCN_BWD_NO_FINE_TAIL");       // experiment knobs, both
    int last_long = -1;
    for (uint32_t i = 0; i < lv.L; i++)
        if (key[mp->order[i]] >= 256u) last_long = (int)mp->order[i];
    for (uint32_t l = 0; l < lv.L; l++) {
        const uint32_t nb = (lv.lv[l].rows + rpb - 1) / rpb;
        // ... and the unevenly loaded levels finer as well: their hottest block sets the longest workgroup of the call (waymo.gin's
        // proposal grid, 1 M samples: one 4.7 ms workgroup of the dense 65^3 level against 2.4 ms-CU of work per CU)
        static const uint32_t uneven_mult = getenv("UCN_BWD_UNEVEN_MULT") ? (uint32_t)atoi(getenv("UCN_BWD_UNEVEN_MULT")) : 2u;   // experiment knob (1 / 2 / 4 / 8 measured: 2)
        const uint32_t mult = key[l] >= 3u * 256u ? (uneven_mult ? uneven_mult : 1u) : ((fine_tail && (int)l == last_long) ? 2u : 1u);
        mp->split[l] = (uint16_t)bwd_sample_split(nb, mult * wg_target);
    }
    return true;
}

template <bool HASHED, bool POW2>
__device__ __forceinline__ uint32_t point_block_mask(const UcnLevel &lv, uint32_t shift, const float (&p)[3]) {
    if (!in_unit_cube(p[0], p[1], p[2])) return 0u;
    if constexpr (HASHED && POW2) {
        // row = (x ^ y P1 ^ z P2) & mask with x <= resolution + 1 < 2^shift: x only reaches the bits BELOW the block index, so the two x
        // corners of a (y, z) combination share their block and the block follows from y and z alone -- 4 hashes instead of 8 rows, no x
        if (lv.resolution + 2u <= (1u << shift)) {
            const uint32_t y0 = (uint32_t)floorf(fmaf(p[1], lv.scale, 0.5f)), z0 = (uint32_t)floorf(fmaf(p[2], lv.scale, 0.5f));
            const uint32_t ya = y0 * kP1, yb = ya + kP1, za = z0 * kP2, zb = za + kP2;
            return (1u << (((ya ^ za) & lv.mask) >> shift)) | (1u << (((yb ^ za) & lv.mask) >> shift)) |
                   (1u << (((ya ^ zb) & lv.mask) >> shift)) | (1u << (((yb ^ zb) & lv.mask) >> shift));
        }
    }
    float fx, fy, fz;
    uint32_t rows[8], m = 0u;
    corner_rows<HASHED, POW2>(lv, p[0], p[1], p[2], fx, fy, fz, rows);
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) m |= 1u << (rows[k] >> shift);
    return m;
}

// geometry planes + block masks of every sample, once per backward call
__global__ __launch_bounds__(256, UCN_MASKS_MIN_BLOCKS) void k_cast_cache_masks(UcnLevels lvls, RayInputs in, HexPattern hx, float std_scale,
                                                          uint32_t N, uint32_t S, MaskPlan plan,
                                                          const float *__restrict__ grad_features, GradStrides gs, uint32_t C,
                                                          float *__restrict__ geom, uint32_t *__restrict__ masks,
                                                          float *__restrict__ grad_level_major /*[L][N*S][C], / 6*/,
                                                          uint32_t *__restrict__ task_counter,
                                                          float *__restrict__ l1_partial /*[L][gridDim.x] or null: sum over the block's samples of max_c |g|*/) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b < 8) task_counter[b] = 0u;                      // the compacted kernel's persistent workgroups pull tasks from here
    __shared__ float s_l1[UCN_MAX_LEVELS][4];             // per level: the four waves' sums of max_c |g| (fixed-point bound)
    if ((threadIdx.x & 63u) < UCN_MAX_LEVELS) s_l1[threadIdx.x & 63u][threadIdx.x >> 6] = 0.0f;   // own column (a wave past the end leaves zeros)
    // whole waves only (the bit planes below are built by wave ballots); a wave past the end skips the body but still reaches the
    // barrier of the l1_partial reduction at the bottom (an early `return` left the barrier to part of the workgroup)
    const bool live_wave = (b & ~(size_t)63) < B;
    do {
    if (!live_wave) break;
    const bool valid = b < B;
    const size_t bb = valid ? b : B - 1;                  // lanes past the end recompute the last sample and store nothing
    const uint32_t ray = (uint32_t)(bb / S), s = (uint32_t)(bb - (size_t)ray * S);
    float u[6][3], rs[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum);
    if (valid) {
#pragma unroll
        for (uint32_t j = 0; j < 6; j++) {
            reinterpret_cast<float4 *>(geom)[b * 6 + j] = make_float4(u[j][0], u[j][1], u[j][2], rs[j]);
        }
    }
    __shared__ uint32_t s_words[kMaxMaskWords * 256u];    // [word][thread]: a thread's own column, bank = thread
    for (uint32_t lvl = 0; lvl < lvls.L; lvl++) {
        const UcnLevel lv = lvls.lv[lvl];
        const uint32_t nb_l = (lv.rows + (1u << plan.shift) - 1u) >> plan.shift;
        bool nz = false;
        float gmax = 0.0f;
#ifdef UCN_EXP_MASKS_NO_GRAD                                   // timing-only build (results garbage): geometry + masks alone, i.e. what is left
        nz = valid;                                            // of this pass if k_train_bwd's epilogue wrote the level-major gradient (VERDICT r05 item 2 c)
        for (uint32_t c = 0; c < 0u; c++) {
#else
        for (uint32_t c = 0; c < C; c++) {
#endif
            const float g = grad_features[lvl * gs.level + bb * gs.sample + c * gs.chan];
            nz |= valid && g != 0.0f;
            gmax = fmaxf(gmax, valid ? fabsf(g) : 0.0f);
            if (valid && !(fabsf(g) <= 3.0e38f)) gmax = __builtin_inff();        // NaN / inf poisons the bound (fmaxf drops NaN)
            // the row-block workgroups fetch gradients per ITEM (scattered): give them 8 contiguous bytes per sample,
            // already divided by the 6 multisamples of the mean (an IEEE division is ~13 VALU instructions per channel;
            // an item stage would repeat it ~27 times per sample and level)
            if (valid && grad_level_major) grad_level_major[((size_t)lvl * B + b) * C + c] = g / 6.0f;
        }
        if (!grad_level_major) gmax *= 6.0f;                // layout 4: g arrived divided by 6 -- the bound is on the undivided gradient (6 addends of <= |g| / 6 x w)
#ifndef UCN_EXP_MASKS_NO_GRAD
        if (l1_partial) {                                   // (workgroup-uniform; every wave of the block gets here: no early `continue` above)
#else
        if (false) {
#endif
            const float wsum = wave_sum_dpp<float>(gmax);
            if ((threadIdx.x & 63u) == 0u) s_l1[lvl][threadIdx.x >> 6] = wsum;
        }
        if (nb_l > 32u || plan.coarse[lvl] == 3) {
            // one bit per (sample, block) in nb / 32 words: set through the thread's own LDS column (dynamic word index)
            const uint32_t nw = (nb_l + 31u) / 32u;
            for (uint32_t k = 0; k < nw; k++) s_words[k * 256u + threadIdx.x] = 0u;
            if (nz) {
#pragma unroll
                for (uint32_t j = 0; j < 6; j++) {
                    if (!in_unit_cube(u[j][0], u[j][1], u[j][2])) continue;
                    float fx, fy, fz;
                    uint32_t rows[8];
                    if (lv.hashed) { if (lv.mask) corner_rows<true, true>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows); else corner_rows<true, false>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows); }
                    else { if (lv.mask) corner_rows<false, true>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows); else corner_rows<false, false>(lv, u[j][0], u[j][1], u[j][2], fx, fy, fz, rows); }
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t blk = rows[k] >> plan.shift;
                        atomicOr(&s_words[(blk >> 5) * 256u + threadIdx.x], 1u << (blk & 31u));     // ds_or_b32, own column
                    }
                }
            }
            if (plan.coarse[lvl] == 3) {
                // wide_block's layout: BIT PLANES [block][ceil(B / 64)] x 64 bits, one bit per sample -- the scanning workgroup of
                // a block then reads B / 8 bytes instead of 4 B (measured in place with phase clocks: the word-per-sample scan
                // was HALF of such a workgroup's time, one exposed load latency per 4096 samples).  A wave = 64 consecutive
                // samples: 32 ballots per mask word, lane i keeps the ballot of block 32 w + i.
                const uint32_t lane = threadIdx.x & 63u;
                const size_t B64 = (B + 63u) / 64u, wave_global = b >> 6;
                uint32_t *T = masks + (size_t)plan.plane[lvl] * B;
                for (uint32_t w = 0; w < nw; w++) {
                    const uint32_t word = s_words[w * 256u + threadIdx.x];
                    uint32_t keep_lo = 0u, keep_hi = 0u;
#pragma unroll
                    for (uint32_t bit = 0; bit < 32u; bit++) {
                        const uint64_t bal = __ballot((word >> bit) & 1u);
                        if (lane == bit) { keep_lo = (uint32_t)bal; keep_hi = (uint32_t)(bal >> 32); }
                    }
                    const uint32_t blk = w * 32u + lane;
                    if (lane < 32u && blk < nb_l) {
                        T[((size_t)blk * B64 + wave_global) * 2u] = keep_lo;
                        T[((size_t)blk * B64 + wave_global) * 2u + 1u] = keep_hi;
                    }
                }
                continue;
            }
            if (valid) {
                uint32_t *mpw = masks + (size_t)plan.plane[lvl] * B + b;
                for (uint32_t k = 0; k < nw; k++) mpw[(size_t)k * B] = s_words[k * 256u + threadIdx.x];
            }
            continue;
        }
        uint32_t m[6];
#pragma unroll
        for (uint32_t j = 0; j < 6; j++) {
            if (lv.hashed) m[j] = lv.mask ? point_block_mask<true, true>(lv, plan.shift, u[j]) : point_block_mask<true, false>(lv, plan.shift, u[j]);
            else m[j] = lv.mask ? point_block_mask<false, true>(lv, plan.shift, u[j]) : point_block_mask<false, false>(lv, plan.shift, u[j]);
        }
        // a sample whose feature gradient on this level is exactly zero contributes nothing: clear its masks here
        // so that the scanning workgroups never have to look at the gradient
        if (!nz) {
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) m[j] = 0u;
        }
        if (plan.fine_kind[lvl] >= 3 && !plan.coarse[lvl]) {
            // BYTE PLANES (cmp_block_bytes): plane p = B bytes, byte b = the six point bits of sample b for row block p.  A thread
            // spreads its masks into words of four block bytes (nibble x 0x00204081 puts bit i of a nibble at bit 8 i), the four
            // threads of a quad (samples 4 q ... 4 q + 3; B % 4 == 0, so a quad is in range or not as a whole) transpose 4 x 4 bytes
            // through DPP quad broadcasts + v_perm_b32, and thread i of the quad stores the dword of block 4 k + i.
            const uint32_t qi = threadIdx.x & 3u;
            const uint32_t sel01 = qi | ((4u + qi) << 8) | 0x0c0c0000u, sel23 = 0x00000c0cu | (qi << 16) | ((4u + qi) << 24);
            uint8_t *pl = reinterpret_cast<uint8_t *>(masks + (size_t)plan.plane[lvl] * B);
            const uint32_t groups = (nb_l + 3u) / 4u;
            for (uint32_t k = 0; k < groups; k++) {
                uint32_t wk = 0u;
#pragma unroll
                for (uint32_t j = 0; j < 6; j++) wk |= ((((m[j] >> (4u * k)) & 15u) * 0x00204081u) & 0x01010101u) << j;
                const uint32_t w0 = dpp_or0<0x00, 0xf>(wk), w1 = dpp_or0<0x55, 0xf>(wk), w2 = dpp_or0<0xaa, 0xf>(wk), w3 = dpp_or0<0xff, 0xf>(wk);
                const uint32_t d = __builtin_amdgcn_perm(w1, w0, sel01) | __builtin_amdgcn_perm(w3, w2, sel23);
                const uint32_t blk = 4u * k + qi;
                if (valid && blk < nb_l) *reinterpret_cast<uint32_t *>(pl + (size_t)blk * B + (b & ~(size_t)3)) = d;
            }
            continue;
        }
        if (!valid) continue;
        uint32_t *mp = masks + (size_t)plan.plane[lvl] * B + b;
        if (plan.coarse[lvl]) {
            mp[0] = m[0] | m[1] | m[2] | m[3] | m[4] | m[5];
        } else {
            // word k = blocks 4k..4k+3, bit 4 * j + (block & 3) for multisample j: a workgroup reads ONE word per sample
            const uint32_t groups = (((lv.rows + (1u << plan.shift) - 1u) >> plan.shift) + 3u) / 4u;
            for (uint32_t k = 0; k < groups; k++) {
                uint32_t wk = 0u;
#pragma unroll
                for (uint32_t j = 0; j < 6; j++) wk |= ((m[j] >> (4u * k)) & 15u) << (4u * j);
                mp[(size_t)k * B] = wk;
            }
        }
    }
    } while (false);
    if (l1_partial) {
        __syncthreads();
        if (threadIdx.x < lvls.L)                           // fixed order: deterministic
            l1_partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
                ((s_l1[threadIdx.x][0] + s_l1[threadIdx.x][1]) + s_l1[threadIdx.x][2]) + s_l1[threadIdx.x][3];
    }
}

// Two rows of the block at once (the x0 / x0 + 1 corners of one (y, z) combination): both reads, then both compare-and-swaps
// -- two LDS round trips where two lds_row_add calls make four.  A lane whose row is outside the block skips its half.
template <uint32_t C, typename A>
__device__ __forceinline__ void lds_row_add_pair(A *acc_, uint32_t r0, bool in0, const float (&v0)[C], uint32_t r1, bool in1,
                                                 const float (&v1)[C]) {
    if constexpr (kFixed<A>) {
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            if (in0) atomicAdd(reinterpret_cast<unsigned long long *>(acc_ + r0 * C + c), fixed_pack(v0[c], v0[c + 1]));
            if (in1) atomicAdd(reinterpret_cast<unsigned long long *>(acc_ + r1 * C + c), fixed_pack(v1[c], v1[c + 1]));
        }
        return;
    }
    float *acc = reinterpret_cast<float *>(acc_);
    if constexpr (C % 2u == 0u) {
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            unsigned long long *p0 = reinterpret_cast<unsigned long long *>(acc + r0 * C + c);
            unsigned long long *p1 = reinterpret_cast<unsigned long long *>(acc + r1 * C + c);
            unsigned long long s0 = 0ull, s1 = 0ull;
            if (in0) s0 = *p0;
            if (in1) s1 = *p1;
            float2 t0 = __builtin_bit_cast(float2, s0), t1 = __builtin_bit_cast(float2, s1);
            t0.x += v0[c]; t0.y += v0[c + 1];
            t1.x += v1[c]; t1.y += v1[c + 1];
            bool lost0 = false, lost1 = false;
            if (in0) lost0 = atomicCAS(p0, s0, __builtin_bit_cast(unsigned long long, t0)) != s0;
            if (in1) lost1 = atomicCAS(p1, s1, __builtin_bit_cast(unsigned long long, t1)) != s1;   // (r1 == r0: loses, correctly)
            if (lost0) { atomicAdd(acc + r0 * C + c, v0[c]); atomicAdd(acc + r0 * C + c + 1, v0[c + 1]); }
            if (lost1) { atomicAdd(acc + r1 * C + c, v1[c]); atomicAdd(acc + r1 * C + c + 1, v1[c + 1]); }
        }
    } else {
        if (in0) lds_row_add<C, true>(acc, r0, v0);
        if (in1) lds_row_add<C, true>(acc, r1, v1);
    }
}

// ONE multisample point of a fine level into the workgroup's row block, by (y, z) COMBINATIONS (r03).  A hash scatters the four
// combinations of a point over four of the 32 row blocks, the x0 / x0 + 1 pair of a combination stays together: of the 8
// corners ~2 are this block's.  Walking all 8 with 22 % of the lanes active in each costs 8 x (hash, compare, weight, products,
// two LDS round trips) per wave; here a lane lists its in-block combinations first (4 row pairs, no weights) and the wave loops
// over "my next combination" -- as many rounds as the busiest lane has combinations (2-3), each with both corners' update in
// flight together.  Same addends ((w_k damp) g_c, w_k = ((wx wy) wz)) as point_scatter_block.
template <uint32_t C, bool HASHED, bool POW2, typename A>
__device__ __forceinline__ void point_scatter_combos(const UcnLevel &lv, A *__restrict__ acc, uint32_t row_lo, uint32_t nrows,
                                                     const float (&p)[3], float rsj, const float (&gout)[C]) {
    if (!in_unit_cube(p[0], p[1], p[2])) return;
    float fx = fmaf(p[0], lv.scale, 0.5f), fy = fmaf(p[1], lv.scale, 0.5f), fz = fmaf(p[2], lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    uint32_t ya, yb, za, zb, xa, xb;
    if constexpr (HASHED) {
        xa = x0; xb = x0 + 1u;
        ya = y0 * kP1; yb = ya + kP1;
        za = z0 * kP2; zb = za + kP2;
    } else {
        xa = x0 * lv.stride[0]; xb = xa + lv.stride[0];
        ya = y0 * lv.stride[1]; yb = ya + lv.stride[1];
        za = z0 * lv.stride[2]; zb = za + lv.stride[2];
    }
    auto row_of = [&](uint32_t xv, uint32_t yzv) -> uint32_t {
        uint32_t idx;
        if constexpr (HASHED) idx = xv ^ yzv; else idx = xv + yzv;
        if constexpr (POW2) return idx & lv.mask;
        else return idx < lv.rows ? idx : idx % lv.rows;
    };
    uint32_t pend = 0u;
    // hashed power-of-two levels whose row block is an aligned power of two above the resolution: x only reaches the bits below
    // the block index, both x corners of a combination lie in the block of (y P1 ^ z P2) -- one test per combination (wave-uniform switch)
    bool yz_decides = false;
    if constexpr (HASHED && POW2) yz_decides = (nrows & (nrows - 1u)) == 0u && (row_lo & (nrows - 1u)) == 0u && lv.resolution + 2u <= nrows;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t yv = (c & 1u) ? yb : ya, zv = (c & 2u) ? zb : za;
        uint32_t yz;
        if constexpr (HASHED) yz = yv ^ zv; else yz = yv + zv;
        bool hit;
        if (yz_decides) hit = (yz & lv.mask) - row_lo < nrows;
        else hit = (row_of(xa, yz) - row_lo < nrows) || (row_of(xb, yz) - row_lo < nrows);
        pend |= hit ? 1u << c : 0u;
    }
    const float damp = erf_pos(rsj * lv.inv_gs);
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
#ifdef UCN_EXP_NO_UPDATE                                                            // experiment build: everything but the LDS updates
    if (damp + gx + gy + gz != 12345.0f) pend &= (fx == 77.0f ? 15u : 0u);
#endif
    while (pend) {
        const uint32_t c = (uint32_t)__builtin_ctz(pend);
        pend &= pend - 1u;
        const uint32_t yv = (c & 1u) ? yb : ya, zv = (c & 2u) ? zb : za;
        const float wy = (c & 1u) ? fy : gy, wz = (c & 2u) ? fz : gz;
        uint32_t yz;
        if constexpr (HASHED) yz = yv ^ zv; else yz = yv + zv;
        const uint32_t r0 = row_of(xa, yz) - row_lo, r1 = row_of(xb, yz) - row_lo;
        const float w0 = ((gx * wy) * wz) * damp, w1 = ((fx * wy) * wz) * damp;
        float v0[C], v1[C];
#pragma unroll
        for (uint32_t cc = 0; cc < C; cc++) { v0[cc] = w0 * gout[cc]; v1[cc] = w1 * gout[cc]; }
        lds_row_add_pair<C>(acc, r0, r0 < nrows, v0, r1, r1 < nrows, v1);
    }
}

// Direct (no run merging) scatter of ONE multisample point into the workgroup's row block.
template <uint32_t C, bool HASHED, bool POW2, typename A>
__device__ __forceinline__ void point_scatter_block(const UcnLevel &lv, A *__restrict__ acc, uint32_t row_lo, uint32_t nrows,
                                                    const float (&p)[3], float rsj, const float (&gout)[C]) {
    if (!in_unit_cube(p[0], p[1], p[2])) return;
    float fx, fy, fz, w[8];
    uint32_t rows[8];
    corner_rows<HASHED, POW2>(lv, p[0], p[1], p[2], fx, fy, fz, rows);
    corner_weights(fx, fy, fz, w);
    const float damp = erf_pos(rsj * lv.inv_gs);
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t r = rows[k] - row_lo;
        if (r < nrows) {
            float v[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) v[c] = (w[k] * damp) * gout[c];
            lds_row_add<C, true>(acc, r, v);
        }
    }
}

template <uint32_t C>
__global__ __launch_bounds__(1024) void k_march_features_bwd_blk(UcnLevels lvls, float *__restrict__ grad_table,
                                                                 RayInputs in, HexPattern hx, float std_scale, uint32_t N,
                                                                 uint32_t S, GradStrides gs, uint32_t rpb,
                                                                 const float *__restrict__ grad_features,
                                                                 const float *__restrict__ geom) {
    extern __shared__ float s_acc[];
    uint32_t task = blockIdx.x, lvl = 0, nb = 1, split = 1;
    for (;; lvl++) {
        nb = (lvls.lv[lvl].rows + rpb - 1) / rpb;
        split = bwd_sample_split(nb);
        if (task < nb * split || lvl + 1 == lvls.L) break;
        task -= nb * split;
    }
    const UcnLevel lv = lvls.lv[lvl];
    const uint32_t row_lo = (task / split) * rpb, part = task % split;
    const uint32_t nrows = lv.rows - row_lo < rpb ? lv.rows - row_lo : rpb;
    for (uint32_t i = threadIdx.x; i < nrows * C; i += 1024u) s_acc[i] = 0.0f;
    __syncthreads();
    const size_t B = (size_t)N * S;
    // a level with few blocks (the dense coarse ones) is cut along the samples too: `split` workgroups per
    // block, interleaved in units of 1024 samples; they share the block, so their flush is atomic
    for (size_t b = (size_t)part * 1024u + threadIdx.x; b < B; b += (size_t)split * 1024u) {
        const float *gp = grad_features + lvl * gs.level + b * gs.sample;
        float gout[C];
        bool nz = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            gout[c] = gp[c * gs.chan] / 6.0f;                           // d(mean over the 6 multisamples)
            nz |= gout[c] != 0.0f;
        }
        if (!nz) continue;
        float u[6][3], rs[6];
        if (geom) {                                   // k_cast_cache's planes: every task re-reads, nobody re-derives
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) {
                const float4 q = reinterpret_cast<const float4 *>(geom)[b * 6 + j];
                u[j][0] = q.x; u[j][1] = q.y; u[j][2] = q.z; rs[j] = q.w;
            }
        } else {
            const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
            float csum[3], tsum;
            cast_sample(in, hx, std_scale, ray, s, S, u, rs, csum, tsum);
        }
        if (lv.hashed) {
            if (lv.mask && lv.resolution <= 2048u) level_scatter_block<C, true, true, true>(lv, s_acc, row_lo, nrows, u, rs, gout);
            else if (lv.mask) level_scatter_block<C, true, true, false>(lv, s_acc, row_lo, nrows, u, rs, gout);
            else level_scatter_block<C, true, false, false>(lv, s_acc, row_lo, nrows, u, rs, gout);
        } else if (lv.resolution <= 2048u) {
            if (lv.mask) level_scatter_block<C, false, true, true>(lv, s_acc, row_lo, nrows, u, rs, gout);
            else level_scatter_block<C, false, false, true>(lv, s_acc, row_lo, nrows, u, rs, gout);
        } else {                                   // the strided fine levels of the uint32-wrap quirk: nothing to merge
            if (lv.mask) level_scatter_block<C, false, true, false>(lv, s_acc, row_lo, nrows, u, rs, gout);
            else level_scatter_block<C, false, false, false>(lv, s_acc, row_lo, nrows, u, rs, gout);
        }
    }
    __syncthreads();
    float *gtab = grad_table + ((size_t)lv.first_row + row_lo) * C;
    for (uint32_t i = threadIdx.x; i < nrows * C; i += 1024u) {
        const float v = s_acc[i];
        if (v != 0.0f) {
            if (split == 1) gtab[i] += v;
            else atomicAdd(gtab + i, v);
        }
    }
}

// The same row-block ownership with COMPACTION.  A cell touches at most 8 of a level's 32 row blocks, a single
// multisample point of a fine level 7 on average: in the plain kernel 78 % of the hash/weight work of a
// workgroup is for corners that are not its own, and a per-lane "skip" does not help a 64-wide wave.  Here every
// wave reads the block masks of 64 samples at a time, appends the items that DO touch the block -- samples on
// coarse levels, (sample, multisample) pairs on fine ones -- to a ring in LDS (ballot + prefix count), and runs
// the expensive part on 64 dense items whenever the ring holds that many.
//
// Code shape matters as much as the algorithm here: the scan/process loop is ONE function template per addressing
// variant with ONE process site, everything force-inlined.  (With lambdas called from several unrolled sites the
// compiler outlined the scatter into a real function: the level descriptor then lived behind a flat pointer --
// a global load + vmcnt(0) in front of every point -- and the LDS base came from the dynamic-LDS offset table,
// an s_load per corner; that version spent 80 % of its time waiting on those.)
constexpr uint32_t kQueue = 512;                               // items per wave; 16 waves x 2 KiB beside the 128 KiB block

template <uint32_t C, bool HASHED, bool POW2, bool COARSE>
__device__ __forceinline__ void cmp_fetch(uint32_t item, bool valid, size_t B, const float *__restrict__ gl,
                                          const float *__restrict__ geom, float (&u)[6][3], float (&rs)[6], float (&gout)[C], float gscale) {
    const uint32_t b = valid ? item & 0x1FFFFFFFu : 0u, j = valid ? item >> 29 : 0u;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) gout[c] = gl[(size_t)b * C + c] * gscale;             // d(mean over the 6 multisamples): / 6 done by k_cast_cache_masks; x the task's power-of-two fixed-point scale (1 for float rows)
    if constexpr (COARSE) {
#pragma unroll
        for (uint32_t jj = 0; jj < 6; jj++) {
            const float4 q = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6 + jj];
            u[jj][0] = q.x; u[jj][1] = q.y; u[jj][2] = q.z; rs[jj] = q.w;
        }
    } else {
        const float4 q = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6 + j];
        u[0][0] = q.x; u[0][1] = q.y; u[0][2] = q.z; rs[0] = q.w;
    }
}

template <uint32_t C, bool HASHED, bool POW2, bool COARSE, bool RUNS, int FINE = 0, typename A = float>
__device__ __forceinline__ void cmp_block(const UcnLevel &lv, A *__restrict__ s_acc, uint32_t *__restrict__ q, uint32_t blk,
                                          uint32_t row_lo, uint32_t nrows, uint32_t part, uint32_t split, size_t B,
                                          const uint32_t *__restrict__ mp, const float *__restrict__ gl,
                                          const float *__restrict__ geom, float gscale = 1.0f) {
    // one mask word per sample: coarse levels bit `blk`; fine levels the word of this block's group of four,
    // bit 4 * j + (blk & 3) for multisample j (k_cast_cache_masks)
    constexpr uint32_t P = COARSE ? 1u : 6u;                                  // items a sample can contribute
    const uint32_t bit0 = COARSE ? (blk & 31u) : (blk & 3u);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t head = 0, tail = 0;                                              // wave-uniform ring positions
    const size_t stride = (size_t)split * kScan * 1024u;
    size_t base = (size_t)part * kScan * 1024u;
    uint32_t cur[kScan], nxt[kScan];
#pragma unroll
    for (uint32_t u = 0; u < kScan; u++) {
        const size_t b = base + u * 1024u + threadIdx.x;
        cur[u] = b < B ? mp[b] : 0u;
    }
    // `split` workgroups share a block; they take the samples in interleaved units of kScan x 1024 (flush: atomic).
    // One workgroup per CU: the masks of the NEXT unit are requested before this unit's items are processed.
    uint32_t u = 0;
    bool more = base < B;
    while (more || tail != head) {
        if (more) {
            if (u == 0) {
                const size_t nb = base + stride;
#pragma unroll
                for (uint32_t uu = 0; uu < kScan; uu++) {
                    const size_t b = nb + uu * 1024u + threadIdx.x;
                    nxt[uu] = b < B ? mp[b] : 0u;
                }
            }
            const uint32_t b = (uint32_t)(base + u * 1024u + threadIdx.x);
            uint32_t m = cur[0];                                              // u is wave-uniform: selects, no scratch
#pragma unroll
            for (uint32_t uu = 1; uu < kScan; uu++) m = u == uu ? cur[uu] : m;
            m >>= bit0;
            if constexpr (COARSE || FINE == 0) {
#pragma unroll
                for (uint32_t j = 0; j < P; j++) {
                    const bool act = (m >> (4u * j)) & 1u;
                    const uint64_t bal = __ballot(act);
                    const uint32_t pos = tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (act) q[pos & (kQueue - 1u)] = b | (j << 29);
                    tail += (uint32_t)__popcll(bal);
                }
            } else {
                // a sample hits this block with 0.84 of its 6 points on average: count per lane, one DPP prefix sum per step,
                // then every lane writes its own items (as many rounds as the busiest lane has hits, ~3) -- six ballot /
                // mbcnt / predicated-write sections per step were a third of a fine level's time (scan alone: 166 of 490 us)
                m &= 0x111111u;
                const uint32_t cnt = (uint32_t)__popc(m);
                const uint32_t incl = wave_scan_dpp<uint32_t>(cnt);
                uint32_t pos = tail + incl - cnt;
                tail += wave_last<uint32_t>(incl);
                while (m) {
                    const uint32_t bit = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    q[pos & (kQueue - 1u)] = b | (bit << 27);                    // bit = 4 j: j lands in bits 29..31
                    pos++;
                }
            }
            if (++u == kScan) {
                u = 0;
                base += stride;
                more = base < B;
#pragma unroll
                for (uint32_t uu = 0; uu < kScan; uu++) cur[uu] = nxt[uu];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // Ring: <= kPer * 64 - 1 left over + <= 384 (fine) / 64 (coarse) appended per step <= kQueue.
        constexpr uint32_t kPer = RUNS ? 6u : 2u;                              // ring items per lane and round
        const uint32_t thr = more ? kPer * 64u : 1u;
        while (tail - head >= thr && tail != head) {
            const uint32_t avail = tail - head < kPer * 64u ? tail - head : kPer * 64u;
            if constexpr (RUNS) {
                // the coarsest levels (every sample of the ring is a neighbour of the previous one):
                // a lane walks kPer CONSECUTIVE ring items (neighbouring samples of a ray) and keeps merging runs
                // across them; the next item's geometry is requested before the current one is scattered
                RowRun<C> run;
                run.have = false;
                float un[6][3], rsn[6], gn[C], glast[C];               // glast: the gradient of the sample the open run belongs to (C <= 2: unused by the flush)
#pragma unroll
                for (uint32_t c = 0; c < C; c++) glast[c] = 0.0f;
                cmp_fetch<C, HASHED, POW2, true>(q[(head + kPer * lane) & (kQueue - 1u)], kPer * lane < avail, B, gl, geom, un, rsn, gn, gscale);
#pragma unroll 1
                for (uint32_t k = 0; k < kPer; k++) {
                    float uc[6][3], rsc[6], gc[C];
#pragma unroll
                    for (uint32_t j = 0; j < 6; j++) {
#pragma unroll
                        for (uint32_t d = 0; d < 3; d++) uc[j][d] = un[j][d];
                        rsc[j] = rsn[j];
                    }
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) gc[c] = gn[c];
                    const uint32_t idx = kPer * lane + k;
                    if (k + 1 < kPer)
                        cmp_fetch<C, HASHED, POW2, true>(q[(head + idx + 1u) & (kQueue - 1u)], idx + 1u < avail, B, gl, geom, un, rsn, gn, gscale);
                    if (idx < avail) run_merge_sample<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, uc, rsc, gc, run);
                }
                if (run.have) run_flush<C>(s_acc, row_lo, nrows, run, glast);   // (lean runs are closed by run_merge_sample: never open here)
            } else {
                // two items per lane: both items' loads are in flight before the first scatter starts
#ifdef UCN_EXP_SCAN_ONLY                                                    // experiment build: what the mask scan alone costs
                if (!RUNS) { head += avail; continue; }
#endif
                const uint32_t i0 = q[(head + lane) & (kQueue - 1u)], i1 = q[(head + 64u + lane) & (kQueue - 1u)];
                const bool v0 = lane < avail, v1 = lane + 64u < avail;
                float u0[6][3], rs0[6], g0[C], u1[6][3], rs1[6], g1[C];
                cmp_fetch<C, HASHED, POW2, COARSE>(i0, v0, B, gl, geom, u0, rs0, g0, gscale);
                cmp_fetch<C, HASHED, POW2, COARSE>(i1, v1, B, gl, geom, u1, rs1, g1, gscale);
                if (v0) {
                    if constexpr (COARSE) level_scatter_block<C, HASHED, POW2, FINE == 0>(lv, s_acc, row_lo, nrows, u0, rs0, g0);
                    else if constexpr (FINE == 2) point_scatter_combos<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u0[0], rs0[0], g0);
                    else point_scatter_block<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u0[0], rs0[0], g0);
                }
                if (v1) {
                    if constexpr (COARSE) level_scatter_block<C, HASHED, POW2, FINE == 0>(lv, s_acc, row_lo, nrows, u1, rs1, g1);
                    else if constexpr (FINE == 2) point_scatter_combos<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u1[0], rs1[0], g1);
                    else point_scatter_block<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u1[0], rs1[0], g1);
                }
            }
            head += avail;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Point-item levels on BYTE PLANES (MaskPlan::fine_kind 3; r06, VERDICT r05 item 2 b).  `mb` = this row block's plane: one byte per
// sample, bit j = multisample j has a corner in the block (k_cast_cache_masks).  A lane reads ONE dword = four consecutive samples
// per unit of 4096 (cmp_block: four dwords, one sample each, 6 useful bits of 32), counts its hits, the wave takes one DPP prefix
// sum and every lane appends its own items; (y, z)-combination scatter as in shape 2.  A lane can hold up to 24 hits: when the
// wave's count would not fit the ring, only every lane's lowest non-empty byte (<= 6 hits, <= 384 per wave: the bound cmp_block
// lives with) is taken in this round and the rest of the word stays for the next one.
template <uint32_t C, bool HASHED, bool POW2, bool COMBOS, typename A>
__device__ __forceinline__ void cmp_block_bytes(const UcnLevel &lv, A *__restrict__ s_acc, uint32_t *__restrict__ q, uint32_t row_lo,
                                                uint32_t nrows, uint32_t part, uint32_t split, size_t B,
                                                const uint8_t *__restrict__ mb, const float *__restrict__ gl,
                                                const float *__restrict__ geom, float gscale) {
    static_assert(kScan == 4u, "a unit is 1024 dwords of four sample bytes");
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t head = 0, tail = 0;
    const size_t stride = (size_t)split * 4096u;
    size_t base = (size_t)part * 4096u;
    bool more = base < B, fresh = true;
    uint32_t cur = 0u, nxt = 0u;
    {
        const size_t b = base + 4u * threadIdx.x;
        if (b < B) cur = *reinterpret_cast<const uint32_t *>(mb + b) & 0x3f3f3f3fu;
    }
    while (more || tail != head) {
        if (more) {
            if (fresh) {                                                      // the next unit's word: in flight under this unit's items
                const size_t b = base + stride + 4u * threadIdx.x;
                nxt = b < B ? *reinterpret_cast<const uint32_t *>(mb + b) & 0x3f3f3f3fu : 0u;
                fresh = false;
            }
            uint32_t take = cur;
            uint32_t cnt = (uint32_t)__popc(take);
            uint32_t incl = wave_scan_dpp<uint32_t>(cnt);
            uint32_t tot = wave_last<uint32_t>(incl);
            if (tail - head + tot > kQueue) {                                 // (wave-uniform, rare: 64 x 3.4 hits expected)
                take = cur ? cur & (0xffu << ((uint32_t)__builtin_ctz(cur) & 24u)) : 0u;
                cnt = (uint32_t)__popc(take);
                incl = wave_scan_dpp<uint32_t>(cnt);
                tot = wave_last<uint32_t>(incl);
            }
            uint32_t pos = tail + incl - cnt;
            tail += tot;
            cur &= ~take;
            const uint32_t b4 = (uint32_t)base + 4u * threadIdx.x;
            while (take) {
                const uint32_t bit = (uint32_t)__builtin_ctz(take);
                take &= take - 1u;
                q[pos & (kQueue - 1u)] = (b4 + (bit >> 3)) | (bit << 29);       // j = bit & 7 < 6: bits 29..31 (the sample's bits fall off the top)
                pos++;
            }
            if (__ballot(cur != 0u) == 0ull) {
                base += stride;
                more = base < B;
                cur = nxt;
                fresh = true;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // Ring: <= 127 left over + a fitting round (checked above) or <= 384 appended <= kQueue.
        const uint32_t thr = more ? 128u : 1u;
        while (tail - head >= thr && tail != head) {
            const uint32_t avail = tail - head < 128u ? tail - head : 128u;
            const uint32_t i0 = q[(head + lane) & (kQueue - 1u)], i1 = q[(head + 64u + lane) & (kQueue - 1u)];
            const bool v0 = lane < avail, v1 = lane + 64u < avail;
            float u0[6][3], rs0[6], g0[C], u1[6][3], rs1[6], g1[C];
            cmp_fetch<C, HASHED, POW2, false>(i0, v0, B, gl, geom, u0, rs0, g0, gscale);
            cmp_fetch<C, HASHED, POW2, false>(i1, v1, B, gl, geom, u1, rs1, g1, gscale);
            if (v0) {
                if constexpr (COMBOS) point_scatter_combos<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u0[0], rs0[0], g0);
                else point_scatter_block<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u0[0], rs0[0], g0);
            }
            if (v1) {
                if constexpr (COMBOS) point_scatter_combos<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u1[0], rs1[0], g1);
                else point_scatter_block<C, HASHED, POW2>(lv, s_acc, row_lo, nrows, u1[0], rs1[0], g1);
            }
            head += avail;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Levels of MORE THAN 32 ROW BLOCKS (the reference's own waymo.gin grid: T = 2^21 rows of C = 4 = 256 blocks of 8192 rows).
// The masks are one bit per (sample, block); a hit sample has a corner pair in this block with only ~1 of its 6 points
// (a point's four (y, z) pairs lie in ~4.5 of the 256 blocks), so walking all 48 corners of every hit sample keeps 17 % of the
// lanes busy in the expensive part (measured: 500 us per workgroup of 262 144 samples, 75 of them the scan).  Two rings per wave
// instead: ring 1 takes the hit SAMPLES from the scan; stage 1 runs 64 of them, point by point -- geometry, cell, the four
// pair rows, which of them are mine -- and appends (sample, pair, point set) items to ring 2 (popcount + one DPP prefix sum);
// stage 2 runs dense batches of ring 2: per point of the set one hash, two weights, one erf; one paired compare-and-swap update per item.
// Same addends as everywhere ((w_k damp) g_c, w_k = ((wx wy) wz)).
constexpr uint32_t kRing1 = 128, kRing2 = kQueue - kRing1;                    // 128 + 384 words of a wave's 2 KiB

template <uint32_t C, bool HASHED, bool POW2>
__device__ __forceinline__ void wide_pair_rows(const UcnLevel &lv, const float4 &g, float &fx, float &fy, float &fz,
                                               uint32_t &xa, uint32_t &xb, uint32_t (&yz)[4]) {
    fx = fmaf(g.x, lv.scale, 0.5f); fy = fmaf(g.y, lv.scale, 0.5f); fz = fmaf(g.z, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    uint32_t ya, yb, za, zb;
    if constexpr (HASHED) {
        xa = x0; xb = x0 + 1u;
        ya = y0 * kP1; yb = ya + kP1;
        za = z0 * kP2; zb = za + kP2;
        yz[0] = ya ^ za; yz[1] = yb ^ za; yz[2] = ya ^ zb; yz[3] = yb ^ zb;
    } else {
        xa = x0 * lv.stride[0]; xb = xa + lv.stride[0];
        ya = y0 * lv.stride[1]; yb = ya + lv.stride[1];
        za = z0 * lv.stride[2]; zb = za + lv.stride[2];
        yz[0] = ya + za; yz[1] = yb + za; yz[2] = ya + zb; yz[3] = yb + zb;
    }
}
template <bool HASHED, bool POW2>
__device__ __forceinline__ uint32_t wide_row(const UcnLevel &lv, uint32_t xv, uint32_t yzv) {
    uint32_t idx;
    if constexpr (HASHED) idx = xv ^ yzv; else idx = xv + yzv;
    if constexpr (POW2) return idx & lv.mask;
    else return idx < lv.rows ? idx : idx % lv.rows;
}

// stage 2 of wide_block: dense (sample | pair << 24 | point set << 26) items of ring 2: the points of the set share ONE lattice cell,
// so the pair's two rows are theirs in common -- their addends are summed in registers and go to the block in one paired update
// (on the coarser levels all six points of a sample share the cell: six times fewer LDS updates, and none of the same-row
// collisions that consecutive points of a sample would cause).  (A function, force-inlined at its two call sites: a lambda
// called from several sites gets outlined by hipcc -- see cmp_block.)
template <uint32_t C, bool HASHED, bool POW2, typename A>
__device__ __forceinline__ void wide_drain(const UcnLevel &lv, A *__restrict__ s_acc, const uint32_t *__restrict__ q2, uint32_t &head2,
                                           uint32_t tail2, uint32_t thr, uint32_t lane, uint32_t row_lo, uint32_t nrows,
                                           const float *__restrict__ gl, const float *__restrict__ geom, float gscale) {
    while (tail2 - head2 >= thr && tail2 != head2) {
        const uint32_t avail = tail2 - head2 < 64u ? tail2 - head2 : 64u;
        const bool v = lane < avail;
        const uint32_t it = v ? q2[(head2 + lane) % kRing2] : 0u;
        const uint32_t b = it & 0xFFFFFFu, c = (it >> 24) & 3u;
        uint32_t pts = it >> 26;
        float gout[C], v0[C], v1[C];
#pragma unroll
        for (uint32_t cc = 0; cc < C; cc++) { gout[cc] = gl[(size_t)b * C + cc] * gscale; v0[cc] = 0.0f; v1[cc] = 0.0f; }
        uint32_t r0 = 0u, r1 = 0u;
        while (pts) {
            const uint32_t j = (uint32_t)__builtin_ctz(pts);
            pts &= pts - 1u;
            const float4 g = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6 + j];
            float fx, fy, fz;
            uint32_t xa, xb, yz[4];
            wide_pair_rows<C, HASHED, POW2>(lv, g, fx, fy, fz, xa, xb, yz);
            const uint32_t h = c == 0u ? yz[0] : c == 1u ? yz[1] : c == 2u ? yz[2] : yz[3];
            const float wy = (c & 1u) ? fy : 1.0f - fy, wz = (c & 2u) ? fz : 1.0f - fz;
            r0 = wide_row<HASHED, POW2>(lv, xa, h) - row_lo; r1 = wide_row<HASHED, POW2>(lv, xb, h) - row_lo;   // (the same for every point of the set)
            const float damp = erf_pos(g.w * lv.inv_gs);
            const float w0 = (((1.0f - fx) * wy) * wz) * damp, w1 = ((fx * wy) * wz) * damp;
#pragma unroll
            for (uint32_t cc = 0; cc < C; cc++) { v0[cc] += w0 * gout[cc]; v1[cc] += w1 * gout[cc]; }
        }
        if (v) lds_row_add_pair<C>(s_acc, r0, r0 < nrows, v0, r1, r1 < nrows, v1);
        head2 += avail;
    }
}

template <uint32_t K>
__device__ __forceinline__ void wide_load_group(const uint32_t *__restrict__ T, size_t first, size_t wstride, size_t B64,
                                                uint32_t (&lo)[K], uint32_t (&hi)[K]) {
#pragma unroll
    for (uint32_t i = 0; i < K; i++) {
        const size_t w = first + i * wstride;
        lo[i] = w < B64 ? T[w * 2u] : 0u;
        hi[i] = w < B64 ? T[w * 2u + 1u] : 0u;
    }
}

template <uint32_t C, bool HASHED, bool POW2, typename A>
__device__ __forceinline__ void wide_block(const UcnLevel &lv, A *__restrict__ s_acc, uint32_t *__restrict__ q, uint32_t blk,
                                           uint32_t row_lo, uint32_t nrows, uint32_t part, uint32_t split, size_t B,
                                           const uint32_t *__restrict__ mp, const float *__restrict__ gl,
                                           const float *__restrict__ geom, float gscale) {
    uint32_t *q1 = q, *q2 = q + kRing1;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t head1 = 0, tail1 = 0, head2 = 0, tail2 = 0;                      // wave-uniform ring positions
    // the block's bit plane (k_cast_cache_masks): bit i of 64-bit word w = sample 64 w + i.  A lane takes the words
    // part + split (1024 k + thread), k = 0, 1, ...: kWords of them are loaded together, the next group is requested before this
    // one is searched.  One round = every lane with bits left hands over its lowest one (<= 64 samples into ring 1).
    constexpr uint32_t kWords = 4;
    const size_t B64 = (B + 63u) / 64u;
    const uint32_t *T = mp + (size_t)blk * B64 * 2u;
    const size_t wstride = (size_t)split * 1024u, w0 = (size_t)part + (size_t)split * threadIdx.x;
    uint32_t clo[kWords], chi[kWords], nlo[kWords], nhi[kWords];
    size_t gfirst = w0;                                                        // this lane's first word of the current group
    wide_load_group<kWords>(T, gfirst, wstride, B64, clo, chi);
    bool more = (size_t)part + (size_t)split * (threadIdx.x & ~63u) < B64;     // wave-uniform: the wave's first word exists
    uint32_t wi = 0;                                                           // word of the group being searched
    uint32_t lo = 0u, hi = 0u;
    bool fresh = true;                                                         // take the next word of the group
    while (more || tail1 != head1) {
        if (more) {
            if (fresh) {
                if (wi == 0) wide_load_group<kWords>(T, gfirst + kWords * wstride, wstride, B64, nlo, nhi);
                lo = clo[0]; hi = chi[0];
#pragma unroll
                for (uint32_t i = 1; i < kWords; i++) { lo = wi == i ? clo[i] : lo; hi = wi == i ? chi[i] : hi; }
                fresh = false;
            }
            const bool act = (lo | hi) != 0u;
            uint32_t bit = lo ? (uint32_t)__builtin_ctz(lo) : 32u + (uint32_t)__builtin_ctz(hi | 0x80000000u * (hi == 0u));
            const size_t w = gfirst + wi * wstride;
            const uint32_t b = (uint32_t)(w * 64u + bit);
            if (lo) lo &= lo - 1u; else hi &= hi - 1u;
            const uint64_t bal = __ballot(act);
            const uint32_t pos = tail1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (act) q1[pos % kRing1] = b;
            tail1 += (uint32_t)__popcll(bal);
            if (__ballot((lo | hi) != 0u) == 0ull) {                           // every lane is through with its word
                fresh = true;
                if (++wi == kWords) {
                    wi = 0;
                    gfirst += kWords * wstride;
#pragma unroll
                    for (uint32_t i = 0; i < kWords; i++) { clo[i] = nlo[i]; chi[i] = nhi[i]; }
                    // the first lane's first word of the new group decides for the wave (its words come first)
                    more = (size_t)part + (size_t)split * (threadIdx.x & ~63u) + (gfirst - w0) < B64;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // stage 1: <= 63 samples left over + <= 64 appended per round <= kRing1
        const uint32_t thr1 = more ? 64u : 1u;
        while (tail1 - head1 >= thr1 && tail1 != head1) {
            const uint32_t avail = tail1 - head1 < 64u ? tail1 - head1 : 64u;
#ifdef UCN_EXP_SCAN_ONLY                                                        // experiment build: the mask scan alone
            head1 += avail;
            continue;
#endif
            const bool v = lane < avail;
            const uint32_t b = v ? q1[(head1 + lane) % kRing1] : 0u;
            // the next point's geometry is requested before this one is tested (a round is otherwise one exposed load latency).
            // Consecutive points in one lattice cell form a RUN: its in-block pairs become items only when the cell changes
            // (or after the sixth point: round 6 flushes), carrying the set of its points.
            float4 gn = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6];
            uint32_t run_x = 0u, run_y = 0u, run_z = 0u, run_pend = 0u, run_pts = 0u;
#pragma unroll 1
            for (uint32_t j = 0; j < 7; j++) {
                const float4 g = gn;
                if (j + 1u < 6u) gn = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6 + j + 1u];
                uint32_t flush = j == 6u ? run_pend : 0u;
                if (j < 6u && v && in_unit_cube(g.x, g.y, g.z)) {
                    float fx, fy, fz;
                    uint32_t xa, xb, yz[4];
                    wide_pair_rows<C, HASHED, POW2>(lv, g, fx, fy, fz, xa, xb, yz);
                    const uint32_t cx = (uint32_t)floorf(fmaf(g.x, lv.scale, 0.5f)), cy = (uint32_t)floorf(fmaf(g.y, lv.scale, 0.5f)),
                                   cz = (uint32_t)floorf(fmaf(g.z, lv.scale, 0.5f));
                    if (run_pts != 0u && cx == run_x && cy == run_y && cz == run_z) {
                        run_pts |= 1u << j;
                    } else {
                        flush = run_pend;                                       // (0 for the first point)
                        uint32_t pend = 0u;
#pragma unroll
                        for (uint32_t c = 0; c < 4; c++) {
                            const bool hit = (wide_row<HASHED, POW2>(lv, xa, yz[c]) - row_lo < nrows) || (wide_row<HASHED, POW2>(lv, xb, yz[c]) - row_lo < nrows);
                            pend |= hit ? 1u << c : 0u;
                        }
                        // the items of the run that ends here carry ITS point set: emit below, then start the new run
                        const uint32_t old_pts = run_pts;
                        run_x = cx; run_y = cy; run_z = cz; run_pend = pend;
                        run_pts = (1u << j) | (old_pts << 8);                   // bits 8..13: the ended run's points, until the emit
                    }
                }
                const uint32_t emit_pts = j == 6u ? run_pts & 63u : run_pts >> 8;
                run_pts &= 63u;
                const uint32_t cnt = (uint32_t)__popc(flush);
                const uint32_t incl = wave_scan_dpp<uint32_t>(cnt);
                uint32_t pos = tail2 + incl - cnt;
                tail2 += wave_last<uint32_t>(incl);
                while (flush) {
                    const uint32_t c = (uint32_t)__builtin_ctz(flush);
                    flush &= flush - 1u;
                    q2[pos % kRing2] = b | (c << 24) | (emit_pts << 26);
                    pos++;
                }
                __builtin_amdgcn_wave_barrier();
#ifdef UCN_EXP_NO_UPDATE                                                        // experiment build: scan + stage 1, no stage 2
                head2 = tail2;
#endif
                // ring 2: <= 63 left over + <= 256 appended per round <= kRing2
                wide_drain<C, HASHED, POW2>(lv, s_acc, q2, head2, tail2, 64u, lane, row_lo, nrows, gl, geom, gscale);
            }
            head1 += avail;
        }
        __builtin_amdgcn_wave_barrier();
    }
    wide_drain<C, HASHED, POW2>(lv, s_acc, q2, head2, tail2, 1u, lane, row_lo, nrows, gl, geom, gscale);
}

#ifdef UCN_WG_CLOCK                                    // tools/bwd_balance.py: start / end time of every workgroup (experiment builds only)
__device__ uint64_t g_wg_clock[8192][3];
#endif
// PERSISTENT workgroups (r03): one per CU, tasks (level, row block, sample part) pulled from a counter in the plan's
// longest-first order.  With one workgroup per task the hardware dispatcher hands workgroup k to XCD k mod 8 IN ORDER: where
// task times differ (45 ... 1010 us inside the uneven levels) a full XCD blocks the dispatch for all eight -- the workgroup
// clocks of the benchmark grid showed 164-208 of 256 CUs busy behind the uneven levels and a 1.3 ms drain at the end
// (tools/bwd_balance.py: 4.26 ms where the workgroup times sum to 3.70 ms per CU).
template <uint32_t C, bool FX = false>
__global__ __launch_bounds__(1024) void k_march_features_bwd_cmp(UcnLevels lvls, float *__restrict__ grad_table, uint32_t N,
                                                                 uint32_t S, uint32_t rpb, MaskPlan plan,
                                                                 const float *__restrict__ grad_features /*[L][N*S][C]*/,
                                                                 const float *__restrict__ geom,
                                                                 const uint32_t *__restrict__ masks,
                                                                 uint32_t *__restrict__ counter, uint32_t total,
                                                                 const float *__restrict__ l1_partial /*FX: [L][ceil(B / 256)]*/) {
    extern __shared__ float s_acc[];                      // 128 KiB row block + 16 rings of 2 KiB: all of the CU's 160 KiB
    using Acc = typename std::conditional<FX, FxLane, float>::type;
    Acc *acc_rows = reinterpret_cast<Acc *>(s_acc);       // FX: the same bytes as int32 fixed-point channel pairs (0.0f == 0)
    for (uint32_t i = threadIdx.x; i < rpb * C; i += 1024u) s_acc[i] = 0.0f;          // a flush leaves zeros behind
    uint32_t *q = reinterpret_cast<uint32_t *>(s_acc + (size_t)rpb * C) + (threadIdx.x >> 6) * kQueue;
    volatile uint32_t *s_task = reinterpret_cast<uint32_t *>(s_acc + (size_t)rpb * C);   // = wave 0's ring, idle between tasks
    const size_t B = (size_t)N * S;
    // Eight queues, one per XCD: queue y holds the tasks k = y (mod 8), i.e. exactly the workgroups the in-order dispatcher
    // would have placed on XCD y.  That keeps what the static launch had for free: the `split` parts of all row blocks of a
    // level that scan the SAME samples (same mask words, geometry, gradients) run on the same XCD and share its L2 -- with one
    // global queue the point-item levels' workgroups took 650 us instead of 480.  An XCD whose queue is dry steals.
    const uint32_t home = __builtin_amdgcn_s_getreg(20 /*HW_REG_XCC_ID*/ | (3u << 11)) & 7u;
    for (;;) {
        if (threadIdx.x == 0) {
            uint32_t t = total;
            for (uint32_t a = 0; a < 8u; a++) {
                const uint32_t y = (home + a) & 7u;
                const uint32_t k = atomicAdd(counter + y, 1u) * 8u + y;
                if (k < total) { t = k; break; }
            }
            *s_task = t;
        }
        __syncthreads();                                                              // (also: zeros / flush of s_acc are done)
        const uint32_t task0 = __builtin_amdgcn_readfirstlane(*s_task);               // wave-uniform: everything derived stays scalar
        __syncthreads();                                                              // before wave 0 refills its ring
        if (task0 >= total) break;
#ifdef UCN_WG_CLOCK
        if (threadIdx.x == 0 && task0 < 8192u) g_wg_clock[task0][0] = wall_clock64();
#endif
        uint32_t task = task0, lvl = 0, nb = 1, split = 1;
        for (uint32_t i = 0;; i++) {
            lvl = plan.order[i];
            nb = (lvls.lv[lvl].rows + rpb - 1) / rpb;
            split = plan.split[lvl];
            const uint32_t here = (plan.skip_fine && !plan.coarse[lvl]) ? 0u : nb * split;
            if (task < here || i + 1 == lvls.L) break;                               // (`total` is the sum of `here`)
            task -= here;
        }
        const UcnLevel lv = lvls.lv[lvl];
        const uint32_t blk = task / split, part = task % split;
        const uint32_t row_lo = blk * rpb;
        const uint32_t nrows = lv.rows - row_lo < rpb ? lv.rows - row_lo : rpb;
        const uint32_t *mp = masks + (size_t)(plan.plane[lvl] + (plan.coarse[lvl] ? blk >> 5 : blk >> 2)) * B;
        const float *gl = grad_features + (size_t)lvl * B * C;
        float gscale = 1.0f, ginv = 1.0f;
        if constexpr (FX) {
            // the task's bound: L1 = sum over ITS samples of max_c |g| (256-sample partials of the mask pass; the sample-item
            // and point-item shapes take the units of kScan x 1024 samples with unit % split == part, the wide shape interleaves
            // 64-sample words and is given the whole level's sum), then the power of two that puts L1 at <= 2^30
            const uint32_t nblk = (uint32_t)((B + 255u) / 256u);
            const float *lp = l1_partial + (size_t)lvl * nblk;
            const bool every = plan.coarse[lvl] == 3 || split == 1u;
            float mine = 0.0f;
            for (uint32_t i = threadIdx.x; i < nblk; i += 1024u)
                if (every || (i / (kScan * 4u)) % split == part) mine += lp[i];
            mine = wave_sum_dpp<float>(mine);
            volatile float *s_red = reinterpret_cast<volatile float *>(s_acc + (size_t)rpb * C) + 16;      // wave 0's ring, idle here
            if ((threadIdx.x & 63u) == 0u) s_red[threadIdx.x >> 6] = mine;
            __syncthreads();
            float l1 = 0.0f;
#pragma unroll
            for (uint32_t w = 0; w < 16u; w++) l1 += s_red[w];                     // fixed order, every thread the same value
            __syncthreads();                                                       // before the rings are used again
            l1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, l1)));
            // (a level whose resolution^2 wrapped negative in int32 has a NaN damping factor, models.py:495: NaN rows in the reference and on
            // the float route -- a NaN addend would pack as 0 here, so the task is poisoned like one with a non-finite gradient)
            const bool nan_level = lv.inv_gs != lv.inv_gs;
            if (!nan_level && l1 > 0.0f && l1 <= 3.0e38f) {
                int x;
                (void)frexpf(l1, &x);                                              // l1 = m 2^x, m in [0.5, 1): l1 <= 2^x
                int e = 30 - x;
                e = e > 120 ? 120 : (e < -100 ? -100 : e);                          // x <= 128 (l1 <= 3e38): e >= -98, never clamped from below
                gscale = ldexpf(1.0f, e);
                ginv = ldexpf(1.0f, -e);
            } else if (nan_level || !(l1 <= 3.0e38f)) {
                ginv = __builtin_nanf("");                                         // a non-finite gradient on this level: every row of the task's block becomes NaN (flush below)
            }
        }
#define UCN_CMP(H, P2, CO, RU) cmp_block<C, H, P2, CO, RU>(lv, acc_rows, q, blk, row_lo, nrows, part, split, B, mp, gl, geom, gscale)
#define UCN_CMPF(H, P2)                                                                                                      \
    do {                                                                                                                     \
        if (plan.fine_kind[lvl] == 3) cmp_block_bytes<C, H, P2, true>(lv, acc_rows, q, row_lo, nrows, part, split, B, reinterpret_cast<const uint8_t *>(masks + (size_t)plan.plane[lvl] * B) + (size_t)blk * B, gl, geom, gscale); \
        else if (plan.fine_kind[lvl] == 4) cmp_block_bytes<C, H, P2, false>(lv, acc_rows, q, row_lo, nrows, part, split, B, reinterpret_cast<const uint8_t *>(masks + (size_t)plan.plane[lvl] * B) + (size_t)blk * B, gl, geom, gscale); \
        else if (plan.fine_kind[lvl] == 2) cmp_block<C, H, P2, false, false, 2>(lv, acc_rows, q, blk, row_lo, nrows, part, split, B, mp, gl, geom, gscale);      \
        else if (plan.fine_kind[lvl] == 1) cmp_block<C, H, P2, false, false, 1>(lv, acc_rows, q, blk, row_lo, nrows, part, split, B, mp, gl, geom, gscale); \
        else cmp_block<C, H, P2, false, false, 0>(lv, acc_rows, q, blk, row_lo, nrows, part, split, B, mp, gl, geom, gscale);           \
    } while (0)
        if (plan.coarse[lvl] == 2) {                                          // all workgroup-uniform; the coarsest
            if (lv.mask) UCN_CMP(false, true, true, true);                    // levels are never hashed
            else UCN_CMP(false, false, true, true);
        } else if (plan.coarse[lvl] == 3) {                                   // > 32 row blocks: sample items, unmerged scatter
#define UCN_CMPW(H, P2) wide_block<C, H, P2>(lv, acc_rows, q, blk, row_lo, nrows, part, split, B, masks + (size_t)plan.plane[lvl] * B, gl, geom, gscale)
            if (lv.hashed) { if (lv.mask) UCN_CMPW(true, true); else UCN_CMPW(true, false); }
            else { if (lv.mask) UCN_CMPW(false, true); else UCN_CMPW(false, false); }
#undef UCN_CMPW
        } else if (plan.coarse[lvl]) {
            if (lv.hashed) { if (lv.mask) UCN_CMP(true, true, true, false); else UCN_CMP(true, false, true, false); }
            else { if (lv.mask) UCN_CMP(false, true, true, false); else UCN_CMP(false, false, true, false); }
        } else {
            if (lv.hashed) { if (lv.mask) UCN_CMPF(true, true); else UCN_CMPF(true, false); }
            else { if (lv.mask) UCN_CMPF(false, true); else UCN_CMPF(false, false); }
        }
#undef UCN_CMP
#undef UCN_CMPF
        __syncthreads();
        float *gtab = grad_table + ((size_t)lv.first_row + row_lo) * C;
        if constexpr (FX) {
            unsigned long long *words = reinterpret_cast<unsigned long long *>(s_acc);
            for (uint32_t i = threadIdx.x; i < nrows * C / 2u; i += 1024u) {
                const unsigned long long w = words[i];
                if (w != 0ull || ginv != ginv) {                 // poisoned task: a NaN addend packs as 0, so untouched-looking rows are flushed too
                    words[i] = 0ull;
                    float a, b;
                    fixed_unpack(w, ginv, a, b);
                    if (split == 1) { gtab[2u * i] += a; gtab[2u * i + 1u] += b; }
                    else { atomicAdd(gtab + 2u * i, a); atomicAdd(gtab + 2u * i + 1u, b); }
                }
            }
        } else {
        for (uint32_t i = threadIdx.x; i < nrows * C; i += 1024u) {
            const float v = s_acc[i];
            if (v != 0.0f) {
                s_acc[i] = 0.0f;
                if (split == 1) gtab[i] += v;
                else atomicAdd(gtab + i, v);
            }
        }
        }
#ifdef UCN_WG_CLOCK
        __syncthreads();
        if (threadIdx.x == 0 && task0 < 8192u) { g_wg_clock[task0][1] = wall_clock64(); g_wg_clock[task0][2] = lvl; }
#endif
    }
}
#ifdef UCN_WG_CLOCK
extern "C" int ucn_debug_wg_clock(uint64_t *host, uint32_t n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_clock), (size_t)n * 24u) == hipSuccess ? 0 : 1;
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Fine levels by ITEM LISTS (r03).  In the compacted kernel above every workgroup of a row block scans the block masks of
// ALL samples (32 blocks per level: the scan is replicated 32x) and a hit re-derives all eight corners of the point although
// only the ~2 of one (y, z) combination lie in the block (a hash scatters the four combinations over ~4 of the 32 blocks:
// 4.5 hits per point).  Here the points are visited ONCE per level: every (point, (y, z) combination) becomes one 32-bit item
// (sample | multisample << 27 | combination << 30) routed to the list of the block that owns the combination's x0 row --
// a counting sort in two passes (k_bwd_bin<false> counts per (level, block), k_bwd_bin_scan turns the counts into offsets and
// into a task table with workgroups PROPORTIONAL to a block's item count -- the strided levels of the uint32-wrap quirk
// load their blocks very unevenly --, k_bwd_bin<true> writes the items through per-workgroup LDS ranks, one global atomic
// per (workgroup, block)).  k_bwd_list then runs dense 64-lane batches over its list: one hash, two x-weights, one erf,
// two 8-byte LDS compare-and-swaps per item; the x0 + 1 corner lands in another block once in ~16 384 items and goes to
// the table by a global atomic.  Same addends as the compacted kernel ((w_k damp) g_c), other order.
constexpr uint32_t kListTasks = 160;                  // workgroups per fine level (128 shared out by item count, + rounding)
constexpr uint32_t kListMaxBlocks = 32;

struct ListPlan {
    uint32_t n_fine;
    uint8_t level[UCN_MAX_LEVELS];          // fine level f -> level index
    uint32_t nb[UCN_MAX_LEVELS];            // row blocks of that level
    size_t cap;                             // items per fine level (= 24 B)
};
// control block (uint32): per fine level f: cnt[32] | cursor[32] | off[33] | tasks[kListTasks][2] (blk | nparts << 8 | part << 16, first item; count via next)
constexpr uint32_t kCtlCnt = 0, kCtlCur = 32, kCtlOff = 64, kCtlTask = 100, kCtlPerLevel = kCtlTask + 3 * kListTasks + 4;

// the x0 row of each of the four (y, z) combinations of a point, and its fractions
template <bool HASHED, bool POW2>
__device__ __forceinline__ void combo_rows(const UcnLevel &lv, float px, float py, float pz, float &fx, float &fy, float &fz,
                                           uint32_t (&r0)[4], uint32_t (&r1)[4]) {
    fx = fmaf(px, lv.scale, 0.5f); fy = fmaf(py, lv.scale, 0.5f); fz = fmaf(pz, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    uint32_t ya, yb, za, zb, xa, xb;
    if constexpr (HASHED) {
        xa = x0; xb = x0 + 1u;
        ya = y0 * kP1; yb = ya + kP1;
        za = z0 * kP2; zb = za + kP2;
    } else {
        xa = x0 * lv.stride[0]; xb = xa + lv.stride[0];
        ya = y0 * lv.stride[1]; yb = ya + lv.stride[1];
        za = z0 * lv.stride[2]; zb = za + lv.stride[2];
    }
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t yv = (c & 1u) ? yb : ya, zv = (c & 2u) ? zb : za;
        uint32_t i0, i1;
        if constexpr (HASHED) { i0 = xa ^ yv ^ zv; i1 = xb ^ yv ^ zv; }
        else { i0 = xa + yv + zv; i1 = xb + yv + zv; }
        if constexpr (POW2) { r0[c] = i0 & lv.mask; r1[c] = i1 & lv.mask; }
        else { r0[c] = i0 < lv.rows ? i0 : i0 % lv.rows; r1[c] = i1 < lv.rows ? i1 : i1 % lv.rows; }
    }
}
__device__ __forceinline__ void combo_rows_any(const UcnLevel &lv, float px, float py, float pz, float &fx, float &fy, float &fz,
                                               uint32_t (&r0)[4], uint32_t (&r1)[4]) {
    if (lv.hashed) {
        if (lv.mask) combo_rows<true, true>(lv, px, py, pz, fx, fy, fz, r0, r1);
        else combo_rows<true, false>(lv, px, py, pz, fx, fy, fz, r0, r1);
    } else {
        if (lv.mask) combo_rows<false, true>(lv, px, py, pz, fx, fy, fz, r0, r1);
        else combo_rows<false, false>(lv, px, py, pz, fx, fy, fz, r0, r1);
    }
}

// pass 1 (WRITE = false): items per (fine level, block);  pass 3 (WRITE = true): the items themselves.
// grid (ceil(B / 256), n_fine); thread = sample.
template <uint32_t C, bool WRITE>
__global__ __launch_bounds__(256) void k_bwd_bin(UcnLevels lvls, ListPlan lp, uint32_t shift, size_t B, const float *__restrict__ geom,
                                                 const float *__restrict__ grad_lm /*[L][B][C]*/, uint32_t *__restrict__ ctl,
                                                 uint32_t *__restrict__ lists) {
    __shared__ uint32_t s_cnt[kListMaxBlocks], s_base[kListMaxBlocks];
    const uint32_t f = blockIdx.y, lvl = lp.level[f];
    const UcnLevel lv = lvls.lv[lvl];
    if (threadIdx.x < kListMaxBlocks) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t slot[24];                                    // blk << 16 | rank in the workgroup, 0xFFFFFFFF = no item
#pragma unroll
    for (uint32_t i = 0; i < 24; i++) slot[i] = 0xFFFFFFFFu;
    bool live = b < B;
    if (live) {
        bool nz = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) nz |= grad_lm[((size_t)lvl * B + b) * C + c] != 0.0f;
        live = nz;                                        // a sample without gradient on this level contributes nothing
    }
    if (live) {
#pragma unroll
        for (uint32_t j = 0; j < 6; j++) {
            const float4 q = reinterpret_cast<const float4 *>(geom)[b * 6 + j];
            if (!in_unit_cube(q.x, q.y, q.z)) continue;
            float fx, fy, fz;
            uint32_t r0[4], r1[4];
            combo_rows_any(lv, q.x, q.y, q.z, fx, fy, fz, r0, r1);
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t blk = r0[c] >> shift;
                const uint32_t rank = atomicAdd(&s_cnt[blk], 1u);
                slot[4 * j + c] = (blk << 16) | rank;
            }
        }
    }
    __syncthreads();
    uint32_t *cl = ctl + (size_t)f * kCtlPerLevel;
    if (threadIdx.x < lp.nb[f] && s_cnt[threadIdx.x])
        s_base[threadIdx.x] = atomicAdd(cl + (WRITE ? kCtlCur : kCtlCnt) + threadIdx.x, s_cnt[threadIdx.x]);
    if constexpr (WRITE) {
        // the workgroup's items are first gathered per block in LDS, then written out as ONE run per block: lanes write
        // consecutive addresses (scattering 24 single words per thread straight to the 32 lists took 2.8 ms per step)
        __shared__ uint32_t s_lbase[kListMaxBlocks + 1], s_items[256 * 24];
        if (threadIdx.x == 0) {
            uint32_t a = 0;
            for (uint32_t k = 0; k < kListMaxBlocks; k++) { s_lbase[k] = a; a += s_cnt[k]; }
            s_lbase[kListMaxBlocks] = a;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < 24; i++) {
            if (slot[i] != 0xFFFFFFFFu) {
                const uint32_t blk = slot[i] >> 16, rank = slot[i] & 0xFFFFu;
                s_items[s_lbase[blk] + rank] = (uint32_t)b | ((i >> 2) << 27) | ((i & 3u) << 30);
            }
        }
        __syncthreads();
        uint32_t *lst = lists + (size_t)f * lp.cap;
        const uint32_t total = s_lbase[kListMaxBlocks];
        for (uint32_t i = threadIdx.x; i < total; i += 256u) {
            uint32_t lo = 0, hi = kListMaxBlocks;                 // the block whose run holds position i
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_lbase[mid] <= i) lo = mid; else hi = mid;
            }
            lst[cl[kCtlOff + lo] + s_base[lo] + (i - s_lbase[lo])] = s_items[i];
        }
    }
}

// pass 2: one workgroup of 64 lanes per fine level: offsets of the blocks' lists, and the task table
__global__ __launch_bounds__(64) void k_bwd_bin_scan(ListPlan lp, uint32_t *__restrict__ ctl) {
    const uint32_t f = blockIdx.x, nb = lp.nb[f];
    uint32_t *cl = ctl + (size_t)f * kCtlPerLevel;
    if (threadIdx.x != 0) return;
    uint32_t total = 0;
    for (uint32_t k = 0; k < nb; k++) { cl[kCtlOff + k] = total; total += cl[kCtlCnt + k]; }
    cl[kCtlOff + nb] = total;
    // parts per block proportional to its share of the items (>= 1 where there are any), 128 in all (+ rounding)
    uint32_t t = 0;
    for (uint32_t k = 0; k < nb; k++) {
        const uint32_t cnt = cl[kCtlCnt + k];
        if (!cnt) continue;
        uint32_t parts = total ? (uint32_t)(((uint64_t)cnt * 128u + total / 2) / total) : 1u;
        parts = parts < 1u ? 1u : parts;
        if (t + parts > kListTasks) parts = kListTasks - t;   // (never: sum of max(1, round(128 share)) over <= 32 blocks <= 160)
        if (!parts) break;
        const uint32_t per = (cnt + parts - 1) / parts;
        for (uint32_t p = 0; p < parts && t < kListTasks; p++) {
            const uint32_t lo = p * per, hi = lo + per < cnt ? lo + per : cnt;
            if (lo >= hi) break;
            cl[kCtlTask + 3 * t + 0] = k | (parts << 8);
            cl[kCtlTask + 3 * t + 1] = cl[kCtlOff + k] + lo;
            cl[kCtlTask + 3 * t + 2] = hi - lo;
            t++;
        }
        if (t >= kListTasks) break;                        // (cannot drop work: parts were clamped so that every block gets >= 1 task
    }                                                      //  only while t < kListTasks; nb <= 32 and the proportional parts sum to <= 144)
    for (; t < kListTasks; t++) cl[kCtlTask + 3 * t + 2] = 0u;
}

template <uint32_t C, bool HASHED, bool POW2>
__device__ __forceinline__ void list_item(const UcnLevel &lv, float *__restrict__ s_acc, float *__restrict__ gtab_level, uint32_t row_lo,
                                          uint32_t nrows, uint32_t item, size_t B, const float *__restrict__ gl, const float *__restrict__ geom) {
    const uint32_t b = item & 0x07FFFFFFu, j = (item >> 27) & 7u, c = item >> 30;
    const float4 q = reinterpret_cast<const float4 *>(geom)[(size_t)b * 6 + j];
    float g[C];
#pragma unroll
    for (uint32_t cc = 0; cc < C; cc++) g[cc] = gl[(size_t)b * C + cc];                     // d(mean over the 6 multisamples), / 6 by k_cast_cache_masks
    float fx = fmaf(q.x, lv.scale, 0.5f), fy = fmaf(q.y, lv.scale, 0.5f), fz = fmaf(q.z, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    const uint32_t yy = y0 + (c & 1u), zz = z0 + (c >> 1);
    uint32_t i0, i1;
    if constexpr (HASHED) { const uint32_t h = (yy * kP1) ^ (zz * kP2); i0 = x0 ^ h; i1 = (x0 + 1u) ^ h; }
    else { const uint32_t h = yy * lv.stride[1] + zz * lv.stride[2]; i0 = x0 * lv.stride[0] + h; i1 = i0 + lv.stride[0]; }
    uint32_t r0, r1;
    if constexpr (POW2) { r0 = i0 & lv.mask; r1 = i1 & lv.mask; }
    else { r0 = i0 < lv.rows ? i0 : i0 % lv.rows; r1 = i1 < lv.rows ? i1 : i1 % lv.rows; }
    // corner weights in the reference's multiplication order ((wx wy) wz), gridencoder.cu:168-180
    const float wy = (c & 1u) ? fy : 1.0f - fy, wz = (c >> 1) ? fz : 1.0f - fz;
    const float damp = erf_pos(q.w * lv.inv_gs);
    const float w0 = ((1.0f - fx) * wy) * wz, w1 = (fx * wy) * wz;
    float v0[C], v1[C];
#pragma unroll
    for (uint32_t cc = 0; cc < C; cc++) { v0[cc] = (w0 * damp) * g[cc]; v1[cc] = (w1 * damp) * g[cc]; }
    lds_row_add<C, true>(s_acc, r0 - row_lo, v0);              // the item was routed by r0: always in this block
    const uint32_t l1 = r1 - row_lo;
    if (l1 < nrows) lds_row_add<C, true>(s_acc, l1, v1);
    else {
#pragma unroll
        for (uint32_t cc = 0; cc < C; cc++) atomicAdd(gtab_level + (size_t)r1 * C + cc, v1[cc]);      // ~1 in 16 384 items
    }
}

// grid (kListTasks, n_fine), 1024 threads, the block's accumulators in LDS
template <uint32_t C>
__global__ __launch_bounds__(1024) void k_bwd_list(UcnLevels lvls, ListPlan lp, float *__restrict__ grad_table, uint32_t rpb, size_t B,
                                                   const float *__restrict__ grad_lm, const float *__restrict__ geom,
                                                   const uint32_t *__restrict__ ctl, const uint32_t *__restrict__ lists) {
    extern __shared__ float s_acc[];
    const uint32_t f = blockIdx.y, lvl = lp.level[f];
    const uint32_t *cl = ctl + (size_t)f * kCtlPerLevel;
    const uint32_t n_items = cl[kCtlTask + 3 * blockIdx.x + 2];
    if (!n_items) return;
    const uint32_t head = cl[kCtlTask + 3 * blockIdx.x + 0], first = cl[kCtlTask + 3 * blockIdx.x + 1];
    const uint32_t blk = head & 0xFFu;
    const UcnLevel lv = lvls.lv[lvl];
    const uint32_t row_lo = blk * rpb;
    const uint32_t nrows = lv.rows - row_lo < rpb ? lv.rows - row_lo : rpb;
    for (uint32_t i = threadIdx.x; i < nrows * C; i += 1024u) s_acc[i] = 0.0f;
    __syncthreads();
    const uint32_t *lst = lists + (size_t)f * lp.cap + first;
    const float *gl = grad_lm + (size_t)lvl * B * C;
    float *gtab = grad_table + (size_t)lv.first_row * C;
    // two items per lane and round: both items' loads are in flight before the first update
    for (uint32_t i = threadIdx.x; i < n_items; i += 2048u) {
        const uint32_t it0 = lst[i], i1 = i + 1024u;
        const bool has1 = i1 < n_items;
        const uint32_t it1 = has1 ? lst[i1] : 0u;
        if (lv.hashed) {
            if (lv.mask) {
                list_item<C, true, true>(lv, s_acc, gtab, row_lo, nrows, it0, B, gl, geom);
                if (has1) list_item<C, true, true>(lv, s_acc, gtab, row_lo, nrows, it1, B, gl, geom);
            } else {
                list_item<C, true, false>(lv, s_acc, gtab, row_lo, nrows, it0, B, gl, geom);
                if (has1) list_item<C, true, false>(lv, s_acc, gtab, row_lo, nrows, it1, B, gl, geom);
            }
        } else if (lv.mask) {
            list_item<C, false, true>(lv, s_acc, gtab, row_lo, nrows, it0, B, gl, geom);
            if (has1) list_item<C, false, true>(lv, s_acc, gtab, row_lo, nrows, it1, B, gl, geom);
        } else {
            list_item<C, false, false>(lv, s_acc, gtab, row_lo, nrows, it0, B, gl, geom);
            if (has1) list_item<C, false, false>(lv, s_acc, gtab, row_lo, nrows, it1, B, gl, geom);
        }
    }
    __syncthreads();
    float *dst = gtab + (size_t)row_lo * C;
    for (uint32_t i = threadIdx.x; i < nrows * C; i += 1024u) {
        const float v = s_acc[i];
        if (v != 0.0f) atomicAdd(dst + i, v);      // several workgroups share a block, and the rare cross-block corners of others land here too
    }
}

// predict_density's featurisation for caller-supplied Gaussians (extract.py / API parity)
template <uint32_t C>
__global__ __launch_bounds__(256) void k_points_features(UcnLevels lvls, const float *__restrict__ table,
                                                         const float *__restrict__ means, const float *__restrict__ stds,
                                                         uint32_t Bn, uint32_t G, int warp, uint32_t lpb,
                                                         float *__restrict__ features, float *__restrict__ coord_out) {
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= Bn) return;
    float u[6][3], rs[6];
    float cs0 = 0.0f, cs1 = 0.0f, cs2 = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (j < G) {
            const float *m = means + (b * G + j) * 3;
            float c0, c1, c2;
            contract_to_unit(m[0], m[1], m[2], stds[b * G + j], warp != 0, u[j][0], u[j][1], u[j][2], rs[j], c0, c1, c2);
            cs0 += c0; cs1 += c1; cs2 += c2;
        } else {
            u[j][0] = u[j][1] = u[j][2] = 0.0f; rs[j] = 1.0f;
        }
    }
    const uint32_t lvl0 = blockIdx.y * lpb;
    const uint32_t lvl1 = lvl0 + lpb < lvls.L ? lvl0 + lpb : lvls.L;
    featurise<C, float>(lvls, table, lvl0, lvl1, u, rs, G, Bn, b, features, false);
    if (blockIdx.y == 0 && coord_out) {
        coord_out[b * 3 + 0] = cs0 / (float)G; coord_out[b * 3 + 1] = cs1 / (float)G; coord_out[b * 3 + 2] = cs2 / (float)G;
    }
}

HexPattern make_hex() {
    HexPattern hx;
    const int order[6] = {0, 2, 4, 3, 5, 1};
    const float third = (float)(M_PI / 3.0), sixth = (float)(M_PI / 6.0), fivethirds = (float)(M_PI * 5.0 / 3.0);
    for (int j = 0; j < 6; j++) {
        const float a = third * (float)order[j];
        hx.ang[j] = a;
        hx.cs[0][j] = cosf(a);
        hx.sn[0][j] = sinf(a);
        const float o = fivethirds - (a + sixth);
        hx.cs[1][j] = cosf(o);
        hx.sn[1][j] = sinf(o);
        hx.cj[j] = (float)(3.0 / sqrt(7.0)) * ((float)(2 * j) / 5.0f - 1.0f);
    }
    return hx;
}

int field_levels(const ucn_field_t *f, UcnLevels *lv) {
    UCN_REQUIRE(f && f->embeddings && f->offsets_host && f->grid_sizes_host, "field: grid pointers missing");
    UCN_REQUIRE(f->level_dim == 1 || f->level_dim == 2 || f->level_dim == 4 || f->level_dim == 8,
                "GridEncoding: C must be 1, 2, 4, or 8.");
    return ucn_build_levels(lv, f->offsets_host, f->grid_sizes_host, f->num_levels, f->level_dim, 3,
                            f->log2_per_level_scale, f->base_resolution, 0, 0);
}

}  // namespace

extern "C" int ucn_march_features(const ucn_field_t *f, const float *sdist, const float *near_, const float *far_,
                                  const float *origins, const float *directions, const float *basis,
                                  const float *radii, const float *flip, const float *spin, float std_scale,
                                  uint32_t N, uint32_t S, uint32_t levels_per_block, int layout,
                                  float *features_out, float *coord_out, float *tmean_out, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (sdist && near_ && far_ && origins && directions && basis && radii && features_out),
                "march_features: null pointer argument");
    UCN_REQUIRE((flip == nullptr) == (spin == nullptr), "march_features: flip and spin come together");
    const bool coresident = (layout & UCN_LAUNCH_CORESIDENT) != 0;
    const bool half_table = (layout & UCN_TABLE_F16) != 0;
    const bool out_bf16 = (layout & UCN_FEATURES_BF16) != 0;
    const bool incoherent = (layout & UCN_RAYS_INCOHERENT) != 0;
    layout &= ~(UCN_LAUNCH_CORESIDENT | UCN_TABLE_F16 | UCN_FEATURES_BF16 | UCN_RAYS_INCOHERENT);
    UCN_REQUIRE(!(half_table && coresident), "march_features: the co-resident launch shape reads fp32 tables");
    UCN_REQUIRE(layout >= 0 && layout <= 2, "march_features: layout must be 0, 1 or 2");
    UCN_REQUIRE(!out_bf16 || (half_table && f->level_dim == 2 && layout != 1),
                "march_features: bf16 features come with half tables, level_dim 2 and a level-major layout");
    if (out_bf16) layout |= 0x10;                      // the kernel's private flags
    if (incoherent) layout |= 0x20;
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (N == 0) return 0;
    const size_t B = (size_t)N * S;
    UCN_REQUIRE(B <= 0xFFFFFF00ull, "march_features: too many samples in one call (%zu)", B);
    const LevelGroups grp = make_groups(lv, levels_per_block);
    const RayInputs in{sdist, near_, far_, origins, directions, basis, radii, flip, spin};
    const HexPattern hx = make_hex();
    hipStream_t st = (hipStream_t)stream;
    // experiment knob (tools/feat_occupancy.py): unused dynamic LDS per workgroup caps the workgroups per CU
    static const size_t dummy_lds = getenv("UCN_FEAT_DUMMY_LDS") ? (size_t)atol(getenv("UCN_FEAT_DUMMY_LDS")) : 0;
    // co-resident shape: 512 threads = two waves per SIMD, 88 KiB of LDS reserved -> ONE such workgroup per CU, and
    // room for one MLP workgroup (72 KiB, one 296-register wave per SIMD) beside it.  The kernel runs at 97 % of its
    // full-occupancy rate with two waves per SIMD (profiles/r02*/occupancy.txt): it is bound by the L2 request rate.
    const uint32_t tpb = coresident ? 512u : 256u;
    const size_t lds = coresident ? 88u * 1024u : dummy_lds;
    const dim3 grid(ucn_div_up(B, tpb), grp.n);
#define UCN_MF2(CC, FEW)                                                                                                  \
    do {                                                                                                                  \
        if (half_table)                                                                                                   \
            hipLaunchKernelGGL((k_march_features<CC, 256, _Float16, FEW>), grid, dim3(256), lds, st, lv,                  \
                               reinterpret_cast<const _Float16 *>(f->embeddings), in, hx, std_scale, N, S, grp, layout,   \
                               features_out, coord_out, tmean_out);                                                       \
        else if (coresident)                                                                                              \
            hipLaunchKernelGGL((k_march_features<CC, 512, float, FEW>), grid, dim3(512), lds, st, lv, f->embeddings, in, hx, std_scale, N, S, \
                               grp, layout, features_out, coord_out, tmean_out);                                          \
        else                                                                                                              \
            hipLaunchKernelGGL((k_march_features<CC, 256, float, FEW>), grid, dim3(256), lds, st, lv, f->embeddings, in, hx, std_scale, N, S, \
                               grp, layout, features_out, coord_out, tmean_out);                                          \
    } while (0)
#define UCN_MF(CC)                                                                                                        \
    do {                                                                                                                  \
        if (lv.L <= 8) UCN_MF2(CC, true);                                                                                 \
        else UCN_MF2(CC, false);                                                                                          \
    } while (0)
    switch (lv.C) {
        case 1: UCN_MF(1); break;
        case 2: UCN_MF(2); break;
        case 4: UCN_MF(4); break;
        case 8: UCN_MF(8); break;
    }
#undef UCN_MF2
#undef UCN_MF
    UCN_LAUNCH_CHECK("march_features");
    return 0;
}

extern "C" int ucn_cast_probe(const float *sdist, const float *near_, const float *far_, const float *origins,
                              const float *directions, const float *basis, const float *radii, const float *flip,
                              const float *spin, float std_scale, uint32_t N, uint32_t S, float *out, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (sdist && near_ && far_ && origins && directions && basis && radii && out),
                "cast_probe: null pointer argument");
    UCN_REQUIRE((flip == nullptr) == (spin == nullptr), "cast_probe: flip and spin come together");
    if (N == 0 || S == 0) return 0;
    const RayInputs in{sdist, near_, far_, origins, directions, basis, radii, flip, spin};
    hipLaunchKernelGGL(k_cast_probe, dim3(ucn_div_up((size_t)N * S, 256)), dim3(256), 0, (hipStream_t)stream, in, make_hex(),
                       std_scale, N, S, out);
    UCN_LAUNCH_CHECK("cast_probe");
    return 0;
}

extern "C" int ucn_contract_probe(const float *means, const float *stds, uint32_t B, float *out_mean, float *out_std,
                                  ucn_stream_t stream) {
    UCN_REQUIRE(B == 0 || (means && stds && out_mean && out_std), "contract_probe: null pointer argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_contract_probe, dim3(ucn_div_up(B, 256)), dim3(256), 0, (hipStream_t)stream, means, stds, B, out_mean,
                       out_std);
    UCN_LAUNCH_CHECK("contract_probe");
    return 0;
}

static bool plan_has_wide(const UcnLevels &lv, const MaskPlan &plan) {      // wide_block items hold the sample in 24 bits
    for (uint32_t l = 0; l < lv.L; l++)
        if (plan.coarse[l] == 3) return true;
    return false;
}
static bool make_list_plan(const UcnLevels &lv, const MaskPlan &plan, uint32_t rpb, size_t B, ListPlan *lp) {
    lp->n_fine = 0;
    lp->cap = 24ull * B;
    if (B >= (1ull << 27)) return false;                     // 27 bits of an item hold the sample
    for (uint32_t l = 0; l < lv.L; l++)
        if (!plan.coarse[l]) {
            lp->level[lp->n_fine] = (uint8_t)l;
            lp->nb[lp->n_fine] = ucn_div_up(lv.lv[l].rows, rpb);
            lp->n_fine++;
        }
    return lp->n_fine > 0;
}
// UCN_BWD_LISTS=1 turns the item-list path on for the fine levels.  OFF by default on measurement (8192 x 128 samples,
// config B, profiles/r03/bwd_lists.txt): the list kernel takes 1.86 ms for the ten fine levels where the compacted kernel
// takes 2.9 -- but the counting sort in front of it costs 0.35 (count) + 2.2 ms (scatter: 250 M items = 1 GB written), and the
// list kernel itself is bound by its three gathers per item (item, 16-byte geometry, 8-byte gradient: 7 GB through the
// texture-address path), which is what the recomputing kernel avoids: 4.4 ms against 2.9.  Kept as a cross-check of the
// compacted kernel (same addends, other route; tests/test_full_size.py runs the adjoint test on it).
static bool bwd_lists_enabled() {
    static const bool on = getenv("UCN_BWD_LISTS") && atoi(getenv("UCN_BWD_LISTS")) != 0;
    return on;
}

// CUs of the current device (the persistent backward launches one workgroup per CU)
static uint32_t device_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local uint32_t cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256u;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = (uint32_t)n;
        cached_dev = dev;
    }
    return cached;
}

extern "C" uint64_t ucn_march_features_backward_ws_floats(const ucn_field_t *f, uint32_t N, uint32_t S) {
    UcnLevels lv;
    if (field_levels(f, &lv)) return 0;
    MaskPlan plan;
    const uint32_t rpb = 128u * 1024u / (lv.C * 4u);
    const size_t B = (size_t)N * S;
    const bool masks = make_mask_plan(lv, rpb, B, &plan);
    MaskPlan plan_fx;                                                                         // the fixed-point call cuts the levels differently: room for either
    const bool masks_fx = masks && make_mask_plan(lv, rpb, B, &plan_fx, true);
    if (masks_fx && plan_fx.n_planes > plan.n_planes) plan.n_planes = plan_fx.n_planes;
    // geometry planes + block-mask planes + a level-major copy of the gradient (layouts 1 and 3)
    uint64_t n = (24ull + (masks ? plan.n_planes + lv.L * lv.C : 0u)) * B + 64u;             // + the task counter
    n += (((uint64_t)lv.L * ucn_div_up(B, 256) + 63u) & ~63ull);                                 // + the 256-sample L1 partials of the fixed-point mode
    ListPlan lp;
    uint64_t n_lists = 0;
    if (masks && make_list_plan(lv, plan, rpb, B, &lp))      // + the item lists of the fine levels and their control block
        n_lists = (uint64_t)lp.n_fine * (lp.cap + kCtlPerLevel) + 64;
    if (masks_fx && make_list_plan(lv, plan_fx, rpb, B, &lp) && (uint64_t)lp.n_fine * (lp.cap + kCtlPerLevel) + 64 > n_lists)
        n_lists = (uint64_t)lp.n_fine * (lp.cap + kCtlPerLevel) + 64;
    n += n_lists;
    return n;
}

// 1 if ucn_march_features_backward(levels_per_block = 0, with a workspace) takes the compacted row-block kernel for this field and call
// size -- the route that reads a layout-4 gradient (its launcher below applies the same test)
extern "C" int ucn_march_features_backward_row_blocks(const ucn_field_t *f, uint32_t N, uint32_t S) {
    UcnLevels lv;
    if (field_levels(f, &lv)) return 0;
    const size_t B = (size_t)N * S;
    const uint32_t rpb = 128u * 1024u / (lv.C * 4u);
    MaskPlan plan;
    return (B > 0 && B < (1ull << 29) && B <= 0xFFFFFF00ull && make_mask_plan(lv, rpb, B, &plan) && !(plan_has_wide(lv, plan) && B >= (1ull << 24))) ? 1 : 0;
}

extern "C" int ucn_march_features_backward(const ucn_field_t *f, const float *sdist, const float *near_, const float *far_,
                                           const float *origins, const float *directions, const float *basis,
                                           const float *radii, const float *flip, const float *spin, float std_scale,
                                           uint32_t N, uint32_t S, uint32_t levels_per_block, int layout,
                                           const float *grad_features, float *grad_embeddings, float *workspace,
                                           ucn_stream_t stream) {
    UCN_REQUIRE(sdist && near_ && far_ && origins && directions && basis && radii && grad_features && grad_embeddings,
                "march_features_backward: null pointer argument");
    UCN_REQUIRE((flip == nullptr) == (spin == nullptr), "march_features_backward: flip and spin come together");
    const bool want_fixed = (layout & UCN_BWD_FIXED_POINT) != 0;
    layout &= ~UCN_BWD_FIXED_POINT;
    const bool prediv = layout == 4;                     // level-major and already / 6: read in place
    if (prediv) layout = 0;
    UCN_REQUIRE(layout == 0 || layout == 1 || layout == 3, "march_features_backward: layout must be 0, 1, 3 or 4");
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (N == 0) return 0;
    const size_t B = (size_t)N * S;
    UCN_REQUIRE(B <= 0xFFFFFF00ull, "march_features_backward: too many samples in one call (%zu)", B);
    // fixed-point row blocks pack channel PAIRS and bound a row's rounding slack by 24 B (48 B addends x 1/2) on top of the 2^30
    // target: C = 1 and calls with 2^30 + 24 B >= 2^31 keep float rows
    const bool fixed = want_fixed && lv.C % 2u == 0u && B <= (1ull << 22);
    static_assert((1ull << 30) + 24ull * (1ull << 22) < (1ull << 31), "fixed-point row blocks: int32 headroom");
    const RayInputs in{sdist, near_, far_, origins, directions, basis, radii, flip, spin};
    const HexPattern hx = make_hex();
    const GradStrides gs = grad_strides(layout, B, lv.L, lv.C);
    hipStream_t st = (hipStream_t)stream;
    if (levels_per_block == 0) {
        // row-block ownership (no global atomics) while the recomputation stays cheap: every block walks
        // all samples, so the cost grows with the number of blocks; above 64 blocks per level the atomic
        // scatter below wins again
        const uint32_t rpb = 128u * 1024u / (lv.C * 4u);
        uint32_t tasks = 0, blocks = 0;
        for (uint32_t l = 0; l < lv.L; l++) {
            const uint32_t nb = ucn_div_up(lv.lv[l].rows, rpb);
            blocks += nb;
            tasks += nb * bwd_sample_split(nb);
        }
        MaskPlan plan;
        if (workspace && B < (1ull << 29) && make_mask_plan(lv, rpb, B, &plan, fixed) && !(plan_has_wide(lv, plan) && B >= (1ull << 24))) {
            tasks = 0;
            for (uint32_t l = 0; l < lv.L; l++) {
                const uint32_t nb = ucn_div_up(lv.lv[l].rows, rpb);
                tasks += nb * plan.split[l];
            }
            // compacting variant: block masks next to the geometry planes, dense items from a per-wave ring in LDS
            uint32_t *masks = reinterpret_cast<uint32_t *>(workspace + 24ull * B);
            float *glm = workspace + (24ull + plan.n_planes) * B;                               // level-major copy, / 6
            uint32_t *task_counter = reinterpret_cast<uint32_t *>(workspace + (24ull + plan.n_planes + (size_t)lv.L * lv.C) * B);
            float *l1_partial = reinterpret_cast<float *>(task_counter + 64);                   // [L][ceil(B / 256)] (fixed-point mode)
            hipLaunchKernelGGL(k_cast_cache_masks, dim3(ucn_div_up(B, 256)), dim3(256), 0, st, lv, in, hx, std_scale, N, S, plan,
                               grad_features, gs, lv.C, workspace, masks, prediv ? nullptr : glm, task_counter, fixed ? l1_partial : nullptr);
            const float *glv = prediv ? grad_features : glm;                                    // [L][B][C], / 6
            const uint32_t cus = device_cu_count();
            ListPlan lp;
            const bool lists = bwd_lists_enabled() && make_list_plan(lv, plan, rpb, B, &lp);
            uint32_t *ctl = nullptr, *items = nullptr;
            if (lists) {
                // fine levels: counting sort of (point, (y, z) combination) items into per-block lists, then k_bwd_list
                ctl = task_counter + 64 + (((size_t)lv.L * ucn_div_up(B, 256) + 63u) & ~(size_t)63u);
                items = ctl + (((size_t)lp.n_fine * kCtlPerLevel + 63u) & ~(size_t)63u);
                if (hipMemsetAsync(ctl, 0, (size_t)lp.n_fine * kCtlPerLevel * sizeof(uint32_t), st) != hipSuccess)
                    return ucn_fail("march_features_backward: hipMemsetAsync failed");
                plan.skip_fine = 1u;
                tasks = 0;
                for (uint32_t l = 0; l < lv.L; l++) {
                    const uint32_t nb = ucn_div_up(lv.lv[l].rows, rpb);
                    if (plan.coarse[l]) tasks += nb * plan.split[l];
                }
            }
#define UCN_MBC(CC)                                                                                              \
    do {                                                                                                         \
        const dim3 bg(ucn_div_up(B, 256), lists ? lp.n_fine : 1u);                                               \
        if (lists) {                                                                                             \
            hipLaunchKernelGGL((k_bwd_bin<CC, false>), bg, dim3(256), 0, st, lv, lp, plan.shift, B, workspace, glv, ctl, items); \
            hipLaunchKernelGGL(k_bwd_bin_scan, dim3(lp.n_fine), dim3(64), 0, st, lp, ctl);                       \
            hipLaunchKernelGGL((k_bwd_bin<CC, true>), bg, dim3(256), 0, st, lv, lp, plan.shift, B, workspace, glv, ctl, items); \
        }                                                                                                        \
        if (tasks && fixed && CC % 2 == 0)                                                                       \
            hipLaunchKernelGGL((k_march_features_bwd_cmp<(CC % 2 == 0 ? CC : 2), true>), dim3(tasks < cus ? tasks : cus), dim3(1024), \
                               (size_t)rpb * CC * 4 + 16 * kQueue * 4, st, lv, grad_embeddings, N, S, rpb, plan, glv, \
                               workspace, masks, task_counter, tasks, l1_partial);                               \
        else if (tasks)                                                                                          \
            hipLaunchKernelGGL((k_march_features_bwd_cmp<CC, false>), dim3(tasks < cus ? tasks : cus), dim3(1024), \
                               (size_t)rpb * CC * 4 + 16 * kQueue * 4, st, lv, grad_embeddings, N, S, rpb, plan, glv, \
                               workspace, masks, task_counter, tasks, nullptr);                                  \
        if (lists)                                                                                               \
            hipLaunchKernelGGL(k_bwd_list<CC>, dim3(kListTasks, lp.n_fine), dim3(1024), (size_t)rpb * CC * 4, st, lv, lp, \
                               grad_embeddings, rpb, B, glv, workspace, ctl, items);                              \
    } while (0)
            switch (lv.C) {
                case 1: UCN_MBC(1); break;
                case 2: UCN_MBC(2); break;
                case 4: UCN_MBC(4); break;
                case 8: UCN_MBC(8); break;
            }
#undef UCN_MBC
            UCN_LAUNCH_CHECK("march_features_backward (row blocks, compacted)");
            return 0;
        }
        UCN_REQUIRE(!prediv, "march_features_backward: layout 4 (pre-divided level-major gradient) is the compacted row-block kernel's input; this call would take a fallback");
        if (blocks <= 64u * lv.L) {
            if (workspace)
                hipLaunchKernelGGL(k_cast_cache, dim3(ucn_div_up(B, 256)), dim3(256), 0, st, in, hx, std_scale, N, S, workspace);
#define UCN_MBB(CC)                                                                                              \
    hipLaunchKernelGGL(k_march_features_bwd_blk<CC>, dim3(tasks), dim3(1024), (size_t)rpb * CC * 4, st, lv,      \
                       grad_embeddings, in, hx, std_scale, N, S, gs, rpb, grad_features, workspace)
            switch (lv.C) {
                case 1: UCN_MBB(1); break;
                case 2: UCN_MBB(2); break;
                case 4: UCN_MBB(4); break;
                case 8: UCN_MBB(8); break;
            }
#undef UCN_MBB
            UCN_LAUNCH_CHECK("march_features_backward (row blocks)");
            return 0;
        }
        levels_per_block = 1;
    }
    UCN_REQUIRE(layout != 3, "march_features_backward: layout 3 is a row-block layout (levels_per_block = 0, <= 64 blocks per level)");
    UCN_REQUIRE(!prediv, "march_features_backward: layout 4 is a row-block layout (levels_per_block = 0 with a workspace)");
    const dim3 grid(ucn_div_up(B, 256), ucn_div_up(lv.L, levels_per_block));
#define UCN_MB(CC)                                                                                                  \
    hipLaunchKernelGGL(k_march_features_bwd<CC>, grid, dim3(256), 0, st, lv, grad_embeddings, in, hx, std_scale, N, \
                       S, levels_per_block, layout, grad_features)
    switch (lv.C) {
        case 1: UCN_MB(1); break;
        case 2: UCN_MB(2); break;
        case 4: UCN_MB(4); break;
        case 8: UCN_MB(8); break;
    }
#undef UCN_MB
    UCN_LAUNCH_CHECK("march_features_backward");
    return 0;
}

extern "C" int ucn_points_features(const ucn_field_t *f, const float *means, const float *stds, uint32_t B, uint32_t G,
                                   int warp, uint32_t levels_per_block, float *features_out, float *coord_out,
                                   ucn_stream_t stream) {
    UCN_REQUIRE(means && stds && features_out, "points_features: null pointer argument");
    UCN_REQUIRE(G >= 1 && G <= 6, "points_features: 1..6 Gaussians per feature, got %u", G);
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (B == 0) return 0;
    if (levels_per_block == 0) levels_per_block = 1;
    const dim3 grid(ucn_div_up(B, 256), ucn_div_up(lv.L, levels_per_block));
    hipStream_t st = (hipStream_t)stream;
#define UCN_PF(CC)                                                                                          \
    hipLaunchKernelGGL(k_points_features<CC>, grid, dim3(256), 0, st, lv, f->embeddings, means, stds, B, G, \
                       warp, levels_per_block, features_out, coord_out)
    switch (lv.C) {
        case 1: UCN_PF(1); break;
        case 2: UCN_PF(2); break;
        case 4: UCN_PF(4); break;
        case 8: UCN_PF(8); break;
    }
#undef UCN_PF
    UCN_LAUNCH_CHECK("points_features");
    return 0;
}
