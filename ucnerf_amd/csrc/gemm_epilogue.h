// Epilogue shared by the dense training kernels (gemm_f32.hip: exact fp32 products; gemm_h3.hip: split-f16 products): accumulator
// tiles of the 32x32 MFMA C/D layout -> Y[M, N] with bias / row-group bias / ReLU / mask, directly or staged through LDS.
#pragma once
#include "ucn_common.h"
#include "wave_dpp.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr uint32_t kGemmAccum = 1u, kGemmRelu = 2u, kGemmMask = 4u, kGemmVec = 16u;

// ---- epilogue of both gemm kernels -------------------------------------------------------------------------------------------------
// The product is formed TRANSPOSED, D[feature][sample] = W_tile X_tile^T (A operand = weight rows from LDS, B operand = this lane's
// sample row): accumulator register r of lane (sample i, half kk) is feature 32 t + (r & 3) + 8 (r >> 2) + 4 kk -- four consecutive
// output columns per register quad, i.e. ONE 16-byte piece per quad instead of four 4-byte pieces spread over four rows.
struct GemmOut {
    const float *bias, *mask;      // mask: y = mask[row][col] > 0 ? y : 0 (the ReLU derivative of the layer below, fused into its d X GEMM)
    const float *rbias;            // row-group bias [M / rgroup, N]: + rbias[row / rgroup][col] (a per-RAY term under a per-sample GEMM)
    float *Y;
    uint32_t ldy, ldm, ldr, rgroup, M, N, flags;
    // r06: a second, 4-wide operand pair added in the epilogue -- Y += X2[M, 4] W2[N, 4]^T (exact fp32 FMAs) before the ReLU / mask.  The
    // sky NeRF has two layers whose input is [256-wide hidden | 3-d point] or whose gradient is [128-wide | 1 density row]
    // (models.py:790-795, :800-806): as a second accumulating GEMM the narrow block cost a full read-modify-write of the [M, 256] output.
    // (staged vector form only: the host requires N = 256 with it -- the scalar tail path is at hipcc's full-unroll limit.)
    const float *x2 = nullptr, *w2 = nullptr;
    uint32_t ldx2 = 0, ldw2 = 0;
    // r06: ReLU derivatives as BIT masks (staged form of the split engine only).  bits_out: the producing GEMM (forward, ReLU) leaves
    // "y > 0" of every output as one bit; bits_in: the d X GEMM of the layer above reads those bits instead of the stored fp32 output
    // (1 GB per [2^20, 256] layer).  Layout = the staged store's own: 32-bit word ((row tile of 32) * (N / 64) + column block) * 64 + lane
    // holds, at bit 4 u + j, the output this lane stores in step u, component j (row 32 tile + 4 u + lane / 16, column 64 block +
    // 4 (lane % 16) + j) -- writer and reader walk the same loop: one coalesced 256-byte store / load per wave and column block, no
    // ballot, no bit is moved.  (first form: 64-bit ballots, word per (u, j): 32 eight-byte stores and as many scalar loads per block.)
    uint32_t *bits_out = nullptr;
    const uint32_t *bits_in = nullptr;
};

// 64-bit words of a bit mask over an [M, N] output (N a multiple of 64)
__host__ __device__ inline unsigned long long gemm_bits_words(unsigned long long M, unsigned N) { return (M + 31u) / 32u * (N / 64u) * 32u; }

// `rb` = the row-group bias of this quad (zeros without one): the staged store fetches it once per column block, not once per row
__device__ __forceinline__ float4 gemm_finish_v(float4 v, const GemmOut &o, uint32_t row, uint32_t col, float4 rb) {      // vector form: col + 3 < N
    if (o.bias) { const float4 bv = *reinterpret_cast<const float4 *>(o.bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
    v.x += rb.x; v.y += rb.y; v.z += rb.z; v.w += rb.w;
    if (o.flags & kGemmRelu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (o.flags & kGemmMask) {
        const float4 m = *reinterpret_cast<const float4 *>(o.mask + (size_t)row * o.ldm + col);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    }
    return v;
}
__device__ __forceinline__ float4 gemm_finish(float4 v, const GemmOut &o, uint32_t row, uint32_t col) {
    float4 rb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.rbias) rb = *reinterpret_cast<const float4 *>(o.rbias + (size_t)(row / o.rgroup) * o.ldr + col);
    return gemm_finish_v(v, o, row, col, rb);
}

// Direct form: a lane stores the quads of its own row (any N, any alignment).  TRACK: returns the maximum of |every value this lane stored|
// (gemm_h3.hip: the absolute maximum of Y, the power-of-two operand scale of the GEMM that consumes it); NaNs do not enter the maximum.
template <uint32_t NT, bool TRACK = false>
__device__ __forceinline__ float gemm_store_direct(const f32x16 (&acc)[NT], const GemmOut &o, uint32_t orow, uint32_t n0, uint32_t kk) {
    float mx = 0.0f;
    if (orow >= o.M) return mx;
    const bool vec = (o.flags & kGemmVec) != 0;
#pragma unroll
    for (uint32_t t = 0; t < NT; t++) {
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t col = n0 + 32u * t + 8u * g + 4u * kk;
            if (col >= o.N) continue;
            float4 v = make_float4(acc[t][4u * g], acc[t][4u * g + 1u], acc[t][4u * g + 2u], acc[t][4u * g + 3u]);
            float *yo = o.Y + (size_t)orow * o.ldy + col;
            if (vec) {
                const float4 f = gemm_finish(v, o, orow, col);
                if constexpr (TRACK) mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f.x)), fmaxf(fabsf(f.y), fabsf(f.z))), fabsf(f.w));
                *reinterpret_cast<float4 *>(yo) = f;
                continue;
            }
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                if (col + c >= o.N) break;
                float y = e[c] + (o.bias ? o.bias[col + c] : 0.f);
                if (o.rbias) y += o.rbias[(size_t)(orow / o.rgroup) * o.ldr + col + c];
                if (o.flags & kGemmRelu) y = fmaxf(y, 0.f);
                if ((o.flags & kGemmMask) && !(o.mask[(size_t)orow * o.ldm + col + c] > 0.f)) y = 0.f;
                if constexpr (TRACK) mx = fmaxf(mx, fabsf(y));
                yo[c] = y;
            }
        }
    }
    return mx;
}

// Staged form (full tiles, 16-byte-aligned rows): a direct store instruction is 32 rows x two 16-byte pieces = 64 partial-line
// requests; written to the wave's own LDS tile as [row][column] and read back row-contiguous, a store instruction covers whole
// 128-byte lines -- the K = 64 layer went from 1.4 to ~3 TB/s of stores.  COLUMN blocks of 64 (two 32-column tiles) per pass (r05): every
// lane writes its row's 8 quads of the block (8 full-width ds_write_b128), the wave reads 32 rows x 256 B back as 8 float4 per lane and
// stores 4 rows x two whole lines per instruction.  (r04 / first r05 form: ROW blocks -- only the lanes of the pass's rows wrote, so a
// tile cost 32 / SR x 4 NT quarter- or half-empty ds_write_b128: 128 per 32 x 256 tile at SR = 8, 40 % of the CU's LDS time.)
// The tile is 32 x (64 + 4) floats = 8.5 KiB per wave.
constexpr uint32_t kStageFloats = 32u * 68u;
template <uint32_t NT, bool TRACK = false, bool X2 = false, int BITS = 0>       // BITS: 1 = write o.bits_out, 2 = mask by o.bits_in
__device__ __forceinline__ float gemm_store_staged(const f32x16 (&acc)[NT], const GemmOut &o, float *tile, uint32_t m0, uint32_t n0,
                                                   uint32_t lane) {
    float mx = 0.0f;
    const size_t bits_row = BITS ? (size_t)(m0 >> 5) * (o.N >> 6) : 0u;
    // row-group bias: a tile of 32 rows meets at most two groups of >= 32 rows -- two (scalar) divisions per tile, the two groups' quads
    // of this lane's columns fetched once per column block, a compare per row (r06; first form: a division and a load per row and quad)
    const bool rb_wide = o.rbias && o.rgroup >= 32u;
    const uint32_t rb_g0 = rb_wide ? (uint32_t)__builtin_amdgcn_readfirstlane(m0) / o.rgroup : 0u, rb_next = (rb_g0 + 1u) * o.rgroup;
    const uint32_t rb_g1 = rb_wide ? min(rb_g0 + 1u, (o.M - 1u) / o.rgroup) : 0u;
    static_assert(NT % 2u == 0u, "column blocks of two tiles");
    constexpr uint32_t RS = 68u;                                   // row stride in floats (64 + 4: the rows of a write fall on different bank groups)
    const uint32_t i = lane & 31u, kk = lane >> 5;
#pragma unroll
    for (uint32_t cb = 0; cb < NT / 2u; cb++) {
#pragma unroll
        for (uint32_t t2 = 0; t2 < 2; t2++)
#pragma unroll
            for (uint32_t g = 0; g < 4; g++) {
                const f32x16 &a = acc[2u * cb + t2];
                *reinterpret_cast<float4 *>(tile + i * RS + 32u * t2 + 8u * g + 4u * kk) =
                    make_float4(a[4u * g], a[4u * g + 1u], a[4u * g + 2u], a[4u * g + 3u]);
            }
        wave_lds_handoff();
        // (X2, a template flag and not a run-time test -- the exact engine's resident kernels have no registers to spare for it: the four
        //  W2 rows of this lane's four columns are the same for every row of the block -- lane & 15 picks the columns -- and are fetched
        //  once per block; per row one 16-byte X2 load and 16 FMAs.  First form: inside the per-row finish, 5 loads per row: the
        //  K = 128 gradient GEMM that carries the density row went from 0.36 + 0.38 ms as two passes to 0.97 as one.)
        float4 w2r[4];
        if constexpr (X2) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) w2r[j] = *reinterpret_cast<const float4 *>(o.w2 + (size_t)(n0 + 64u * cb + 4u * (lane & 15u) + j) * o.ldw2);
        }
        float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0;
        if (rb_wide) {
            const float *rp = o.rbias + n0 + 64u * cb + 4u * (lane & 15u);
            rb0 = *reinterpret_cast<const float4 *>(rp + (size_t)rb_g0 * o.ldr);
            rb1 = *reinterpret_cast<const float4 *>(rp + (size_t)rb_g1 * o.ldr);
        }
        const size_t widx = BITS ? (bits_row + (n0 >> 6) + cb) * 64u + lane : 0u;
        uint32_t word = 0u;
        if constexpr (BITS == 2) {
            if (m0 < o.M) word = o.bits_in[widx];                  // (a wave wholly past the last row owns no words)
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const uint32_t f = lane + 64u * u, r = f >> 4, c4 = f & 15u;
            const uint32_t ro = m0 + r, col = n0 + 64u * cb + 4u * c4;
            float4 v = *reinterpret_cast<const float4 *>(tile + r * RS + 4u * c4);
            if (ro < o.M) {
                if constexpr (X2) {
                    const float4 xv = *reinterpret_cast<const float4 *>(o.x2 + (size_t)ro * o.ldx2);
                    v.x += xv.x * w2r[0].x + xv.y * w2r[0].y + xv.z * w2r[0].z + xv.w * w2r[0].w;
                    v.y += xv.x * w2r[1].x + xv.y * w2r[1].y + xv.z * w2r[1].z + xv.w * w2r[1].w;
                    v.z += xv.x * w2r[2].x + xv.y * w2r[2].y + xv.z * w2r[2].z + xv.w * w2r[2].w;
                    v.w += xv.x * w2r[3].x + xv.y * w2r[3].y + xv.z * w2r[3].z + xv.w * w2r[3].w;
                }
                float4 rb = ro >= rb_next ? rb1 : rb0;
                if (o.rbias && !rb_wide) rb = *reinterpret_cast<const float4 *>(o.rbias + (size_t)(ro / o.rgroup) * o.ldr + col);
                float4 f = gemm_finish_v(v, o, ro, col, rb);
                if constexpr (BITS == 2) {
                    f.x = (word & (1u << (4u * u))) ? f.x : 0.f; f.y = (word & (2u << (4u * u))) ? f.y : 0.f;
                    f.z = (word & (4u << (4u * u))) ? f.z : 0.f; f.w = (word & (8u << (4u * u))) ? f.w : 0.f;
                }
                if constexpr (BITS == 1)
                    word |= ((f.x > 0.f ? 1u : 0u) | (f.y > 0.f ? 2u : 0u) | (f.z > 0.f ? 4u : 0u) | (f.w > 0.f ? 8u : 0u)) << (4u * u);
                if constexpr (TRACK) mx = fmaxf(fmaxf(fmaxf(mx, fabsf(f.x)), fmaxf(fabsf(f.y), fabsf(f.z))), fabsf(f.w));
                *reinterpret_cast<float4 *>(o.Y + (size_t)ro * o.ldy + col) = f;
            }
        }
        if constexpr (BITS == 1) {
            if (m0 < o.M) o.bits_out[widx] = word;
        }
        wave_lds_handoff();
    }
    return mx;
}

template <uint32_t NT>
__device__ __forceinline__ void gemm_init_acc(f32x16 (&acc)[NT], const GemmOut &o, uint32_t orow, uint32_t n0, uint32_t kk) {
    const bool vec = (o.flags & kGemmVec) != 0;
#pragma unroll
    for (uint32_t t = 0; t < NT; t++) {
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t col = n0 + 32u * t + 8u * g + 4u * kk;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((o.flags & kGemmAccum) && orow < o.M) {
                const float *yi = o.Y + (size_t)orow * o.ldy + col;
                if (vec) { if (col < o.N) v = *reinterpret_cast<const float4 *>(yi); }
                else {
                    if (col < o.N) v.x = yi[0];
                    if (col + 1u < o.N) v.y = yi[1];
                    if (col + 2u < o.N) v.z = yi[2];
                    if (col + 3u < o.N) v.w = yi[3];
                }
            }
            acc[t][4u * g] = v.x; acc[t][4u * g + 1u] = v.y; acc[t][4u * g + 2u] = v.z; acc[t][4u * g + 3u] = v.w;
        }
    }
}

}  // namespace
