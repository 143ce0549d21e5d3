// Density + colour MLP of one field, split-f16 MFMA variant (mlp_mode 1; engine: mlp_ring.h).
// Network: models.py:507-508, :581, :599-674 of /root/reference/nerf/internal/models.py.
//
// 1. COMPOSED LAYERS.  The reference's bottleneck x = W_d1 h0 + b_d1 (models.py:508, Linear(64, 256) with
//    NO activation behind it) is consumed only by linear maps -- the first colour layer and, through the
//    skip connection, the second (models.py:599-656; bottleneck_noise = 0 is enforced at construction).
//    So, exactly in real arithmetic,
//        h1 = relu( (W_c0[:, x] W_d1) h0 + W_c0[:, dir] d + (W_c0[:, x] b_d1 + b_c0) )
//        h2 = relu( W_c1[:, h1] h1 + (W_c1[:, x] W_d1) h0 + W_c1[:, dir] d + (W_c1[:, x] b_d1 + b_c1) )
//        raw density = W_d1[0, :] h0 + b_d1[0]
//    and the 256-wide x never has to exist: ucn_field_pack composes the two 256x64 products once per
//    weight update (double accumulation, then rounded to fp32), and the kernel runs 118,784 MACs per
//    sample instead of 229,632.  The difference to the reference is the fp32 rounding of x (~6e-8
//    relative) -- below the split-f16 product error.  (The bottleneck OUTPUT of predict_density is served
//    by the fp32 kernel.)
// 2. 3 x v_mfma_f32_32x32x16_f16 per 16 k's instead of 8 x v_mfma_f32_32x32x2_f32 (5.3x the rate).
// 3. The 27 view-direction inputs are one 32-wide input tile (k < 27: pos_enc(viewdir), k = 27: the
//    constant 1 whose weight column is the composed bias, k > 27: 0), produced per ray by k_dir_enc.
// 4. OUTPUT-PAIR-MAJOR ORDER (round 2).  A wave computes two output tiles at a time and finishes them before the
//    next two: the 256-wide layers never hold more than two pairs of accumulators, h1 exists only as split f16
//    operands (128 VGPRs) and h2 is consumed by the rgb head pair by pair.  Everything that is not an MFMA rides in
//    the shadow of the next pair's MFMAs: the ReLU + hi/lo split of a finished h1 pair, the rgb-head dot products
//    of a finished h2 pair.  Registers: ~300 per lane (round 1: 451), which is what lets two waves of the
//    featurisation kernel (104 registers each) share a SIMD with an MLP wave (DESIGN.md "Co-residency").
// 5. POWER-OF-TWO LAYER SCALES.  f16 operands cover 2^-24 .. 65504 and the low halves are exact only while they are
//    normal (|v| >= 2^-3).  ReLU layers are positively homogeneous, so every layer's activations can be carried at
//    a power-of-two scale chosen at pack time: activations h_k are computed as 2^e_k h_k, the packed weights of a
//    layer are W 2^(e_out - e_in) (exact), the VALU heads undo the last scale (exact).  e_k = min(e_typ, e_safe):
//    e_typ centres an estimate of the typical magnitude (second-moment propagation through the weights) at 2^4,
//    e_safe keeps a RIGOROUS bound of |h_k| (row sums of |W| times the bound of the layer below, from max |table|)
//    under 2^15 -- so no finite weights and table can overflow an operand, and a freshly initialised model
//    (|features| ~ 1e-4) or a trained one with large pre-activations both keep fp32-class products
//    (tests/test_gpu_parity.py::test_split_f16_range).
//
// Packed layout (floats at ucn_field_t::packed + pstream; 1 KiB = 256-float groups):
//   side table, kSideGroups groups:
//       [0,64)     b_d0 2^e_h0 as two bias tiles [tile][h][16]
//       [64,128)   density head W_d1[0, :] 2^-e_h0 in accumulator-slot order;  [128] b_d1[0];  [129] 2^e_in
//       [256,1280) rgb head {w_r, w_g, w_b, 0} 2^-e_h2 per accumulator slot of h2
//   stream (pairs [hi][lo] of groups; two output tiles o2 per double step, o2 innermost):
//       S0  density layer 0          [s < 4][o2]                                W_d0 2^(e_h0 - e_in)
//       A   composed layer 0         [otp][it < 3][s][o2]                       M_A = [W_c0x W_d1 | W_c0 dir | bias | 0]
//       B   for each output pair otp: skip part [it < 3][s][o2] of M_B = [W_c1x W_d1 | W_c1 dir | bias | 0],
//                                     then      [it < NTW][s][o2] of W_c1[:, 0:NW]
#include "field_plan.h"
#include "wave_dpp.h"
#include "mlp_ring.h"

namespace {

constexpr int kKS = (int)kFirstSteps;
constexpr int kDirExp = 10;                   // the direction tile (|values| <= 1 and the constant 1) is carried at 2^10
enum { E_IN = 0, E_H0 = 1, E_H1 = 2, E_H2 = 3, E_COUNT = 4 };

// per-ray direction tile: out[ray][k] = 2^kDirExp * pos_enc(viewdir, 0, deg, append_identity)[k] (coord.py:214-225)
// for k < ndir, 2^kDirExp for k == ndir, 0 beyond
__global__ __launch_bounds__(256) void k_dir_enc(const float *__restrict__ viewdirs, uint32_t ndir, uint32_t N,
                                                 float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N * 32u) return;
    const uint32_t ray = i >> 5, k = i & 31u;
    const uint32_t deg = (ndir - 3u) / 6u;
    float v;
    if (k < 3u) {
        v = viewdirs[ray * 3 + k];
    } else if (k < ndir) {
        const uint32_t kk = (k - 3u) % (3u * deg), a = kk % 3u, si = kk / 3u;
        const float sc = viewdirs[ray * 3 + a] * (float)(1u << si);
        v = (k - 3u) < 3u * deg ? sinf(sc) : sinf(sc + 1.5707963705062866f);
    } else {
        v = k == ndir ? 1.0f : 0.0f;
    }
    out[i] = ldexpf(v, kDirExp);
}

// M[j][c], j < NW, c < kCompCols:  c < 64: sum_i Wout[j][col0+i] W_d1[i][c]  |  c-64 < ND: Wout[j][col0+NB+c-64]
//                                  c == 64+ND: bias_out[j] + sum_i Wout[j][col0+i] b_d1[i]  |  else 0
__global__ __launch_bounds__(256) void k_compose(const float *__restrict__ Wout, uint32_t ldo, uint32_t col0,
                                                 const float *__restrict__ bias_out, const float *__restrict__ Wd1,
                                                 const float *__restrict__ bd1, uint32_t NB, uint32_t ND, uint32_t NW,
                                                 float *__restrict__ M) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= NW * kCompCols) return;
    const uint32_t j = i / kCompCols, c = i - j * kCompCols;
    const float *wr = Wout + (size_t)j * ldo + col0;
    float v = 0.0f;
    if (c < 64u) {
        double acc = 0.0;
        for (uint32_t k = 0; k < NB; k++) acc += (double)wr[k] * (double)Wd1[(size_t)k * 64 + c];
        v = (float)acc;
    } else if (c - 64u < ND) {
        v = wr[NB + c - 64u];
    } else if (c == 64u + ND) {
        double acc = (double)bias_out[j];
        for (uint32_t k = 0; k < NB; k++) acc += (double)wr[k] * (double)bd1[k];
        v = (float)acc;
    }
    M[i] = v;
}

// ---------------------------------------------------------------- layer scales (comment 5 at the top)
__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ x, uint64_t n, uint32_t *__restrict__ out_bits) {
    float m = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const float v = fabsf(x[i]);
        m = v > m ? v : m;            // NaN compares false: a NaN entry does not poison the bound (it poisons the output)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63u) == 0u) atomicMax(out_bits, __float_as_uint(m));    // non-negative floats order like uints
}

__device__ __forceinline__ int scale_exponent(float bound, float typical) {
    const bool ok_b = bound > 0.0f && bound < 3.0e38f;
    const bool ok_t = typical > 0.0f && typical < 3.0e38f;
    int e_safe = ok_b ? 14 - ilogbf(bound) : 0;                 // 2^e_safe * bound < 2^15
    int e_typ = ok_t ? 4 - ilogbf(typical) : e_safe;            // 2^e_typ * typical in [2^4, 2^5)
    int e = e_typ < e_safe ? e_typ : e_safe;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// One workgroup.  bound_k: rigorous max |h_k| given |features| <= m0 and |direction tile| <= 1;  typ_k: sqrt of the
// mean second moment of h_k assuming independent inputs of second moment typ_(k-1)^2 (1/2 for the direction
// encoding), halved by the ReLU.
__global__ __launch_bounds__(256) void k_field_scales(const float *__restrict__ Wd0, const float *__restrict__ bd0, uint32_t F,
                                                      const float *__restrict__ MA, const float *__restrict__ MB,
                                                      const float *__restrict__ Wc1, uint32_t ldc1, uint32_t NW, uint32_t ND,
                                                      const uint32_t *__restrict__ m0_bits, int *__restrict__ exps,
                                                      float *__restrict__ side, float *__restrict__ report) {
    __shared__ float s_max[4], s_sum[4];
    const uint32_t t = threadIdx.x;
    auto reduce = [&](float bnd, float sq, uint32_t rows, float &bound, float &typ) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bnd = fmaxf(bnd, __shfl_xor(bnd, o, 64));
            sq += __shfl_xor(sq, o, 64);
        }
        __syncthreads();
        if ((t & 63u) == 0u) { s_max[t >> 6] = bnd; s_sum[t >> 6] = sq; }
        __syncthreads();
        bound = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        typ = sqrtf(0.5f * (s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / (float)rows);
    };
    const float m0 = __uint_as_float(*m0_bits), t0 = 0.5f * m0;
    float bnd = 0.0f, sq = 0.0f, b_h0, t_h0, b_h1, t_h1, b_h2, t_h2;
    if (t < 64u) {
        float sa = 0.0f, s2 = 0.0f;
        for (uint32_t j = 0; j < F; j++) { const float w = Wd0[t * F + j]; sa += fabsf(w); s2 += w * w; }
        bnd = sa * m0 + fabsf(bd0[t]);
        sq = s2 * t0 * t0 + bd0[t] * bd0[t];
    }
    reduce(bnd, sq, 64u, b_h0, t_h0);
    auto composed = [&](const float *M, float &b_out, float &q_out) {
        float sa = 0.0f, s2 = 0.0f, da = 0.0f, d2 = 0.0f;
        for (uint32_t j = 0; j < 64u; j++) { const float w = M[t * kCompCols + j]; sa += fabsf(w); s2 += w * w; }
        for (uint32_t j = 64u; j < 64u + ND; j++) { const float w = M[t * kCompCols + j]; da += fabsf(w); d2 += w * w; }
        const float bias = M[t * kCompCols + 64u + ND];
        b_out = sa * b_h0 + da + fabsf(bias);
        q_out = s2 * t_h0 * t_h0 + 0.5f * d2 + bias * bias;
    };
    bnd = sq = 0.0f;
    if (t < NW) composed(MA, bnd, sq);
    reduce(bnd, sq, NW, b_h1, t_h1);
    bnd = sq = 0.0f;
    if (t < NW) {
        composed(MB, bnd, sq);
        float sa = 0.0f, s2 = 0.0f;
        for (uint32_t j = 0; j < NW; j++) { const float w = Wc1[(size_t)t * ldc1 + j]; sa += fabsf(w); s2 += w * w; }
        bnd += sa * b_h1;
        sq += s2 * t_h1 * t_h1;
    }
    reduce(bnd, sq, NW, b_h2, t_h2);
    if (t == 0u) {
        exps[E_IN] = scale_exponent(m0, t0);
        exps[E_H0] = scale_exponent(b_h0, t_h0);
        exps[E_H1] = scale_exponent(b_h1, t_h1);
        exps[E_H2] = scale_exponent(b_h2, t_h2);
        side[129] = ldexpf(1.0f, exps[E_IN]);
        report[0] = m0; report[1] = b_h0; report[2] = b_h1; report[3] = b_h2;
        report[4] = t0; report[5] = t_h0; report[6] = t_h1; report[7] = t_h2;
    }
}

// ---------------------------------------------------------------- packers (all scale by exact powers of two)
__device__ __forceinline__ void store_split(float w, uint32_t part, _Float16 *dst) {
    const _Float16 hh = (_Float16)w;
    *dst = part == 0u ? hh : (_Float16)(w - (float)hh);
}
// Weight pairs: dst halfs
//   [(((otp*n_in + it)*2 + s)*2 + o2)*2 + part][lane][e] =
//        part(2^x V[32(row_tile0 + 2otp + o2) + (lane&31)][col0 + 32it + perm(8s+e, lane>>5)])
// with perm(r, g) = (r&3) + 8(r>>2) + 4g (the accumulator-register order of a 32x32 MFMA tile: the 16 k's of a
// k-step are the rows of the previous layer's C/D registers 8s..8s+7), part 0 = f16(v), part 1 = f16(v - f16(v)),
// V = W for col < ld and 0 beyond, x = exps[e_out] - (col < split_col ? exps[e_lo] : e_hi_const).
__global__ __launch_bounds__(256) void k_pack_pairs(const float *__restrict__ W, uint32_t ld, uint32_t col0, uint32_t row_tile0,
                                                    uint32_t nt_out, uint32_t nt_in, const int *__restrict__ exps, int e_out,
                                                    int e_lo, uint32_t split_col, int e_hi_const, _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = nt_out * nt_in * 4u * 512u;
    if (i >= total) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, o2 = (grp >> 1) & 1u, s = (grp >> 2) & 1u, it = (grp >> 3) % nt_in;
    const uint32_t ot = 2u * ((grp >> 3) / nt_in) + o2;
    const uint32_t r = 8u * s + e;
    const uint32_t row = 32u * (row_tile0 + ot) + (lane & 31u);
    const uint32_t col = col0 + 32u * it + (r & 3u) + 8u * (r >> 2) + 4u * (lane >> 5);
    const int x = exps[e_out] - (col < split_col ? exps[e_lo] : e_hi_const);
    const float w = col < ld ? ldexpf(W[(size_t)row * ld + col], x) : 0.0f;
    store_split(w, part, dst + i);
}
// First layer (inputs in natural order k = 16s + 8g + e, zero-padded to KS k-steps):
//   dst[((s*2 + ot)*2 + part)][lane][e] = part(2^(e_h0 - e_in) W[32ot + (lane&31)][16s + 8(lane>>5) + e])
__global__ __launch_bounds__(256) void k_pack_first_s(const float *__restrict__ W, uint32_t F, uint32_t KS,
                                                      const int *__restrict__ exps, _Float16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 2u * KS * 2u * 512u) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t part = grp & 1u, ot = (grp >> 1) & 1u, s = grp >> 2;
    const uint32_t k = 16u * s + 8u * (lane >> 5) + e;
    const float w = k < F ? ldexpf(W[(size_t)(32u * ot + (lane & 31u)) * F + k], exps[E_H0] - exps[E_IN]) : 0.0f;
    store_split(w, part, dst + i);
}
// side table: bias tiles of the first layer, density head, rgb head
__global__ __launch_bounds__(256) void k_pack_side(const float *__restrict__ bd0, const float *__restrict__ Wd1,
                                                   const float *__restrict__ bd1, const float *__restrict__ Wrgb, uint32_t NW,
                                                   const int *__restrict__ exps, float *__restrict__ side) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 64u) {                                           // [tile][h][r]
        const uint32_t r = i & 15u, h = (i >> 4) & 1u, t = i >> 5;
        side[i] = ldexpf(bd0[acc_row(t, r, h)], exps[E_H0]);
    } else if (i < 128u) {                                   // slot = (it*16 + r)*2 + h
        const uint32_t k = i - 64u, h = k & 1u, r = (k >> 1) & 15u, it = k >> 5;
        side[i] = ldexpf(Wd1[acc_row(it, r, h)], -exps[E_H0]);                  // row 0 of W_d1 [NB, 64]
    } else if (i == 128u) {
        side[i] = bd1[0];
    } else if (i >= 256u && i < 256u + NW * 4u) {            // ((t*16 + r)*2 + h)*4 + c
        const uint32_t k = i - 256u, c = k & 3u, slot = k >> 2;
        const uint32_t h = slot & 1u, r = (slot >> 1) & 15u, t = slot >> 5;
        side[i] = c < 3u ? ldexpf(Wrgb[(size_t)c * NW + acc_row(t, r, h)], -exps[E_H2]) : 0.0f;
    }
}

// ---------------------------------------------------------------- the kernel
__device__ __forceinline__ void dma_group(const float *g, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_byte), "v"(voff), "s"(g) : "memory");
}

// four accumulator registers 4Q..4Q+3 of h2 tile T through ReLU into the rgb head's partial sums
template <int T, int Q>
__device__ __forceinline__ void head_quarter(const f32x16 &acc, const float *side, int h, float &s0, float &s1, float &s2) {
    const float4 *pr = reinterpret_cast<const float4 *>(side + 256) + T * 32 + h;
#pragma unroll
    for (int r = 4 * Q; r < 4 * Q + 4; r++) {
        const float4 w = pr[2 * r];
        const float v = relu_bits(acc[r]);
        s0 = fmaf(v, w.x, s0);
        s1 = fmaf(v, w.y, s1);
        s2 = fmaf(v, w.z, s2);
    }
}
// eighth E (0..7) of the rgb head of a finished pair: tile E / 4, registers 4 (E % 4) ..
template <int T0, int E>
__device__ __forceinline__ void head_eighth(const f32x16 (&pair)[2], const float *side, int h, float &s0, float &s1, float &s2) {
    head_quarter<T0 + E / 4, E % 4>(pair[E / 4], side, h, s0, s1, s2);
}

#ifdef UCN_EXP_STOP        // experiment builds: leave the kernel after a phase (timing by difference; results are garbage)
#define UCN_STOP_AT(k, accpair)                                                                       \
    do {                                                                                              \
        if constexpr (UCN_EXP_STOP == (k)) {                                                          \
            float keep = 0.0f;                                                                        \
            for (int r = 0; r < 16; r++) keep += accpair[0][r] + accpair[1][r];                       \
            if (live && h == 0) a.rgb[(size_t)oi * 3] = keep;                                         \
            ring.drain();                                                                             \
            __syncthreads();                                                                          \
            return;                                                                                   \
        }                                                                                             \
    } while (0)
#else
#define UCN_STOP_AT(k, accpair) do { } while (0)
#endif
#ifdef UCN_EXP_TIMING      // experiment builds: s_memtime stamps of workgroup 64's wave 0 into the (otherwise unused) bottleneck buffer
#define UCN_STAMP(i) do { if (blockIdx.x == UCN_EXP_TIMING && threadIdx.x == 0) a.bott[i] = (float)(__builtin_readcyclecounter() - t_begin); } while (0)
#else
#define UCN_STAMP(i) do { } while (0)
#endif
template <int NTW, bool RGB, int CHUNK, int STAGE>
__global__ __launch_bounds__(256) void k_field_mlp_h(MlpArgs a) {
#ifdef UCN_EXP_TIMING
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) float s_lds[];   // [side table][ring]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // every wave runs the whole program (workgroup-wide barriers); waves past the end of the batch
    // compute on a clamped sample and store nothing
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 32u;
    bool live = b0 + j < a.B;
    uint32_t b = live ? b0 + j : a.B - 1;
    if (a.idx) {                                         // compacted colour pass: slot -> sample through the alive list
        const uint32_t cnt = *a.count;
        if (blockIdx.x * 128u >= cnt) return;            // workgroup-uniform, before any barrier or DMA
        live = b0 + j < cnt;
        b = a.idx[live ? b0 + j : cnt - 1];
    }
    const uint32_t oi = out_index(a, b);                 // position in the [ray][sample]-ordered outputs

    constexpr int NP = NTW / 2;                          // output pairs of a hidden layer
    constexpr int GA = 4 * kKS;                          // S0: kKS double steps
    constexpr int GB = GA + NP * 24;                     // A: NP pairs x 6 double steps
    constexpr int PB = 24 + NTW * 8;                     // B: per pair 6 + 2 NTW double steps
    constexpr int NG = RGB ? GB + NP * PB : GA;

    // ---- the wave's only global loads: its feature values and the ray's direction tile
    float fv[kKS][8];
    load_features<kKS>(a, b, h, fv);
    f32x16 ev;
    if constexpr (RGB) {
        const float4 *ep = reinterpret_cast<const float4 *>(a.dir_bias + (size_t)ray_index(a, b) * 32 + 4 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = ep[2 * r4];
            ev[4 * r4 + 0] = v.x; ev[4 * r4 + 1] = v.y; ev[4 * r4 + 2] = v.z; ev[4 * r4 + 3] = v.w;
        }
    }
    // ---- weight DMA: side table, then the first kLead chunks of the ring
    const float *side = s_lds;
    Ring<NG, CHUNK, 4, kRingSlots, kLead, STAGE> ring(a.packed + a.pstream + kSideGroups * 256, s_lds + kSideGroups * 256, lane, wave);
    {
        const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)s_lds;
#pragma unroll
        for (int i = 0; i < kSideGroups / 4; i++)
            dma_group(a.packed + a.pstream + (size_t)(i * 4 + wave) * 256, lbase + (uint32_t)(i * 4 + wave) * 1024u, (uint32_t)lane * 16u);
    }
    rstatic_for<kLead>([&](auto c) { ring.template issue_chunk<c.value>(); });
    UCN_STAMP(0);
    if constexpr (STAGE > 0) ring.drain();          // the side table came by DMA
    ring.template boundary<0>();                    // side table + chunk 0 landed (the later chunks stay in flight)
    UCN_STAMP(1);
    if constexpr (RGB) {
        // (r05) the direction tile's loads are PINNED as landed here, at the top where the stream is only just starting: left to the compiler
        // their wait sits in front of the tile's first use, behind the density stage -- a `s_waitcnt vmcnt(3..0)` that, counted against a
        // hardware counter full of LDS-DMA pieces the compiler cannot see, drained the whole weight look-ahead once per pass
#pragma unroll
        for (int r = 0; r < 16; r++) asm volatile("" : "+v"(ev[r]));
    }
    const float in_scale = side[129];

    // ---- density layer 0: F -> 64, ReLU (accumulators start from the bias tiles)
    f32x16 acc0[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float4 *pb = reinterpret_cast<const float4 *>(side + t * 32 + h * 16);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = pb[r4];
            acc0[t][4 * r4 + 0] = v.x; acc0[t][4 * r4 + 1] = v.y; acc0[t][4 * r4 + 2] = v.z; acc0[t][4 * r4 + 3] = v.w;
        }
    }
    OpPipe pipe;
    pipe_prime<NG>(pipe, ring);
    {
        h8 fhi[kKS], flo[kKS];
#pragma unroll
        for (int s = 0; s < kKS; s++) {
#pragma unroll
            for (int e = 0; e < 8; e++) fv[s][e] *= in_scale;
            rsplit8(fv[s], fhi[s], flo[s]);
        }
        rstatic_for<kKS>([&](auto s) { dstep<4 * s.value, NG>(acc0[0], acc0[1], fhi[s.value], flo[s.value], pipe, ring); });
        relu_tile(acc0[0]);
        relu_tile(acc0[1]);
        UCN_STAMP(2);
    }
    if constexpr (RGB) UCN_STOP_AT(1, acc0);
    // ---- raw density = row 0 of the second density layer, on the VALU (models.py:508,581): this lane holds
    //      32 of the 64 hidden units of its sample
    {
        const float *pd = side + 64;
        float part = 0.0f;
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int r = 0; r < 16; r++) part = fmaf(acc0[it][r], pd[(it * 16 + r) * 2 + h], part);
        const float raw = xor32_sum(part) + pd[64];
        if (live && h == 0 && a.idx == nullptr) a.density[oi] = softplus(raw + a.density_bias);
    }
    if constexpr (RGB) {
        HPair in[3];                                         // h0 (2 tiles) and the direction tile
#pragma unroll
        for (int s = 0; s < 2; s++) {
            split_half<false>(acc0[0], s, in[0]);
            split_half<false>(acc0[1], s, in[1]);
            split_half<false>(ev, s, in[2]);
        }
        UCN_STAMP(3);
        // ---- A: composed colour layer 0, pair by pair; pair p - 1 is ReLU'd and split under pair p's MFMAs
        HPair h1s[NTW];
        f32x16 acc[2][2];                                    // pair q of the whole program lives in acc[q % 2]
        rstatic_for<NP>([&](auto pc) {
            constexpr int p = pc.value;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[p % 2][0][r] = acc[p % 2][1][r] = 0.0f;       // biases ride in the direction tile
            rstatic_for<6>([&](auto ic) {                                                 // [it < 3][s]
                constexpr int i = ic.value, G = GA + (p * 6 + i) * 4;
                if constexpr (p > 0 && i < 4)
                    dstep<G, NG, 7, 0>(acc[p % 2][0], acc[p % 2][1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring,
                                       [&] { split_half<true>(acc[(p - 1) % 2][i / 2], i % 2, h1s[2 * (p - 1) + i / 2]); });
                else
                    dstep<G, NG>(acc[p % 2][0], acc[p % 2][1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring);
            });
        });
        UCN_STOP_AT(2, acc[(NP - 1) % 2]);
        UCN_STAMP(4);
        // ---- B: colour layer 1, pair by pair: skip part (reads `in`), then the hidden part (reads h1s).  Under the skip
        //      part's MFMAs: the split of A's last pair (p = 0) or the rgb head of pair p - 1
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        rstatic_for<NP>([&](auto pc) {
            constexpr int p = pc.value, q = NP + p, qp = q % 2, qq = (q - 1) % 2;   // this pair's and the previous pair's slot
#pragma unroll
            for (int r = 0; r < 16; r++) acc[qp][0][r] = acc[qp][1][r] = 0.0f;
            rstatic_for<6>([&](auto ic) {
                constexpr int i = ic.value, G = GB + p * PB + i * 4;
                if constexpr (p == 0 && i < 4)
                    dstep<G, NG, 7, 0>(acc[qp][0], acc[qp][1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring,
                                       [&] { split_half<true>(acc[qq][i / 2], i % 2, h1s[2 * (NP - 1) + i / 2]); });
                else if constexpr (p > 0)
                    dstep<G, NG, 3, 4>(acc[qp][0], acc[qp][1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring,
                                       [&] { head_eighth<2 * (p - 1), i>(acc[qq], side, h, s0, s1, s2); });
                else
                    dstep<G, NG>(acc[qp][0], acc[qp][1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring);
            });
            UCN_STAMP(5 + 2 * p);
            rstatic_for<2 * NTW>([&](auto ic) {                                           // [it < NTW][s]
                constexpr int i = ic.value, G = GB + p * PB + 24 + i * 4;
                if constexpr (p > 0 && i < 2)
                    dstep<G, NG, 3, 4>(acc[qp][0], acc[qp][1], h1s[i / 2].hi[i % 2], h1s[i / 2].lo[i % 2], pipe, ring,
                                       [&] { head_eighth<2 * (p - 1), 6 + i>(acc[qq], side, h, s0, s1, s2); });
                else
                    dstep<G, NG>(acc[qp][0], acc[qp][1], h1s[i / 2].hi[i % 2], h1s[i / 2].lo[i % 2], pipe, ring);
            });
            UCN_STOP_AT(3 + p, acc[qp]);
        });
        UCN_STAMP(13);
        rstatic_for<8>([&](auto e) { head_eighth<2 * (NP - 1), e.value>(acc[(2 * NP - 1) % 2], side, h, s0, s1, s2); });
        UCN_STAMP(14);
        // ---- rgb: sigmoid + padding (models.py:657-674)
        s0 = xor32_sum(s0);
        s1 = xor32_sum(s1);
        s2 = xor32_sum(s2);
        if (live && h == 0) {
            const float pad = a.rgb_padding;
            const float v[3] = {s0 + a.b_rgb[0], s1 + a.b_rgb[1], s2 + a.b_rgb[2]};
            const size_t o = (size_t)oi * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float sg = 1.0f / (1.0f + expf(-(a.rgb_premult * v[c] + a.rgb_bias)));
                a.rgb[o + c] = sg * (1.0f + 2.0f * pad) - pad;
            }
        }
    }
}


// ---- 8-wave variant: 512-thread workgroups, TWO waves per SIMD, 256 samples per pass over the weight stream.
// k_field_mlp_h above streams 464 KiB of weights through LDS per 128 samples, and the LDS-DMA path lands ~10 B per
// clock and CU (25 GB/s: cycle stamps in profiles/r02*/mlp_timeline.txt; the same figure as MI355X_MICROARCH.md's
// "ldsdma-fill" row): 4 KiB per double step = 393 cycles where the six MFMAs need 192 -- the kernel sits at that floor.
// Here eight waves share the stream, so the DMA moves half the bytes per sample and the two waves of a SIMD take turns
// on the matrix pipe: one computes while the other splits / runs the rgb head / waits for LDS.  That needs <= 256
// registers per lane: ONE accumulator pair, nothing in MFMA shadows (the partner wave is the shadow), a short
// operand pipe.
template <int NTW, int NWAVES, int CHUNK, int SLOTS, int LEAD, int DEPTH>
__global__ __launch_bounds__(NWAVES * 64, 2) void k_field_mlp_h8(MlpArgs a) {
#ifdef UCN_EXP_TIMING
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define UCN_STAMP8(i) do { if (blockIdx.x == UCN_EXP_TIMING && (threadIdx.x & 255) == 0) a.bott[(threadIdx.x >> 8) * 32 + (i)] = (float)(__builtin_readcyclecounter() - t_begin); } while (0)
#else
#define UCN_STAMP8(i) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) float s_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t b0 = (blockIdx.x * NWAVES + wave) * 32u;
    bool live = b0 + j < a.B;
    uint32_t b = live ? b0 + j : a.B - 1;
    if (a.idx) {                                         // compacted colour pass (see k_field_mlp_h)
        const uint32_t cnt = *a.count;
        if (blockIdx.x * (NWAVES * 32u) >= cnt) return;
        live = b0 + j < cnt;
        b = a.idx[live ? b0 + j : cnt - 1];
    }
    const uint32_t oi = out_index(a, b);
    constexpr int NP = NTW / 2;
    constexpr int GA = 4 * kKS;
    constexpr int GB = GA + NP * 24;
    constexpr int PB = 24 + NTW * 8;
    constexpr int NG = GB + NP * PB;

    float fv[kKS][8];
    load_features<kKS>(a, b, h, fv);
    f32x16 ev;
    {
        const float4 *ep = reinterpret_cast<const float4 *>(a.dir_bias + (size_t)ray_index(a, b) * 32 + 4 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = ep[2 * r4];
            ev[4 * r4 + 0] = v.x; ev[4 * r4 + 1] = v.y; ev[4 * r4 + 2] = v.z; ev[4 * r4 + 3] = v.w;
        }
    }
    const float *side = s_lds;
    Ring<NG, CHUNK, NWAVES, SLOTS, LEAD> ring(a.packed + a.pstream + kSideGroups * 256, s_lds + kSideGroups * 256, lane, wave);
    {
        const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)s_lds;
#pragma unroll
        for (int i = 0; i < kSideGroups / NWAVES; i++)
            dma_group(a.packed + a.pstream + (size_t)(i * NWAVES + wave) * 256, lbase + (uint32_t)(i * NWAVES + wave) * 1024u, (uint32_t)lane * 16u);
    }
    UCN_STAMP8(0);
    rstatic_for<LEAD>([&](auto c) { ring.template issue_chunk<c.value>(); });
    UCN_STAMP8(1);
    ring.template boundary<0>();
    UCN_STAMP8(2);
    // (r05) the direction tile's loads are PINNED as landed here, at the top where the stream is only just starting: left to the compiler
    // their wait sits in front of the tile's first use, behind the density stage -- a `s_waitcnt vmcnt(3..0)` that, counted against a
    // hardware counter full of LDS-DMA pieces the compiler cannot see, drained the whole weight look-ahead once per pass
#pragma unroll
    for (int r = 0; r < 16; r++) asm volatile("" : "+v"(ev[r]));
    const float in_scale = side[129];

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float4 *pb = reinterpret_cast<const float4 *>(side + t * 32 + h * 16);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = pb[r4];
            acc[t][4 * r4 + 0] = v.x; acc[t][4 * r4 + 1] = v.y; acc[t][4 * r4 + 2] = v.z; acc[t][4 * r4 + 3] = v.w;
        }
    }
    OpPipeD<DEPTH> pipe;
    pipe_prime_d<NG>(pipe, ring);
    {
        h8 fhi[kKS], flo[kKS];
#pragma unroll
        for (int s = 0; s < kKS; s++) {
#pragma unroll
            for (int e = 0; e < 8; e++) fv[s][e] *= in_scale;
            rsplit8(fv[s], fhi[s], flo[s]);
        }
        rstatic_for<kKS>([&](auto s) { dstep_d<4 * s.value, NG>(acc[0], acc[1], fhi[s.value], flo[s.value], pipe, ring); });
    }
    {
        const float *pd = side + 64;
        float part = 0.0f;
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int r = 0; r < 16; r++) part = fmaf(relu_bits(acc[it][r]), pd[(it * 16 + r) * 2 + h], part);
        const float raw = xor32_sum(part) + pd[64];
        if (live && h == 0 && a.idx == nullptr) a.density[oi] = softplus(raw + a.density_bias);
    }
    UCN_STAMP8(3);
    HPair in[3];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        split_half<true>(acc[0], s, in[0]);
        split_half<true>(acc[1], s, in[1]);
        split_half<false>(ev, s, in[2]);
    }
    UCN_STAMP8(4);
    HPair h1s[NTW];
    rstatic_for<NP>([&](auto pc) {
        constexpr int p = pc.value;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][r] = acc[1][r] = 0.0f;
        rstatic_for<6>([&](auto ic) {
            constexpr int i = ic.value, G = GA + (p * 6 + i) * 4;
            dstep_d<G, NG>(acc[0], acc[1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring);
        });
#pragma unroll
        for (int s = 0; s < 2; s++) {
            split_half<true>(acc[0], s, h1s[2 * p]);
            split_half<true>(acc[1], s, h1s[2 * p + 1]);
        }
    });
    UCN_STAMP8(5);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    rstatic_for<NP>([&](auto pc) {
        constexpr int p = pc.value;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][r] = acc[1][r] = 0.0f;
        rstatic_for<6>([&](auto ic) {
            constexpr int i = ic.value, G = GB + p * PB + i * 4;
            dstep_d<G, NG>(acc[0], acc[1], in[i / 2].hi[i % 2], in[i / 2].lo[i % 2], pipe, ring);
        });
        rstatic_for<2 * NTW>([&](auto ic) {
            constexpr int i = ic.value, G = GB + p * PB + 24 + i * 4;
            dstep_d<G, NG>(acc[0], acc[1], h1s[i / 2].hi[i % 2], h1s[i / 2].lo[i % 2], pipe, ring);
        });
        UCN_STAMP8(6 + 2 * p);
        rstatic_for<8>([&](auto e) { head_eighth<2 * p, e.value>(acc, side, h, s0, s1, s2); });
        UCN_STAMP8(7 + 2 * p);
    });
    s0 = xor32_sum(s0);
    s1 = xor32_sum(s1);
    s2 = xor32_sum(s2);
    if (live && h == 0) {
        const float pad = a.rgb_padding;
        const float v[3] = {s0 + a.b_rgb[0], s1 + a.b_rgb[1], s2 + a.b_rgb[2]};
        const size_t o = (size_t)oi * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float sg = 1.0f / (1.0f + expf(-(a.rgb_premult * v[c] + a.rgb_bias)));
            a.rgb[o + c] = sg * (1.0f + 2.0f * pad) - pad;
        }
    }
}

}  // namespace

int ucn_h_pack(const ucn_field_t *f, const PackPlan &pl, hipStream_t st) {
    const uint32_t NB = f->n_bottleneck, NW = f->n_width, ND = f->n_dir;
    float *side = f->packed + pl.pstream;
    float *stream = side + kSideGroups * 256;
    float *MA = f->packed + pl.pcomp, *MB = MA + (size_t)NW * kCompCols;
    float *scratch = MB + (size_t)NW * kCompCols;                       // [0] max |table| bits, [1..4] exponents, [8..15] report
    int *exps = reinterpret_cast<int *>(scratch + 1);
    const uint64_t n_floats = (uint64_t)pl.n_groups * 256;
    hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up(n_floats, 256)), dim3(256), 0, st, side, (uint32_t)n_floats);
    hipLaunchKernelGGL(k_fill_zero, dim3(1), dim3(256), 0, st, scratch, 16u);
    // composed matrices (fp32, double-accumulated)
    const dim3 cg(ucn_div_up((uint64_t)NW * kCompCols, 256));
    hipLaunchKernelGGL(k_compose, cg, dim3(256), 0, st, f->w_c0, NB + ND, 0u, f->b_c0, f->w_d1, f->b_d1, NB, ND, NW, MA);
    hipLaunchKernelGGL(k_compose, cg, dim3(256), 0, st, f->w_c1, NW + NB + ND, NW, f->b_c1, f->w_d1, f->b_d1, NB, ND, NW, MB);
    // layer scales
    const uint64_t n_emb = (uint64_t)f->offsets_host[f->num_levels] * f->level_dim;
    hipLaunchKernelGGL(k_absmax, dim3(1024), dim3(256), 0, st, f->embeddings, n_emb, reinterpret_cast<uint32_t *>(scratch));
    hipLaunchKernelGGL(k_field_scales, dim3(1), dim3(256), 0, st, f->w_d0, f->b_d0, pl.F, MA, MB, f->w_c1, NW + NB + ND, NW, ND,
                       reinterpret_cast<const uint32_t *>(scratch), exps, side, scratch + 8);
    hipLaunchKernelGGL(k_pack_side, dim3(ucn_div_up(256u + NW * 4u, 256)), dim3(256), 0, st, f->b_d0, f->w_d1, f->b_d1, f->w_rgb, NW,
                       exps, side);
    uint64_t g = 0;                                                     // position in groups
    auto pairs = [&](const float *W, uint32_t ld, uint32_t row_tile0, uint32_t nto, uint32_t nti, int e_out, int e_lo,
                     uint32_t split_col) {
        hipLaunchKernelGGL(k_pack_pairs, dim3(ucn_div_up((uint64_t)nto * nti * 2048, 256)), dim3(256), 0, st, W, ld, 0u,
                           row_tile0, nto, nti, exps, e_out, e_lo, split_col, kDirExp, reinterpret_cast<_Float16 *>(stream + g * 256));
        g += (uint64_t)nto * nti * 4;
    };
    hipLaunchKernelGGL(k_pack_first_s, dim3(ucn_div_up(2ull * kFirstSteps * 1024, 256)), dim3(256), 0, st, f->w_d0, pl.F,
                       kFirstSteps, exps, reinterpret_cast<_Float16 *>(stream));
    g += 2 * kFirstSteps * 2;                                            // S0
    pairs(MA, kCompCols, 0, pl.NTW, 3, E_H1, E_H0, 64u);                 // A: all pairs, [otp][it][s][o2]
    for (uint32_t otp = 0; otp < pl.NTW / 2; otp++) {                    // B: per output pair
        pairs(MB, kCompCols, 2 * otp, 2, 3, E_H2, E_H0, 64u);
        pairs(f->w_c1, NW + NB + ND, 2 * otp, 2, pl.NTW, E_H2, E_H1, 0xFFFFFFFFu);
    }
    UCN_REQUIRE(g == stream_groups_h(pl.NTW), "field_pack: internal stream length mismatch (%llu)", (unsigned long long)g);
    UCN_LAUNCH_CHECK("field_pack (split-f16)");
    return 0;
}

int ucn_h_dir_enc(const ucn_field_t *f, const float *viewdirs, uint32_t N, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_dir_enc, dim3(ucn_div_up((uint64_t)N * 32, 256)), dim3(256), 0, st, viewdirs, f->n_dir, N, out);
    UCN_LAUNCH_CHECK("field_dir_bias (split-f16)");
    return 0;
}

int ucn_h_launch(const PackPlan &pl, const MlpArgs &a, dim3 grid, hipStream_t st) {
#ifndef UCN_EXP_TIMING
    UCN_REQUIRE(a.bott == nullptr, "field_mlp: mlp_mode 1 composes the bottleneck away; request bottleneck_out with mlp_mode 0");
#endif
    // a.small_ring: the 64 KiB ring (+ side table = 72 KiB), so that one 512-thread featurisation workgroup holding
    // 88 KiB can share the CU (DESIGN.md "Co-residency"); default: the 128 KiB ring
#define UCN_MLP_H(NTW_, RGB_)                                                                                              \
    do {                                                                                                                   \
        if (a.small_ring)                                                                                                  \
            hipLaunchKernelGGL((k_field_mlp_h<NTW_, RGB_, kRingChunkSmall, 0>), grid, dim3(256), ring_lds_bytes(kRingChunkSmall), st, a); \
        else                                                                                                               \
            hipLaunchKernelGGL((k_field_mlp_h<NTW_, RGB_, kRingChunk, UCN_MLP4_STAGE>), grid, dim3(256), ring_lds_bytes(kRingChunk), st, a); \
    } while (0)
#ifndef UCN_MLP8_DEPTH
#define UCN_MLP8_DEPTH 2
#endif
#ifndef UCN_MLP4_STAGE
#define UCN_MLP4_STAGE 0
#endif
#ifndef UCN_MLP8_CHUNK
#define UCN_MLP8_CHUNK 32
#define UCN_MLP8_SLOTS 4
#define UCN_MLP8_LEAD 2
#endif
#ifndef UCN_MLP2_CHUNK          // 64 KiB ring of the two-workgroups-per-CU kernel
#define UCN_MLP2_CHUNK 16
#define UCN_MLP2_SLOTS 4
#define UCN_MLP2_LEAD 2
#endif
    // default for the 256-wide colour field: 4-wave workgroups of <= 256 registers, TWO per CU (72 KiB of LDS each), so
    // every SIMD holds two waves of DIFFERENT workgroups: one wave's MFMAs run while the other splits, reads LDS, issues
    // DMA or sits in its prologue.  Within one wave nothing overlaps an MFMA (tools/mfma_valu_bench.hip: 33 cycles per MFMA
    // alone, 33 + 2.6 per VALU instruction behind it), and the two waves of ONE workgroup move in lockstep between the
    // ring's barriers.  UCN_MLP_WAVES = 8 / 1: the 8-wave and the one-workgroup-per-CU kernels (experiments).
    static const int waves = getenv("UCN_MLP_WAVES") ? atoi(getenv("UCN_MLP_WAVES")) : 4;
    if (a.rgb != nullptr && pl.NTW == 8 && !a.small_ring && waves == 8) {
        const dim3 grid8(ucn_div_up(a.B, 256));
        hipLaunchKernelGGL((k_field_mlp_h8<8, 8, UCN_MLP8_CHUNK, UCN_MLP8_SLOTS, UCN_MLP8_LEAD, UCN_MLP8_DEPTH>), grid8, dim3(512),
                           (kSideGroups + UCN_MLP8_SLOTS * UCN_MLP8_CHUNK) * 1024, st, a);
        UCN_LAUNCH_CHECK("field_mlp (split-f16, 8 waves)");
        return 0;
    }
    if (a.rgb != nullptr && pl.NTW == 8 && (waves == 4 || a.small_ring)) {     // (also the co-resident shape: 72 KiB, 254 registers)
        hipLaunchKernelGGL((k_field_mlp_h8<8, 4, UCN_MLP2_CHUNK, UCN_MLP2_SLOTS, UCN_MLP2_LEAD, UCN_MLP8_DEPTH>), grid, dim3(256),
                           (kSideGroups + UCN_MLP2_SLOTS * UCN_MLP2_CHUNK) * 1024, st, a);
        UCN_LAUNCH_CHECK("field_mlp (split-f16, 2 workgroups per CU)");
        return 0;
    }
    if (a.rgb == nullptr) {
        if (pl.NTW == 8) UCN_MLP_H(8, false);
        else UCN_MLP_H(2, false);
    } else {
        if (pl.NTW == 8) UCN_MLP_H(8, true);
        else UCN_MLP_H(2, true);
    }
#undef UCN_MLP_H
    UCN_LAUNCH_CHECK("field_mlp (split-f16)");
    return 0;
}
