// Shared between field_mlp.hip (fp32-input MFMA, mlp_mode 0) and field_mlp_h.hip (split-f16 MFMA,
// mlp_mode 1): layout of ucn_field_t::packed, kernel arguments, small device helpers.
#pragma once
#include "mfma_chain.h"
#include "wave_dpp.h"
#include "mlp_ring.h"

struct PackPlan {
    uint64_t p0, pstream, phead, pcomp, total;        // float offsets into ucn_field_t::packed
    uint32_t F, KQ, NTB, NTW, n_groups;               // n_groups: weight-stream length of the ACTIVE mode
    bool prop;
};

// stream lengths (1 KiB groups), before padding to whole LDS chunks
static inline uint32_t stream_groups_f32(uint32_t NTB, uint32_t NTW) {
    return NTB * 2 * 4 + NTW * NTB * 4 + NTW * (NTB * 4 + NTW * 4);
}
// split-f16 stream (field_mlp_h.hip), 4 groups per double step: first layer (kFirstSteps k-steps), composed layer 0
// (NTW/2 output pairs x 3 input tiles x 2 k-steps), colour layer 1 (per output pair: the 3 composed input tiles,
// then the NTW hidden tiles).  A side table of kSideGroups groups sits in front of it.
constexpr uint32_t kFirstSteps = 4;                   // first layer: F <= 16*kFirstSteps inputs (zero padded)
constexpr uint32_t kCompCols = 96;                    // composed layers: [64 hidden | 27 direction | bias | 0...]
static inline uint32_t stream_groups_h(uint32_t NTW) {
    return 4 * (kFirstSteps + 3 * NTW + NTW * (3 + NTW));
}

static inline int make_plan(const ucn_field_t *f, PackPlan *pl) {
    UCN_REQUIRE(f, "field: null descriptor");
    UCN_REQUIRE(f->mlp_mode <= 1, "field: mlp_mode must be 0 (fp32 MFMA) or 1 (split-f16 MFMA)");
    pl->F = f->num_levels * f->level_dim;
    UCN_REQUIRE(pl->F % 2 == 0 && pl->F >= 2, "field: num_levels*level_dim must be even, got %u", pl->F);
    pl->KQ = pl->F / 2;
    pl->prop = f->n_bottleneck == 1;
    pl->p0 = 0;
    uint64_t o = 2ull * pl->KQ * 64;
    const uint64_t o_h = 2ull * ((pl->F + 15) / 16) * 2 * 256;   // split-f16 first layer: [ot][s][hi,lo] x 1 KiB
    if (o_h > o) o = o_h;
    o = (o + 255) & ~255ull;                           // keep the stream 1 KiB aligned
    if (pl->prop) {
        pl->NTB = pl->NTW = 0;
        pl->n_groups = 0;
        pl->pstream = o;
        pl->phead = o; o += 64;
        pl->pcomp = o;
    } else {
        UCN_REQUIRE((f->n_bottleneck == 256 && f->n_width == 256) || (f->n_bottleneck == 64 && f->n_width == 64),
                    "field: supported (bottleneck_width, net_width_viewdirs) are (256,256) and (64,64), got (%u,%u)",
                    f->n_bottleneck, f->n_width);
        UCN_REQUIRE(f->w_c0 && f->w_c1 && f->w_rgb && f->b_c0 && f->b_c1 && f->b_rgb, "field: colour MLP weights missing");
        UCN_REQUIRE(f->n_dir >= 3 && f->n_dir <= 27, "field: n_dir must be in [3,27] (one 32-wide tile incl. the bias slot)");
        UCN_REQUIRE(f->mlp_mode == 0 || pl->F <= 16 * kFirstSteps,
                    "field: mlp_mode 1 supports num_levels*level_dim <= %u, got %u", 16 * kFirstSteps, pl->F);
        pl->NTB = f->n_bottleneck / 32;
        pl->NTW = f->n_width / 32;
        const uint32_t g0 = (stream_groups_f32(pl->NTB, pl->NTW) + kChunkGroups - 1) / kChunkGroups * kChunkGroups;
        const uint32_t g1 = kSideGroups + (stream_groups_h(pl->NTW) + kRingPad - 1) / kRingPad * kRingPad + kRingPad;   // + slack: the prologue's DMA may run past a short stream
        pl->n_groups = f->mlp_mode == 1 ? g1 : g0;
        pl->pstream = o; o += (uint64_t)(g0 > g1 ? g0 : g1) * 256;
        pl->phead = o; o += (uint64_t)pl->NTW * 128;
        pl->pcomp = o; o += 2ull * f->n_width * kCompCols + 16;        // composed fp32 matrices + scale scratch (mode 1)
    }
    pl->total = o;
    return 0;
}

struct MlpArgs {
    const float *feat;        // [L][B][C]
    const float *packed;
    const float *b_d0, *b_d1, *b_c0, *b_c1, *b_rgb;
    const float *dir_bias;    // mode 0: [rays][2][NW] per-ray biases; mode 1: [rays][32] direction encoding + 1.0
    float *density, *rgb, *bott;
    uint32_t B, spr, C, F, n_chunks;
    uint32_t cshift;                 // log2(C) when C is 1, 2, 4 or 8 and the features span < 4 GiB (load_features), else 0xFF
    uint32_t n_rays, rays_fastest;   // rays_fastest: feature index b = s*n_rays + ray (else ray*spr + s)
    uint32_t small_ring;             // mode 1: 64 KiB weight ring (co-resident launches)
    const uint32_t *idx, *count;     // compacted colour pass: tile slot i evaluates sample idx[i], i < *count (else NULL)
    uint64_t p0, pstream, phead;
    float density_bias, rgb_premult, rgb_bias, rgb_padding;
};

// The wave's feature values from feat [L][B][C]: lane (j, h) supplies k = 16 s + 8 h + e of its sample b.  For C in
// {1, 2, 4, 8} (8 % C == 0) the (level, channel) of k splits into a wave-uniform part -- s, e: scalar arithmetic, a scalar
// base pointer per element -- and ONE per-lane byte offset shared by all elements: no vector integer work per load.
// (The general form below divides k by C per element: 35 u32 divisions, ~650 of the 2860 VALU instructions a tile of the
// NeRF-level kernel issued.  Removing them did not move the kernel time -- 167-171 ms per frame before and after: the
// prologue of one workgroup runs under the MFMA phase of the other one on the CU -- so this is tidiness, not speed.)
template <int KS>
__device__ __forceinline__ void load_features(const MlpArgs &a, uint32_t b, int h, float (&fv)[KS][8]) {
    if (a.cshift != 0xFFu) {
        const uint32_t BC = a.B * a.C, cm = a.C - 1u;
        const uint32_t mine = b * a.C + (uint32_t)h * ((8u >> a.cshift) * BC);          // elements; < 2^30 (host check)
        const bool whole = (a.F & 15u) == 0u;                                            // then k < F is wave-uniform
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t kk = 16u * s + e;
                const float *row = a.feat + ((size_t)(kk >> a.cshift) * BC + (kk & cm));     // uniform
                const bool ok = whole ? kk < a.F : kk + 8u * h < a.F;
                fv[s][e] = ok ? row[mine] : 0.0f;
            }
        return;
    }
#pragma unroll
    for (int s = 0; s < KS; s++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t k = 16u * s + 8u * h + e, l = k / a.C, c = k - l * a.C;
            fv[s][e] = k < a.F ? a.feat[((size_t)l * a.B + b) * a.C + c] : 0.0f;
        }
}

// feature index b -> ray and -> position in the [N,S]-shaped outputs
__device__ __forceinline__ uint32_t ray_index(const MlpArgs &a, uint32_t b) {
    return a.rays_fastest ? b % a.n_rays : b / a.spr;
}
__device__ __forceinline__ uint32_t out_index(const MlpArgs &a, uint32_t b) {
    return a.rays_fastest ? (b % a.n_rays) * a.spr + b / a.n_rays : b;
}
__device__ __forceinline__ float softplus(float x) {
    return x > 20.0f ? x : log1pf(expf(x));        // F.softplus, beta = 1, threshold = 20
}

// rgb head NW -> 3 on the VALU (weights {w_r, w_g, w_b, 0} per accumulator slot), sigmoid, padding
// (models.py:657-674); h2 is the pre-ReLU accumulator of the last hidden layer.
template <int NTW>
__device__ __forceinline__ void rgb_head(const f32x16 (&h2)[NTW], const MlpArgs &a, uint32_t b, bool live, int h) {
    const float4 *pr = reinterpret_cast<const float4 *>(a.packed + a.phead);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float4 w = pr[(t * 16 + r) * 2 + h];
            const float v = fmaxf(h2[t][r], 0.0f);
            s0 = fmaf(v, w.x, s0);
            s1 = fmaf(v, w.y, s1);
            s2 = fmaf(v, w.z, s2);
        }
    s0 = xor32_sum(s0);
    s1 = xor32_sum(s1);
    s2 = xor32_sum(s2);
    if (live && h == 0) {
        const float pad = a.rgb_padding;
        float v[3] = {s0 + a.b_rgb[0], s1 + a.b_rgb[1], s2 + a.b_rgb[2]};
        const size_t o = (size_t)out_index(a, b) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float sg = 1.0f / (1.0f + expf(-(a.rgb_premult * v[c] + a.rgb_bias)));
            a.rgb[o + c] = sg * (1.0f + 2.0f * pad) - pad;
        }
    }
}

// implemented in field_mlp_h.hip
int ucn_h_pack(const ucn_field_t *f, const PackPlan &pl, hipStream_t st);
int ucn_h_dir_enc(const ucn_field_t *f, const float *viewdirs, uint32_t N, float *out, hipStream_t st);
int ucn_h_launch(const PackPlan &pl, const MlpArgs &a, dim3 grid, hipStream_t st);
