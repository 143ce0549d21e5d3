// Density + colour MLP of one field on the fp32 matrix cores (engine: mfma_chain.h).
//
// Replaces models.py:507-508 (density_layer), :581 (softplus), :599-674 (view-dependent colour
// MLP with skip, sigmoid, rgb padding) of /root/reference/nerf/internal/models.py.
//
// The two colour layers are interleaved per 32-neuron hidden tile (a tile of h1 is consumed into
// the second layer's accumulators as soon as it is complete), so only x (NB), the second layer's
// accumulators (NW) and one h1 tile are live: 272 accumulator registers at width 256 = one wave
// per SIMD on the 512-entry unified VGPR/AGPR file.  The 27 view-direction inputs are per RAY,
// not per sample: ucn_field_dir_bias folds them into a per-ray bias and only K = 256 / 512 go
// through the matrix cores.  Per sample: F*64 + 64*NB + NB*NW + (NW+NB)*NW MACs on MFMA, 3*NW on
// VALU (215,808 MAC at F=32, NB=NW=256).
//
// Weight stream order (ucn_field_pack writes it, k_field_mlp consumes it):
//   S1  density layer 1      [ot < NTB][it < 2  ][r4]          rows of W_d1
//   S2  colour layer 1, x    [ot < NTW][it < NTB][r4]          W_c1[:, NW:NW+NB]
//   S3  for each hidden tile t < NTW:
//         colour layer 0     [it < NTB][r4]                    rows 32t.. of W_c0[:, 0:NB]
//         colour layer 1, h1 [ot < NTW][r4]                    W_c1[:, 32t:32t+32]
// padded to a whole number of LDS chunks.
#include "mfma_chain_h.h"

namespace {

struct PackPlan {
    uint64_t p0, pstream, phead, total;               // float offsets into ucn_field_t::packed
    uint32_t F, KQ, NTB, NTW, n_groups;
    bool prop;
};

int make_plan(const ucn_field_t *f, PackPlan *pl) {
    UCN_REQUIRE(f, "field: null descriptor");
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
    } else {
        UCN_REQUIRE((f->n_bottleneck == 256 && f->n_width == 256) || (f->n_bottleneck == 64 && f->n_width == 64),
                    "field: supported (bottleneck_width, net_width_viewdirs) are (256,256) and (64,64), got (%u,%u)",
                    f->n_bottleneck, f->n_width);
        UCN_REQUIRE(f->w_c0 && f->w_c1 && f->w_rgb && f->b_c0 && f->b_c1 && f->b_rgb, "field: colour MLP weights missing");
        pl->NTB = f->n_bottleneck / 32;
        pl->NTW = f->n_width / 32;
        const uint32_t g = pl->NTB * 2 * 4 + pl->NTW * pl->NTB * 4 + pl->NTW * (pl->NTB * 4 + pl->NTW * 4);
        pl->n_groups = (g + kChunkGroups - 1) / kChunkGroups * kChunkGroups;
        pl->pstream = o; o += (uint64_t)pl->n_groups * 256;
        pl->phead = o; o += (uint64_t)pl->NTW * 128;
    }
    pl->total = o;
    return 0;
}

// first layer: dst[(ot*KQ + q)*64 + lane] = W[32ot + (lane&31)][2q + (lane>>5)]
__global__ __launch_bounds__(256) void k_pack_first(const float *__restrict__ W, uint32_t F, float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t KQ = F / 2;
    if (i >= 2u * KQ * 64u) return;
    const uint32_t lane = i & 63u, q = (i >> 6) % KQ, ot = (i >> 6) / KQ;
    dst[i] = W[(size_t)(32u * ot + (lane & 31u)) * F + 2u * q + (lane >> 5)];
}

// ---------------------------------------------------------------- per-ray direction bias
// enc = pos_enc(viewdir, 0, deg, append_identity)  (coord.py:214-225);  out[ray][0][n] =
// sum_k Wc0[n][NB+k] enc[k],  out[ray][1][n] = sum_k Wc1[n][NW+NB+k] enc[k].
constexpr int kMaxDir = 3 + 6 * 8;
template <int NW>
__global__ __launch_bounds__(256) void k_dir_bias(const float *__restrict__ viewdirs, const float *__restrict__ wc0,
                                                  const float *__restrict__ wc1, uint32_t nb, uint32_t ndir,
                                                  uint32_t N, float *__restrict__ out) {
    constexpr int RB = 16;                       // rays per workgroup
    constexpr int GROUPS = 256 / NW;             // ray sub-groups working in parallel
    __shared__ float enc[RB][kMaxDir + 1];
    const uint32_t ray0 = blockIdx.x * RB;
    const uint32_t deg = (ndir - 3u) / 6u;
    for (uint32_t i = threadIdx.x; i < RB * ndir; i += 256u) {
        const uint32_t rl = i / ndir, k = i - rl * ndir;
        const uint32_t ray = ray0 + rl < N ? ray0 + rl : N - 1;
        float v;
        if (k < 3u) {
            v = viewdirs[ray * 3 + k];
        } else {
            const uint32_t kk = (k - 3u) % (3u * deg), a = kk % 3u, si = kk / 3u;
            const float sc = viewdirs[ray * 3 + a] * (float)(1u << si);
            v = (k - 3u) < 3u * deg ? sinf(sc) : sinf(sc + 1.5707963705062866f);
        }
        enc[rl][k] = v;
    }
    __syncthreads();
    const uint32_t n = threadIdx.x % NW, grp = threadIdx.x / NW;
    const float *r0 = wc0 + (size_t)n * (nb + ndir) + nb;
    const float *r1 = wc1 + (size_t)n * (NW + nb + ndir) + NW + nb;
    for (uint32_t rl = grp; rl < RB; rl += GROUPS) {
        if (ray0 + rl >= N) break;
        float s0 = 0.0f, s1 = 0.0f;
        for (uint32_t k = 0; k < ndir; k++) {
            const float ev = enc[rl][k];
            s0 = fmaf(r0[k], ev, s0);
            s1 = fmaf(r1[k], ev, s1);
        }
        out[((size_t)(ray0 + rl) * 2 + 0) * NW + n] = s0;
        out[((size_t)(ray0 + rl) * 2 + 1) * NW + n] = s1;
    }
}

// ---------------------------------------------------------------- the MLP kernels
__device__ __forceinline__ float softplus(float x) {
    return x > 20.0f ? x : log1pf(expf(x));        // F.softplus, beta = 1, threshold = 20
}

struct MlpArgs {
    const float *feat;        // [L][B][C]
    const float *packed;
    const float *b_d0, *b_d1, *b_c0, *b_c1, *b_rgb;
    const float *dir_bias;    // [rays][2][NW]
    float *density, *rgb, *bott;
    uint32_t B, spr, C, F, n_chunks;
    uint64_t p0, pstream, phead;
    float density_bias, rgb_premult, rgb_bias, rgb_padding;
};

// density layer 0: F -> 64 with the B operand straight from the [L][B][C] feature buffer
__device__ __forceinline__ void first_layer(f32x16 (&h0)[2], const MlpArgs &a, uint32_t b, int lane, int h) {
    init_bias<2>(h0, a.b_d0, nullptr, h);
    const float *p0 = a.packed + a.p0;
    const uint32_t KQ = a.F / 2;
    for (uint32_t q = 0; q < KQ; q++) {
        const uint32_t k = 2u * q + h, l = k / a.C, c = k - l * a.C;
        const float bv = a.feat[((size_t)l * a.B + b) * a.C + c];
        const float a0 = p0[(size_t)q * 64 + lane], a1 = p0[((size_t)KQ + q) * 64 + lane];
        h0[0] = mfma32(a0, bv, h0[0]);
        h0[1] = mfma32(a1, bv, h0[1]);
    }
    relu_tile(h0[0]);
    relu_tile(h0[1]);
}

// PropMLP: F -> 64 -> 1 (models.py:438-441 with disable_rgb)
__global__ __launch_bounds__(256) void k_prop_mlp(MlpArgs a) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t b0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 32u;
    if (b0 >= a.B) return;                                   // wave-uniform, no barriers in this kernel
    const bool live = b0 + j < a.B;
    const uint32_t b = live ? b0 + j : a.B - 1;
    f32x16 h0[2];
    first_layer(h0, a, b, lane, h);
    // 64 -> 1 head on the VALU: this lane holds 32 of the 64 hidden units of its sample
    const float *pd = a.packed + a.phead;
    float part = 0.0f;
#pragma unroll
    for (int it = 0; it < 2; it++)
#pragma unroll
        for (int r = 0; r < 16; r++) part = fmaf(h0[it][r], pd[(it * 16 + r) * 2 + h], part);
    const float raw = (part + __shfl_xor(part, 32, 64)) + a.b_d1[0];
    if (live && h == 0) a.density[b] = softplus(raw + a.density_bias);
}

// Hidden tile T of the interleaved colour layers (compile-time recursion, see mfma_chain.h)
template <int T, int NTB, int NTW>
__device__ __forceinline__ void hidden_tiles(const int G3, const int per_tile, f32x16 (&h2)[NTW],
                                             const f32x16 (&x)[NTB], const float *__restrict__ b_c0,
                                             const float *__restrict__ db, int h, WeightStream &ws) {
    f32x16 h1;
    init_tile(h1, T, b_c0, db, h);
    chain_one<NTB>(G3 + T * per_tile, h1, x, ws);
    relu_tile(h1);
    chain_from_one<NTW>(G3 + T * per_tile + NTB * 4, h2, h1, ws);
    if constexpr (T + 1 < NTW) hidden_tiles<T + 1, NTB, NTW>(G3, per_tile, h2, x, b_c0, db, h, ws);
}

template <int NTB, int NTW>
__global__ __launch_bounds__(256) void k_field_mlp(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // 2 x 32 KiB weight chunks
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // Every wave of the workgroup runs the whole program (the weight stream's barriers are
    // workgroup-wide); waves past the end of the batch compute on a clamped sample and store nothing.
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 32u;
    const bool live = b0 + j < a.B;
    const uint32_t b = live ? b0 + j : a.B - 1;

    WeightStream ws{a.packed + a.pstream, s_w, lane, wave, a.n_chunks};
    ws.issue(0);

    f32x16 h0[2];
    first_layer(h0, a, b, lane, h);

    // stream positions (compile-time)
    constexpr int G1 = 0;                          // S1: density layer 1
    constexpr int G2 = G1 + NTB * 2 * 4;           // S2: colour layer 1, x part
    constexpr int G3 = G2 + NTW * NTB * 4;         // S3: per hidden tile
    constexpr int PER_TILE = NTB * 4 + NTW * 4;

    // ---- density layer 1: 64 -> NB, no activation; x[0] is the raw density (models.py:508)
    f32x16 x[NTB];
    init_bias<NTB>(x, a.b_d1, nullptr, h);
    chain<NTB, 2>(G1, x, h0, ws);
    if (live && h == 0) a.density[b] = softplus(x[0][0] + a.density_bias);
    if (a.bott && live) {
#pragma unroll
        for (int t = 0; t < NTB; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.bott[(size_t)b * (NTB * 32) + acc_row(t, r, h)] = x[t][r];
    }
    if (a.rgb == nullptr) {                       // density-only query: uniform over the grid
        __syncthreads();                          // retire the in-flight weight DMA before exit
        return;
    }
    const float *db = a.dir_bias + (size_t)(b / a.spr) * 2 * (NTW * 32);
    // ---- colour layer 1 accumulators, seeded with bias + direction term + the skip (x) part
    f32x16 h2[NTW];
    init_bias<NTW>(h2, a.b_c1, db + NTW * 32, h);
    chain<NTW, NTB>(G2, h2, x, ws);
    // ---- per hidden tile: colour layer 0 tile -> ReLU -> straight into colour layer 1
    hidden_tiles<0, NTB, NTW>(G3, PER_TILE, h2, x, a.b_c0, db, h, ws);
    // ---- rgb head NW -> 3 on the VALU, then sigmoid and padding (models.py:657-674)
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
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (live && h == 0) {
        const float pad = a.rgb_padding;
        float v[3] = {s0 + a.b_rgb[0], s1 + a.b_rgb[1], s2 + a.b_rgb[2]};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float sg = 1.0f / (1.0f + expf(-(a.rgb_premult * v[c] + a.rgb_bias)));
            a.rgb[(size_t)b * 3 + c] = sg * (1.0f + 2.0f * pad) - pad;
        }
    }
}

// ------------------------------------------------------------------ split-f16 variant (mlp_mode 1)
// Same program and weight-stream order as k_field_mlp; operands are (hi, lo) f16 pairs, see
// mfma_chain_h.h.  x is split tile by tile as it is produced, so the fp32 copy never exists in full.
template <int T, int NTB, int NTW>
__device__ __forceinline__ void hidden_tiles_h(const int G3, const int per_tile, f32x16 (&h2)[NTW],
                                               const HTile (&x)[NTB], const float *__restrict__ b_c0,
                                               const float *__restrict__ db, int h, WeightStream &ws) {
    f32x16 h1;
    init_tile(h1, T, b_c0, db, h);
    chain_one_h<NTB>(G3 + T * per_tile, h1, x, ws);
    relu_tile(h1);
    HTile h1s;
    split_tile(h1, h1s);
    chain_from_one_h<NTW>(G3 + T * per_tile + NTB * 4, h2, h1s, ws);
    if constexpr (T + 1 < NTW) hidden_tiles_h<T + 1, NTB, NTW>(G3, per_tile, h2, x, b_c0, db, h, ws);
}

template <int OT, int NTB>
__device__ __forceinline__ void density_tiles_h(const int G1, HTile (&x)[NTB], const HTile (&h0)[2], const MlpArgs &a,
                                                uint32_t b, bool live, int h, WeightStream &ws) {
    f32x16 acc;
    init_tile(acc, OT, a.b_d1, nullptr, h);
    chain_one_h<2>(G1 + OT * 2 * 4, acc, h0, ws);
    if (OT == 0 && live && h == 0) a.density[b] = softplus(acc[0] + a.density_bias);     // models.py:508,581
    if (a.bott && live) {
#pragma unroll
        for (int r = 0; r < 16; r++) a.bott[(size_t)b * (NTB * 32) + acc_row(OT, r, h)] = acc[r];
    }
    split_tile(acc, x[OT]);
    if constexpr (OT + 1 < NTB) density_tiles_h<OT + 1, NTB>(G1, x, h0, a, b, live, h, ws);
}

template <int NTB, int NTW>
__global__ __launch_bounds__(256) void k_field_mlp_h(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 32u;
    const bool live = b0 + j < a.B;
    const uint32_t b = live ? b0 + j : a.B - 1;

    WeightStream ws{a.packed + a.pstream, s_w, lane, wave, a.n_chunks};
    ws.issue(0);

    // ---- density layer 0: F -> 64.  B operand: features k = 16s + 8h + e, split on the fly.
    HTile h0[2];
    {
        f32x16 acc[2];
        init_bias<2>(acc, a.b_d0, nullptr, h);
        const h8 *p0 = reinterpret_cast<const h8 *>(a.packed + a.p0);
        const uint32_t KS = (a.F + 15) / 16;
        for (uint32_t s = 0; s < KS; s++) {
            h8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t k = 16u * s + 8u * h + e;
                float v = 0.0f;
                if (k < a.F) {
                    const uint32_t l = k / a.C, c = k - l * a.C;
                    v = a.feat[((size_t)l * a.B + b) * a.C + c];
                }
                const _Float16 hv = (_Float16)v;
                bh[e] = hv;
                bl[e] = (_Float16)(v - (float)hv);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ot++) {
                const h8 a_hi = p0[((size_t)(ot * KS + s) * 2 + 0) * 64 + lane];
                const h8 a_lo = p0[((size_t)(ot * KS + s) * 2 + 1) * 64 + lane];
                acc[ot] = mfma16h(a_hi, bh, acc[ot]);
                acc[ot] = mfma16h(a_hi, bl, acc[ot]);
                acc[ot] = mfma16h(a_lo, bh, acc[ot]);
            }
        }
        relu_tile(acc[0]);
        relu_tile(acc[1]);
        split_tile(acc[0], h0[0]);
        split_tile(acc[1], h0[1]);
    }
    constexpr int G1 = 0;
    constexpr int G2 = G1 + NTB * 2 * 4;
    constexpr int G3 = G2 + NTW * NTB * 4;
    constexpr int PER_TILE = NTB * 4 + NTW * 4;

    HTile x[NTB];
    density_tiles_h<0, NTB>(G1, x, h0, a, b, live, h, ws);
    if (a.rgb == nullptr) {
        __syncthreads();
        return;
    }
    const float *db = a.dir_bias + (size_t)(b / a.spr) * 2 * (NTW * 32);
    f32x16 h2[NTW];
    init_bias<NTW>(h2, a.b_c1, db + NTW * 32, h);
    chain_h<NTW, NTB>(G2, h2, x, ws);
    hidden_tiles_h<0, NTB, NTW>(G3, PER_TILE, h2, x, a.b_c0, db, h, ws);
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
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (live && h == 0) {
        const float pad = a.rgb_padding;
        float v[3] = {s0 + a.b_rgb[0], s1 + a.b_rgb[1], s2 + a.b_rgb[2]};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float sg = 1.0f / (1.0f + expf(-(a.rgb_premult * v[c] + a.rgb_bias)));
            a.rgb[(size_t)b * 3 + c] = sg * (1.0f + 2.0f * pad) - pad;
        }
    }
}

}  // namespace

extern "C" uint64_t ucn_field_packed_floats(const ucn_field_t *f) {
    PackPlan pl;
    if (make_plan(f, &pl)) return 0;
    return pl.total;
}

extern "C" int ucn_field_pack(const ucn_field_t *f, ucn_stream_t stream) {
    PackPlan pl;
    if (int rc = make_plan(f, &pl)) return rc;
    UCN_REQUIRE(f->packed && f->w_d0 && f->w_d1 && f->b_d0 && f->b_d1, "field_pack: density MLP weights / packed buffer missing");
    hipStream_t st = (hipStream_t)stream;
    UCN_REQUIRE(f->mlp_mode <= 1, "field: mlp_mode must be 0 (fp32 MFMA) or 1 (split-f16 MFMA)");
    if (!pl.prop && f->mlp_mode == 1) {
        // split-f16 layout: same stream order, every 1 KiB group holds 64 lanes x 8 halfs of one
        // (k-step, hi|lo) instead of 64 lanes x 4 floats of one r4
        const uint32_t NB = f->n_bottleneck, NW = f->n_width, ND = f->n_dir, KS = (pl.F + 15) / 16;
        hipLaunchKernelGGL(k_pack_first_h, dim3(ucn_div_up(2ull * KS * 2 * 512, 256)), dim3(256), 0, st, f->w_d0, pl.F, KS,
                           reinterpret_cast<_Float16 *>(f->packed + pl.p0));
        uint64_t off = pl.pstream;
        auto chainpack = [&](const float *W, uint32_t ld, uint32_t col0, uint32_t row_tile0, uint32_t nto, uint32_t nti) {
            hipLaunchKernelGGL(k_pack_chain_h, dim3(ucn_div_up((uint64_t)nto * nti * 2048, 256)), dim3(256), 0, st, W, ld,
                               col0, row_tile0, nto, nti, reinterpret_cast<_Float16 *>(f->packed + off));
            off += (uint64_t)nto * nti * 1024;
        };
        chainpack(f->w_d1, 64, 0, 0, pl.NTB, 2);
        chainpack(f->w_c1, NW + NB + ND, NW, 0, pl.NTW, pl.NTB);
        for (uint32_t t = 0; t < pl.NTW; t++) {
            chainpack(f->w_c0, NB + ND, 0, t, 1, pl.NTB);
            chainpack(f->w_c1, NW + NB + ND, 32 * t, 0, pl.NTW, 1);
        }
        const uint64_t end = pl.pstream + (uint64_t)pl.n_groups * 256;
        if (off < end)
            hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up(end - off, 256)), dim3(256), 0, st, f->packed + off, (uint32_t)(end - off));
        hipLaunchKernelGGL(k_pack_head, dim3(ucn_div_up((uint64_t)NW * 4, 256)), dim3(256), 0, st, f->w_rgb, NW, 0u, NW,
                           3u, 4u, f->packed + pl.phead);
        UCN_LAUNCH_CHECK("field_pack (split-f16)");
        return 0;
    }
    hipLaunchKernelGGL(k_pack_first, dim3(ucn_div_up(2ull * pl.KQ * 64, 256)), dim3(256), 0, st, f->w_d0, pl.F, f->packed + pl.p0);
    if (pl.prop) {
        hipLaunchKernelGGL(k_pack_head, dim3(1), dim3(256), 0, st, f->w_d1, 64u, 0u, 64u, 1u, 1u, f->packed + pl.phead);
    } else {
        const uint32_t NB = f->n_bottleneck, NW = f->n_width, ND = f->n_dir;
        uint64_t off = pl.pstream;
        auto chainpack = [&](const float *W, uint32_t ld, uint32_t col0, uint32_t row_tile0, uint32_t nto, uint32_t nti) {
            hipLaunchKernelGGL(k_pack_chain, dim3(ucn_div_up((uint64_t)nto * nti * 1024, 256)), dim3(256), 0, st, W, ld,
                               col0, row_tile0, nto, nti, f->packed + off);
            off += (uint64_t)nto * nti * 1024;
        };
        chainpack(f->w_d1, 64, 0, 0, pl.NTB, 2);                         // S1
        chainpack(f->w_c1, NW + NB + ND, NW, 0, pl.NTW, pl.NTB);         // S2
        for (uint32_t t = 0; t < pl.NTW; t++) {                          // S3
            chainpack(f->w_c0, NB + ND, 0, t, 1, pl.NTB);                //   rows of tile t, all of x
            chainpack(f->w_c1, NW + NB + ND, 32 * t, 0, pl.NTW, 1);      //   all rows, the tile's 32 columns
        }
        const uint64_t end = pl.pstream + (uint64_t)pl.n_groups * 256;
        if (off < end)
            hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up(end - off, 256)), dim3(256), 0, st, f->packed + off, (uint32_t)(end - off));
        hipLaunchKernelGGL(k_pack_head, dim3(ucn_div_up((uint64_t)NW * 4, 256)), dim3(256), 0, st, f->w_rgb, NW, 0u, NW,
                           3u, 4u, f->packed + pl.phead);
    }
    UCN_LAUNCH_CHECK("field_pack");
    return 0;
}

extern "C" int ucn_field_dir_bias(const ucn_field_t *f, const float *viewdirs, uint32_t N, float *dir_bias_out,
                                  ucn_stream_t stream) {
    PackPlan pl;
    if (int rc = make_plan(f, &pl)) return rc;
    UCN_REQUIRE(!pl.prop, "field_dir_bias: field has no colour MLP (disable_rgb)");
    UCN_REQUIRE(viewdirs && dir_bias_out, "field_dir_bias: null pointer argument");
    UCN_REQUIRE(f->n_dir >= 3 && (f->n_dir - 3) % 6 == 0 && f->n_dir <= (uint32_t)kMaxDir, "field_dir_bias: n_dir must be 3+6*deg, got %u", f->n_dir);
    if (N == 0) return 0;
    const dim3 grid(ucn_div_up(N, 16));
    hipStream_t st = (hipStream_t)stream;
    if (f->n_width == 256)
        hipLaunchKernelGGL(k_dir_bias<256>, grid, dim3(256), 0, st, viewdirs, f->w_c0, f->w_c1, f->n_bottleneck, f->n_dir, N, dir_bias_out);
    else
        hipLaunchKernelGGL(k_dir_bias<64>, grid, dim3(256), 0, st, viewdirs, f->w_c0, f->w_c1, f->n_bottleneck, f->n_dir, N, dir_bias_out);
    UCN_LAUNCH_CHECK("field_dir_bias");
    return 0;
}

extern "C" int ucn_field_mlp(const ucn_field_t *f, const float *features, uint32_t B, uint32_t samples_per_ray,
                             const float *dir_bias, float *density_out, float *rgb_out, float *bottleneck_out,
                             ucn_stream_t stream) {
    PackPlan pl;
    if (int rc = make_plan(f, &pl)) return rc;
    UCN_REQUIRE(features && density_out && f->packed, "field_mlp: null pointer argument");
    UCN_REQUIRE(samples_per_ray >= 1, "field_mlp: samples_per_ray must be >= 1");
    UCN_REQUIRE(pl.prop ? rgb_out == nullptr : true, "field_mlp: a disable_rgb field has no colour output");
    UCN_REQUIRE(rgb_out == nullptr || dir_bias != nullptr, "field_mlp: colour output needs the per-ray direction bias");
    if (B == 0) return 0;
    MlpArgs a;
    a.feat = features; a.packed = f->packed;
    a.b_d0 = f->b_d0; a.b_d1 = f->b_d1; a.b_c0 = f->b_c0; a.b_c1 = f->b_c1; a.b_rgb = f->b_rgb;
    a.dir_bias = dir_bias; a.density = density_out; a.rgb = rgb_out; a.bott = bottleneck_out;
    a.B = B; a.spr = samples_per_ray; a.C = f->level_dim; a.F = pl.F;
    a.n_chunks = pl.n_groups / kChunkGroups;
    a.p0 = pl.p0; a.pstream = pl.pstream; a.phead = pl.phead;
    a.density_bias = f->density_bias; a.rgb_premult = f->rgb_premultiplier; a.rgb_bias = f->rgb_bias;
    a.rgb_padding = f->rgb_padding;
    const dim3 grid(ucn_div_up(B, 128));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 2 * kChunkGroups * 256 * sizeof(float);
    UCN_REQUIRE(f->mlp_mode <= 1, "field: mlp_mode must be 0 (fp32 MFMA) or 1 (split-f16 MFMA)");
    if (pl.prop) hipLaunchKernelGGL(k_prop_mlp, grid, dim3(256), 0, st, a);
    else if (f->mlp_mode == 1 && pl.NTB == 8) hipLaunchKernelGGL((k_field_mlp_h<8, 8>), grid, dim3(256), lds, st, a);
    else if (f->mlp_mode == 1) hipLaunchKernelGGL((k_field_mlp_h<2, 2>), grid, dim3(256), lds, st, a);
    else if (pl.NTB == 8) hipLaunchKernelGGL((k_field_mlp<8, 8>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_field_mlp<2, 2>), grid, dim3(256), lds, st, a);
    UCN_LAUNCH_CHECK("field_mlp");
    return 0;
}
