// Density + colour MLP of one field on the matrix cores (engine: mfma_chain.h) -- entry points,
// the exact-fp32 kernel (mlp_mode 0) and the proposal-field kernel.
//
// Replaces models.py:507-508 (density_layer), :581 (softplus), :599-674 (view-dependent colour
// MLP with skip, sigmoid, rgb padding) of /root/reference/nerf/internal/models.py.
//
// The two colour layers are interleaved per 32-neuron hidden tile (a tile of h1 is consumed into
// the second layer's accumulators as soon as it is complete), so only x (NB), the second layer's
// accumulators (NW) and one h1 tile are live: 272 accumulator registers at width 256 = one wave
// per SIMD on the 512-entry unified VGPR/AGPR file.  Per sample: F*64 + 64*NB + NB*NW + (NW+NB)*NW
// MACs on MFMA, 3*NW on VALU (215,808 MAC at F=32, NB=NW=256).
//
// mlp_mode 0 (this file): v_mfma_f32_32x32x2_f32, exact fp32 products; the 27 view-direction inputs
//   are per RAY and are folded into a per-ray bias by ucn_field_dir_bias.  Measured 126 TFLOP/s =
//   0.80 of the fp32-MFMA peak (157 TF = the fp32 VECTOR rate on CDNA4).
// mlp_mode 1 (field_mlp_h.hip): split-f16 operands, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate.
//
// Weight stream order of mode 0 (ucn_field_pack writes it, k_field_mlp consumes it):
//   S1  density layer 1      [ot < NTB][it < 2  ][r4]          rows of W_d1
//   S2  colour layer 1, x    [ot < NTW][it < NTB][r4]          W_c1[:, NW:NW+NB]
//   S3  for each hidden tile t < NTW:
//         colour layer 0     [it < NTB][r4]                    rows 32t.. of W_c0[:, 0:NB]
//         colour layer 1, h1 [ot < NTW][r4]                    W_c1[:, 32t:32t+32]
// padded to a whole number of LDS chunks.
#include "field_plan.h"
#include "wave_dpp.h"

namespace {

// first layer: dst[(ot*KQ + q)*64 + lane] = W[32ot + (lane&31)][2q + (lane>>5)]
__global__ __launch_bounds__(256) void k_pack_first(const float *__restrict__ W, uint32_t F, float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t KQ = F / 2;
    if (i >= 2u * KQ * 64u) return;
    const uint32_t lane = i & 63u, q = (i >> 6) % KQ, ot = (i >> 6) / KQ;
    dst[i] = W[(size_t)(32u * ot + (lane & 31u)) * F + 2u * q + (lane >> 5)];
}

// ---------------------------------------------------------------- per-ray direction bias (mode 0)
// enc = pos_enc(viewdir, 0, deg, append_identity)  (coord.py:214-225);  out[ray][0][n] =
// sum_k Wc0[n][NB+k] enc[k],  out[ray][1][n] = sum_k Wc1[n][NW+NB+k] enc[k].
constexpr int kMaxDir = 27;
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

// ---------------------------------------------------------------- kernels
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
    const float raw = xor32_sum(part) + a.b_d1[0];
    if (live && h == 0) a.density[out_index(a, b)] = softplus(raw + a.density_bias);
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
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // 2 x 64 KiB weight chunks
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

    constexpr int G1 = 0;                          // S1: density layer 1
    constexpr int G2 = G1 + NTB * 2 * 4;           // S2: colour layer 1, x part
    constexpr int G3 = G2 + NTW * NTB * 4;         // S3: per hidden tile
    constexpr int PER_TILE = NTB * 4 + NTW * 4;

    // ---- density layer 1: 64 -> NB, no activation; x[0] is the raw density (models.py:508)
    f32x16 x[NTB];
    init_bias<NTB>(x, a.b_d1, nullptr, h);
    chain<NTB, 2>(G1, x, h0, ws);
    if (live && h == 0) a.density[out_index(a, b)] = softplus(x[0][0] + a.density_bias);
    if (a.bott && live) {
#pragma unroll
        for (int t = 0; t < NTB; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.bott[(size_t)out_index(a, b) * (NTB * 32) + acc_row(t, r, h)] = x[t][r];
    }
    if (a.rgb == nullptr) {                       // density-only query: uniform over the grid
        ws.drain();                               // retire the in-flight weight DMA before exit
        __syncthreads();
        return;
    }
    const float *db = a.dir_bias + (size_t)ray_index(a, b) * 2 * (NTW * 32);
    // ---- colour layer 1 accumulators, seeded with bias + direction term + the skip (x) part
    f32x16 h2[NTW];
    init_bias<NTW>(h2, a.b_c1, db + NTW * 32, h);
    chain<NTW, NTB>(G2, h2, x, ws);
    // ---- per hidden tile: colour layer 0 tile -> ReLU -> straight into colour layer 1
    hidden_tiles<0, NTB, NTW>(G3, PER_TILE, h2, x, a.b_c0, db, h, ws);
    rgb_head<NTW>(h2, a, b, live, h);
}

}  // namespace

extern "C" uint64_t ucn_field_packed_floats(const ucn_field_t *f) {
    PackPlan pl;
    if (make_plan(f, &pl)) return 0;
    return pl.total;
}

extern "C" uint64_t ucn_field_dir_floats(const ucn_field_t *f, uint32_t N) {
    PackPlan pl;
    if (make_plan(f, &pl) || pl.prop) return 0;
    return f->mlp_mode == 1 ? (uint64_t)N * 32 : (uint64_t)N * 2 * f->n_width;
}

extern "C" int ucn_field_pack(const ucn_field_t *f, ucn_stream_t stream) {
    PackPlan pl;
    if (int rc = make_plan(f, &pl)) return rc;
    UCN_REQUIRE(f->packed && f->w_d0 && f->w_d1 && f->b_d0 && f->b_d1, "field_pack: density MLP weights / packed buffer missing");
    hipStream_t st = (hipStream_t)stream;
    if (!pl.prop && f->mlp_mode == 1) return ucn_h_pack(f, pl, st);
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
    UCN_REQUIRE(N == 0 || (viewdirs && dir_bias_out), "field_dir_bias: null pointer argument");
    UCN_REQUIRE((f->n_dir - 3) % 6 == 0, "field_dir_bias: n_dir must be 3+6*deg, got %u", f->n_dir);
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (f->mlp_mode == 1) return ucn_h_dir_enc(f, viewdirs, N, dir_bias_out, st);
    const dim3 grid(ucn_div_up(N, 16));
    if (f->n_width == 256)
        hipLaunchKernelGGL(k_dir_bias<256>, grid, dim3(256), 0, st, viewdirs, f->w_c0, f->w_c1, f->n_bottleneck, f->n_dir, N, dir_bias_out);
    else
        hipLaunchKernelGGL(k_dir_bias<64>, grid, dim3(256), 0, st, viewdirs, f->w_c0, f->w_c1, f->n_bottleneck, f->n_dir, N, dir_bias_out);
    UCN_LAUNCH_CHECK("field_dir_bias");
    return 0;
}

// set (and cleared) by ucn_field_rgb_compacted around its call of ucn_field_mlp on the same host thread
static thread_local const uint32_t *g_mlp_idx = nullptr, *g_mlp_count = nullptr;

extern "C" int ucn_field_mlp(const ucn_field_t *f, const float *features, uint32_t B, uint32_t samples_per_ray,
                             int rays_fastest, const float *dir_bias, float *density_out, float *rgb_out,
                             float *bottleneck_out, ucn_stream_t stream) {
    PackPlan pl;
    if (int rc = make_plan(f, &pl)) return rc;
    UCN_REQUIRE(B == 0 || (features && density_out && f->packed), "field_mlp: null pointer argument");
    UCN_REQUIRE(samples_per_ray >= 1, "field_mlp: samples_per_ray must be >= 1");
    UCN_REQUIRE(pl.prop ? rgb_out == nullptr : true, "field_mlp: a disable_rgb field has no colour output");
    UCN_REQUIRE(B == 0 || rgb_out == nullptr || dir_bias != nullptr, "field_mlp: colour output needs the per-ray direction terms");
    if (B == 0) return 0;
    MlpArgs a;
    a.feat = features; a.packed = f->packed;
    a.b_d0 = f->b_d0; a.b_d1 = f->b_d1; a.b_c0 = f->b_c0; a.b_c1 = f->b_c1; a.b_rgb = f->b_rgb;
    a.dir_bias = dir_bias; a.density = density_out; a.rgb = rgb_out; a.bott = bottleneck_out;
    UCN_REQUIRE(!(rays_fastest & 1) || B % samples_per_ray == 0, "field_mlp: B must be rays x samples_per_ray");
    a.B = B; a.spr = samples_per_ray; a.C = f->level_dim; a.F = pl.F;
    a.cshift = 0xFFu;
    if ((a.C == 1 || a.C == 2 || a.C == 4 || a.C == 8) && (uint64_t)B * a.C * (pl.F / a.C + 2) < (1ull << 30))
        a.cshift = a.C == 1 ? 0u : a.C == 2 ? 1u : a.C == 4 ? 2u : 3u;
    a.n_rays = B / samples_per_ray; a.rays_fastest = (rays_fastest & 1) ? 1u : 0u;
    a.small_ring = (rays_fastest & UCN_LAUNCH_CORESIDENT) ? 1u : 0u;
    a.idx = g_mlp_idx; a.count = g_mlp_count;
    a.n_chunks = pl.n_groups / kChunkGroups;
    a.p0 = pl.p0; a.pstream = pl.pstream; a.phead = pl.phead;
    a.density_bias = f->density_bias; a.rgb_premult = f->rgb_premultiplier; a.rgb_bias = f->rgb_bias;
    a.rgb_padding = f->rgb_padding;
    const dim3 grid(ucn_div_up(B, 128));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 2 * kChunkGroups * 256 * sizeof(float);
    if (pl.prop) hipLaunchKernelGGL(k_prop_mlp, grid, dim3(256), 0, st, a);
    else if (f->mlp_mode == 1) return ucn_h_launch(pl, a, grid, st);
    else if (pl.NTB == 8) hipLaunchKernelGGL((k_field_mlp<8, 8>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_field_mlp<2, 2>), grid, dim3(256), lds, st, a);
    UCN_LAUNCH_CHECK("field_mlp");
    return 0;
}

extern "C" int ucn_field_rgb_compacted(const ucn_field_t *f, const float *features, uint32_t B, uint32_t samples_per_ray,
                                       int rays_fastest, const float *dir_bias, const uint32_t *idx, const uint32_t *count,
                                       float *rgb_out, ucn_stream_t stream) {
    UCN_REQUIRE(f && f->mlp_mode == 1 && f->n_bottleneck != 1, "field_rgb_compacted: needs a colour field in mlp_mode 1");
    UCN_REQUIRE(B == 0 || (idx && count && rgb_out), "field_rgb_compacted: null pointer argument");
    g_mlp_idx = idx; g_mlp_count = count;
    // density_out is not written in this mode; any non-null pointer satisfies the common argument checks
    const int rc = ucn_field_mlp(f, features, B, samples_per_ray, rays_fastest, dir_bias, rgb_out, rgb_out, nullptr, stream);
    g_mlp_idx = g_mlp_count = nullptr;
    return rc;
}
