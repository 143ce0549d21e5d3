// Density + colour MLP of one field, split-f16 MFMA variant (mlp_mode 1; engine: mfma_chain_h.h).
// Same network and interleaving as field_mlp.hip (models.py:507-508, :581, :599-674 of
// /root/reference/nerf/internal/models.py); differences that matter for speed:
//   * 3 x v_mfma_f32_32x32x16_f16 per 16 k's instead of 8 x v_mfma_f32_32x32x2_f32 (5.3x the rate);
//   * the 27 view-direction inputs are one more 32-wide INPUT tile of both colour layers
//     (k < 27: pos_enc(viewdir), k = 27: the constant 1 whose weight column is the layer's bias,
//     k > 27: 0), produced per ray by k_dir_enc -- 128 B per ray instead of 2 KiB of folded biases;
//   * all weights AND biases come through the LDS stream; the only global loads of a wave are its
//     feature values and its direction tile, issued before the first barrier;
//   * the whole network is ONE pipelined segment of the stream: A operands are requested kDepth steps
//     ahead, across layer boundaries.
//
// Stream (1 KiB groups; a "pair" is [hi][lo]):
//   B   groups 0-1: bias tiles [b_d0: 2 tiles][b_d1: NTB tiles]
//   S0  density layer 0   pairs [s < kFirstSteps][o2]
//   S1  density layer 1   pairs [otp < NTB/2][it < 2][s][o2]                           rows of W_d1
//   S2  colour layer 1, skip part: pairs [otp < NTW/2][it < NTB+1][s][o2]   W_c1[:, NW : NW+NB+27 | b_c1]
//   S3  per PAIR of hidden tiles (t, t+1):
//         colour layer 0   pairs [it < NTB+1][s][o2]                   rows 32t..32t+63 of W_c0 | b_c0
//         colour layer 1   pairs [otp < NTW/2][it < 2][s][o2]          W_c1[:, 32t : 32t+64]
//   S4  rgb head {w_r, w_g, w_b, 0} per accumulator slot (4 groups)
// Output tiles always go in pairs (o2 innermost): consecutive MFMAs alternate between two accumulators.
#include "field_plan.h"
#include "mfma_chain_h.h"

namespace {

constexpr int kKS = (int)kFirstSteps;

// per-ray direction tile: out[ray][k] = pos_enc(viewdir, 0, deg, append_identity)[k] (coord.py:214-225)
// for k < ndir, 1 for k == ndir, 0 beyond
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
    out[i] = v;
}

template <int T, int NTB, int NTW>
__device__ __forceinline__ void hidden_tiles_h(const int G3, const int GEND, f32x16 (&h2)[NTW], const HTile (&xin)[NTB + 1],
                                               APipe &p, WeightStream &ws) {
    // hidden tiles T, T+1 together (two independent accumulators, see dstep_h)
    constexpr int PER_PAIR = (NTB + 1) * 8 + NTW * 8;
    const int base = G3 + (T / 2) * PER_PAIR;
    f32x16 h1[2];
#pragma unroll
    for (int r = 0; r < 16; r++) h1[0][r] = h1[1][r] = 0.0f;   // b_c0 rides in the direction tile's slot 27
    chain_h<2, NTB + 1>(base, GEND, h1, xin, p, ws);
    relu_tile(h1[0]);
    relu_tile(h1[1]);
    HTile t[2];
    split_tile(h1[0], t[0]);
    split_tile(h1[1], t[1]);
    chain_h<NTW, 2>(base + (NTB + 1) * 8, GEND, h2, t, p, ws);
    if constexpr (T + 2 < NTW) hidden_tiles_h<T + 2, NTB, NTW>(G3, GEND, h2, xin, p, ws);
}

template <int NTB, int NTW>
__global__ __launch_bounds__(256) void k_field_mlp_h(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // 2 x 64 KiB weight chunks
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // every wave runs the whole program (workgroup-wide barriers); waves past the end of the batch
    // compute on a clamped sample and store nothing
    const uint32_t b0 = (blockIdx.x * 4u + wave) * 32u;
    const bool live = b0 + j < a.B;
    const uint32_t b = live ? b0 + j : a.B - 1;
    const uint32_t oi = out_index(a, b);                 // position in the [ray][sample]-ordered outputs

    constexpr int G0 = 2;                                // after the bias tiles
    constexpr int G1 = G0 + 2 * kKS * 2;
    constexpr int G2 = G1 + NTB * 2 * 4;
    constexpr int G3 = G2 + NTW * (NTB + 1) * 4;
    constexpr int GH = G3 + NTW * ((NTB + 1) * 4 + NTW * 4);   // end of the pipelined segment, rgb head

    // ---- the wave's only global loads: its feature values and the ray's direction tile
    float fv[kKS][8];
#pragma unroll
    for (int s = 0; s < kKS; s++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t k = 16u * s + 8u * h + e, l = k / a.C, c = k - l * a.C;
            fv[s][e] = k < a.F ? a.feat[((size_t)l * a.B + b) * a.C + c] : 0.0f;
        }
    f32x16 ev;
    if (a.rgb) {
        const float4 *ep = reinterpret_cast<const float4 *>(a.dir_bias + (size_t)ray_index(a, b) * 32 + 4 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = ep[2 * r4];
            ev[4 * r4 + 0] = v.x; ev[4 * r4 + 1] = v.y; ev[4 * r4 + 2] = v.z; ev[4 * r4 + 3] = v.w;
        }
    }
    WeightStream ws{a.packed + a.pstream, s_w, lane, wave, a.n_chunks};
    ws.issue(0);
    ws.sync();
    if (kChunkGroups < GH + kTailGroups) ws.piece_unchecked(1, 0);   // the rest of chunk 1 rides on pipe_fetch
    static_assert(GH % kChunkGroups == 0 || GH % kChunkGroups + 4 <= kChunkGroups, "rgb head straddles a chunk");

    // ---- accumulators of both density layers start from their biases (chunk 0 is resident now)
    f32x16 acc0[2], x[NTB];
    bias_tile_h(0, acc0[0], h, ws);
    bias_tile_h(1, acc0[1], h, ws);
#pragma unroll
    for (int t = 0; t < NTB; t++) bias_tile_h(2 + t, x[t], h, ws);
    APipe p;
    pipe_prime(G0, GH, p, ws);

    // ---- density layer 0: F -> 64, ReLU
    HTile h0[2];
    {
        h8 fhi[kKS], flo[kKS];
#pragma unroll
        for (int s = 0; s < kKS; s++) split8(fv[s], fhi[s], flo[s]);
#pragma unroll
        for (int s = 0; s < kKS; s++) dstep_h(G0 + 4 * s, GH, acc0[0], acc0[1], fhi[s], flo[s], p, ws);
        relu_tile(acc0[0]);
        relu_tile(acc0[1]);
        split_tile(acc0[0], h0[0]);
        split_tile(acc0[1], h0[1]);
    }

    // ---- density layer 1: 64 -> NB, no activation; x[0] is the raw density (models.py:508,581)
    chain_h<NTB, 2>(G1, GH, x, h0, p, ws);
    if (live && h == 0) a.density[oi] = softplus(x[0][0] + a.density_bias);
    if (a.bott && live) {
#pragma unroll
        for (int t = 0; t < NTB; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) a.bott[(size_t)oi * (NTB * 32) + acc_row(t, r, h)] = x[t][r];
    }
    if (a.rgb == nullptr) {                       // density-only query: uniform over the grid
        ws.drain();                               // retire the in-flight weight DMA before exit
        __syncthreads();
        return;
    }
    HTile xin[NTB + 1];
#pragma unroll
    for (int t = 0; t < NTB; t++) split_tile(x[t], xin[t]);
    split_tile(ev, xin[NTB]);

    // ---- colour layers
    f32x16 h2[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) h2[t][r] = 0.0f;     // b_c1 rides in the direction tile's slot 27
    chain_h<NTW, NTB + 1>(G2, GH, h2, xin, p, ws);
    hidden_tiles_h<0, NTB, NTW>(G3, GH, h2, xin, p, ws);

    // ---- rgb head NW -> 3 on the VALU, weights broadcast from the stream's tail (models.py:657-674)
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int g = GH + t / 2;                            // 64 float4 slots per group = 2 tiles
        if (t % 2 == 0 && g % kChunkGroups == 0) ws.sync();   // (its DMA rode on the previous chunk's requests)
        const float4 *pr = reinterpret_cast<const float4 *>(ws.group_ptr(g)) + (t % 2) * 32 + h;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float4 w = pr[2 * r];
            const float v = fmaxf(h2[t][r], 0.0f);
            s0 = fmaf(v, w.x, s0);
            s1 = fmaf(v, w.y, s1);
            s2 = fmaf(v, w.z, s2);
        }
    }
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
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
    float *base = f->packed + pl.pstream;
    const uint64_t n_floats = (uint64_t)pl.n_groups * 256;
    hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up(n_floats, 256)), dim3(256), 0, st, base, (uint32_t)n_floats);
    uint64_t g = 0;                                                     // position in groups
    auto chainpack = [&](const float *W, uint32_t ld, uint32_t col0, uint32_t row_tile0, uint32_t nto, uint32_t nti,
                         const float *bias) {
        hipLaunchKernelGGL(k_pack_chain_h, dim3(ucn_div_up((uint64_t)nto * nti * 2048, 256)), dim3(256), 0, st, W, ld,
                           col0, row_tile0, nto, nti, bias, reinterpret_cast<_Float16 *>(base + g * 256));
        g += (uint64_t)nto * nti * 4;
    };
    // bias tiles
    hipLaunchKernelGGL(k_pack_bias_h, dim3(1), dim3(256), 0, st, f->b_d0, 2u, base);
    hipLaunchKernelGGL(k_pack_bias_h, dim3(ucn_div_up(pl.NTB * 32, 256)), dim3(256), 0, st, f->b_d1, pl.NTB, base + 64);
    g += 2;
    // S0
    hipLaunchKernelGGL(k_pack_first_h, dim3(ucn_div_up(2ull * kFirstSteps * 1024, 256)), dim3(256), 0, st, f->w_d0, pl.F,
                       kFirstSteps, reinterpret_cast<_Float16 *>(base + g * 256));
    g += 2 * kFirstSteps * 2;
    // S1
    chainpack(f->w_d1, 64, 0, 0, pl.NTB, 2, nullptr);
    // S2: x columns, then the direction columns + bias column (col == ld) + zeros
    chainpack(f->w_c1, NW + NB + ND, NW, 0, pl.NTW, pl.NTB + 1, f->b_c1);
    // S3
    for (uint32_t t = 0; t < pl.NTW; t += 2) {
        chainpack(f->w_c0, NB + ND, 0, t, 2, pl.NTB + 1, f->b_c0);
        chainpack(f->w_c1, NW + NB + ND, 32 * t, 0, pl.NTW, 2, nullptr);
    }
    // S4
    hipLaunchKernelGGL(k_pack_head, dim3(ucn_div_up((uint64_t)NW * 4, 256)), dim3(256), 0, st, f->w_rgb, NW, 0u, NW, 3u,
                       4u, base + g * 256);
    g += 4;
    UCN_REQUIRE(g == stream_groups_h(pl.NTB, pl.NTW), "field_pack: internal stream length mismatch (%llu)", (unsigned long long)g);
    UCN_LAUNCH_CHECK("field_pack (split-f16)");
    return 0;
}

int ucn_h_dir_enc(const ucn_field_t *f, const float *viewdirs, uint32_t N, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_dir_enc, dim3(ucn_div_up((uint64_t)N * 32, 256)), dim3(256), 0, st, viewdirs, f->n_dir, N, out);
    UCN_LAUNCH_CHECK("field_dir_bias (split-f16)");
    return 0;
}

int ucn_h_launch(const PackPlan &pl, const MlpArgs &a, dim3 grid, hipStream_t st) {
    const size_t lds = 2 * kChunkGroups * 256 * sizeof(float);
    if (pl.NTB == 8) hipLaunchKernelGGL((k_field_mlp_h<8, 8>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_field_mlp_h<2, 2>), grid, dim3(256), lds, st, a);
    UCN_LAUNCH_CHECK("field_mlp (split-f16)");
    return 0;
}
