// Training step of the UC-NeRF sky layer: forward with the activations the backward
// needs, the dgrad chain, and the compositing forward / backward, as hand-written kernels.
//
// Replaces, for a training step under bf16 autocast (train.py:165-171; scripts/train_waymo.sh:11 turns model_sky on),
// the eager evaluation of models.py:326-337 (call), :852-904 (render_rays), :743-820 (NeRF: 8 x 256, skip at 4, views
// branch) and :822-850 (raw2outputs) of /root/reference/nerf/internal/models.py and autograd's way back through them:
// 563 k MAC per sample x 120 samples per ray x 3 (forward, dgrad, wgrad) -- the largest FLOP term of the reference's
// shipped training step.  As ~80 library launches with 0.5 GB activations between each it took 80 ms of a 91 ms step.
//
//   k_sky_train_fwd   the rendering kernel's bf16 pair-chain sequence (sky.hip k_sky_mlp_bf) in the training kernels'
//                     shape (4 waves, 64 KiB LDS-DMA ring + 14 KiB side table, two workgroups per CU); a wave keeps its
//                     32 samples in registers through all ten layers; every hidden activation is written ONCE, as the
//                     bf16 B operand the next layer consumed, into one [M, 2240] buffer h_0 .. h_7 | aux | hv, where
//                     aux (32) = (p, 1, embed(cam_dir), 0) holds the rest of the reference's concatenated layer inputs
//                     plus the constant-1 column: every layer's weight + bias gradient is then ONE pass of
//                     ucn_wgrad_bf16 (wgrad.hip) over [h_{l-1} | aux].  ReLU masks as 16 bits per tile.
//   k_sky_composite_bwd   d rgb_map / d (colour logits, sigma) per sample (suffix-sum form of the transmittance gradient)
//   k_sky_train_bwd   the chain backwards on transposed fragments: dv = (W_rgb^T g) m_v, d7 = ([Mv_h | w_alpha]^T [dv | g]) m_7,
//                     d_l = (W_{l+1}^T d_{l+1}) m_l; pre-activation gradients stored once, bf16, [M, 2240].
// feature_linear (no activation behind it) stays composed into the views layer as in rendering: the host forms
// Mv = W_view[:, :256] W_feat with differentiable torch ops, so autograd carries d Mv back to both factors -- the
// `feature` activation is neither computed nor stored.  Weight gradients: csrc/wgrad.hip over the two buffers.
#include "pack_split.h"
#include "sky_layout.h"

namespace {

constexpr int kTrFrags = 984;                                           // fragments of either stream
constexpr int kTrPadded = (kTrFrags + kTChunk - 1) / kTChunk * kTChunk; // 992
// ring geometry per kernel (slots x 16 KiB, chunks requested LEAD ahead): the field kernels' 4 x 16 KiB / 2 ahead is cut for two
// workgroups per CU; UCN_SKY_{FWD,BWD}_{SLOTS,LEAD} are build knobs (tools/build_variant.sh)
#ifndef UCN_SKY_FWD_SLOTS
#define UCN_SKY_FWD_SLOTS 6
#endif
#ifndef UCN_SKY_FWD_LEAD
#define UCN_SKY_FWD_LEAD 4
#endif
#ifndef UCN_SKY_BWD_OCC
#define UCN_SKY_BWD_OCC 1
#endif
#ifndef UCN_SKY_BWD_SLOTS
#define UCN_SKY_BWD_SLOTS (UCN_SKY_BWD_OCC == 1 ? 6 : kTSlots)
#endif
#ifndef UCN_SKY_BWD_LEAD
#define UCN_SKY_BWD_LEAD (UCN_SKY_BWD_OCC == 1 ? 4 : kTLead)
#endif
constexpr int kFwdSlots = UCN_SKY_FWD_SLOTS, kBwdSlots = UCN_SKY_BWD_SLOTS;
static_assert(UCN_SKY_FWD_SLOTS >= UCN_SKY_FWD_LEAD + 2 && UCN_SKY_BWD_SLOTS >= UCN_SKY_BWD_LEAD + 2, "a chunk is refilled two boundaries after its last reader");
// UCN_SKY_{FWD,BWD}_STAGE: 0 = the weight stream reaches LDS by LDS-DMA (global_load_lds), n > 0 = by plain 16-byte loads into n staging
// registers per lane and ds_write_b128 n pieces later (mlp_ring.h)
#ifndef UCN_SKY_FWD_STAGE
#define UCN_SKY_FWD_STAGE 0
#endif
#ifndef UCN_SKY_BWD_STAGE
#define UCN_SKY_BWD_STAGE 0
#endif
#ifndef UCN_SKY_FWD_WAVES
#define UCN_SKY_FWD_WAVES 4
#endif
constexpr int kFwdWaves = UCN_SKY_FWD_WAVES;
using STRing = Ring<kTrPadded, kTChunk, UCN_SKY_FWD_WAVES, UCN_SKY_FWD_SLOTS, UCN_SKY_FWD_LEAD, UCN_SKY_FWD_STAGE>;
// waves per workgroup of the backward kernel: 8 waves share ONE weight ring, so the LDS-DMA stream (the whole 1 MB of fragments per
// pass; ~25 GB/s per CU is what the DMA path lands) is paid once per 256 samples instead of once per 128
#ifndef UCN_SKY_BWD_WAVES
#define UCN_SKY_BWD_WAVES 4
#endif
constexpr int kBwdWaves = UCN_SKY_BWD_WAVES;
using SBRing = Ring<kTrPadded, kTChunk, UCN_SKY_BWD_WAVES, UCN_SKY_BWD_SLOTS, UCN_SKY_BWD_LEAD, UCN_SKY_BWD_STAGE>;
// forward stream positions (A-fragments [otp][it][s][o2]): pts_linears 1..4, 5 (9 input tiles), 6, 7, views (9 tiles)
constexpr int kFL[7] = {0, 128, 256, 384, 512, 656, 784};
constexpr int kFV = 912;
// backward stream: W_rgb^T (2 pairs x 1 tile) | [Mv_h | w_alpha]^T (4 pairs x 5 tiles) | W7^T, W6^T, M5_h^T, W4^T .. W1^T
constexpr int kGV = 0, kG7 = 8, kGL = 88;
// activation buffer (bf16 [M, kActLd]): h_0 .. h_7 (256 each) | aux (32) | hv (128) | pad
constexpr int kActBlock = 256, kActAux = 8 * kActBlock, kActHv = kActAux + 32, kActLd = 2240;    // rows of 35 x 128 bytes
// gradient buffer (bf16 [M, kDlLd]): d0 .. d7 (256 each) | dv (128) | g (32: d logits r, g, b, d sigma, 0...) | pad
constexpr int kDlV = 2048, kDlG = kDlV + 128, kDlLd = 2240;                              // 35 x 128 bytes
// packed buffer (bytes): forward stream | backward stream | side table | scratch matrices of the two composite stages
constexpr uint64_t kPkBwd = (uint64_t)kTrPadded * 1024, kPkSide = 2 * kPkBwd, kPkB7 = kPkSide + kSideFloats * 4;
constexpr uint64_t kPkBv = kPkB7 + 256 * 160 * 4, kPkBytes = kPkBv + 128 * 32 * 4;

struct SkyTrainArgs {
    const uint8_t *packed;
    const float *aux;            // [N,32] per ray: [0,0,0,1, embed(cam_dir) (27), 0]
    const float *origins, *dirs, *far_, *t_vals;
    uint32_t N;
    float *raw;                  // [N*120, 4] = colour logits (3), sigma
    uint16_t *act;               // [M, kActLd] bf16
    uint4 *mask;                 // [8][M][2]: ReLU masks of h0 .. h7, 16 bits per tile, 8 tiles
    uint2 *mask_v;               // [M][2]: of hv (4 tiles)
};

// sky_far = 1.5 * near[0] with near = batch.far (models.py:328-330), read on the device: no host sync in the step
__device__ __forceinline__ float inv_sky_far_of(const float *far_) { return 1.0f / (1.5f * far_[0]); }

constexpr int kFwdFrags = kFV + 72;              // fragments the forward chain consumes
// UCN_SKY_FWD_PIPE: fragments requested kWAhead ahead of their MFMAs through a register pipe (bf_tiles.h tile_pair_pf)
#ifndef UCN_SKY_FWD_PIPE
#define UCN_SKY_FWD_PIPE 1
#endif
template <int P>
__device__ __forceinline__ bf8 (&pick9(bf8 (&a)[9][2], bf8 (&b)[9][2]))[9][2] {
    if constexpr (P == 0) return a;
    else return b;
}

// UCN_SKY_FWD_OCC: workgroups per CU the register budget of the forward kernel is cut for (build knob, tools/build_variant.sh).
// r05: ONE.  At two per CU (256 registers) the kernel spilled 190 registers to scratch, and every scratch reload made the compiler put
// `s_waitcnt vmcnt(0)` in front of it (96 inside the chain) -- each draining the weight stream's look-ahead: 2.39 ms.  One workgroup
// per CU with the whole register file, a 6-slot ring requested 4 chunks ahead, the pairs stored through LDS as whole lines
// (store_pair_staged): 1.79 ms.  Then the two changes that mattered (profiles/r05/sky_train_variants.txt):
//   * with ONE wave per SIMD every instruction is an issue slot nobody else fills, and the chain was "ds_read the fragment, wait for
//     it, MFMA" with the same four registers for every fragment -- the LDS latency in front of each of the 984 MFMAs.  The fragment
//     pipe (bf_tiles.h tile_pair_pf: three requests ahead, order pinned per step): 1.56 -> 1.27 ms;
//   * hipcc's default keeps MFMA accumulators in AGPRs, which only MFMAs can touch: each value the epilogue converts / masks costs a
//     v_accvgpr_read first -- 1984 of 11317 instructions per wave.  -mllvm -amdgpu-mfma-vgpr-form (build.sh): 1.80 -> 1.60 ms.
// Measured and NOT kept: an 8-slot ring 6 ahead (1.83 before the pipe), a per-layer [M, 256] activation layout (no change: DRAM page
// locality is not it), 8-wave workgroups sharing one ring (1.50 against 1.29 with the pipe), two workgroups per CU on a 3-slot ring
// (1.92), the register-staged weight stream instead of LDS-DMA (2.8), chunk waits that count the younger stores (nothing once the
// pipe was in, and unsafe: mlp_ring.h boundary), two sample tiles per wave (k_sky_train_fwd2 below: 1.92).  Without its activation
// stores the kernel takes 1.16 ms: 9.5 k instructions per wave and 32 samples against 984 MFMAs -- on this part VALU / scalar issue
// adds to MFMA time (DESIGN.md "Execution model of a gfx950 SIMD"), so what is left is instruction count.
#ifndef UCN_SKY_FWD_OCC
#define UCN_SKY_FWD_OCC 1
#endif
#ifndef UCN_SKY_FWD_STAGED
#define UCN_SKY_FWD_STAGED (UCN_SKY_FWD_OCC == 1)
#endif
__global__ __launch_bounds__(64 * UCN_SKY_FWD_WAVES, UCN_SKY_FWD_WAVES == 8 ? 1 : UCN_SKY_FWD_OCC) void k_sky_train_fwd(SkyTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // weight ring + side table (14 KiB) [+ 4 staging tiles]
    const float *side = s_w + kFwdSlots * kTChunk * 256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t M = a.N * (uint32_t)kSkySamples;
    const uint32_t b0 = (blockIdx.x * (uint32_t)kFwdWaves + wave) * 32u;
    constexpr bool kStaged = UCN_SKY_FWD_STAGED;                   // the activation pairs leave through LDS (bf_tiles.h)
    uint8_t *stage = reinterpret_cast<uint8_t *>(s_w + kFwdSlots * kTChunk * 256 + kSideFloats) + wave * kStageTile;
    const uint32_t n_rows = b0 < M ? (M - b0 < 32u ? M - b0 : 32u) : 0u;
    const bool live = b0 + j < M;
    const uint32_t b = live ? b0 + j : M - 1;
    const uint32_t ray = b / kSkySamples, s = b - ray * kSkySamples;

    const float tv = a.t_vals[s];
    const float z = a.far_[ray] * (1.0f - tv) + inv_sky_far_of(a.far_) * tv;          // models.py:872
    const float px = a.origins[ray * 3 + 0] + a.dirs[ray * 3 + 0] * z;
    const float py = a.origins[ray * 3 + 1] + a.dirs[ray * 3 + 1] * z;
    const float pz = a.origins[ray * 3 + 2] + a.dirs[ray * 3 + 2] * z;

    bf8 XA[9][2], XB[9][2];            // two activation buffers of 8 tiles + the auxiliary tile in slot 8 of both
    {
        f32x16 av;
        const float4 *ap = reinterpret_cast<const float4 *>(a.aux + (size_t)ray * 32 + 4 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = ap[2 * r4];
            av[4 * r4 + 0] = v.x; av[4 * r4 + 1] = v.y; av[4 * r4 + 2] = v.z; av[4 * r4 + 3] = v.w;
        }
        if (h == 0) { av[0] = px; av[1] = py; av[2] = pz; }
        XA[8][0] = XB[8][0] = to_b(av, 0, false);
        XA[8][1] = XB[8][1] = to_b(av, 1, false);
    }
    uint16_t *row = a.act;
    // the auxiliary tile, once per sample: the skip connection's point, the view encoding and the constant 1 are the second
    // column block of every layer's weight-gradient GEMM (ucn_wgrad_bf16 takes B as two blocks)
    store_tile(row + kActAux, kActLd, b, 0, h, XA[8], live);

    STRing ring(reinterpret_cast<const float *>(a.packed), s_w, lane, wave);
    {   // side table: 14 pieces of 1 KiB, DMA'd once, ahead of the ring's chunks (vmcnt completes in order)
        const uint32_t lside = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)s_w + (uint32_t)(kFwdSlots * kTChunk) * 1024u;
        const float *gside = reinterpret_cast<const float *>(a.packed + kPkSide);
#pragma unroll
        for (int k = 0; k < (kSideFloats / 256 + kFwdWaves - 1) / kFwdWaves; k++) {
            const int piece = k * kFwdWaves + wave;
            if (piece < kSideFloats / 256)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :
                             : "s"(lside + piece * 1024u), "v"(lane * 16u), "s"(gside + piece * 256)
                             : "memory");
        }
    }
    ring_start(ring);
    ring.template boundary<0>();                    // side table + chunk 0 landed

    // ---- layer 0 (3 -> 256), fp32 on the VALU, rounded into XA
    {
        const float4 *p0 = reinterpret_cast<const float4 *>(side + kSL0) + h;
        uint32_t mk[4];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float4 w = p0[(t * 16 + r) * 2];
                acc[r] = fmaf(w.z, pz, fmaf(w.y, py, fmaf(w.x, px, w.w)));
            }
            XA[t][0] = to_b(acc, 0, true);
            XA[t][1] = to_b(acc, 1, true);
            if (t % 2 == 0) mk[t / 2] = mask16(acc);
            else mk[t / 2] |= mask16(acc) << 16;
            if (t % 2 == 1) {
#ifdef UCN_EXP_COMPACT_ACT
                if constexpr (kStaged) store_pair_staged(stage, row, 256, b0, n_rows, t - 1, lane, XA[t - 1], XA[t]);
#else
                if constexpr (kStaged) store_pair_staged(stage, row, kActLd, b0, n_rows, t - 1, lane, XA[t - 1], XA[t]);
#endif
                else store_two<true>(row, kActLd, b, t - 1, h, XA[t - 1], XA[t], live);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) a.mask[((size_t)0 * M + b) * 2 + h] = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    }

    constexpr bool kPipe = UCN_SKY_FWD_PIPE;
    bf8 wp[kWSlots];                                                    // fragment pipe (tile_pair_pf)
    if constexpr (kPipe) sfor<kWAhead>([&](auto g) { frag_fetch<g.value, kFwdFrags>(ring, wp); });
    float sig = 0.0f;                                             // alpha head partial (this lane's 128 neurons)
    const float *pa = side + kSAlpha + h;
    sfor<7>([&](auto lic) {
        constexpr int li = lic.value, NT_IN = li == 4 ? 9 : 8;
        bf8 (&in)[9][2] = pick9<li % 2>(XA, XB);
        bf8 (&out)[9][2] = pick9<(li + 1) % 2>(XA, XB);
        uint32_t mk[4];
        sfor<4>([&](auto pc) {
            constexpr int pr = pc.value;
            f32x16 cur[2];
            if constexpr (kBiasIdx[li] >= 0) {
                side_bias_tile(side, kSB + kBiasIdx[li] * 256, 2 * pr, cur[0], h);
                side_bias_tile(side, kSB + kBiasIdx[li] * 256, 2 * pr + 1, cur[1], h);
            } else {
                zero_acc(cur[0]);
                zero_acc(cur[1]);
            }
            if constexpr (kPipe) tile_pair_pf<2, NT_IN, kFL[li] + pr * NT_IN * 4, kFwdFrags>(ring, wp, cur, reinterpret_cast<const bf8(&)[NT_IN][2]>(in));
            else tile_pair<2, NT_IN, kFL[li] + pr * NT_IN * 4>(ring, cur, reinterpret_cast<const bf8(&)[NT_IN][2]>(in));
            if constexpr (li == 6) {                              // alpha head on the fp32 ReLU output of layer 7
                alpha_partial<2 * pr, 0>(cur[0], pa, sig);
                alpha_partial<2 * pr, 1>(cur[0], pa, sig);
                alpha_partial<2 * pr + 1, 0>(cur[1], pa, sig);
                alpha_partial<2 * pr + 1, 1>(cur[1], pa, sig);
            }
#pragma unroll
            for (int o = 0; o < 2; o++) {
                out[2 * pr + o][0] = to_b(cur[o], 0, true);
                out[2 * pr + o][1] = to_b(cur[o], 1, true);
            }
            mk[pr] = mask16(cur[0]) | (mask16(cur[1]) << 16);
            asm volatile("" : "+v"(mk[pr]));                      // computed HERE: left alone it may sink into the `live` block at the layer's end
#ifdef UCN_EXP_COMPACT_ACT   // timing-only experiment: every layer its own [M, 256] matrix (the host still reads the interleaved layout: results garbage)
            if constexpr (kStaged) store_pair_staged(stage, row + (size_t)(li + 1) * M * 256, 256, b0, n_rows, 2 * pr, lane, out[2 * pr], out[2 * pr + 1]);
#else
            if constexpr (kStaged) store_pair_staged(stage, row + (li + 1) * kActBlock, kActLd, b0, n_rows, 2 * pr, lane, out[2 * pr], out[2 * pr + 1]);
#endif
            else store_two<true>(row + (li + 1) * kActBlock, kActLd, b, 2 * pr, h, out[2 * pr], out[2 * pr + 1], live);
            // a pair's epilogue (conversion, mask, store) is finished before the next pair's chain starts: left to itself the
            // scheduler carries accumulators and store operands across (588 bytes of scratch per lane at the 256-register cap)
            __builtin_amdgcn_sched_barrier(0);
        });
        if (live) a.mask[((size_t)(li + 1) * M + b) * 2 + h] = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    });
    // ---- views layer: [h7 (8 tiles) | aux] -> 128, 2 pair chains; h7 = buffer 1 (7 layers), then the rgb head per pair
    bf8 (&h7)[9][2] = pick9<1>(XA, XB);
    sig = xor32_sum(sig) + side[kSAlpha + 256];
    const float4 *prgb = reinterpret_cast<const float4 *>(side + kSRgb) + h;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    uint32_t mkv[2];
    sfor<2>([&](auto pc) {
        constexpr int pr = pc.value;
        f32x16 v[2];
        zero_acc(v[0]);
        zero_acc(v[1]);
        if constexpr (kPipe) tile_pair_pf<2, 9, kFV + pr * 36, kFwdFrags>(ring, wp, v, h7);
        else tile_pair<2, 9, kFV + pr * 36>(ring, v, h7);
        bf8 hv[2][2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            hv[o][0] = to_b(v[o], 0, true);
            hv[o][1] = to_b(v[o], 1, true);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float4 w = prgb[((2 * pr + o) * 16 + r) * 2];
                const float x = fmaxf(v[o][r], 0.0f);
                c0 = fmaf(x, w.x, c0); c1 = fmaf(x, w.y, c1); c2 = fmaf(x, w.z, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        mkv[pr] = mask16(v[0]) | (mask16(v[1]) << 16);
        if constexpr (kStaged) store_pair_staged(stage, row + kActHv, kActLd, b0, n_rows, 2 * pr, lane, hv[0], hv[1]);
        else store_two<true>(row + kActHv, kActLd, b, 2 * pr, h, hv[0], hv[1], live);
    });
    if (live) a.mask_v[(size_t)b * 2 + h] = make_uint2(mkv[0], mkv[1]);
    c0 = xor32_sum(c0); c1 = xor32_sum(c1); c2 = xor32_sum(c2);
    const float *brgb = side + kSRgb + 512;
    if (live && h == 0)
        *reinterpret_cast<float4 *>(a.raw + (size_t)b * 4) = make_float4(c0 + brgb[0], c1 + brgb[1], c2 + brgb[2], sig);
}

// ---- the forward kernel with TWO sample tiles per wave (UCN_SKY_FWD_TILES = 2; r05) ------------------------------------------------
// Same chain, same stream, same outputs; a wave carries 64 samples and every weight fragment feeds two MFMAs (bf_tiles.h tile_pair2):
// the 1 MB weight stream is paid once per 256 samples of a workgroup.  Register plan per lane (one workgroup per CU, 512 registers):
//   AGPRs: activation buffer A of both tiles (128) + the per-ray tiles (16) + a pair's accumulators (64)
//   VGPRs: activation buffer B of both tiles (128) + the fragment pipe (16) + epilogue temporaries
// Layer li reads A and writes B (li even) or reads B and writes A (li odd; the VALU's results are moved over, 128 v_accvgpr_write).
#ifndef UCN_SKY_FWD_TILES
#define UCN_SKY_FWD_TILES 1
#endif
template <int P>
__device__ __forceinline__ bf8 (&pick8(bf8 (&a)[8][2], bf8 (&b)[8][2]))[8][2] {
    if constexpr (P == 0) return a;
    else return b;
}
__global__ __launch_bounds__(256, 1) void k_sky_train_fwd2(SkyTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // weight ring + side table + 4 staging tiles
    const float *side = s_w + kFwdSlots * kTChunk * 256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t M = a.N * (uint32_t)kSkySamples;
    uint8_t *stage = reinterpret_cast<uint8_t *>(s_w + kFwdSlots * kTChunk * 256 + kSideFloats) + wave * kStageTile;
    uint32_t b0[2], n_rows[2], b[2];
    bool live[2];
    float px[2], py[2], pz[2];
    bf8 XA[2][8][2], XB[2][8][2], AUX[2][2];      // XA, AUX: AGPRs (only ever operands of the class-annotated MFMA); XB: VGPRs
    uint16_t *row = a.act;
#pragma unroll
    for (int st = 0; st < 2; st++) {
        b0[st] = ((blockIdx.x * 4u + wave) * 2u + st) * 32u;
        n_rows[st] = b0[st] < M ? (M - b0[st] < 32u ? M - b0[st] : 32u) : 0u;
        live[st] = b0[st] + j < M;
        b[st] = live[st] ? b0[st] + j : M - 1;
        const uint32_t ray = b[st] / kSkySamples, s = b[st] - ray * kSkySamples;
        const float tv = a.t_vals[s];
        const float z = a.far_[ray] * (1.0f - tv) + inv_sky_far_of(a.far_) * tv;          // models.py:872
        px[st] = a.origins[ray * 3 + 0] + a.dirs[ray * 3 + 0] * z;
        py[st] = a.origins[ray * 3 + 1] + a.dirs[ray * 3 + 1] * z;
        pz[st] = a.origins[ray * 3 + 2] + a.dirs[ray * 3 + 2] * z;
        f32x16 av;
        const float4 *ap = reinterpret_cast<const float4 *>(a.aux + (size_t)ray * 32 + 4 * h);
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 v = ap[2 * r4];
            av[4 * r4 + 0] = v.x; av[4 * r4 + 1] = v.y; av[4 * r4 + 2] = v.z; av[4 * r4 + 3] = v.w;
        }
        if (h == 0) { av[0] = px[st]; av[1] = py[st]; av[2] = pz[st]; }
        bf8 t[2] = {to_b(av, 0, false), to_b(av, 1, false)};
        store_tile(row + kActAux, kActLd, b[st], 0, h, t, live[st]);
        AUX[st][0] = to_agpr(t[0]);
        AUX[st][1] = to_agpr(t[1]);
    }
    STRing ring(reinterpret_cast<const float *>(a.packed), s_w, lane, wave);
    {   // side table: 14 pieces of 1 KiB, DMA'd once, ahead of the ring's chunks (vmcnt completes in order)
        const uint32_t lside = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)s_w + (uint32_t)(kFwdSlots * kTChunk) * 1024u;
        const float *gside = reinterpret_cast<const float *>(a.packed + kPkSide);
#pragma unroll
        for (int k = 0; k < (kSideFloats / 256 + 3) / 4; k++) {
            const int piece = k * 4 + wave;
            if (piece < kSideFloats / 256)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :
                             : "s"(lside + piece * 1024u), "v"(lane * 16u), "s"(gside + piece * 256)
                             : "memory");
        }
    }
    ring_start(ring);
    ring.template boundary<0>();                    // side table + chunk 0 landed

    // ---- layer 0 (3 -> 256), fp32 on the VALU, rounded into XA
    {
        const float4 *p0 = reinterpret_cast<const float4 *>(side + kSL0) + h;
        uint32_t mk[2][4];
        sfor<4>([&](auto pc) {
            constexpr int pr = pc.value;
            sfor<2>([&](auto stc) {
                constexpr int st = stc.value;
                bf8 o[2][2];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int t = 2 * pr + q;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float4 w = p0[(t * 16 + r) * 2];
                        acc[r] = fmaf(w.z, pz[st], fmaf(w.y, py[st], fmaf(w.x, px[st], w.w)));
                    }
                    o[q][0] = to_b(acc, 0, true);
                    o[q][1] = to_b(acc, 1, true);
                    if (q == 0) mk[st][pr] = mask16(acc);
                    else mk[st][pr] |= mask16(acc) << 16;
                }
                asm volatile("" : "+v"(mk[st][pr]));              // computed HERE (left alone it sinks into the `live` block at the layer's end,
                //                                                   the pair's 32 accumulator values with it: 536 bytes of scratch)
                store_pair_staged(stage, row, kActLd, b0[st], n_rows[st], 2 * pr, lane, o[0], o[1]);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    XA[st][2 * pr + q][0] = to_agpr(o[q][0]);
                    XA[st][2 * pr + q][1] = to_agpr(o[q][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int st = 0; st < 2; st++)
            if (live[st]) a.mask[((size_t)0 * M + b[st]) * 2 + h] = make_uint4(mk[st][0], mk[st][1], mk[st][2], mk[st][3]);
    }

    bf8 wp[kWSlots];
    sfor<kWAhead>([&](auto g) { frag_fetch<g.value, kFwdFrags>(ring, wp); });
    float sig[2] = {0.0f, 0.0f};                                  // alpha head partials (this lane's 128 neurons)
    const float *pa = side + kSAlpha + h;
    sfor<7>([&](auto lic) {
        constexpr int li = lic.value, NT_IN = li == 4 ? 9 : 8;
        constexpr bool INA = li % 2 == 0;                         // reads XA, writes XB
        bf8 (&in0)[8][2] = pick8<li % 2>(XA[0], XB[0]);
        bf8 (&in1)[8][2] = pick8<li % 2>(XA[1], XB[1]);
        bf8 (&out0)[8][2] = pick8<(li + 1) % 2>(XA[0], XB[0]);
        bf8 (&out1)[8][2] = pick8<(li + 1) % 2>(XA[1], XB[1]);
        uint32_t mk[2][4];
        sfor<4>([&](auto pc) {
            constexpr int pr = pc.value;
            f32x16 c0[2], c1[2];
            constexpr bool ZERO = kBiasIdx[li] < 0;
            if constexpr (!ZERO) {
                side_bias_tile(side, kSB + kBiasIdx[li] * 256, 2 * pr, c0[0], h);
                side_bias_tile(side, kSB + kBiasIdx[li] * 256, 2 * pr + 1, c0[1], h);
#pragma unroll
                for (int r = 0; r < 16; r++) { c1[0][r] = c0[0][r]; c1[1][r] = c0[1][r]; }
                pair_ready(c0, c1);
            } else {
                asm volatile("s_nop 7");
            }
            tile_pair2<NT_IN, kFL[li] + pr * NT_IN * 4, kFwdFrags, INA, ZERO>(ring, wp, c0, c1, in0, in1, AUX[0], AUX[1]);
            pair_settle(c0, c1);
            sfor<2>([&](auto stc) {
                constexpr int st = stc.value;
                f32x16 (&c)[2] = *(st == 0 ? &c0 : &c1);
                if constexpr (li == 6) {                          // alpha head on the fp32 ReLU output of layer 7
                    alpha_partial<2 * pr, 0>(c[0], pa, sig[st]); alpha_partial<2 * pr, 1>(c[0], pa, sig[st]);
                    alpha_partial<2 * pr + 1, 0>(c[1], pa, sig[st]); alpha_partial<2 * pr + 1, 1>(c[1], pa, sig[st]);
                }
                bf8 o[2][2];
#pragma unroll
                for (int q = 0; q < 2; q++) { o[q][0] = to_b(c[q], 0, true); o[q][1] = to_b(c[q], 1, true); }
                mk[st][pr] = mask16(c[0]) | (mask16(c[1]) << 16);
                asm volatile("" : "+v"(mk[st][pr]));
                store_pair_staged(stage, row + (li + 1) * kActBlock, kActLd, b0[st], n_rows[st], 2 * pr, lane, o[0], o[1]);
                bf8 (&out)[8][2] = *(st == 0 ? &out0 : &out1);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    out[2 * pr + q][0] = INA ? o[q][0] : to_agpr(o[q][0]);
                    out[2 * pr + q][1] = INA ? o[q][1] : to_agpr(o[q][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int st = 0; st < 2; st++)
            if (live[st]) a.mask[((size_t)(li + 1) * M + b[st]) * 2 + h] = make_uint4(mk[st][0], mk[st][1], mk[st][2], mk[st][3]);
    });
    // ---- views layer: [h7 (8 tiles) | aux] -> 128, 2 pair chains; h7 = XB (7 layers), then the rgb head per pair
    sig[0] = xor32_sum(sig[0]) + side[kSAlpha + 256];
    sig[1] = xor32_sum(sig[1]) + side[kSAlpha + 256];
    const float4 *prgb = reinterpret_cast<const float4 *>(side + kSRgb) + h;
    float cc[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    uint32_t mkv[2][2];
    sfor<2>([&](auto pc) {
        constexpr int pr = pc.value;
        f32x16 v0[2], v1[2];
        asm volatile("s_nop 7");
        tile_pair2<9, kFV + pr * 36, kFwdFrags, false, true>(ring, wp, v0, v1, XB[0], XB[1], AUX[0], AUX[1]);
        pair_settle(v0, v1);
        sfor<2>([&](auto stc) {
            constexpr int st = stc.value;
            f32x16 (&v)[2] = *(st == 0 ? &v0 : &v1);
            bf8 hv[2][2];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                hv[o][0] = to_b(v[o], 0, true);
                hv[o][1] = to_b(v[o], 1, true);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float4 w = prgb[((2 * pr + o) * 16 + r) * 2];
                    const float x = fmaxf(v[o][r], 0.0f);
                    cc[st][0] = fmaf(x, w.x, cc[st][0]); cc[st][1] = fmaf(x, w.y, cc[st][1]); cc[st][2] = fmaf(x, w.z, cc[st][2]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mkv[st][pr] = mask16(v[0]) | (mask16(v[1]) << 16);
            asm volatile("" : "+v"(mkv[st][pr]));
            store_pair_staged(stage, row + kActHv, kActLd, b0[st], n_rows[st], 2 * pr, lane, hv[0], hv[1]);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    const float *brgb = side + kSRgb + 512;
#pragma unroll
    for (int st = 0; st < 2; st++) {
        if (live[st]) a.mask_v[(size_t)b[st] * 2 + h] = make_uint2(mkv[st][0], mkv[st][1]);
        const float r0 = xor32_sum(cc[st][0]), r1 = xor32_sum(cc[st][1]), r2 = xor32_sum(cc[st][2]);
        if (live[st] && h == 0)
            *reinterpret_cast<float4 *>(a.raw + (size_t)b[st] * 4) = make_float4(r0 + brgb[0], r1 + brgb[1], r2 + brgb[2], sig[st]);
    }
}

struct SkyTrainBwdArgs {
    const uint8_t *packed;
    const float *graw;           // [M, 4] fp32: d loss / d (colour logits, sigma)
    const uint4 *mask;
    const uint2 *mask_v;
    uint16_t *dl;                // [M, kDlLd] bf16
    uint32_t M;
};


// UCN_SKY_BWD_OCC: workgroups per CU the backward kernel's registers are cut for.  r05: ONE, as the forward kernel and for the same
// reasons -- the whole register file (every layer's ReLU masks preloaded: no compiler-visible load, hence no compiler-placed vmcnt,
// inside the weight stream), the fragment pipe (tile_pair_pf), a 6-slot ring 4 chunks ahead, pairs stored as whole lines through LDS
// and counted at the chunk waits.
#ifndef UCN_SKY_BWD_OCC
#define UCN_SKY_BWD_OCC 1
#endif
#ifndef UCN_SKY_BWD_PIPE
#define UCN_SKY_BWD_PIPE (UCN_SKY_BWD_OCC == 1)
#endif
#ifndef UCN_SKY_BWD_STAGED
#define UCN_SKY_BWD_STAGED (UCN_SKY_BWD_OCC == 1)
#endif
__global__ __launch_bounds__(64 * UCN_SKY_BWD_WAVES, UCN_SKY_BWD_WAVES == 8 ? 1 : UCN_SKY_BWD_OCC) void k_sky_train_bwd(SkyTrainBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // weight ring [+ one staging tile per wave]
    constexpr bool kPipe = UCN_SKY_BWD_PIPE, kStaged = UCN_SKY_BWD_STAGED, kPreMask = UCN_SKY_BWD_OCC == 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t b0 = (blockIdx.x * (uint32_t)kBwdWaves + wave) * 32u;
    const uint32_t n_rows = b0 < a.M ? (a.M - b0 < 32u ? a.M - b0 : 32u) : 0u;
    const uint32_t s0 = b0 + j;
    const bool live = s0 < a.M;
    const uint32_t b = live ? s0 : a.M - 1;
    uint8_t *stage = reinterpret_cast<uint8_t *>(s_w + kBwdSlots * kTChunk * 256) + wave * kStageTile;
    uint4 mall[8];                                                // kPreMask: every layer's ReLU masks, before the stream starts
    if constexpr (kPreMask) {
#pragma unroll
        for (int l = 0; l < 8; l++) mall[l] = a.mask[((size_t)l * a.M + b) * 2 + h];
    }
    const uint2 mv = a.mask_v[(size_t)b * 2 + h];
    uint4 mcur = kPreMask ? mall[7] : a.mask[((size_t)7 * a.M + b) * 2 + h];
    float4 ghead = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (h == 0) ghead = *reinterpret_cast<const float4 *>(a.graw + (size_t)b * 4);
    if constexpr (kPreMask) {                                     // the loads above are complete before the first DMA piece is counted
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SBRing ring(reinterpret_cast<const float *>(a.packed + kPkBwd), s_w, lane, wave);
    ring_start(ring);
    // ---- the head gradients as one input tile: columns 0..3 = d logits (r, g, b), d sigma -- registers 0..3 of wave half 0
    //      in accumulator order and in natural order alike
    bf8 dv[5][2];                      // tiles 0..3: dv (filled below), 4: the head-gradient tile
    {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) { v[0] = ghead.x; v[1] = ghead.y; v[2] = ghead.z; v[3] = ghead.w; }
        dv[4][0] = pack8(v);
        const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        dv[4][1] = pack8(z);
        store_tile(a.dl + kDlG, kDlLd, b, 0, h, dv[4], live);
    }
    ring.template boundary<0>();
    bf8 wp[kWSlots];                                                    // fragment pipe (tile_pair_pf)
    if constexpr (kPipe) sfor<kWAhead>([&](auto g) { frag_fetch<g.value, kTrFrags>(ring, wp); });
    // ---- through the rgb layer and the views layer's ReLU
    sfor<2>([&](auto pc) {
        constexpr int pr = pc.value;
        f32x16 acc[2];
        zero_acc(acc[0]);
        zero_acc(acc[1]);
        if constexpr (kPipe) tile_pair_pf<2, 1, kGV + 4 * pr, kTrFrags>(ring, wp, acc, reinterpret_cast<const bf8(&)[1][2]>(dv[4]));
        else tile_pair<2, 1, kGV + 4 * pr>(ring, acc, reinterpret_cast<const bf8(&)[1][2]>(dv[4]));
        const uint32_t mw = pr == 0 ? mv.x : mv.y;
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t bits = (mw >> (16 * o)) & 0xFFFFu;
            dv[2 * pr + o][0] = to_b_masked(acc[o], 0, bits);
            dv[2 * pr + o][1] = to_b_masked(acc[o], 1, bits);
        }
        if constexpr (kStaged) store_pair_staged(stage, a.dl + kDlV, kDlLd, b0, n_rows, 2 * pr, lane, dv[2 * pr], dv[2 * pr + 1]);
        else store_two<true>(a.dl + kDlV, kDlLd, b, 2 * pr, h, dv[2 * pr], dv[2 * pr + 1], live);
        __builtin_amdgcn_sched_barrier(0);
    });
    // ---- through the (composed) views layer and the alpha head into h7, then down the trunk.  Stage q = 0..7 produces
    //      d_{7-q}: q = 0 reads [dv | g] (5 tiles), the others the previous stage's 8 tiles; DA / DB ping-pong.
    bf8 DA[8][2], DB[8][2];
    sfor<8>([&](auto qc) {
        constexpr int q = qc.value, l = 7 - q;
        bf8 (&out)[8][2] = pick8<q % 2>(DA, DB);
        const uint4 mk = kPreMask ? mall[l] : mcur;
        if constexpr (l > 0 && !kPreMask) mcur = a.mask[((size_t)(l - 1) * a.M + b) * 2 + h];   // the next stage's masks, one stage ahead
        const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
        sfor<4>([&](auto pc) {
            constexpr int pr = pc.value;
            f32x16 acc[2];
            zero_acc(acc[0]);
            zero_acc(acc[1]);
            if constexpr (kPipe) {
                if constexpr (q == 0) tile_pair_pf<2, 5, kG7 + 20 * pr, kTrFrags>(ring, wp, acc, dv);
                else tile_pair_pf<2, 8, kGL + 128 * (q - 1) + 32 * pr, kTrFrags>(ring, wp, acc, pick8<(q + 1) % 2>(DA, DB));
            } else {
                if constexpr (q == 0) tile_pair<2, 5, kG7 + 20 * pr>(ring, acc, dv);
                else tile_pair<2, 8, kGL + 128 * (q - 1) + 32 * pr>(ring, acc, pick8<(q + 1) % 2>(DA, DB));
            }
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const uint32_t bits = (mw[pr] >> (16 * o)) & 0xFFFFu;
                out[2 * pr + o][0] = to_b_masked(acc[o], 0, bits);
                out[2 * pr + o][1] = to_b_masked(acc[o], 1, bits);
            }
            if constexpr (kStaged) store_pair_staged(stage, a.dl + l * 256, kDlLd, b0, n_rows, 2 * pr, lane, out[2 * pr], out[2 * pr + 1]);
            else store_two<true>(a.dl + l * 256, kDlLd, b, 2 * pr, h, out[2 * pr], out[2 * pr + 1], live);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// raw2outputs backwards (models.py:822-850): rgb_map = sum_s w_s sigmoid(y_s), w_s = alpha_s T_s, T_{s+1} = T_s (1 - alpha_s + 1e-10),
// alpha_s = 1 - exp(-relu(sigma_s) dist_s).  One thread per ray, two passes over its 120 samples:
//   d / d y_s      = g . (w_s c_s (1 - c_s))                        per channel
//   d / d alpha_s  = g . (T_s c_s) - (sum_{k > s} g . (w_k c_k)) / (1 - alpha_s + 1e-10)
//   d / d sigma_s  = d / d alpha_s * dist_s exp(-sigma_s dist_s)    for sigma_s > 0, else 0
__global__ __launch_bounds__(64) void k_sky_composite_bwd(const float *__restrict__ raw, const float *__restrict__ dirs,
                                                          const float *__restrict__ far_, const float *__restrict__ t_vals,
                                                          const float *__restrict__ g_out, uint32_t N, float *__restrict__ g_raw) {
    const uint32_t ray = blockIdx.x * 64u + threadIdx.x;
    if (ray >= N) return;
    const float inv_sky_far = inv_sky_far_of(far_);
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float nr = far_[ray];
    const float g0 = g_out[ray * 3 + 0], g1 = g_out[ray * 3 + 1], g2 = g_out[ray * 3 + 2];
    float total = 0.0f;
    for (int pass = 0; pass < 2; pass++) {
        float T = 1.0f, prefix = 0.0f;
        float z = nr * (1.0f - t_vals[0]) + inv_sky_far * t_vals[0];
        for (int s = 0; s < kSkySamples; s++) {
            float dist, zn = z;
            if (s + 1 < kSkySamples) {
                zn = nr * (1.0f - t_vals[s + 1]) + inv_sky_far * t_vals[s + 1];
                dist = zn - z;
            } else {
                dist = 1e10f;
            }
            dist = dist * dn;
            const size_t o = ((size_t)ray * kSkySamples + s) * 4;
            const float4 v = *reinterpret_cast<const float4 *>(raw + o);
            const float sg = fmaxf(v.w, 0.0f);
            const float e = expf(-sg * dist);
            const float alpha = 1.0f - e;
            const float w = alpha * T;
            const float cr = 1.0f / (1.0f + expf(-v.x)), cg = 1.0f / (1.0f + expf(-v.y)), cb = 1.0f / (1.0f + expf(-v.z));
            const float gc = (g0 * cr + g1 * cg) + g2 * cb;
            prefix += w * gc;
            if (pass == 1) {
                const float keep = (1.0f - alpha) + 1e-10f;
                const float dalpha = T * gc - (total - prefix) / keep;
                const float dsig = v.w > 0.0f ? dalpha * dist * e : 0.0f;
                *reinterpret_cast<float4 *>(g_raw + o) =
                    make_float4(g0 * w * cr * (1.0f - cr), g1 * w * cg * (1.0f - cg), g2 * w * cb * (1.0f - cb), dsig);
            }
            T = T * ((1.0f - alpha) + 1e-10f);
            z = zn;
        }
        total = prefix;
    }
}

// The same gradients with one WAVE per ray (r06; the default): lane l holds samples 2 l and 2 l + 1 (coalesced 32-byte pieces of the
// ray's 1920 bytes), the transmittance is an exclusive prefix PRODUCT and the suffix sum an inclusive prefix sum across the wave (DPP
// scans, wave_dpp.h).  The one-thread-per-ray form above is 128 waves of 240 dependent iterations each for a batch of 8192 rays: beside
// the field's kernels on the other stream it sat on the sky stream's critical path for 3 ms (profiles/r06/train_top_heads.txt).  A scan
// multiplies / adds in tree order: the results differ from the serial form's in the last bits (the forward pass, whose pixel values
// are compared with the rendering kernel's, keeps the serial order).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or1(float v) {          // the DPP-selected lane's value, 1 where there is none
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_scan_product(float v) {
    v *= dpp_or1<0x111, 0xf>(v);
    v *= dpp_or1<0x112, 0xf>(v);
    v *= dpp_or1<0x114, 0xf>(v);
    v *= dpp_or1<0x118, 0xf>(v);
    v *= dpp_or1<0x142, 0xa>(v);
    v *= dpp_or1<0x143, 0xc>(v);
    return v;
}
__global__ __launch_bounds__(256) void k_sky_composite_bwd_wave(const float *__restrict__ raw, const float *__restrict__ dirs,
                                                                const float *__restrict__ far_, const float *__restrict__ t_vals,
                                                                const float *__restrict__ g_out, uint32_t N, float *__restrict__ g_raw) {
    static_assert(kSkySamples <= 128 && kSkySamples % 2 == 0, "two samples per lane");
    const uint32_t ray = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (ray >= N) return;                                           // (wave-uniform)
    const float inv_sky_far = inv_sky_far_of(far_);
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float nr = far_[ray];
    const float g0 = g_out[ray * 3 + 0], g1 = g_out[ray * 3 + 1], g2 = g_out[ray * 3 + 2];
    const bool live = 2u * lane < (uint32_t)kSkySamples;
    const uint32_t s0 = live ? 2u * lane : 0u;
    float zv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const uint32_t s = s0 + q < (uint32_t)kSkySamples ? s0 + q : (uint32_t)kSkySamples - 1u;
        zv[q] = nr * (1.0f - t_vals[s]) + inv_sky_far * t_vals[s];
    }
    const size_t o = ((size_t)ray * kSkySamples + s0) * 4;
    float4 v[2];
    v[0] = *reinterpret_cast<const float4 *>(raw + o);
    v[1] = *reinterpret_cast<const float4 *>(raw + o + 4);
    float e[2], alpha[2], keep[2], dist[2], c[2][3], gc[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        dist[q] = (s0 + q + 1u < (uint32_t)kSkySamples ? zv[q + 1] - zv[q] : 1e10f) * dn;
        const float sg = fmaxf(v[q].w, 0.0f);
        e[q] = expf(-sg * dist[q]);
        alpha[q] = 1.0f - e[q];
        keep[q] = live ? (1.0f - alpha[q]) + 1e-10f : 1.0f;
        c[q][0] = 1.0f / (1.0f + expf(-v[q].x)); c[q][1] = 1.0f / (1.0f + expf(-v[q].y)); c[q][2] = 1.0f / (1.0f + expf(-v[q].z));
        gc[q] = (g0 * c[q][0] + g1 * c[q][1]) + g2 * c[q][2];
    }
    // transmittance in front of the lane's first sample: the exclusive product of the lanes below
    const float incl = wave_scan_product(keep[0] * keep[1]);
    const float below = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, incl), 0x138, 0xf, 0xf, false));
    float T[2] = {below, below * keep[0]};
    float w[2] = {alpha[0] * T[0], alpha[1] * T[1]};
    const float wg0 = live ? w[0] * gc[0] : 0.0f, wg1 = live ? w[1] * gc[1] : 0.0f;
    const float sum_incl = wave_scan_dpp<float>(wg0 + wg1);
    const float total = wave_last<float>(sum_incl);
    const float prefix[2] = {(sum_incl - (wg0 + wg1)) + wg0, sum_incl};
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const float dalpha = T[q] * gc[q] - (total - prefix[q]) / keep[q];
        const float dsig = v[q].w > 0.0f ? dalpha * dist[q] * e[q] : 0.0f;
        *reinterpret_cast<float4 *>(g_raw + o + 4 * q) =
            make_float4(g0 * w[q] * c[q][0] * (1.0f - c[q][0]), g1 * w[q] * c[q][1] * (1.0f - c[q][1]), g2 * w[q] * c[q][2] * (1.0f - c[q][2]), dsig);
    }
}

// the forward compositing with the device-side far plane (k_sky_composite of sky.hip takes it from the host)
__global__ __launch_bounds__(64) void k_sky_composite_dev(const float *__restrict__ raw, const float *__restrict__ dirs,
                                                          const float *__restrict__ far_, const float *__restrict__ t_vals,
                                                          uint32_t N, float *__restrict__ out) {
    const uint32_t ray = blockIdx.x * 64u + threadIdx.x;
    if (ray >= N) return;
    const float inv_sky_far = inv_sky_far_of(far_);
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float nr = far_[ray];
    float T = 1.0f, r = 0.0f, g = 0.0f, bl = 0.0f;
    float z = nr * (1.0f - t_vals[0]) + inv_sky_far * t_vals[0];
    for (int s = 0; s < kSkySamples; s++) {
        float dist, zn = z;
        if (s + 1 < kSkySamples) {
            zn = nr * (1.0f - t_vals[s + 1]) + inv_sky_far * t_vals[s + 1];
            dist = zn - z;
        } else {
            dist = 1e10f;
        }
        dist = dist * dn;
        const float4 v = *reinterpret_cast<const float4 *>(raw + ((size_t)ray * kSkySamples + s) * 4);
        const float alpha = 1.0f - expf(-fmaxf(v.w, 0.0f) * dist);
        const float w = alpha * T;
        r += w * (1.0f / (1.0f + expf(-v.x)));
        g += w * (1.0f / (1.0f + expf(-v.y)));
        bl += w * (1.0f / (1.0f + expf(-v.z)));
        T = T * ((1.0f - alpha) + 1e-10f);
        z = zn;
    }
    out[ray * 3 + 0] = r; out[ray * 3 + 1] = g; out[ray * 3 + 2] = bl;
}

// bf16 A-fragments [otp][it][s][o2][lane][8] of a logical matrix V[row][col] = src[row * rs + col * cs] (0 outside
// nrows x ncols): lane (row = lane & 31, g = lane >> 5) element e holds V[32 (2 otp + o2) + row][32 it + perm(8 s + e, g)],
// perm(r, g) = (r & 3) + 8 (r >> 2) + 4 g -- the accumulator order of the layer that produced the input tile.
__global__ __launch_bounds__(256) void k_pack_frag_bf(const float *__restrict__ src, uint32_t rs, uint32_t cs, uint32_t nrows,
                                                      uint32_t ncols, uint32_t nt_out, uint32_t nt_in, __bf16 *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nt_out * nt_in * 2u * 512u) return;
    const uint32_t e = i & 7u, lane = (i >> 3) & 63u, grp = i >> 9;
    const uint32_t o2 = grp & 1u, s = (grp >> 1) & 1u, it = (grp >> 2) % nt_in, ot = 2u * ((grp >> 2) / nt_in) + o2;
    const uint32_t r = 8u * s + e;
    const uint32_t row = 32u * ot + (lane & 31u), col = 32u * it + (r & 3u) + 8u * (r >> 2) + 4u * (lane >> 5);
    dst[i] = (__bf16)((row < nrows && col < ncols) ? src[(size_t)row * rs + (size_t)col * cs] : 0.0f);
}

// scratch matrices of the two composite backward stages:
//   B7 [256][160]: columns 0..127 = Mv[:, n]^T (the composed views layer, hidden part), column 131 = w_alpha[n] (meets d sigma)
//   Bv [128][32]:  columns 0..2 = W_rgb[c][n]
__global__ __launch_bounds__(256) void k_sky_train_scratch(const float *__restrict__ mv, const float *__restrict__ w_alpha,
                                                           const float *__restrict__ w_rgb, float *__restrict__ B7,
                                                           float *__restrict__ Bv) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 256u * 160u) {
        const uint32_t n = i / 160u, c = i - n * 160u;
        B7[i] = c < 128u ? mv[(size_t)c * 288 + n] : (c == 131u ? w_alpha[n] : 0.0f);
    }
    if (i < 128u * 32u) {
        const uint32_t n = i >> 5, c = i & 31u;
        Bv[i] = c < 3u ? w_rgb[(size_t)c * 128 + n] : 0.0f;
    }
}

__global__ __launch_bounds__(64) void k_sky_train_scalars(const float *__restrict__ b_alpha, const float *__restrict__ b_rgb,
                                                          float *__restrict__ side) {
    if (threadIdx.x == 0) side[kSAlpha + 256] = b_alpha[0];
    if (threadIdx.x < 3) side[kSRgb + 512 + threadIdx.x] = b_rgb[threadIdx.x];
}

}  // namespace

extern "C" uint64_t ucn_sky_train_packed_bytes(void) { return kPkBytes; }
extern "C" uint32_t ucn_sky_train_act_ld(void) { return kActLd; }
extern "C" uint32_t ucn_sky_train_grad_ld(void) { return kDlLd; }

extern "C" int ucn_sky_train_pack(const ucn_sky_train_t *s, ucn_stream_t stream) {
    UCN_REQUIRE(s && s->packed, "sky_train_pack: null descriptor / packed buffer");
    for (int i = 0; i < 8; i++) UCN_REQUIRE(i == 5 || (s->w_pts[i] && s->b_pts[i]), "sky_train_pack: pts_linears.%d missing", i);
    UCN_REQUIRE(s->m5 && s->mv && s->w_alpha && s->b_alpha && s->w_rgb && s->b_rgb, "sky_train_pack: composed / head weights missing");
    hipStream_t st = (hipStream_t)stream;
    uint8_t *pk = reinterpret_cast<uint8_t *>(s->packed);
    __bf16 *fwd = reinterpret_cast<__bf16 *>(pk), *bwd = reinterpret_cast<__bf16 *>(pk + kPkBwd);
    float *side = reinterpret_cast<float *>(pk + kPkSide), *B7 = reinterpret_cast<float *>(pk + kPkB7), *Bv = reinterpret_cast<float *>(pk + kPkBv);
    // zero: the streams' tails (8 padding fragments each) and the side table's gaps
    hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up((uint64_t)(kTrPadded - kTrFrags) * 256, 256)), dim3(256), 0, st,
                       reinterpret_cast<float *>(fwd + (size_t)kTrFrags * 512), (uint32_t)((kTrPadded - kTrFrags) * 256));
    hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up((uint64_t)(kTrPadded - kTrFrags) * 256, 256)), dim3(256), 0, st,
                       reinterpret_cast<float *>(bwd + (size_t)kTrFrags * 512), (uint32_t)((kTrPadded - kTrFrags) * 256));
    hipLaunchKernelGGL(k_fill_zero, dim3(ucn_div_up(kSideFloats, 256)), dim3(256), 0, st, side, (uint32_t)kSideFloats);
    hipLaunchKernelGGL(k_sky_train_scratch, dim3(ucn_div_up(256 * 160, 256)), dim3(256), 0, st, s->mv, s->w_alpha, s->w_rgb, B7, Bv);
    auto pack = [&](const float *src, uint32_t rs, uint32_t cs, uint32_t nr, uint32_t nc, uint32_t nto, uint32_t nti, __bf16 *dst) {
        hipLaunchKernelGGL(k_pack_frag_bf, dim3(ucn_div_up((uint64_t)nto * nti * 1024, 256)), dim3(256), 0, st, src, rs, cs, nr, nc,
                           nto, nti, dst);
    };
    const int plain[6] = {1, 2, 3, 4, 6, 7};
    for (int i = 0; i < 6; i++) {
        pack(s->w_pts[plain[i]], 256, 1, 256, 256, 8, 8, fwd + (size_t)kFL[plain[i] - 1] * 512);
        hipLaunchKernelGGL(k_pack_bias_h, dim3(1), dim3(256), 0, st, s->b_pts[plain[i]], 8u, side + kSB + i * 256);
    }
    pack(s->m5, 288, 1, 256, 288, 8, 9, fwd + (size_t)kFL[4] * 512);
    pack(s->mv, 288, 1, 128, 288, 4, 9, fwd + (size_t)kFV * 512);
    // backward: transposed views of the same matrices, in the order the dgrad chain consumes them
    pack(Bv, 32, 1, 128, 32, 4, 1, bwd + (size_t)kGV * 512);
    pack(B7, 160, 1, 256, 160, 8, 5, bwd + (size_t)kG7 * 512);
    const int down[7] = {7, 6, 5, 4, 3, 2, 1};                     // stage q = 1..7: W_{down[q-1]}^T
    for (int q = 0; q < 7; q++) {
        const int l = down[q];
        if (l == 5) pack(s->m5, 1, 288, 256, 256, 8, 8, bwd + (size_t)(kGL + 128 * q) * 512);       // M5[:, :256]^T
        else pack(s->w_pts[l], 1, 256, 256, 256, 8, 8, bwd + (size_t)(kGL + 128 * q) * 512);
    }
    hipLaunchKernelGGL(k_pack_in3, dim3(4), dim3(256), 0, st, s->w_pts[0], 3u, s->b_pts[0], 256u, side + kSL0);
    hipLaunchKernelGGL(k_pack_head, dim3(1), dim3(256), 0, st, s->w_alpha, 256u, 0u, 256u, 1u, 1u, side + kSAlpha);
    hipLaunchKernelGGL(k_pack_head, dim3(2), dim3(256), 0, st, s->w_rgb, 128u, 0u, 128u, 3u, 4u, side + kSRgb);
    hipLaunchKernelGGL(k_sky_train_scalars, dim3(1), dim3(64), 0, st, s->b_alpha, s->b_rgb, side);
    UCN_LAUNCH_CHECK("sky_train_pack");
    return 0;
}

extern "C" int ucn_sky_train_fwd(const void *packed, const float *origins, const float *directions, const float *cam_dirs,
                                 const float *far_, const float *t_vals, uint32_t N, float *aux_ws, float *raw, void *act,
                                 void *mask, void *mask_v, float *sky_rgb_out, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (packed && origins && directions && cam_dirs && far_ && t_vals && aux_ws && raw && act && mask && mask_v && sky_rgb_out),
                "sky_train_fwd: null pointer argument");
    if (N == 0) return 0;
    UCN_REQUIRE((uint64_t)N * kSkySamples < 0xFFFFFF00ull, "sky_train_fwd: too many samples");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sky_aux, dim3(ucn_div_up((uint64_t)N * 32, 256)), dim3(256), 0, st, cam_dirs, N, aux_ws);
    SkyTrainArgs a{reinterpret_cast<const uint8_t *>(packed), aux_ws, origins, directions, far_, t_vals, N, raw,
                   reinterpret_cast<uint16_t *>(act), reinterpret_cast<uint4 *>(mask), reinterpret_cast<uint2 *>(mask_v)};
    const uint64_t M = (uint64_t)N * kSkySamples;
    const size_t lds = ((size_t)kFwdSlots * kTChunk * 256 + kSideFloats) * sizeof(float) + (UCN_SKY_FWD_STAGED ? kFwdWaves * kStageTile : 0);
    const char *tiles_env = getenv("UCN_SKY_FWD_TILES");        // per call: the A/B tools flip it inside one process
    const int tiles = tiles_env ? atoi(tiles_env) : UCN_SKY_FWD_TILES;
    if (tiles == 2 && UCN_SKY_FWD_OCC == 1 && kFwdWaves == 4) hipLaunchKernelGGL(k_sky_train_fwd2, dim3(ucn_div_up(M, 256)), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(k_sky_train_fwd, dim3(ucn_div_up(M, 32 * kFwdWaves)), dim3(64 * kFwdWaves), lds, st, a);
    hipLaunchKernelGGL(k_sky_composite_dev, dim3(ucn_div_up(N, 64)), dim3(64), 0, st, raw, directions, far_, t_vals, N, sky_rgb_out);
    UCN_LAUNCH_CHECK("sky_train_fwd");
    return 0;
}

extern "C" int ucn_sky_train_bwd(const void *packed, const float *g_sky_rgb, const float *raw, const float *directions,
                                 const float *far_, const float *t_vals, uint32_t N, const void *mask, const void *mask_v,
                                 float *g_raw_ws, void *grad, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (packed && g_sky_rgb && raw && directions && far_ && t_vals && mask && mask_v && g_raw_ws && grad),
                "sky_train_bwd: null pointer argument");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t M = (uint64_t)N * kSkySamples;
    const char *serial = getenv("UCN_SKY_COMP_SERIAL");           // A/B switch (read per call): 1 = one thread per ray
    if (serial && atoi(serial) == 1)
        hipLaunchKernelGGL(k_sky_composite_bwd, dim3(ucn_div_up(N, 64)), dim3(64), 0, st, raw, directions, far_, t_vals, g_sky_rgb, N, g_raw_ws);
    else
        hipLaunchKernelGGL(k_sky_composite_bwd_wave, dim3(ucn_div_up(N, 4)), dim3(256), 0, st, raw, directions, far_, t_vals, g_sky_rgb, N,
                           g_raw_ws);
    SkyTrainBwdArgs a{reinterpret_cast<const uint8_t *>(packed), g_raw_ws, reinterpret_cast<const uint4 *>(mask),
                      reinterpret_cast<const uint2 *>(mask_v), reinterpret_cast<uint16_t *>(grad), (uint32_t)M};
    hipLaunchKernelGGL(k_sky_train_bwd, dim3(ucn_div_up(M, 32 * kBwdWaves)), dim3(64 * kBwdWaves),
                       (size_t)kBwdSlots * kTChunk * 1024 + (UCN_SKY_BWD_STAGED ? kBwdWaves * kStageTile : 0), st, a);
    UCN_LAUNCH_CHECK("sky_train_bwd");
    return 0;
}
