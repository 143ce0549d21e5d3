// Training-time FORWARD of the NeRF field's dense layers as one kernel: the five layers
// of models.py:507-674 in the reference's own (uncomposed) topology, bf16 operands with fp32 accumulation -- what the
// reference's `accelerator.autocast()` makes of them -- with every hidden activation the backward needs written once:
//
//   h0 = relu(W_d0 feat + b_d0)            [64]      density layer 0
//   x  = W_d1 h0 + b_d1                    [256]     bottleneck; x[0] = raw density
//   h1 = relu(W0x x  + pr0[ray])           [256]     pr0 = W0e enc + b0, per RAY (computed by the caller, [N, 256])
//   h2 = relu(W1h h1 + W1x x + pr1[ray])   [256]     the skip connection as a second block of the same GEMM
//        (evaluated as (W0x W_d1) h0 and (W1x W_d1) h0 with W b_d1 inside pr0 / pr1: see k_train_fwd)
//   y  = W_rgb h2 + b_rgb                  [3]       pre-sigmoid colour
//
// As library GEMMs these layers are HBM-bound on their activations (1 GB in + out per 256x256 product, plus the
// elementwise passes between them); here a wave keeps its 32 samples' activations in registers from the 128-byte
// feature row to the three colour logits and HBM sees each activation exactly once, as a store.
//
// One wave = 32 samples (the N of v_mfma_f32_32x32x16_bf16), 4 waves per workgroup.  Weights are packed by the host
// into MFMA A-fragments in consumption order (frag = 64 lanes x 8 bf16 = 1 KiB; k-order permuted to the accumulator
// layout of the producing layer, exactly like the rendering engine's packers): 436 fragments per tile, staged through
// LDS in 64 KiB chunks shared by the workgroup's four waves (reading them per wave straight from L2 was 5x slower).
#include "bf_tiles.h"

namespace {


struct TrainFwdArgs {
    const float *feat;        // [M, F] fp32, F <= 64 (sample-major, what k_march_features writes for the GEMMs)
    const uint4 *w;           // packed fragments: L0 (2x1x2), L1 (8x2x2), L2 (8x8x2), L3 (8x16x2), L4 (1x8x2)
    const float *bias_d0, *bias_d1, *bias_rgb;   // accumulator order: [tile][h][16]
    const float *pr0, *pr1;   // [N, 8][2][16] fp32, accumulator order per ray
    uint16_t *h0, *x, *h1, *h2;                  // bf16 [M, 64] / [M, 256] row-major, row strides ld_h0 / ld_act elements
    const uint16_t *ray_cols; // [N, 32] bf16 per-RAY columns copied to every sample's row at ray_dst (row stride ld_act), or NULL
    uint16_t *ray_dst;
    uint16_t *fb;             // [M, F] bf16 copy of the features (the density layer's weight-gradient operand) or NULL
    float *raw, *y;           // [M], [M, 3]: raw density and colour logits, or (head) density and rgb
    uint32_t ld_h0, ld_act, ld_fb;
    int store;                // 0: inference -- no activation / mask stores (h0, x, h1, h2, m0, m1, m2 are NULL)
    int feat_bf16_in;         // (inference form) feat holds [L][B] pairs of bf16 (level_dim 2): what UCN_FEATURES_BF16 wrote
    uint32_t n_rays, level_dim;   // level_dim != 0 (inference only): feat is the rendering gather's [L][B][C] with b = s * n_rays + ray
    //                              (rays fastest) and the wave's 32 lanes are 32 consecutive b; outputs stay [ray][sample]
    int head;                 // 1: raw := softplus(raw + density_bias), y := sigmoid(premult y + rgb_bias) (1 + 2 pad) - pad
    float density_bias, rgb_premult, rgb_bias, rgb_padding;
    uint32_t *m0;             // [M][2] : ReLU masks of h0, bit 16 t + r of wave half h  (2 tiles)
    uint4 *m1, *m2;           // [M][2] : ReLU masks of h1 / h2, 16 bits per tile, 8 tiles
    uint32_t M, S, F;
};


// UCN_TRAIN_OCC: workgroups per CU both kernels are cut for (build knob); the ring follows it (two per CU: 4 x 16 KiB, 2 chunks ahead;
// one per CU: 6 x 16 KiB, 4 ahead)
#ifndef UCN_TRAIN_OCC
#define UCN_TRAIN_OCC 2
#endif
// UCN_TRAIN_WAVES: waves per workgroup of the two training kernels (4: two workgroups per CU; 8: ONE workgroup per CU whose eight waves
// share one ring -- half the LDS-DMA weight stream per sample, a 6-slot ring, still two waves per SIMD)
#ifndef UCN_TRAIN_WAVES
#define UCN_TRAIN_WAVES 8
#endif
constexpr int kTrainWaves = UCN_TRAIN_WAVES;
// UCN_TRAIN_STAGED (r05): activation / gradient pairs leave through a per-wave LDS tile as whole 128-byte lines (bf_tiles.h
// store_pair_staged) instead of one 64-byte sector per lane: the forward kernel ran 0.255 ms without its stores and 0.553 with them.
// Two workgroups per CU then have room for a 3-slot ring only (3 x 16 + 9.25 side + 18 staging = 75.25 KiB each), which by itself
// costs 2-4 % (measured: 0.563 / 0.577 against 0.553 / 0.553 ms).
#ifndef UCN_TRAIN_STAGED
#define UCN_TRAIN_STAGED 1
#endif
#ifndef UCN_TRAIN_SLOTS
#define UCN_TRAIN_SLOTS (UCN_TRAIN_OCC == 1 || UCN_TRAIN_WAVES == 8 ? 6 : (UCN_TRAIN_STAGED ? 3 : kTSlots))
#endif
#ifndef UCN_TRAIN_LEAD
#define UCN_TRAIN_LEAD (UCN_TRAIN_OCC == 1 || UCN_TRAIN_WAVES == 8 ? 4 : (UCN_TRAIN_STAGED ? 1 : kTLead))
#endif
constexpr int kFtSlots = UCN_TRAIN_SLOTS, kFtLead = UCN_TRAIN_LEAD;
// UCN_TRAIN_PIPE (r05): weight fragments requested kWAhead ahead of their MFMAs through a register pipe (bf_tiles.h tile_pair_pf)
#ifndef UCN_TRAIN_PIPE
#define UCN_TRAIN_PIPE 1
#endif
#ifndef UCN_TRAIN_PAIR_FWD
#define UCN_TRAIN_PAIR_FWD 1
#endif
#ifndef UCN_TRAIN_PAIR_BWD
#define UCN_TRAIN_PAIR_BWD 0
#endif
// Weight staging = the rendering engine's DMA ring (mlp_ring.h): a 64 KiB LDS ring of 4 x 16 KiB chunks filled by
// global_load_lds two chunks ahead, one piece per four MFMAs, one barrier per chunk -- 64 KiB and <= 256 registers per
// wave, so that TWO workgroups share a CU: at one wave per SIMD (the first version: two 64 KiB buffers, 456 registers)
// both kernels spent 75 % of their wave-cycles waiting (profiles/r02c/pmc_table_train.txt).
constexpr int kFragsMax = 2 * 2 * 2 + 8 * 2 * 2 + 8 * 8 * 2 + 8 * 16 * 2 + 1 * 8 * 2;   // 440 with two feature tiles, 436 with one
constexpr int kFragsPadded = (kFragsMax + kTChunk - 1) / kTChunk * kTChunk;               // 448: the stream is zero-padded
using TRing = Ring<kFragsPadded, kTChunk, kTrainWaves, kFtSlots, kFtLead>;                            // the backward's stream
// the forward's stream (composed colour layers, see k_train_fwd): 2 NTF 2 + 32 + 32 + 4 (40 + 4) = 244 / 248 fragments
constexpr int kFwdFragsMax = 2 * 2 * 2 + 32 + 4 * 12 + 4 * 48;     // with the direction tile in the stream (inference): 276 / 280
constexpr int kFwdPadded = (kFwdFragsMax + kTChunk - 1) / kTChunk * kTChunk;              // 288
using FRing = Ring<kFwdPadded, kTChunk, kTrainWaves, kFtSlots, kFtLead>;
template <int P, int NT_IN, int G0, int NG, class RING>
__device__ __forceinline__ void tile_pair_sel(RING &ring, bf8 (&wp)[4], f32x16 (&acc)[P], const bf8 (&in)[NT_IN][2]) {
    if constexpr (UCN_TRAIN_PIPE != 0) tile_pair_pf<P, NT_IN, G0, NG>(ring, wp, acc, in);
    else tile_pair<P, NT_IN, G0>(ring, acc, in);
}


#ifndef UCN_TRAIN_FWD_WGS
#define UCN_TRAIN_FWD_WGS UCN_TRAIN_OCC
#endif
// AUX (inference with rays-fastest lanes): the per-ray direction term is NOT pre-multiplied by the caller (pr0 / pr1 would
// be 2 KiB per LANE there, 64 KiB of loads per wave); the ray's 32-column tile [dir_enc (27), 1, 0...] (a.ray_cols) enters
// the two colour layers as one more input tile whose column 27 carries the layer bias, like in the rendering kernel.
// The AUX form is inference only: its store paths fold away at compile time (88 -> 79 -> 70.5 ms per frame with them gone).
// kInferWaves: workgroup shape of that form -- 4 = two 4-wave workgroups per CU like the training form; 8 / 12 = ONE
// workgroup around a larger ring (the sky layer's bf16 shape; here 12 waves need 168 registers and spill: 72.6 ms, 8: 73.7).
#ifndef UCN_INFER_WAVES
#define UCN_INFER_WAVES 4
#endif
constexpr int kInferWaves = UCN_INFER_WAVES;
using IRing = Ring<(kFwdFragsMax + 2 * kInferWaves - 1) / (2 * kInferWaves) * (2 * kInferWaves), kInferWaves == 4 ? kTChunk : 2 * kInferWaves,
                   kInferWaves, kFtSlots, kFtLead>;
static_assert(IRing::kChunks * IRing::kChunk <= kFwdPadded, "the packed forward stream is padded to kFwdPadded fragments");
template <bool AUX> struct FwdShape { using ring = FRing; static constexpr int waves = kTrainWaves, wgs = kTrainWaves == 8 ? 1 : UCN_TRAIN_FWD_WGS; };
template <> struct FwdShape<true> { using ring = IRing; static constexpr int waves = kInferWaves, wgs = kInferWaves == 4 ? 2 : 1; };

// SIDE (r05, training form): the layer biases and the wave's two per-ray rows (pr0 / pr1: all 32 samples of a wave belong to one ray
// when S % 32 == 0) are copied to LDS ONCE, before the chain, and every pair's accumulator start comes from there.  As global loads
// inside the chain (r02-r04) each of the 12 pair starts carried a compiler-inserted `s_waitcnt vmcnt(0)` in front of its first MFMA
// -- the compiler cannot see the hand-placed LDS-DMA stream, so its wait drained the whole weight look-ahead, twelve times per pass.
#ifndef UCN_TRAIN_FWD_SIDE
#define UCN_TRAIN_FWD_SIDE 1
#endif
constexpr int kSideBias = 16 + 64, kSideWave = 128;                    // float4: [bias_d0 (16) | bias_d1 (64)], then per wave [pr0 (64) | pr1 (64)]
constexpr size_t kFwdSideBytes = (size_t)(kSideBias + kTrainWaves * kSideWave) * 16;
constexpr size_t kStageBytes = UCN_TRAIN_STAGED ? kTrainWaves * kStageTile : 0;           // one staging tile per wave (training forms)

template <int NTF, bool AUX = false, bool SIDE = false>   // feature tiles: F <= 32 * NTF
__global__ __launch_bounds__(64 * FwdShape<AUX>::waves, FwdShape<AUX>::wgs) void k_train_fwd(TrainFwdArgs aa) {
    static_assert(!(AUX && SIDE), "the side table is the training form's");
    TrainFwdArgs a = aa;
    if constexpr (AUX) a.store = 0;
    else a.level_dim = 0;                 // the training form reads sample-major features (checked by the entry point)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform: the DMA addresses live in SGPRs
    const int j = lane & 31, h = lane >> 5;
    const uint32_t s0 = (blockIdx.x * (uint32_t)FwdShape<AUX>::waves + wave) * 32u + j;
    const bool live = s0 < a.M;
    const uint32_t bq = live ? s0 : a.M - 1;             // position in the feature buffer
    const uint32_t ray = a.level_dim ? bq % a.n_rays : bq / a.S;
    const uint32_t sample = a.level_dim ? ray * a.S + bq / a.n_rays : bq;   // position in the [ray][sample] outputs
    extern __shared__ __attribute__((aligned(16))) float s_w[];      // the weight ring [| side table] [| one staging tile per wave]
    constexpr bool kStaged = UCN_TRAIN_STAGED != 0 && !AUX;
    const uint32_t wb0 = (blockIdx.x * (uint32_t)FwdShape<AUX>::waves + wave) * 32u;           // the wave's first sample
    const uint32_t n_rows = wb0 < a.M ? (a.M - wb0 < 32u ? a.M - wb0 : 32u) : 0u;
    uint8_t *stage = reinterpret_cast<uint8_t *>(s_w) + kFtSlots * kTChunk * 1024 + (SIDE ? kFwdSideBytes : 0) + wave * kStageTile;
    typename FwdShape<AUX>::ring ring(reinterpret_cast<const float *>(a.w), s_w, lane, wave);
    auto store_fwd = [&](uint16_t *dst, uint32_t ld, int tp, const bf8 (&t0)[2], const bf8 (&t1)[2]) {
        if constexpr (kStaged) store_pair_staged(stage, dst, ld, wb0, n_rows, tp, lane, t0, t1);
        else store_two<UCN_TRAIN_PAIR_FWD != 0>(dst, ld, sample, tp, h, t0, t1, live);
    };
    ring_start(ring);
    const float *side_b0 = nullptr, *side_b1 = nullptr, *side_p0 = nullptr, *side_p1 = nullptr;
    if constexpr (SIDE) {
        float4 *side = reinterpret_cast<float4 *>(s_w + kFtSlots * kTChunk * 256);
        if (threadIdx.x < 16) side[threadIdx.x] = reinterpret_cast<const float4 *>(a.bias_d0)[threadIdx.x];
        else if (threadIdx.x < (uint32_t)kSideBias) side[threadIdx.x] = reinterpret_cast<const float4 *>(a.bias_d1)[threadIdx.x - 16];
        const uint32_t w0 = (blockIdx.x * (uint32_t)kTrainWaves + wave) * 32u;                  // the wave's first sample: its ray is every lane's
        const uint32_t wray = (w0 < a.M ? w0 : a.M - 1) / a.S;
        float4 *wp = side + kSideBias + wave * kSideWave;
        wp[lane] = reinterpret_cast<const float4 *>(a.pr0)[(size_t)wray * 64 + lane];
        wp[64 + lane] = reinterpret_cast<const float4 *>(a.pr1)[(size_t)wray * 64 + lane];
        ring_wait_lds<0>();                                                                     // written before boundary<0>'s barrier
        side_b0 = reinterpret_cast<const float *>(side);
        side_b1 = side_b0 + 64;
        side_p0 = reinterpret_cast<const float *>(wp);
        side_p1 = side_p0 + 256;
    }

    // ---- features: lane (j, h) supplies k = 16 s + 8 h + e of its sample
    bf8 fin[NTF][2];
#pragma unroll
    for (int ft = 0; ft < NTF; ft++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t k = 32u * ft + 16u * s + 8u * h + e;
                if (AUX && a.feat_bf16_in) {                   // bf16 pairs as the gather left them: one 4-byte load per level, no conversion
                    if (e % 2 == 0) {
                        const uint32_t t = k < a.F ? reinterpret_cast<const uint32_t *>(a.feat)[(size_t)(k >> 1) * a.M + bq] : 0u;
                        v[e] = __uint_as_float(t << 16);
                        v[e + 1] = __uint_as_float(t & 0xFFFF0000u);
                    }
                } else if (a.level_dim == 2u) {                // one 8-byte load per level
                    if (e % 2 == 0) {
                        const float2 t = k < a.F ? *reinterpret_cast<const float2 *>(a.feat + ((size_t)(k >> 1) * a.M + bq) * 2) : make_float2(0.0f, 0.0f);
                        v[e] = t.x;
                        v[e + 1] = t.y;
                    }
                } else if (a.level_dim) {
                    const uint32_t l = k / a.level_dim, c = k - l * a.level_dim;
                    v[e] = k < a.F ? a.feat[((size_t)l * a.M + bq) * a.level_dim + c] : 0.0f;
                } else {
                    v[e] = k < a.F ? a.feat[(size_t)sample * a.F + k] : 0.0f;
                }
            }
            fin[ft][s] = pack8(v);
#ifndef UCN_EXP_NOAUXSTORE
            if (a.fb && live && a.F % 8 == 0 && 32u * ft + 16u * s + 8u * h < a.F)
                *reinterpret_cast<uint4 *>(a.fb + (size_t)sample * a.ld_fb + 32u * ft + 16u * s + 8u * h) = __builtin_bit_cast(uint4, fin[ft][s]);
#endif
        }
#ifndef UCN_EXP_NOAUXSTORE
    if (!AUX && a.ray_cols && live) {       // lane (j, h): columns 16 h .. 16 h + 15 of its sample's row
        const uint4 *src = reinterpret_cast<const uint4 *>(a.ray_cols + (size_t)ray * 32 + 16 * h);
        uint4 *dst = reinterpret_cast<uint4 *>(a.ray_dst + (size_t)sample * a.ld_act + 16 * h);
        dst[0] = src[0];
        dst[1] = src[1];
    }
#endif
    ring.template boundary<0>();            // chunk 0 and the feature loads above land together
    // The bottleneck x = W_d1 h0 + b_d1 has NO activation behind it (models.py:508) and only linear maps consume it: the
    // colour layers take it COMPOSED, (W0x W_d1) h0 and (W1x W_d1) h0 (the host forms the two 256 x 64 products and folds
    // W b_d1 into the per-ray terms every step), exactly like the rendering kernel.  x itself is still computed -- the
    // backward's weight gradients and the density head need it -- but only to be stored: 244 fragments instead of 436, and
    // the 64 registers that held x as an operand are free, which is what lets two workgroups share a CU without spills.
    // stream positions: L0 | L1: 4 pairs x 8 | L2': 4 pairs x 8 | 4 x (L3' pair: [h1 (8 tiles) | h0 (2 tiles)] = 40, then the
    // rgb layer's fragments for the two h2 tiles just finished: 4)
    constexpr int NA = AUX ? 1 : 0;                        // extra input tiles of the colour layers
    constexpr int P2 = (2 + NA) * 4, P3 = (10 + NA) * 4;   // fragments per output pair of L2' / L3'
    constexpr int G1 = 2 * NTF * 2, G2 = G1 + 32, G3 = G2 + 4 * P2;
    constexpr int NGF = G3 + 4 * (P3 + 4);                 // fragments this kernel consumes
    bf8 wp[kWSlots];                                             // fragment pipe
    if constexpr (UCN_TRAIN_PIPE != 0) sfor<kWAhead>([&](auto g) { frag_fetch<g.value, NGF>(ring, wp); });
    // ---- density layer 0
    bf8 hin[10 + NA][2];                   // tiles 0..7: h1 (filled below), 8..9: h0, (10: the ray's direction tile)
    bf8 (&h0)[2][2] = reinterpret_cast<bf8(&)[2][2]>(hin[8]);
    if constexpr (AUX) {
        // accumulator-order operand of a 32-wide natural row: k-step s = pieces {0-3, 8-11} / {16-19, 24-27} + 4 h
        const uint2 *src = reinterpret_cast<const uint2 *>(a.ray_cols + (size_t)ray * 32 + 4 * h);
        const uint2 p0 = src[0], p1 = src[2], p2 = src[4], p3 = src[6];
        hin[10][0] = __builtin_bit_cast(bf8, make_uint4(p0.x, p0.y, p1.x, p1.y));
        hin[10][1] = __builtin_bit_cast(bf8, make_uint4(p2.x, p2.y, p3.x, p3.y));
    }
    {
        f32x16 a0[2];
        load_acc((SIDE ? side_b0 : a.bias_d0) + (0 * 2 + h) * 16, a0[0]);
        load_acc((SIDE ? side_b0 : a.bias_d0) + (1 * 2 + h) * 16, a0[1]);
        tile_pair_sel<2, NTF, 0, NGF>(ring, wp, a0, fin);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            h0[t][0] = to_b(a0[t], 0, true);
            h0[t][1] = to_b(a0[t], 1, true);
        }
        if (a.store) store_fwd(a.h0, a.ld_h0, 0, h0[0], h0[1]);
        if (live && a.store) a.m0[(size_t)sample * 2 + h] = mask16(a0[0]) | (mask16(a0[1]) << 16);
    }
    // ---- density layer 1 -> bottleneck x (no activation), raw density = x[0]
    bf8 x0[2];                              // tile 0, k-step 0 of x: its first value is the raw density
    sfor<4>([&](auto pp) {
        constexpr int p = pp.value;
        f32x16 acc[2];
        load_acc((SIDE ? side_b1 : a.bias_d1) + ((2 * p) * 2 + h) * 16, acc[0]);
        load_acc((SIDE ? side_b1 : a.bias_d1) + ((2 * p + 1) * 2 + h) * 16, acc[1]);
        tile_pair_sel<2, 2, G1 + 8 * p, NGF>(ring, wp, acc, h0);
        bf8 xp[2][2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            xp[o][0] = to_b(acc[o], 0, false);
            xp[o][1] = to_b(acc[o], 1, false);
        }
        if constexpr (p == 0) x0[0] = xp[0][0];
        if (a.store && a.x) store_fwd(a.x, a.ld_act, 2 * p, xp[0], xp[1]);     // (r04: x = NULL -- its weight gradients are formed from h0, see _FusedHeads)
    });
    if (live && h == 0) {                   // row 0 = accumulator register 0 of tile 0 in wave-half 0, AFTER its bf16 rounding
        const uint4 q = __builtin_bit_cast(uint4, x0[0]);
        const float rawv = __uint_as_float(q.x << 16);
        // head: F.softplus (beta 1, threshold 20) of the bf16-rounded linear output, in fp32 like autocast runs it
        const float z = rawv + a.density_bias;
        a.raw[sample] = !a.head ? rawv : (z > 20.0f ? z : log1pf(__expf(z)));
    }
    // ---- colour layer 0: x -> h1
    {
        uint32_t mk[4];
        sfor<4>([&](auto pp) {
            constexpr int p = pp.value;
            f32x16 acc[2];
            if constexpr (AUX) {
                zero_acc(acc[0]);
                zero_acc(acc[1]);
            } else {
                if constexpr (SIDE) {
                    load_acc(side_p0 + (2 * p) * 32 + h * 16, acc[0]);
                    load_acc(side_p0 + (2 * p + 1) * 32 + h * 16, acc[1]);
                } else {
                    load_acc(a.pr0 + ((size_t)ray * 8 + 2 * p) * 32 + h * 16, acc[0]);
                    load_acc(a.pr0 + ((size_t)ray * 8 + 2 * p + 1) * 32 + h * 16, acc[1]);
                }
            }
            tile_pair_sel<2, 2 + NA, G2 + P2 * p, NGF>(ring, wp, acc, reinterpret_cast<const bf8(&)[2 + NA][2]>(hin[8]));
#pragma unroll
            for (int o = 0; o < 2; o++) {
                hin[2 * p + o][0] = to_b(acc[o], 0, true);
                hin[2 * p + o][1] = to_b(acc[o], 1, true);
            }
            if (a.store) store_fwd(a.h1, a.ld_act, 2 * p, hin[2 * p], hin[2 * p + 1]);
            mk[p] = mask16(acc[0]) | (mask16(acc[1]) << 16);
        });
        if (live && a.store) a.m1[(size_t)sample * 2 + h] = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    }
    // ---- colour layer 1: [h1, x] -> h2, and the rgb layer (3 rows of one padded output tile) on each finished pair
    float y3[3] = {a.bias_rgb[h * 16 + 0], a.bias_rgb[h * 16 + 1], a.bias_rgb[h * 16 + 2]};   // rows 0..2 live in wave half 0
    {
        uint32_t mk[4];
        sfor<4>([&](auto pp) {
            constexpr int p = pp.value;
            f32x16 acc[2];
            if constexpr (AUX) {
                zero_acc(acc[0]);
                zero_acc(acc[1]);
            } else {
                if constexpr (SIDE) {
                    load_acc(side_p1 + (2 * p) * 32 + h * 16, acc[0]);
                    load_acc(side_p1 + (2 * p + 1) * 32 + h * 16, acc[1]);
                } else {
                    load_acc(a.pr1 + ((size_t)ray * 8 + 2 * p) * 32 + h * 16, acc[0]);
                    load_acc(a.pr1 + ((size_t)ray * 8 + 2 * p + 1) * 32 + h * 16, acc[1]);
                }
            }
            tile_pair_sel<2, 10 + NA, G3 + (P3 + 4) * p, NGF>(ring, wp, acc, hin);
            bf8 hp[2][2];
#pragma unroll
            for (int o = 0; o < 2; o++) {
                hp[o][0] = to_b(acc[o], 0, true);
                hp[o][1] = to_b(acc[o], 1, true);
            }
            if (a.store) store_fwd(a.h2, a.ld_act, 2 * p, hp[0], hp[1]);
            mk[p] = mask16(acc[0]) | (mask16(acc[1]) << 16);
            f32x16 yo[1];                          // transient: four MFMAs, then only its three real rows are kept
            zero_acc(yo[0]);
            tile_pair_sel<1, 2, G3 + (P3 + 4) * p + P3, NGF>(ring, wp, yo, hp);
            y3[0] += yo[0][0]; y3[1] += yo[0][1]; y3[2] += yo[0][2];
        });
        if (live && a.store) a.m2[(size_t)sample * 2 + h] = make_uint4(mk[0], mk[1], mk[2], mk[3]);
    }
    if (live && h == 0) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            float v = y3[e];
            if (a.head) v = fmaf(1.0f / (1.0f + __expf(-fmaf(a.rgb_premult, v, a.rgb_bias))), 1.0f + 2.0f * a.rgb_padding, -a.rgb_padding);
            a.y[(size_t)sample * 3 + e] = v;
        }
    }
}

// The same chain backwards (dgrad): transposed weights in the same fragment stream format, the forward's ReLU masks
// instead of the activations, every pre-activation gradient the weight-gradient GEMMs need stored once:
//   d1 = (Wr^T gy) * m2          [256]      gx  = W1x^T d1 + W0x^T d0 (+ g_raw on feature 0)   [256]
//   d0 = (W1h^T d1) * m1         [256]      gh0 = (Wd1^T gx) * m0 [64],   gfeat = Wd0^T gh0    [F] fp32
struct TrainBwdArgs {
    const void *gy;           // [M, 3] bf16 logit gradients, or (head) fp32 gradients of rgb
    const void *graw;         // [M] bf16 gradient of raw, or (head) fp32 gradient of density; or NULL
    int head;                 // 1: the activation derivatives are applied here from the saved outputs density / rgb
    const float *density, *rgb;
    float rgb_premult, rgb_padding;
    const uint4 *w;           // fragments: Wr^T (8x1x2), W1h^T (8x8x2), [W1x^T | W0x^T] (8x16x2), Wd1^T (2x8x2), Wd0^T (1x2x2)
    const uint32_t *m0;
    const uint4 *m1, *m2;
    uint16_t *d1, *d0, *gx, *gh0;     // bf16 [M,256] x3, [M,64]
    uint16_t *dy;                     // bf16 [M, ld_dy] | NULL: the colour-logit gradient this kernel consumed (columns 0..2), column 3 = the density head's
    uint32_t ld_dy;                   // row stride of dy in elements (4, or 32: a zero-padded tile that ucn_wgrad_bf16 takes as its A operand)
    float *gfeat;                     // [M, F]; lm: [F / 2][M][2], every value / 6
    uint32_t M, F;
    uint32_t lm;                      // 1 (level_dim 2) / 2 (level_dim 4): the feature gradient as ucn_march_features_backward's layout 4 wants it (r06, VERDICT r05 item 2 c)
};


template <int NTF>
__global__ __launch_bounds__(64 * kTrainWaves, kTrainWaves == 8 ? 1 : UCN_TRAIN_OCC) void k_train_bwd(TrainBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t s0 = (blockIdx.x * (uint32_t)kTrainWaves + wave) * 32u + j;
    const bool live = s0 < a.M;
    const uint32_t sample = live ? s0 : a.M - 1;
    extern __shared__ __attribute__((aligned(16))) float s_w[];      // the weight ring [| one staging tile per wave]
    constexpr bool kStaged = UCN_TRAIN_STAGED != 0;
    const uint32_t wb0 = (blockIdx.x * (uint32_t)kTrainWaves + wave) * 32u;   // the wave's first sample
    const uint32_t n_rows = wb0 < a.M ? (a.M - wb0 < 32u ? a.M - wb0 : 32u) : 0u;
    uint8_t *stage = reinterpret_cast<uint8_t *>(s_w) + kFtSlots * kTChunk * 1024 + wave * kStageTile;
    auto store_bwd = [&](uint16_t *dst, uint32_t ld, int tp, const bf8 (&t0)[2], const bf8 (&t1)[2]) {
        if constexpr (kStaged) store_pair_staged(stage, dst, ld, wb0, n_rows, tp, lane, t0, t1);
        else store_two<UCN_TRAIN_PAIR_BWD != 0>(dst, ld, sample, tp, h, t0, t1, live);
    };
    TRing ring(reinterpret_cast<const float *>(a.w), s_w, lane, wave);
    ring_start(ring);
    // ---- colour logit gradients: k = 0..2 of k-step 0, wave half 0
    bf8 gin[1][2];
    {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < 3; e++) {
                if (a.head) {
                    const float k = 1.0f + 2.0f * a.rgb_padding;
                    const float sg = (a.rgb[(size_t)sample * 3 + e] + a.rgb_padding) / k;
                    v[e] = reinterpret_cast<const float *>(a.gy)[(size_t)sample * 3 + e] * k * a.rgb_premult * sg * (1.0f - sg);
                } else {
                    v[e] = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(a.gy)[(size_t)sample * 3 + e] << 16);
                }
            }
        }
        gin[0][0] = pack8(v);
        if (a.dy && live && h == 0) {
            const uint4 q = __builtin_bit_cast(uint4, gin[0][0]);
            *reinterpret_cast<uint2 *>(a.dy + (size_t)sample * a.ld_dy) = make_uint2(q.x, q.y);
        }
        const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        gin[0][1] = pack8(z);
    }
    const uint4 mk2 = a.m2[(size_t)sample * 2 + h], mk1 = a.m1[(size_t)sample * 2 + h];
    const uint32_t mk0 = a.m0[(size_t)sample * 2 + h];
    // the density head's gradient, fetched HERE with the masks (r05): as a global load in the middle of the chain it carried a
    // compiler-inserted `s_waitcnt vmcnt(0)` that drained the weight stream's look-ahead (the compiler cannot see the LDS-DMA)
    float gr_head = 0.0f;
    if (a.graw && h == 0) {
        // head: d softplus(z) / dz = sigmoid(z) = -expm1(-softplus(z)) (no cancellation in empty space), from the saved density; rounded to bf16 like
        // the gradient the per-layer path hands to the bottleneck's GEMM
        if (a.head) gr_head = (float)(__bf16)(reinterpret_cast<const float *>(a.graw)[sample] * (-expm1f(-a.density[sample])));
        else gr_head = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(a.graw)[sample] << 16);
    }
    const uint32_t m2w[4] = {mk2.x, mk2.y, mk2.z, mk2.w}, m1w[4] = {mk1.x, mk1.y, mk1.z, mk1.w};
    ring.template boundary<0>();
    // stream positions: Wr^T: 4 pairs x 4 | W1h^T: 4 pairs x 32 | 4 x ([W1x^T | W0x^T] pair: 64, then Wd1^T's fragments for the
    // two gx tiles just finished: 8) | Wd0^T
    constexpr int H1 = 16, H2 = H1 + 128, H4 = H2 + 4 * 72;
    constexpr int NGF = H4 + 4 * NTF;                      // fragments this kernel consumes
    bf8 wp[kWSlots];                                             // fragment pipe
    if constexpr (UCN_TRAIN_PIPE != 0) sfor<kWAhead>([&](auto g) { frag_fetch<g.value, NGF>(ring, wp); });
    bf8 din[16][2];                          // tiles 0..7: d1, 8..15: d0  (the order of [W1x^T | W0x^T])
    // ---- through the rgb layer and the second hidden layer's ReLU
    sfor<4>([&](auto pp) {
        constexpr int p = pp.value;
        f32x16 acc[2];
        zero_acc(acc[0]);
        zero_acc(acc[1]);
        tile_pair_sel<2, 1, 4 * p, NGF>(ring, wp, acc, gin);
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t bits = (m2w[p] >> (16 * o)) & 0xFFFFu;
            din[2 * p + o][0] = to_b_masked(acc[o], 0, bits);
            din[2 * p + o][1] = to_b_masked(acc[o], 1, bits);
        }
        store_bwd(a.d1, 256, 2 * p, din[2 * p], din[2 * p + 1]);
    });
    // ---- through W1h and the first hidden layer's ReLU
    sfor<4>([&](auto pp) {
        constexpr int p = pp.value;
        f32x16 acc[2];
        zero_acc(acc[0]);
        zero_acc(acc[1]);
        tile_pair_sel<2, 8, H1 + 32 * p, NGF>(ring, wp, acc, reinterpret_cast<const bf8(&)[8][2]>(din[0]));
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const uint32_t bits = (m1w[p] >> (16 * o)) & 0xFFFFu;
            din[8 + 2 * p + o][0] = to_b_masked(acc[o], 0, bits);
            din[8 + 2 * p + o][1] = to_b_masked(acc[o], 1, bits);
        }
        store_bwd(a.d0, 256, 2 * p, din[8 + 2 * p], din[9 + 2 * p]);
    });
    // ---- both paths into the bottleneck (plus the density head's column), and density layer 1 backwards on each
    //      finished pair of gx tiles
    f32x16 a0[2];
    zero_acc(a0[0]);
    zero_acc(a0[1]);
    sfor<4>([&](auto pp) {
        constexpr int p = pp.value;
        f32x16 acc[2];
        zero_acc(acc[0]);
        zero_acc(acc[1]);
        tile_pair_sel<2, 16, H2 + 72 * p, NGF>(ring, wp, acc, din);
        if constexpr (p == 0) {
            if (a.graw && h == 0) {
                const float gr = gr_head;
                acc[0][0] += gr;
                // (r04) the density head's gradient at the bottleneck's feature 0, kept as column 3 of dy: with gx not stored the
                // host forms its share of d W_d1[0, :] and d b_d1[0] from dy[:, 3]^T h0 (wave half 0 holds feature 0 of its sample in register 0 of tile 0)
                if (a.dy && live) a.dy[(size_t)sample * a.ld_dy + 3] = (uint16_t)(__float_as_uint((float)(__bf16)gr) >> 16);
            }
        }
        bf8 gp[2][2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            gp[o][0] = to_b(acc[o], 0, false);
            gp[o][1] = to_b(acc[o], 1, false);
        }
        if (a.gx) store_bwd(a.gx, 256, 2 * p, gp[0], gp[1]);                  // (r04: gx = NULL -- the bottleneck's weight gradient is formed from d0^T h0, d1^T h0)
        tile_pair_sel<2, 2, H2 + 72 * p + 64, NGF>(ring, wp, a0, gp);
    });
    // ---- ReLU of h0
    bf8 gh0[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint32_t bits = (mk0 >> (16 * t)) & 0xFFFFu;
        gh0[t][0] = to_b_masked(a0[t], 0, bits);
        gh0[t][1] = to_b_masked(a0[t], 1, bits);
    }
    store_bwd(a.gh0, 64, 0, gh0[0], gh0[1]);
    // ---- density layer 0 backwards: the feature gradient, fp32, natural feature order
    f32x16 gf[NTF];
#pragma unroll
    for (int ft = 0; ft < NTF; ft++) zero_acc(gf[ft]);
    tile_pair_sel<NTF, 2, H4, NGF>(ring, wp, gf, gh0);
    if (live) {
#pragma unroll
        for (int ft = 0; ft < NTF; ft++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t f0 = 32u * ft + 8u * q + 4u * h;            // features f0 .. f0 + 3 = registers 4q .. 4q + 3
                if (a.lm) {
                    // level-major pairs (level_dim 2: features 2 l, 2 l + 1 = level l), divided by the 6 multisamples of the mean with the
                    // same IEEE division the mask pass of the table gradient applied to this layout's row-major form: 32 lanes = 32
                    // consecutive samples = 256 contiguous bytes per level
                    if (f0 + 3 < a.F && a.lm == 1u) {
                        float2 *o = reinterpret_cast<float2 *>(a.gfeat);
                        o[(size_t)(f0 / 2u) * a.M + sample] = make_float2(gf[ft][4 * q] / 6.0f, gf[ft][4 * q + 1] / 6.0f);
                        o[(size_t)(f0 / 2u + 1u) * a.M + sample] = make_float2(gf[ft][4 * q + 2] / 6.0f, gf[ft][4 * q + 3] / 6.0f);
                    } else if (f0 + 3 < a.F) {                                 // level_dim 4 (the reference's own waymo.gin grid): one level per quad
                        reinterpret_cast<float4 *>(a.gfeat)[(size_t)(f0 / 4u) * a.M + sample] =
                            make_float4(gf[ft][4 * q] / 6.0f, gf[ft][4 * q + 1] / 6.0f, gf[ft][4 * q + 2] / 6.0f, gf[ft][4 * q + 3] / 6.0f);
                    }
                } else if (f0 + 3 < a.F && a.F % 4 == 0) {
                    *reinterpret_cast<float4 *>(a.gfeat + (size_t)sample * a.F + f0) =
                        make_float4(gf[ft][4 * q], gf[ft][4 * q + 1], gf[ft][4 * q + 2], gf[ft][4 * q + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (f0 + e < a.F) a.gfeat[(size_t)sample * a.F + f0 + e] = gf[ft][4 * q + e];
                }
            }
    }
}

}  // namespace

extern "C" uint64_t ucn_train_fwd_fragments(void) { return (uint64_t)kFragsPadded; }     // backward: 436 / 440 used; forward: 244 / 248

extern "C" int ucn_train_fwd(const float *feat, uint32_t F, const void *packed, const float *bias_d0, const float *bias_d1,
                             const float *bias_rgb, const float *pr0, const float *pr1, uint32_t N, uint32_t S, void *h0, void *x,
                             void *h1, void *h2, uint32_t act_ld, const void *ray_cols, void *ray_dst, void *feat_bf16, const float *head,
                             float *raw, float *y, uint32_t *m0, void *m1, void *m2, uint32_t feat_level_dim, ucn_stream_t stream) {
    const uint64_t M = (uint64_t)N * S;
    if (M == 0) return 0;
    UCN_REQUIRE(feat && packed && bias_d0 && bias_d1 && bias_rgb && raw && y, "train_fwd: null pointer argument");
    const bool aux = !pr0 && !pr1;           // direction tile in the stream instead of pre-multiplied per-ray terms
    const bool bf16_in = (feat_level_dim & UCN_FEAT_BF16) != 0;
    feat_level_dim &= ~(uint32_t)UCN_FEAT_BF16;
    UCN_REQUIRE(!bf16_in || (feat_level_dim == 2 && !pr0 && !pr1), "train_fwd: bf16 features are the inference form's, level_dim 2");
    UCN_REQUIRE(aux || (pr0 && pr1), "train_fwd: pr0 and pr1 come together");
    UCN_REQUIRE(!aux || (ray_cols && !h0 && !x && !h1 && !h2), "train_fwd: the in-stream direction tile is an inference form: ray_cols, no stores");
    const bool store = h0 || x || h1 || h2 || m0 || m1 || m2;
    UCN_REQUIRE(!store || (h0 && h1 && h2 && m0 && m1 && m2),
                "train_fwd: the activation / mask outputs come together (all, or none = inference; x alone may be NULL: it is linear in h0)");
    UCN_REQUIRE(feat_level_dim == 0 || (aux && !store && !feat_bf16 && F % feat_level_dim == 0),
                "train_fwd: level-major rays-fastest features are the inference layout (in-stream direction tile, no stores), F a multiple of the level dim");
    UCN_REQUIRE(F >= 1 && F <= 64, "train_fwd: 1..64 input features, got %u", F);
    UCN_REQUIRE(M < 0xFFFFFF00ull, "train_fwd: too many samples");
    UCN_REQUIRE(act_ld == 0 || (act_ld >= 256 && act_ld % 8 == 0), "train_fwd: act_ld = %u (0, or >= 256 and a multiple of 8)", act_ld);
    UCN_REQUIRE(!ray_cols || aux || (ray_dst && act_ld && act_ld % 8 == 0), "train_fwd: ray_cols needs ray_dst and act_ld %% 8 == 0");
    UCN_REQUIRE(!feat_bf16 || F % 8 == 0, "train_fwd: the bf16 feature copy needs F %% 8 == 0, got %u", F);
    TrainFwdArgs a{feat, (const uint4 *)packed, bias_d0, bias_d1, bias_rgb, pr0, pr1, (uint16_t *)h0, (uint16_t *)x, (uint16_t *)h1,
                   (uint16_t *)h2, (const uint16_t *)ray_cols, (uint16_t *)ray_dst, (uint16_t *)feat_bf16, raw, y, act_ld ? act_ld : 64u, act_ld ? act_ld : 256u, act_ld ? act_ld : F, store ? 1 : 0, bf16_in ? 1 : 0, N, feat_level_dim, head != nullptr,
                   head ? head[0] : 0.0f, head ? head[1] : 1.0f, head ? head[2] : 0.0f, head ? head[3] : 0.0f,
                   m0, (uint4 *)m1, (uint4 *)m2, (uint32_t)M, S, F};
    const dim3 grid(ucn_div_up(M, 32 * kTrainWaves));
    const size_t lds = kFtSlots * kTChunk * 1024;
    if (aux) {
        const dim3 igrid(ucn_div_up(M, 32 * kInferWaves)), iblock(64 * kInferWaves);
        const size_t ilds = (size_t)IRing::kSlots * IRing::kChunk * 1024;
        if (F <= 32) hipLaunchKernelGGL((k_train_fwd<1, true>), igrid, iblock, ilds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_train_fwd<2, true>), igrid, iblock, ilds, (hipStream_t)stream, a);
    } else if (UCN_TRAIN_FWD_SIDE && S % 32u == 0u && (((uintptr_t)bias_d0 | (uintptr_t)bias_d1 | (uintptr_t)pr0 | (uintptr_t)pr1) & 15u) == 0u) {
        // a wave's 32 samples are one ray's: biases + the wave's per-ray rows from an LDS side table (see k_train_fwd, SIDE)
        if (F <= 32) hipLaunchKernelGGL((k_train_fwd<1, false, true>), grid, dim3(64 * kTrainWaves), lds + kFwdSideBytes + kStageBytes, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_train_fwd<2, false, true>), grid, dim3(64 * kTrainWaves), lds + kFwdSideBytes + kStageBytes, (hipStream_t)stream, a);
    } else {
        if (F <= 32) hipLaunchKernelGGL((k_train_fwd<1, false>), grid, dim3(64 * kTrainWaves), lds + kStageBytes, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_train_fwd<2, false>), grid, dim3(64 * kTrainWaves), lds + kStageBytes, (hipStream_t)stream, a);
    }
    UCN_LAUNCH_CHECK("train_fwd");
    return 0;
}

extern "C" int ucn_train_bwd(const void *gy, const void *graw, const float *head, const float *density, const float *rgb,
                             const void *packed_t, const uint32_t *m0, const void *m1, const void *m2, uint32_t N, uint32_t S, uint32_t F,
                             void *d1, void *d0, void *gx, void *gh0, void *dy, uint32_t dy_ld, float *gfeat, ucn_stream_t stream) {
    const uint64_t M = (uint64_t)N * S;
    if (M == 0) return 0;
    UCN_REQUIRE(gy && packed_t && m0 && m1 && m2 && d1 && d0 && gh0 && gfeat, "train_bwd: null pointer argument");
    const uint32_t lm = (F & UCN_GFEAT_LEVEL_MAJOR) ? 1u : ((F & UCN_GFEAT_LEVEL_MAJOR4) ? 2u : 0u);
    F &= ~(uint32_t)(UCN_GFEAT_LEVEL_MAJOR | UCN_GFEAT_LEVEL_MAJOR4);
    UCN_REQUIRE(F >= 1 && F <= 64, "train_bwd: 1..64 input features, got %u", F);
    UCN_REQUIRE(!lm || F % 4u == 0u, "train_bwd: the level-major feature gradient needs F %% 4 == 0 (pairs of level_dim 2), got %u", F);
    UCN_REQUIRE(M < 0xFFFFFF00ull, "train_bwd: too many samples");
    UCN_REQUIRE(!head || (density && rgb), "train_bwd: head mode needs the forward's density and rgb");
    TrainBwdArgs a{gy, graw, head != nullptr, density, rgb, head ? head[1] : 1.0f, head ? head[3] : 0.0f, (const uint4 *)packed_t, m0,
                   (const uint4 *)m1, (const uint4 *)m2,
                   (uint16_t *)d1, (uint16_t *)d0, (uint16_t *)gx, (uint16_t *)gh0, (uint16_t *)dy, dy_ld ? dy_ld : 4u, gfeat, (uint32_t)M, F, lm};
    if (F <= 32) hipLaunchKernelGGL(k_train_bwd<1>, dim3(ucn_div_up(M, 32 * kTrainWaves)), dim3(64 * kTrainWaves), kFtSlots * kTChunk * 1024 + kStageBytes, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_train_bwd<2>, dim3(ucn_div_up(M, 32 * kTrainWaves)), dim3(64 * kTrainWaves), kFtSlots * kTChunk * 1024 + kStageBytes, (hipStream_t)stream, a);
    UCN_LAUNCH_CHECK("train_bwd");
    return 0;
}
