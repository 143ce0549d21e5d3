// Training-time forward and backward of the PROPOSAL field's dense part:
// models.py:507-516 with disable_rgb = True -- Linear(F, 64) + ReLU, Linear(64, 1), softplus(raw + density_bias) -- on
// the F = levels x channels features of the proposal grid (12 in BASELINE's configs, 24 on the reference's own waymo.gin grid: r06).
//
// As library ops this tiny MLP is ~45 launches per step (two GEMMs whose outputs are 64 and 1 columns wide, casts,
// bias / ReLU / softplus passes, and in the backward a one-row weight-gradient GEMM that the library serves through a
// slow host path) -- 0.7 ms for 0.9 GFLOP.  There is no matrix-core shape in it worth the operand shuffles: the kernels
// below are plain VALU code, laid out so that no cross-lane reduction is needed anywhere:
//
//   k_prop_fwd        thread = sample.  Weights broadcast from LDS; the hidden activations never leave registers.
//   k_prop_bwd_feat   thread = sample.  Recomputes the hidden layer (cheaper than storing [M, 64]) and forms
//                     d feat = W0^T (g_raw w1 * relu'), [M, F].
//   k_prop_bwd_weight lane = hidden unit, a wave walks a slab of samples whose rows it reads as LDS broadcasts: every
//                     lane accumulates ITS row of d W0, d b0, d w1 in registers.  Per-workgroup partial sums go to a
//                     workspace, k_prop_reduce adds them in a fixed order (deterministic).
//
// `round_bf16` reproduces what the reference's accelerator.autocast() does to these layers: operands and layer outputs
// rounded to bf16, fp32 accumulation, softplus in fp32.  0 = plain fp32 (the G10 parity path).
#include <hip/hip_runtime.h>

#include "../../include/ucnerf_march.h"
#include "ucn_common.h"

namespace {

constexpr int kHidden = 64;

__device__ __forceinline__ float bf16r(float v, int on) {
    if (!on) return v;
    return (float)(__bf16)v;                       // v_cvt_pk_bf16_f32: round to nearest even
}

struct PropArgs {
    const float *feat;        // [M, F]
    const float *W0, *b0;     // [64, F], [64]
    const float *w1, *b1;     // [64], [1]
    float density_bias;
    int round_bf16;
    uint32_t M, F;
    float *density;           // [M]            (fwd: out, bwd: in)
    const float *g_density;   // [M]
    float *gfeat;             // [M, F]
    float *partial;           // [n_wg][64 * (FP + 2) + 1]
    uint32_t slab;            // samples per workgroup of k_prop_bwd_weight
    uint32_t n_rays, level_dim;   // level_dim != 0 (forward / inference only): feat is the rendering gather's [L][B][C] with
    //                              b = s * n_rays + ray; thread m = b, the density goes to [ray][sample]
};

// weights into LDS, rounded like the GEMM operands: W0 padded to FP columns
template <int FP>
__device__ __forceinline__ void stage_weights(const PropArgs &a, float *s_w0, float *s_b0, float *s_w1, float &b1) {
    for (uint32_t i = threadIdx.x; i < kHidden * FP; i += blockDim.x) {
        const uint32_t k = i / FP, c = i % FP;
        s_w0[i] = c < a.F ? bf16r(a.W0[k * a.F + c], a.round_bf16) : 0.0f;
    }
    for (uint32_t i = threadIdx.x; i < kHidden; i += blockDim.x) {
        s_b0[i] = bf16r(a.b0[i], a.round_bf16);
        s_w1[i] = bf16r(a.w1[i], a.round_bf16);
    }
    b1 = bf16r(a.b1[0], a.round_bf16);
    __syncthreads();
}

template <int FP>
__device__ __forceinline__ void load_features(const PropArgs &a, uint32_t m, float (&f)[FP]) {
#pragma unroll
    for (int c = 0; c < FP; c++) {
        size_t at = (size_t)m * a.F + c;
        if (a.level_dim) at = ((size_t)((uint32_t)c / a.level_dim) * a.M + m) * a.level_dim + (uint32_t)c % a.level_dim;
        f[c] = (uint32_t)c < a.F ? bf16r(a.feat[at], a.round_bf16) : 0.0f;
    }
}

template <int FP>
__device__ __forceinline__ float hidden_pre(const float *s_w0, const float *s_b0, int k, const float (&f)[FP]) {
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < FP; c += 4) {
        const float4 w = *reinterpret_cast<const float4 *>(s_w0 + k * FP + c);      // same address in every lane: broadcast
        acc = fmaf(w.x, f[c], acc); acc = fmaf(w.y, f[c + 1], acc); acc = fmaf(w.z, f[c + 2], acc); acc = fmaf(w.w, f[c + 3], acc);
    }
    return acc + s_b0[k];
}

template <int FP>
__global__ __launch_bounds__(256) void k_prop_fwd(PropArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w0[kHidden * FP];
    __shared__ float s_b0[kHidden], s_w1[kHidden];
    float b1;
    stage_weights<FP>(a, s_w0, s_b0, s_w1, b1);
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= a.M) return;
    float f[FP];
    load_features<FP>(a, m, f);
    float raw = 0.0f;
#pragma unroll 8
    for (int k = 0; k < kHidden; k++) {
        const float h = fmaxf(bf16r(hidden_pre<FP>(s_w0, s_b0, k, f), a.round_bf16), 0.0f);
        raw = fmaf(s_w1[k], h, raw);
    }
    const float z = bf16r(raw + b1, a.round_bf16) + a.density_bias;
    const uint32_t spr = a.M / (a.n_rays ? a.n_rays : 1u);
    const uint32_t o = a.level_dim ? (m % a.n_rays) * spr + m / a.n_rays : m;
    a.density[o] = z > 20.0f ? z : log1pf(__expf(z));                                 // F.softplus, beta 1, threshold 20
}

// d softplus(z) / dz = sigmoid(z) = -expm1(-softplus(z)) (expm1: 1 - exp cancels for small densities)
__device__ __forceinline__ float g_raw_of(const PropArgs &a, uint32_t m) {
    return bf16r(a.g_density[m] * (-expm1f(-a.density[m])), a.round_bf16);
}

template <int FP>
__global__ __launch_bounds__(256) void k_prop_bwd_feat(PropArgs a) {
    __shared__ __attribute__((aligned(16))) float s_w0[kHidden * FP];
    __shared__ float s_b0[kHidden], s_w1[kHidden];
    float b1;
    stage_weights<FP>(a, s_w0, s_b0, s_w1, b1);
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= a.M) return;
    float f[FP], gf[FP];
    load_features<FP>(a, m, f);
#pragma unroll
    for (int c = 0; c < FP; c++) gf[c] = 0.0f;
    const float g = g_raw_of(a, m);
#pragma unroll 4
    for (int k = 0; k < kHidden; k++) {
        const float pre = bf16r(hidden_pre<FP>(s_w0, s_b0, k, f), a.round_bf16);
        const float gh = pre > 0.0f ? bf16r(g * s_w1[k], a.round_bf16) : 0.0f;
#pragma unroll
        for (int c = 0; c < FP; c += 4) {
            const float4 w = *reinterpret_cast<const float4 *>(s_w0 + k * FP + c);
            gf[c] = fmaf(w.x, gh, gf[c]); gf[c + 1] = fmaf(w.y, gh, gf[c + 1]);
            gf[c + 2] = fmaf(w.z, gh, gf[c + 2]); gf[c + 3] = fmaf(w.w, gh, gf[c + 3]);
        }
    }
#pragma unroll
    for (int c = 0; c < FP; c++)
        if ((uint32_t)c < a.F) a.gfeat[(size_t)m * a.F + c] = bf16r(gf[c], a.round_bf16);
}

// lane = hidden unit k.  A wave walks its slab 64 samples at a time: every lane fetches ONE sample's row (features, g_raw)
// with ordinary vector loads and parks it in the wave's LDS tile; the 64 rows are then read back as broadcasts (every
// lane the same address), so each lane sees every sample without a cross-lane operation.  (A first version read the rows
// through scalar loads: one exposed s_load latency per sample, 198 us per call.)
template <int FP>
__global__ __launch_bounds__(256) void k_prop_bwd_weight(PropArgs a) {
    constexpr int RS = FP + 4;                                // LDS row: FP features, g_raw, 3 pad (16-byte rows)
    __shared__ __attribute__((aligned(16))) float s_rows[4][64 * RS];
    const int k = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float w0[FP], gw0[FP];
#pragma unroll
    for (int c = 0; c < FP; c++) {
        w0[c] = (uint32_t)c < a.F ? bf16r(a.W0[k * a.F + c], a.round_bf16) : 0.0f;
        gw0[c] = 0.0f;
    }
    const float b0 = bf16r(a.b0[k], a.round_bf16), w1 = bf16r(a.w1[k], a.round_bf16);
    float gb0 = 0.0f, gw1 = 0.0f, gb1 = 0.0f;
    const uint32_t per_wave = a.slab / 4u;
    const uint32_t lo = blockIdx.x * a.slab + wave * per_wave;
    const uint32_t hi = lo + per_wave < a.M ? lo + per_wave : a.M;
    float *rows = s_rows[wave];
    for (uint32_t m0 = lo; m0 < hi; m0 += 64u) {
        const uint32_t m = m0 + (uint32_t)k;
        float f[FP];
        float g = 0.0f;
        if (m < hi) {
            load_features<FP>(a, m, f);
            g = g_raw_of(a, m);
        } else {
#pragma unroll
            for (int c = 0; c < FP; c++) f[c] = 0.0f;         // g = 0: the padded samples add nothing
        }
#pragma unroll
        for (int c = 0; c < FP; c += 4) *reinterpret_cast<float4 *>(rows + k * RS + c) = make_float4(f[c], f[c + 1], f[c + 2], f[c + 3]);
        rows[k * RS + FP] = g;
        __builtin_amdgcn_wave_barrier();                      // the tile is private to the wave; LDS ops of a wave are ordered
#pragma unroll 4
        for (int j = 0; j < 64; j++) {
            float fj[FP];
#pragma unroll
            for (int c = 0; c < FP; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(rows + j * RS + c);
                fj[c] = v.x; fj[c + 1] = v.y; fj[c + 2] = v.z; fj[c + 3] = v.w;
            }
            const float gj = rows[j * RS + FP];
            float pre = 0.0f;
#pragma unroll
            for (int c = 0; c < FP; c++) pre = fmaf(w0[c], fj[c], pre);
            pre = bf16r(pre + b0, a.round_bf16);
            const float h = fmaxf(pre, 0.0f);
            const float gh = pre > 0.0f ? bf16r(gj * w1, a.round_bf16) : 0.0f;
#pragma unroll
            for (int c = 0; c < FP; c++) gw0[c] = fmaf(gh, fj[c], gw0[c]);
            gb0 += gh;
            gw1 = fmaf(gj, h, gw1);
            gb1 += gj;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the workgroup's four waves -> one partial row: [64][FP] dW0, [64] db0, [64] dw1, [1] db1
    constexpr int ROW = kHidden * (FP + 2) + 1;
    __shared__ float s_p[4][ROW];
#pragma unroll
    for (int c = 0; c < FP; c++) s_p[wave][k * FP + c] = gw0[c];
    s_p[wave][kHidden * FP + k] = gb0;
    s_p[wave][kHidden * (FP + 1) + k] = gw1;
    if (k == 0) s_p[wave][kHidden * (FP + 2)] = gb1;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)ROW; i += 256u)
        a.partial[(size_t)blockIdx.x * ROW + i] = (s_p[0][i] + s_p[1][i]) + (s_p[2][i] + s_p[3][i]);
}

// out[i] = sum over workgroups of partial[wg][i] in a fixed order: 64 columns x 16 interleaved groups of workgroups per
// block, the groups added in order through LDS.  Output layout = the partial row, FP columns squeezed to F.
template <int FP>
__global__ __launch_bounds__(1024) void k_prop_reduce(const float *__restrict__ partial, uint32_t n_wg, uint32_t F, float *gW0, float *gb0,
                                                       float *gw1, float *gb1) {
    constexpr int ROW = kHidden * (FP + 2) + 1;
    __shared__ float s_g[16][64];
    const uint32_t col = threadIdx.x & 63u, grp = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * 64u + col;
    float s = 0.0f;
    if (i < (uint32_t)ROW)
        for (uint32_t w = grp; w < n_wg; w += 16u) s += partial[(size_t)w * ROW + i];
    s_g[grp][col] = s;
    __syncthreads();
    if (grp != 0 || i >= (uint32_t)ROW) return;
#pragma unroll
    for (int g = 1; g < 16; g++) s += s_g[g][col];
    if (i < (uint32_t)(kHidden * FP)) {
        const uint32_t k = i / FP, c = i % FP;
        if (c < F) gW0[k * F + c] = s;
    } else if (i < (uint32_t)(kHidden * (FP + 1))) gb0[i - kHidden * FP] = s;
    else if (i < (uint32_t)(kHidden * (FP + 2))) gw1[i - kHidden * (FP + 1)] = s;
    else gb1[0] = s;
}

constexpr uint32_t kSlab = 1024;     // samples per workgroup of the weight pass (256 per wave)

int check_shapes(uint32_t F, uint32_t hidden, uint64_t M) {
    UCN_REQUIRE(hidden == (uint32_t)kHidden, "prop_train: hidden width %u (this kernel: 64)", hidden);
    UCN_REQUIRE(F >= 1 && F <= 24, "prop_train: 1..24 input features, got %u", F);
    UCN_REQUIRE(M < 0xFFFF0000ull, "prop_train: too many samples");
    return 0;
}

}  // namespace

#define UCN_PROP_DISPATCH(FP_, ...)                 \
    switch (FP_) {                                  \
        case 4: { constexpr int FPC = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int FPC = 8; __VA_ARGS__; } break;   \
        case 12: { constexpr int FPC = 12; __VA_ARGS__; } break; \
        case 16: { constexpr int FPC = 16; __VA_ARGS__; } break; \
        case 20: { constexpr int FPC = 20; __VA_ARGS__; } break; \
        default: { constexpr int FPC = 24; __VA_ARGS__; } break; \
    }

extern "C" int ucn_prop_train_fwd(const float *feat, uint32_t F, uint32_t hidden, const float *W0, const float *b0, const float *w1,
                                  const float *b1, float density_bias, int round_bf16, uint64_t M, float *density,
                                  uint32_t n_rays, uint32_t feat_level_dim, ucn_stream_t stream) {
    if (int rc = check_shapes(F, hidden, M)) return rc;
    if (M == 0) return 0;
    UCN_REQUIRE(feat && W0 && b0 && w1 && b1 && density, "prop_train_fwd: null pointer argument");
    UCN_REQUIRE(feat_level_dim == 0 || (n_rays && M % n_rays == 0 && F % feat_level_dim == 0),
                "prop_train_fwd: level-major rays-fastest features need n_rays | M and level_dim | F");
    PropArgs a{feat, W0, b0, w1, b1, density_bias, round_bf16, (uint32_t)M, F, density, nullptr, nullptr, nullptr, 0, n_rays, feat_level_dim};
    const uint32_t fp = (F + 3u) & ~3u;
    UCN_PROP_DISPATCH(fp, hipLaunchKernelGGL(k_prop_fwd<FPC>, dim3(ucn_div_up(M, 256)), dim3(256), 0, (hipStream_t)stream, a));
    UCN_LAUNCH_CHECK("prop_train_fwd");
    return 0;
}

extern "C" uint64_t ucn_prop_train_bwd_ws_floats(uint32_t F, uint64_t M) {
    const uint32_t fp = (F + 3u) & ~3u;
    return (uint64_t)ucn_div_up(M, kSlab) * (kHidden * (fp + 2) + 1);
}

extern "C" int ucn_prop_train_bwd(const float *feat, uint32_t F, uint32_t hidden, const float *W0, const float *b0, const float *w1,
                                  const float *b1, float density_bias, int round_bf16, uint64_t M, const float *density,
                                  const float *g_density, float *gfeat /*[M,F] | NULL*/, float *gW0, float *gb0, float *gw1, float *gb1,
                                  float *workspace, ucn_stream_t stream) {
    if (int rc = check_shapes(F, hidden, M)) return rc;
    UCN_REQUIRE(gW0 && gb0 && gw1 && gb1, "prop_train_bwd: null pointer argument");
    if (M == 0) {
        (void)hipMemsetAsync(gW0, 0, sizeof(float) * kHidden * F, (hipStream_t)stream);
        (void)hipMemsetAsync(gb0, 0, sizeof(float) * kHidden, (hipStream_t)stream);
        (void)hipMemsetAsync(gw1, 0, sizeof(float) * kHidden, (hipStream_t)stream);
        (void)hipMemsetAsync(gb1, 0, sizeof(float), (hipStream_t)stream);
        return 0;
    }
    UCN_REQUIRE(feat && W0 && b0 && w1 && b1 && density && g_density && workspace, "prop_train_bwd: null pointer argument");
    PropArgs a{feat, W0, b0, w1, b1, density_bias, round_bf16, (uint32_t)M, F, const_cast<float *>(density), g_density, gfeat, workspace, kSlab, 0, 0};
    const uint32_t fp = (F + 3u) & ~3u, n_wg = (uint32_t)ucn_div_up(M, kSlab);
    if (gfeat) {
        UCN_PROP_DISPATCH(fp, hipLaunchKernelGGL(k_prop_bwd_feat<FPC>, dim3(ucn_div_up(M, 256)), dim3(256), 0, (hipStream_t)stream, a));
        UCN_LAUNCH_CHECK("prop_train_bwd (features)");
    }
    UCN_PROP_DISPATCH(fp, hipLaunchKernelGGL(k_prop_bwd_weight<FPC>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, a));
    UCN_LAUNCH_CHECK("prop_train_bwd (weights)");
    UCN_PROP_DISPATCH(fp, hipLaunchKernelGGL(k_prop_reduce<FPC>, dim3(ucn_div_up(kHidden * (FPC + 2) + 1, 64)), dim3(1024), 0,
                                             (hipStream_t)stream, workspace, n_wg, F, gW0, gb0, gw1, gb1));
    UCN_LAUNCH_CHECK("prop_train_bwd (reduce)");
    return 0;
}
