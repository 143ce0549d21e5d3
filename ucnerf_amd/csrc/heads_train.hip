// The tail of the training step with the colour-correction head and the sky layer on (scripts/train_waymo.sh:11-12): per-ray affine
// colour correction + sky blend and the three ray-wise losses around it, forward and backward.  As eager torch ops this tail was ~280
// launches of 2-8 us on 8192-row tensors between the field's forward and backward kernels (~2 ms of an 18 ms step); here each piece is
// one launch forward + one backward.  References: models.py:339-363 (affine + sky blend), train_utils.py:171-230 (data loss),
// train_utils.py:149-157 (sky loss), train_utils.py:159-169 (identity loss).
#include "ucn_common.h"

namespace {

// rgb' = A rgb + t  (+ opac (A_sky sky + t_sky)), opac = 1 - acc_last; A, A_sky per RAY, row-major [3, 4] = [A | t]
__global__ __launch_bounds__(256) void k_affine_blend_fwd(const float *__restrict__ rgb, const float *__restrict__ aff,
                                                          const float *__restrict__ acc_last, const float *__restrict__ sky,
                                                          const float *__restrict__ aff_sky, uint32_t N, float *__restrict__ out) {
    const uint32_t ray = blockIdx.x * 256u + threadIdx.x;
    if (ray >= N) return;
    const float *A = aff + (size_t)ray * 12;
    const float r = rgb[ray * 3 + 0], g = rgb[ray * 3 + 1], b = rgb[ray * 3 + 2];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = ((A[c * 4 + 0] * r + A[c * 4 + 1] * g) + A[c * 4 + 2] * b) + A[c * 4 + 3];
    if (sky) {
        const float opac = 1.0f - acc_last[ray];
        const float *B = aff_sky + (size_t)ray * 12;
        const float sr = sky[ray * 3 + 0], sg = sky[ray * 3 + 1], sb = sky[ray * 3 + 2];
#pragma unroll
        for (int c = 0; c < 3; c++) o[c] += opac * (((B[c * 4 + 0] * sr + B[c * 4 + 1] * sg) + B[c * 4 + 2] * sb) + B[c * 4 + 3]);
    }
    out[ray * 3 + 0] = o[0]; out[ray * 3 + 1] = o[1]; out[ray * 3 + 2] = o[2];
}

// gradients of the above w.r.t. everything; the optional outputs ACCUMULATE (the same A / sky / acc serve every level's call)
__global__ __launch_bounds__(256) void k_affine_blend_bwd(const float *__restrict__ g_out, const float *__restrict__ rgb,
                                                          const float *__restrict__ aff, const float *__restrict__ acc_last,
                                                          const float *__restrict__ sky, const float *__restrict__ aff_sky, uint32_t N,
                                                          int accumulate, float *__restrict__ g_rgb, float *__restrict__ g_aff,
                                                          float *__restrict__ g_acc, float *__restrict__ g_sky, float *__restrict__ g_aff_sky) {
    const uint32_t ray = blockIdx.x * 256u + threadIdx.x;
    if (ray >= N) return;
    const float *A = aff + (size_t)ray * 12;
    const float go[3] = {g_out[ray * 3 + 0], g_out[ray * 3 + 1], g_out[ray * 3 + 2]};
    const float v[3] = {rgb[ray * 3 + 0], rgb[ray * 3 + 1], rgb[ray * 3 + 2]};
#pragma unroll
    for (int k = 0; k < 3; k++) g_rgb[ray * 3 + k] = (go[0] * A[0 * 4 + k] + go[1] * A[1 * 4 + k]) + go[2] * A[2 * 4 + k];
    float *gA = g_aff + (size_t)ray * 12;
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float t = go[c] * (k < 3 ? v[k] : 1.0f);
            gA[c * 4 + k] = accumulate ? gA[c * 4 + k] + t : t;
        }
    }
    if (sky) {
        const float opac = 1.0f - acc_last[ray];
        const float *B = aff_sky + (size_t)ray * 12;
        const float s[3] = {sky[ray * 3 + 0], sky[ray * 3 + 1], sky[ray * 3 + 2]};
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) dot += go[c] * (((B[c * 4 + 0] * s[0] + B[c * 4 + 1] * s[1]) + B[c * 4 + 2] * s[2]) + B[c * 4 + 3]);
        g_acc[ray] = accumulate ? g_acc[ray] - dot : -dot;                       // d opac / d acc = -1
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float t = opac * ((go[0] * B[0 * 4 + k] + go[1] * B[1 * 4 + k]) + go[2] * B[2 * 4 + k]);
            g_sky[ray * 3 + k] = accumulate ? g_sky[ray * 3 + k] + t : t;
        }
        float *gB = g_aff_sky + (size_t)ray * 12;
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float t = opac * go[c] * (k < 3 ? s[k] : 1.0f);
                gB[c * 4 + k] = accumulate ? gB[c * 4 + k] + t : t;
            }
        }
    }
}

// deterministic sum of one value per thread over a 1024-thread workgroup (fixed tree); the result is valid in thread 0
template <class T>
__device__ __forceinline__ T wg_sum_1024(T v, T *scratch /*[1024]*/) {
    scratch[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t s = 512u; s > 0u; s >>= 1) {
        if (threadIdx.x < s) scratch[threadIdx.x] += scratch[threadIdx.x + s];
        __syncthreads();
    }
    const T r = scratch[0];
    __syncthreads();
    return r;
}

constexpr uint32_t kMaxLossLevels = 4;
struct LevelPtrs {
    const float *p[kMaxLossLevels];
};
struct LevelOut {
    float *p[kMaxLossLevels];
};
struct LevelWeights {                                   // loss = sum_l  w_mse[l] mse_l + w_charb[l] charb_l
    float w_mse[kMaxLossLevels], w_charb[kMaxLossLevels];
};

// train_utils.py:171-230: per level l  mse_l = sum(m r^2) / sum(m),  charb_l = sum(m sqrt(r^2 + pad^2)) / sum(m),  r = rgb_l - target,
// m = lossmult (per ray, broadcast over the channels; NULL = 1).  ONE workgroup (the batch is 8192 rays): fixed-order sums.
// out: [L][2] = {mse, charb}, the denominator at [2 L], the weighted loss at [2 L + 1].
__global__ __launch_bounds__(1024) void k_data_loss_fwd(LevelPtrs rgb, uint32_t L, LevelWeights w, const float *__restrict__ target,
                                                        const float *__restrict__ mult, uint32_t N, float pad2, float *__restrict__ out) {
    __shared__ float scratch[1024];
    float den = 0.0f, loss = 0.0f;
    for (uint32_t i = threadIdx.x; i < N * 3u; i += 1024u) den += mult ? mult[i / 3u] : 1.0f;
    den = wg_sum_1024(den, scratch);
    for (uint32_t l = 0; l < L; l++) {
        float a = 0.0f, c = 0.0f;
        for (uint32_t i = threadIdx.x; i < N * 3u; i += 1024u) {
            const float m = mult ? mult[i / 3u] : 1.0f, r = rgb.p[l][i] - target[i], r2 = r * r;
            a += m * r2;
            c += m * sqrtf(r2 + pad2);
        }
        a = wg_sum_1024(a, scratch);
        c = wg_sum_1024(c, scratch);
        if (threadIdx.x == 0) { out[2 * l] = a / den; out[2 * l + 1] = c / den; }
        loss += w.w_mse[l] * (a / den) + w.w_charb[l] * (c / den);
    }
    if (threadIdx.x == 0) { out[2 * L] = den; out[2 * L + 1] = loss; }
}
// g: [1] = d / d loss
__global__ __launch_bounds__(256) void k_data_loss_bwd(LevelPtrs rgb, LevelOut g_rgb, uint32_t L, LevelWeights w, const float *__restrict__ target,
                                                       const float *__restrict__ mult, uint32_t N, float pad2,
                                                       const float *__restrict__ fwd_out, const float *__restrict__ g) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N * 3u) return;
    const float m = (mult ? mult[i / 3u] : 1.0f) / fwd_out[2 * L];
    for (uint32_t l = 0; l < L; l++) {
        const float r = rgb.p[l][i] - target[i], r2 = r * r;
        g_rgb.p[l][i] = g[0] * m * (w.w_mse[l] * 2.0f * r + w.w_charb[l] * r / sqrtf(r2 + pad2));
    }
}

// train_utils.py:149-157: sum over the levels of mean BCE(clip(acc_l, 1e-3, 1 - 1e-3), 1 - sky_seg)
__global__ __launch_bounds__(1024) void k_sky_loss_fwd(LevelPtrs acc, uint32_t L, const float *__restrict__ sky_segs, uint32_t N,
                                                       float *__restrict__ out) {
    __shared__ float scratch[1024];
    float tot = 0.0f;
    for (uint32_t l = 0; l < L; l++) {
        float s = 0.0f;
        for (uint32_t i = threadIdx.x; i < N; i += 1024u) {
            const float a = fminf(fmaxf(acc.p[l][i], 1e-3f), 0.999f), t = 1.0f - sky_segs[i];
            s -= t * logf(a) + (1.0f - t) * logf(1.0f - a);
        }
        s = wg_sum_1024(s, scratch);
        tot += s / (float)N;
    }
    if (threadIdx.x == 0) out[0] = tot;
}
__global__ __launch_bounds__(256) void k_sky_loss_bwd(LevelPtrs acc, LevelOut g_acc, uint32_t L, const float *__restrict__ sky_segs,
                                                      uint32_t N, const float *__restrict__ g) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const float t = 1.0f - sky_segs[i], go = g[0] / (float)N;
    for (uint32_t l = 0; l < L; l++) {
        const float raw = acc.p[l][i];
        const bool inside = raw >= 1e-3f && raw <= 0.999f;              // clip's gradient: 1 inside (torch: bounds included), 0 outside
        const float a = fminf(fmaxf(raw, 1e-3f), 0.999f);
        g_acc.p[l][i] = inside ? go * (a - t) / (a * (1.0f - a)) : 0.0f;
    }
}

// train_utils.py:159-169: mean over [N, 3, 4] of |eye - A| (+ |eye - A_sky|), float64 like the reference's eye
__global__ __launch_bounds__(1024) void k_identity_loss_fwd(const float *__restrict__ A, const float *__restrict__ B, uint32_t N,
                                                            double *__restrict__ out) {
    __shared__ double scratch[1024];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < N * 12u; i += 1024u) {
        const uint32_t e = i % 12u;
        const double eye = (e == 0u || e == 5u || e == 10u) ? 1.0 : 0.0;
        s += fabs(eye - (double)A[i]) + (B ? fabs(eye - (double)B[i]) : 0.0);
    }
    s = wg_sum_1024(s, scratch);
    if (threadIdx.x == 0) out[0] = s / ((double)N * 12.0);
}
__global__ __launch_bounds__(256) void k_identity_loss_bwd(const float *__restrict__ A, const float *__restrict__ B, uint32_t N,
                                                           const double *__restrict__ g, float *__restrict__ gA, float *__restrict__ gB) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N * 12u) return;
    const uint32_t e = i % 12u;
    const double eye = (e == 0u || e == 5u || e == 10u) ? 1.0 : 0.0, go = g[0] / ((double)N * 12.0);
    const double da = eye - (double)A[i];
    gA[i] = (float)(da > 0.0 ? -go : da < 0.0 ? go : 0.0);                    // d|eye - a| / da = -sign(eye - a)
    if (B) {
        const double db = eye - (double)B[i];
        gB[i] = (float)(db > 0.0 ? -go : db < 0.0 ? go : 0.0);
    }
}

}  // namespace

extern "C" int ucn_affine_blend(const float *g_out, const float *rgb, const float *affine, const float *acc_last, const float *sky_rgb,
                                const float *affine_sky, uint32_t N, int accumulate, float *out_or_g_rgb, float *g_affine, float *g_acc,
                                float *g_sky, float *g_affine_sky, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (rgb && affine && out_or_g_rgb), "affine_blend: null pointer argument");
    UCN_REQUIRE(!sky_rgb || (acc_last && affine_sky), "affine_blend: the sky blend needs acc and the sky affine");
    UCN_REQUIRE(!g_out || (g_affine && (!sky_rgb || (g_acc && g_sky && g_affine_sky))), "affine_blend: backward needs every gradient buffer");
    if (N == 0) return 0;
    if (!g_out)
        hipLaunchKernelGGL(k_affine_blend_fwd, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, rgb, affine, acc_last, sky_rgb,
                           affine_sky, N, out_or_g_rgb);
    else
        hipLaunchKernelGGL(k_affine_blend_bwd, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, g_out, rgb, affine, acc_last,
                           sky_rgb, affine_sky, N, accumulate, out_or_g_rgb, g_affine, g_acc, g_sky, g_affine_sky);
    UCN_LAUNCH_CHECK("affine_blend");
    return 0;
}

extern "C" int ucn_data_loss(const float *const *rgb_levels_host, uint32_t L, const float *w_mse_host, const float *w_charb_host,
                             const float *target, const float *lossmult, uint32_t N, float charb_padding, float *fwd_out /*[2 L + 2]*/,
                             const float *g /*[1] or NULL*/, float *const *g_rgb_levels_host, ucn_stream_t stream) {
    UCN_REQUIRE(L >= 1 && L <= kMaxLossLevels, "data_loss: 1 to %u levels", kMaxLossLevels);
    UCN_REQUIRE(rgb_levels_host && w_mse_host && w_charb_host && target && fwd_out && ((g == nullptr) == (g_rgb_levels_host == nullptr)),
                "data_loss: null pointer argument");
    if (N == 0) return 0;
    LevelPtrs in;
    LevelOut go;
    LevelWeights w;
    for (uint32_t l = 0; l < kMaxLossLevels; l++) {
        in.p[l] = l < L ? rgb_levels_host[l] : nullptr;
        go.p[l] = (g && l < L) ? g_rgb_levels_host[l] : nullptr;
        w.w_mse[l] = l < L ? w_mse_host[l] : 0.0f;
        w.w_charb[l] = l < L ? w_charb_host[l] : 0.0f;
    }
    const float pad2 = charb_padding * charb_padding;
    if (!g) hipLaunchKernelGGL(k_data_loss_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, L, w, target, lossmult, N, pad2, fwd_out);
    else hipLaunchKernelGGL(k_data_loss_bwd, dim3(ucn_div_up((uint64_t)N * 3u, 256)), dim3(256), 0, (hipStream_t)stream, in, go, L, w, target,
                            lossmult, N, pad2, fwd_out, g);
    UCN_LAUNCH_CHECK("data_loss");
    return 0;
}

extern "C" int ucn_sky_loss(const float *const *acc_levels_host, uint32_t L, const float *sky_segs, uint32_t N, float *loss_out,
                            const float *g /*[1] or NULL*/, float *const *g_acc_levels_host, ucn_stream_t stream) {
    UCN_REQUIRE(L >= 1 && L <= kMaxLossLevels, "sky_loss: 1 to %u levels", kMaxLossLevels);
    UCN_REQUIRE(acc_levels_host && sky_segs && ((g == nullptr) == (g_acc_levels_host == nullptr)) && (g || loss_out), "sky_loss: null pointer argument");
    if (N == 0) return 0;
    LevelPtrs in;
    LevelOut go;
    for (uint32_t l = 0; l < kMaxLossLevels; l++) {
        in.p[l] = l < L ? acc_levels_host[l] : nullptr;
        go.p[l] = (g && l < L) ? g_acc_levels_host[l] : nullptr;
    }
    if (!g) hipLaunchKernelGGL(k_sky_loss_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, L, sky_segs, N, loss_out);
    else hipLaunchKernelGGL(k_sky_loss_bwd, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, in, go, L, sky_segs, N, g);
    UCN_LAUNCH_CHECK("sky_loss");
    return 0;
}

extern "C" int ucn_identity_loss(const float *affine, const float *affine_sky, uint32_t N, double *loss_out, const double *g /*[1] or NULL*/,
                                 float *g_affine, float *g_affine_sky, ucn_stream_t stream) {
    UCN_REQUIRE(affine && (g || loss_out) && (!g || (g_affine && (!affine_sky || g_affine_sky))), "identity_loss: null pointer argument");
    if (N == 0) return 0;
    if (!g) hipLaunchKernelGGL(k_identity_loss_fwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, affine, affine_sky, N, loss_out);
    else hipLaunchKernelGGL(k_identity_loss_bwd, dim3(ucn_div_up((uint64_t)N * 12u, 256)), dim3(256), 0, (hipStream_t)stream, affine, affine_sky,
                            N, g, g_affine, g_affine_sky);
    UCN_LAUNCH_CHECK("identity_loss");
    return 0;
}
