// Image error metrics of an evaluation pass on the device (consumers of the rendered frame).
//
// Replaces MetricHarness.__call__ of /root/reference/nerf/internal/image.py:114-133 for its two closed-form metrics (LPIPS
// needs the downloaded VGG weights of the third-party `lpips` package and stays out).  The reference quantises both images
// to uint8, takes PSNR over the colour image (skimage.metrics.peak_signal_noise_ratio, data_range 255) and SSIM over the
// grey images (cv2.cvtColor RGB2GRAY, skimage.metrics.structural_similarity with its defaults: 7 x 7 uniform window, sample
// covariance, K1 = 0.01, K2 = 0.03, mean over the image cropped by 3 pixels) -- on the host, after copying the frame back.
// Both libraries are third-party and absent from this image: the algorithms are restated from their published definitions
// (oracle/metrics.py, same arithmetic in float64) and pinned device == restatement; parity with the reference's numbers is
// otherwise unpinned.
//   quantisation:  pred -> uint8(clip(pred, 0, 1) * 255) (truncation, image.py:119), gt -> uint8(gt * 255)
//   grey (cv2, 8-bit): (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14
//   SSIM window sums are exact integers (49 products of bytes), so every pixel's S is the same float64 expression as the
//   restatement's; block partial sums are added in block order by one thread.
#include "ucn_common.h"

namespace {

__device__ __forceinline__ uint32_t quant_pred(float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f); }
__device__ __forceinline__ uint32_t quant_gt(float v) { return (uint32_t)(uint8_t)(int)(v * 255.0f); }

// pass 1: uint8 grey images + per-block sums of squared colour differences
__global__ __launch_bounds__(256) void k_metrics_quant(const float *__restrict__ pred, const float *__restrict__ gt, uint32_t n,
                                                       uint8_t *__restrict__ gp, uint8_t *__restrict__ gg, double *__restrict__ blk_se) {
    __shared__ double s_part[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double se = 0.0;
    if (i < n) {
        uint32_t p[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            p[c] = quant_pred(pred[(size_t)i * 3 + c]);
            g[c] = quant_gt(gt[(size_t)i * 3 + c]);
            const double d = (double)p[c] - (double)g[c];
            se += d * d;
        }
        gp[i] = (uint8_t)((p[0] * 4899u + p[1] * 9617u + p[2] * 1868u + 8192u) >> 14);
        gg[i] = (uint8_t)((g[0] * 4899u + g[1] * 9617u + g[2] * 1868u + 8192u) >> 14);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) blk_se[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// pass 2: S of every interior pixel (window fully inside), per-block sums
__global__ __launch_bounds__(256) void k_metrics_ssim(const uint8_t *__restrict__ gp, const uint8_t *__restrict__ gg, uint32_t H, uint32_t W,
                                                      double *__restrict__ blk_s) {
    __shared__ double s_part[4];
    const uint32_t wi = W - 6u, hi = H - 6u;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double S = 0.0;
    if (i < wi * hi) {
        const uint32_t y = i / wi + 3u, x = i - (i / wi) * wi + 3u;
        uint32_t sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = -3; dy <= 3; dy++) {
            const uint8_t *rp = gp + (size_t)(y + dy) * W + x - 3, *rg = gg + (size_t)(y + dy) * W + x - 3;
#pragma unroll
            for (int dx = 0; dx < 7; dx++) {
                const uint32_t a = rp[dx], b = rg[dx];
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        }
        const double NP = 49.0, cov_norm = NP / (NP - 1.0);
        const double ux = (double)sx / NP, uy = (double)sy / NP;
        const double vx = cov_norm * ((double)sxx / NP - ux * ux), vy = cov_norm * ((double)syy / NP - uy * uy);
        const double vxy = cov_norm * ((double)sxy / NP - ux * uy);
        const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
        S = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o, 64);
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = S;
    __syncthreads();
    if (threadIdx.x == 0) blk_s[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ void k_metrics_final(const double *__restrict__ blk_se, uint32_t nb_se, const double *__restrict__ blk_s, uint32_t nb_s,
                                double n_values, double n_interior, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double se = 0.0, s = 0.0;
    for (uint32_t i = 0; i < nb_se; i++) se += blk_se[i];
    for (uint32_t i = 0; i < nb_s; i++) s += blk_s[i];
    const double mse = se / n_values;
    out[0] = 10.0 * log10(255.0 * 255.0 / mse);           // +inf for identical images, like skimage
    out[1] = s / n_interior;
    out[2] = mse;
}

}  // namespace

extern "C" uint64_t ucn_image_metrics_ws_bytes(uint32_t H, uint32_t W) {
    const uint64_t n = (uint64_t)H * W, nb = (n + 255) / 256;
    return 2 * n + 2 * nb * sizeof(double) + 64;
}

extern "C" int ucn_image_metrics(const float *pred, const float *gt, uint32_t H, uint32_t W, void *workspace,
                                 double *out /*DEVICE [3]: psnr, ssim, mse (uint8 scale)*/, ucn_stream_t stream) {
    UCN_REQUIRE(pred && gt && workspace && out, "image_metrics: null pointer argument");
    UCN_REQUIRE(H >= 7 && W >= 7, "image_metrics: the SSIM window needs at least 7 x 7 pixels, got %u x %u", H, W);
    const uint64_t n = (uint64_t)H * W;
    UCN_REQUIRE(n < 0x7FFFFF00ull, "image_metrics: image too large");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nb = (uint32_t)((n + 255) / 256), n_in = (H - 6u) * (W - 6u), nb_in = (n_in + 255u) / 256u;
    uint8_t *gp = reinterpret_cast<uint8_t *>(workspace), *gg = gp + n;
    double *blk_se = reinterpret_cast<double *>(reinterpret_cast<uintptr_t>(gg + n + 63) & ~(uintptr_t)63), *blk_s = blk_se + nb;
    hipLaunchKernelGGL(k_metrics_quant, dim3(nb), dim3(256), 0, st, pred, gt, (uint32_t)n, gp, gg, blk_se);
    hipLaunchKernelGGL(k_metrics_ssim, dim3(nb_in), dim3(256), 0, st, gp, gg, H, W, blk_s);
    hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(64), 0, st, blk_se, nb, blk_s, nb_in, (double)n * 3.0, (double)n_in, out);
    UCN_LAUNCH_CHECK("image_metrics");
    return 0;
}
