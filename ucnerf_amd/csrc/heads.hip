// Small heads around the ray-march: generic dense layer (brightness MLP), per-ray affine colour
// correction + sky blend.  ref: extrinsic_optimizer.py:4-48, models.py:339-363.
#include "ucn_common.h"

namespace {

// y[m][n] = act(b[n] + sum_k x[m][k] w[n][k]); one thread per output, k-sequential fp32 fmaf.
// Used for M <= a few hundred rows (one row per camera latent), so no tiling.
__global__ __launch_bounds__(256) void k_dense(const float *__restrict__ x, const float *__restrict__ w,
                                               const float *__restrict__ b, uint32_t M, uint32_t K, uint32_t Nout,
                                               int relu, float *__restrict__ y) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= M * Nout) return;
    const uint32_t m = i / Nout, n = i - m * Nout;
    float s = 0.0f;
    for (uint32_t k = 0; k < K; k++) s = fmaf(x[(size_t)m * K + k], w[(size_t)n * K + k], s);
    s += b ? b[n] : 0.0f;
    y[i] = relu ? fmaxf(s, 0.0f) : s;
}

// rgb' = A rgb + t  (+ (1 - sum_s w_last) * (A_sky sky + t_sky)); A row-major [3,4] = [A | t]
__global__ __launch_bounds__(256) void k_apply_affine(const float *__restrict__ rgb_in, const float *__restrict__ aff,
                                                      const int64_t *__restrict__ row_of, const float *__restrict__ w_last,
                                                      uint32_t S, const float *__restrict__ sky, const float *__restrict__ aff_sky,
                                                      uint32_t N, float *__restrict__ rgb_out) {
    const uint32_t ray = blockIdx.x * 256u + threadIdx.x;
    if (ray >= N) return;
    const size_t row = row_of ? (size_t)row_of[ray] : 0;
    const float *A = aff + row * 12;
    const float r = rgb_in[ray * 3 + 0], g = rgb_in[ray * 3 + 1], b = rgb_in[ray * 3 + 2];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = ((A[c * 4 + 0] * r + A[c * 4 + 1] * g) + A[c * 4 + 2] * b) + A[c * 4 + 3];
    if (sky) {
        float acc = 0.0f;
        for (uint32_t s = 0; s < S; s++) acc += w_last[(size_t)ray * S + s];
        const float opac = 1.0f - acc;
        const float *B = aff_sky + row * 12;
        const float sr = sky[ray * 3 + 0], sg = sky[ray * 3 + 1], sb = sky[ray * 3 + 2];
#pragma unroll
        for (int c = 0; c < 3; c++)
            o[c] += opac * (((B[c * 4 + 0] * sr + B[c * 4 + 1] * sg) + B[c * 4 + 2] * sb) + B[c * 4 + 3]);
    }
    rgb_out[ray * 3 + 0] = o[0]; rgb_out[ray * 3 + 1] = o[1]; rgb_out[ray * 3 + 2] = o[2];
}

}  // namespace

extern "C" int ucn_dense(const float *x, const float *w, const float *b, uint32_t M, uint32_t K, uint32_t Nout, int relu,
                         float *y, ucn_stream_t stream) {
    UCN_REQUIRE(x && w && y, "dense: null pointer argument");
    if ((uint64_t)M * Nout == 0) return 0;
    hipLaunchKernelGGL(k_dense, dim3(ucn_div_up((uint64_t)M * Nout, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, M, K, Nout, relu, y);
    UCN_LAUNCH_CHECK("dense");
    return 0;
}

extern "C" int ucn_apply_affine(const float *rgb_in, const float *affine, const int64_t *ray_to_row,
                                const float *weights_last, uint32_t S, const float *sky_rgb, const float *affine_sky,
                                uint32_t N, float *rgb_out, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (rgb_in && affine && rgb_out), "apply_affine: null pointer argument");
    UCN_REQUIRE(!sky_rgb || (weights_last && affine_sky), "apply_affine: the sky blend needs weights and the sky affine");
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_apply_affine, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, rgb_in, affine, ray_to_row,
                       weights_last, S, sky_rgb, affine_sky, N, rgb_out);
    UCN_LAUNCH_CHECK("apply_affine");
    return 0;
}
