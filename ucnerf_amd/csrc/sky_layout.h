// Shared by sky.hip (rendering) and sky_train.hip (training step): the sky NeRF's sample count, the 14 KiB side table
// that sits in LDS behind the weight ring (biases as accumulator tiles, the 3-wide input layer and the narrow heads as
// VALU operands in accumulator-slot order), and the small kernels / device helpers both kernels use.
// ref /root/reference/nerf/internal/models.py:743-820 (NeRF, D = 8, W = 256, skips = [4], multires_view = 4), :852-904.
#pragma once
#include "bf_tiles.h"
#include "wave_dpp.h"

namespace {

constexpr int kSkySamples = 120;
// ---- side table (floats), resident in LDS behind the ring
constexpr int kSB = 0;            // 6 x 256: biases of pts_linears 1,2,3,4,6,7 as bias tiles [t][h][16]
constexpr int kSL0 = 1536;        // 256 x {w0,w1,w2,b} of pts_linears.0, accumulator-slot order
constexpr int kSAlpha = 2560;     // 256 alpha_linear weights (slot order), then b_alpha
constexpr int kSRgb = 2820;       // 128 x {w_r,w_g,w_b,0} (slot order), then b_rgb[3]
constexpr int kSideFloats = 3584; // 14 KiB
constexpr int kBiasIdx[7] = {0, 1, 2, 3, -1, 4, 5};              // side-table bias block; layer 5's bias rides in the aux tile

// alpha head (256 -> 1, VALU) on the fp32 ReLU output of layer 7, taken half a tile at a time while that half is
// being split anyway: sig += sum_e relu(acc[8S+e]) * w[slot(TILE, 8S+e)]
template <int TILE, int S>
__device__ __forceinline__ void alpha_partial(const f32x16 &acc, const float *__restrict__ pa_h, float &sig) {
#pragma unroll
    for (int e = 0; e < 8; e++) sig = fmaf(relu_bits(acc[8 * S + e]), pa_h[(TILE * 16 + 8 * S + e) * 2], sig);
}

__device__ __forceinline__ void side_bias_tile(const float *side, int off, int tile, f32x16 &acc, int h) {
    const float4 *p = reinterpret_cast<const float4 *>(side + off + tile * 32 + h * 16);
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
        const float4 v = p[r4];
        acc[4 * r4 + 0] = v.x; acc[4 * r4 + 1] = v.y; acc[4 * r4 + 2] = v.z; acc[4 * r4 + 3] = v.w;
    }
}

// per-ray auxiliary tile: [0, 0, 0, 1, embed(cam_dir) = x, sin(f x), cos(f x) for f in 1,2,4,8 (27), 0]
__global__ __launch_bounds__(256) void k_sky_aux(const float *__restrict__ cam, uint32_t N, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N * 32u) return;
    const uint32_t ray = i >> 5, k = i & 31u;
    float v = 0.0f;
    if (k == 3u) v = 1.0f;
    else if (k >= 4u && k < 31u) {
        const uint32_t e = k - 4u;
        if (e < 3u) v = cam[ray * 3 + e];
        else {
            const uint32_t kk = e - 3u, a = kk % 3u, fn = (kk / 3u) & 1u, fi = kk / 6u;
            const float x = cam[ray * 3 + a] * (float)(1u << fi);
            v = fn ? cosf(x) : sinf(x);
        }
    }
    out[i] = v;
}


}  // namespace
