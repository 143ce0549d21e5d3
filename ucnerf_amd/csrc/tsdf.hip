// TSDF fusion of rendered depth / colour images into a voxel volume.
//
// Replaces TSDF.integrate_tsdf of /root/reference/nerf/tsdf.py:115-219: the reference projects ALL voxels into
// every view with two batched GEMMs ([B,4,N] and [B,3,N] temporaries), samples depth / colour with
// F.grid_sample(mode="nearest", padding_mode="zeros", align_corners=False), and then walks the views sequentially
// with boolean-mask gathers and scatters over the whole volume (5 full-volume passes per view).  Here one thread owns
// one voxel: its running value / weight / colour stay in registers across the views of the call, every view is a
// projection (12 FMAs), one or four scattered loads and a running weighted mean -- the same sequential update
// order per voxel as the reference's loop over the batch.  Bound: one coalesced read-modify-write of the volume
// (20-36 B per voxel) per CALL instead of per view.
#include "ucn_common.h"

namespace {

__global__ __launch_bounds__(256) void k_tsdf_integrate(const float *__restrict__ world, uint32_t N,
                                                        const float *__restrict__ w2c, const float *__restrict__ K,
                                                        const float *__restrict__ depth, const float *__restrict__ color,
                                                        uint32_t B, uint32_t H, uint32_t W, float truncation,
                                                        float *__restrict__ values, float *__restrict__ weights,
                                                        float *__restrict__ colors) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= N) return;
    const float x = world[v], y = world[(size_t)N + v], z = world[2 * (size_t)N + v], w = world[3 * (size_t)N + v];
    float val = values[v], wt = weights[v];
    float col[3] = {0.0f, 0.0f, 0.0f};
    if (color) { col[0] = colors[(size_t)v * 3]; col[1] = colors[(size_t)v * 3 + 1]; col[2] = colors[(size_t)v * 3 + 2]; }
    const float k00 = K[0], k01 = K[1], k02 = K[2], k10 = K[3], k11 = K[4], k12 = K[5];
    for (uint32_t i = 0; i < B; i++) {
        const float *m = w2c + (size_t)i * 12;                       // rows 0..2 of inverse(c2w[i])
        const float xc = ((m[0] * x + m[1] * y) + m[2] * z) + m[3] * w;
        const float yc = -(((m[4] * x + m[5] * y) + m[6] * z) + m[7] * w);        // tsdf.py:150-152: y and z flipped
        const float zc = -(((m[8] * x + m[9] * y) + m[10] * z) + m[11] * w);
        const float px = xc / zc, py = yc / zc, pz = zc / zc;        // tsdf.py:162: [x, y, z] / z
        const float u = (k00 * px + k01 * py) + k02 * pz;
        const float vv = (k10 * px + k11 * py) + k12 * pz;
        // tsdf.py:169 grid = 2 p / size - 1, then grid_sample's unnormalisation ((g + 1) size - 1) / 2 and nearbyint
        const float gx = 2.0f * u / (float)W - 1.0f, gy = 2.0f * vv / (float)H - 1.0f;
        const float fx = nearbyintf(((gx + 1.0f) * (float)W - 1.0f) / 2.0f);
        const float fy = nearbyintf(((gy + 1.0f) * (float)H - 1.0f) / 2.0f);
        const bool inb = fx >= 0.0f && fx <= (float)(W - 1) && fy >= 0.0f && fy <= (float)(H - 1);   // NaN: false
        const size_t pix = inb ? (size_t)fy * W + (size_t)fx : 0;
        const float sd = inb ? depth[(size_t)i * H * W + pix] : 0.0f;
        const float dist = sd - zc;
        const float tsdf = fminf(fmaxf(dist / truncation, -1.0f), 1.0f);
        if (zc > 0.0f && sd > 0.0f && dist > -truncation) {          // tsdf.py:194
            const float total = wt + 1.0f;
            val = (val * wt + tsdf * 1.0f) / total;
            if (color) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float sc = color[((size_t)i * 3 + c) * H * W + pix];
                    col[c] = (col[c] * wt + sc * 1.0f) / total;
                }
            }
            wt = total;
        }
    }
    values[v] = val;
    weights[v] = wt;
    if (color) { colors[(size_t)v * 3] = col[0]; colors[(size_t)v * 3 + 1] = col[1]; colors[(size_t)v * 3 + 2] = col[2]; }
}

}  // namespace

extern "C" int ucn_tsdf_integrate(const float *voxel_world /*[4][N]*/, uint32_t N, const float *w2c /*[B][3][4]*/,
                                  const float *K /*[3][3]*/, const float *depth /*[B][H][W]*/,
                                  const float *color /*[B][3][H][W] | NULL*/, uint32_t B, uint32_t H, uint32_t W,
                                  float truncation, float *values, float *weights, float *colors, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || B == 0 || (voxel_world && w2c && K && depth && values && weights), "tsdf_integrate: null pointer argument");
    UCN_REQUIRE(color == nullptr || colors != nullptr, "tsdf_integrate: colour images need the colour volume");
    UCN_REQUIRE(truncation > 0.0f, "tsdf_integrate: truncation must be positive");
    if (N == 0 || B == 0) return 0;
    hipLaunchKernelGGL(k_tsdf_integrate, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, voxel_world, N, w2c, K, depth,
                       color, B, H, W, truncation, values, weights, colors);
    UCN_LAUNCH_CHECK("tsdf_integrate");
    return 0;
}
