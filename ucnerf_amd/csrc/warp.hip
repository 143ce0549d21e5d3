// Virtual-pose depth warping of a reference frame into a (virtual) source camera.
//
//   k_img_warp      <- train_utils.py:19-55 `img_warping` (and the first half of :58-98 `img_warping_for_depth`)
//   k_warp_owner / k_warp_scatter <- train_utils.py:93-96: depth_tgt[y, x] = z, later pixels overwrite earlier ones
//
// The reference runs this on the CPU inside the data loader, on a full 1280x1920 depth map per training step
// (datasets.py:511-529: back-project every pixel with its depth, move it into the source camera, project, keep the
// pixels that land inside the frame).  One thread per pixel here; the 4x4 relative pose and the intrinsics travel as
// kernel arguments.  The reference's two [3,3] x [3,HW] products are BLAS sgemm calls whose k = 3 reductions may be
// fused (FMA) or not depending on the host library, so parity for the projected coordinates is a tolerance (a few
// float32 ulp of a pixel coordinate), not bit-exactness; the fused form is used here.
#include "ucn_common.h"

namespace {

struct WarpCam {
    float R[9], t[3], K[9];
};

__device__ __forceinline__ float dot3(const float *a, float x, float y, float z) { return fmaf(a[2], z, fmaf(a[1], y, a[0] * x)); }

__global__ __launch_bounds__(256) void k_img_warp(const float *__restrict__ depth, WarpCam cam, uint32_t H, uint32_t W,
                                                  float *__restrict__ pts, uint8_t *__restrict__ mask, float *__restrict__ zsrc) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= H * W) return;
    const uint32_t row = i / W, col = i - row * W;
    const float d = depth[i];
    const float X = ((float)col - cam.K[2]) / cam.K[0], Y = ((float)row - cam.K[5]) / cam.K[4];      // :34-35
    const float px = X * d, py = Y * d, pz = 1.0f * d;                                                 // :36-37
    const float qx = dot3(cam.R + 0, px, py, pz) + cam.t[0];                                           // :41
    const float qy = dot3(cam.R + 3, px, py, pz) + cam.t[1];
    const float qz = dot3(cam.R + 6, px, py, pz) + cam.t[2];
    const float nx = qx / qz, ny = qy / qz, nz = qz / qz;                                              // :43
    const float u = dot3(cam.K + 0, nx, ny, nz), v = dot3(cam.K + 3, nx, ny, nz);                      // :45
    pts[(size_t)i * 2 + 0] = u;
    pts[(size_t)i * 2 + 1] = v;
    const bool valid = u >= 0.0f && v >= 0.0f && u < (float)W - 0.5f && v < (float)H - 0.5f;           // :48-50
    mask[i] = (d > 0.0f && valid) ? 1 : 0;                                                             // :28, :51
    if (zsrc) zsrc[i] = qz;
}

// depth_tgt[int(v), int(u)] = z for the masked pixels in row-major order, the LAST writer winning (the sequential
// semantics of the reference's indexed assignment): the owner of a target pixel is the largest source index.
__global__ __launch_bounds__(256) void k_warp_owner(const float *__restrict__ pts, const uint8_t *__restrict__ mask, uint32_t H,
                                                    uint32_t W, uint32_t *__restrict__ owner) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= H * W || !mask[i]) return;
    const uint32_t x = (uint32_t)(long long)pts[(size_t)i * 2 + 0], y = (uint32_t)(long long)pts[(size_t)i * 2 + 1];   // .to(long)
    atomicMax(owner + (size_t)y * W + x, i + 1u);
}
__global__ __launch_bounds__(256) void k_warp_scatter(const float *__restrict__ pts, const uint8_t *__restrict__ mask,
                                                      const float *__restrict__ zsrc, const uint32_t *__restrict__ owner, uint32_t H,
                                                      uint32_t W, float *__restrict__ depth_tgt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= H * W || !mask[i]) return;
    const uint32_t x = (uint32_t)(long long)pts[(size_t)i * 2 + 0], y = (uint32_t)(long long)pts[(size_t)i * 2 + 1];
    const size_t tgt = (size_t)y * W + x;
    if (owner[tgt] == i + 1u) depth_tgt[tgt] = zsrc[i];
}

}  // namespace

extern "C" int ucn_img_warping(const float *depth, const float *rel_pose_host, const float *intrinsic_host, uint32_t H, uint32_t W,
                               float *pts_out, uint8_t *mask_out, float *z_src_out, ucn_stream_t stream) {
    if ((uint64_t)H * W == 0) return 0;
    UCN_REQUIRE(depth && rel_pose_host && intrinsic_host && pts_out && mask_out, "img_warping: null pointer argument");
    UCN_REQUIRE((uint64_t)H * W < 0xFFFFFFFFull, "img_warping: frame too large");
    WarpCam cam;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            cam.R[r * 3 + c] = rel_pose_host[r * 4 + c];
            cam.K[r * 3 + c] = intrinsic_host[r * 3 + c];
        }
        cam.t[r] = rel_pose_host[r * 4 + 3];
    }
    hipLaunchKernelGGL(k_img_warp, dim3(ucn_div_up((uint64_t)H * W, 256)), dim3(256), 0, (hipStream_t)stream, depth, cam, H, W, pts_out,
                       mask_out, z_src_out);
    UCN_LAUNCH_CHECK("img_warping");
    return 0;
}

extern "C" int ucn_warp_scatter_depth(const float *pts, const uint8_t *mask, const float *z_src, uint32_t H, uint32_t W,
                                      uint32_t *owner_ws, float *depth_tgt, ucn_stream_t stream) {
    if ((uint64_t)H * W == 0) return 0;
    UCN_REQUIRE(pts && mask && z_src && owner_ws && depth_tgt, "warp_scatter_depth: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ucn_div_up((uint64_t)H * W, 256));
    (void)hipMemsetAsync(owner_ws, 0, (size_t)H * W * 4, st);
    (void)hipMemsetAsync(depth_tgt, 0, (size_t)H * W * 4, st);                       // torch.zeros_like (:93)
    hipLaunchKernelGGL(k_warp_owner, grid, dim3(256), 0, st, pts, mask, H, W, owner_ws);
    hipLaunchKernelGGL(k_warp_scatter, grid, dim3(256), 0, st, pts, mask, z_src, owner_ws, H, W, depth_tgt);
    UCN_LAUNCH_CHECK("warp_scatter_depth");
    return 0;
}
