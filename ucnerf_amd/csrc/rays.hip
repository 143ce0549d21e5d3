// On-GPU ray generation for full frames and pixel batches.
//
//   k_generate_rays <- camera_utils.py:448-557 (pixels_to_rays: perspective pinhole, no distortion, no NDC -- the
//                      Waymo / UC-NeRF configuration, datasets.py:855) + datasets.py:421-447 (_make_ray_batch:
//                      cam_dirs and the broadcast near / far / lossmult / cam_idx columns)
//
// The reference does this with numpy on DataLoader workers, in FLOAT64 (integer pixel + .5 promotes everything), and
// casts the batch to float32 at the very end (datasets.py:476); 64 bytes per ray then cross PCIe (157 MB per
// 1920x1280 frame).  Here one thread derives one ray from two small per-camera matrices, in float64 with the
// reference's operation order (explicit mul / add, -ffp-contract=off; IEEE double division and sqrt), and rounds once
// at the store -- the float32 batch is bit-identical to the reference's (tests/golden/rays.npz).  The kernel is a pure
// streaming write: 68 B per ray (76 with the image plane) out, 8 B in when pixel coordinates are given.
#include "ucn_common.h"

namespace {

struct RayOut {
    float *origins, *directions, *viewdirs, *radii, *imageplane, *cam_dirs, *near_, *far_, *lossmult, *cam_idx;
};

__device__ __forceinline__ void matvec3(const double *__restrict__ A, uint32_t ld, double x, double y, double z, double (&o)[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = (A[i * ld + 0] * x + A[i * ld + 1] * y) + A[i * ld + 2] * z;
}
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt((x * x + y * y) + z * z); }

__global__ __launch_bounds__(256) void k_generate_rays(const int32_t *__restrict__ pix_x, const int32_t *__restrict__ pix_y,
                                                       const int32_t *__restrict__ cam_idx, int32_t cam_scalar,
                                                       const double *__restrict__ pixtocams, const double *__restrict__ camtoworlds,
                                                       uint32_t width, uint32_t n, float near_v, float far_v, RayOut out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t cam = cam_idx ? cam_idx[i] : cam_scalar;
    const double *P = pixtocams + (size_t)cam * 9, *M = camtoworlds + (size_t)cam * 12;
    const double x = (double)(pix_x ? pix_x[i] : (int32_t)(i % width));          // camera_utils.py:368-370 'xy' meshgrid
    const double y = (double)(pix_y ? pix_y[i] : (int32_t)(i / width));
    // pixel centre and its +1 neighbours in x and y (:487-495), inverse intrinsics (:501), OpenCV -> OpenGL (:531)
    double c[3][3], d[3][3];
    matvec3(P, 3, x + 0.5, y + 0.5, 1.0, c[0]);
    matvec3(P, 3, (x + 1.0) + 0.5, y + 0.5, 1.0, c[1]);
    matvec3(P, 3, x + 0.5, (y + 1.0) + 0.5, 1.0, c[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c[k][1] = -c[k][1];
        c[k][2] = -c[k][2];
        matvec3(M, 4, c[k][0], c[k][1], c[k][2], d[k]);                          // camera rotation (:538)
    }
    const double nd = norm3(d[0][0], d[0][1], d[0][2]);
    const double dxn = norm3(d[1][0] - d[0][0], d[1][1] - d[0][1], d[1][2] - d[0][2]);      // :548-549
    const double dyn = norm3(d[2][0] - d[0][0], d[2][1] - d[0][1], d[2][2] - d[0][2]);
    const double radius = (0.5 * (dxn + dyn)) * 2.0 / 3.4641016151377544;        // :562, np.sqrt(12)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        out.origins[(size_t)i * 3 + k] = (float)M[k * 4 + 3];
        out.directions[(size_t)i * 3 + k] = (float)d[0][k];
        out.viewdirs[(size_t)i * 3 + k] = (float)(d[0][k] / nd);                 // :544
        out.cam_dirs[(size_t)i * 3 + k] = (float)(-M[k * 4 + 2]);               // datasets.py:446
    }
    out.radii[i] = (float)radius;
    if (out.imageplane) {
        out.imageplane[(size_t)i * 2 + 0] = (float)c[0][0];
        out.imageplane[(size_t)i * 2 + 1] = (float)c[0][1];
    }
    if (out.near_) out.near_[i] = near_v;
    if (out.far_) out.far_[i] = far_v;
    if (out.lossmult) out.lossmult[i] = 1.0f;
    if (out.cam_idx) out.cam_idx[i] = (float)cam;
}

}  // namespace

extern "C" int ucn_generate_rays(const int32_t *pix_x, const int32_t *pix_y, const int32_t *cam_idx, int32_t cam_idx_scalar,
                                 const double *pixtocams, const double *camtoworlds, uint32_t n_cams, uint32_t width,
                                 uint32_t height, uint32_t n_rays, float near_, float far_, float *origins, float *directions,
                                 float *viewdirs, float *radii, float *imageplane, float *cam_dirs, float *near_out,
                                 float *far_out, float *lossmult_out, float *cam_idx_out, ucn_stream_t stream) {
    if (n_rays == 0) return 0;                            // an empty batch has null data pointers
    UCN_REQUIRE(pixtocams && camtoworlds && origins && directions && viewdirs && radii && cam_dirs,
                "generate_rays: null pointer argument");
    UCN_REQUIRE((pix_x == nullptr) == (pix_y == nullptr), "generate_rays: pix_x and pix_y come together");
    UCN_REQUIRE(pix_x || (width > 0 && (uint64_t)width * height == n_rays),
                "generate_rays: without pixel coordinates n_rays must be width * height (%u x %u != %u)", width, height, n_rays);
    UCN_REQUIRE(n_cams > 0 && (cam_idx || (cam_idx_scalar >= 0 && (uint32_t)cam_idx_scalar < n_cams)),
                "generate_rays: camera index %d out of range [0, %u)", cam_idx_scalar, n_cams);
    const RayOut out{origins, directions, viewdirs, radii, imageplane, cam_dirs, near_out, far_out, lossmult_out, cam_idx_out};
    hipLaunchKernelGGL(k_generate_rays, dim3(ucn_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, pix_x, pix_y, cam_idx,
                       cam_idx_scalar, pixtocams, camtoworlds, width ? width : 1u, n_rays, near_, far_, out);
    UCN_LAUNCH_CHECK("generate_rays");
    return 0;
}
