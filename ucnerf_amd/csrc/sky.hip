// UC-NeRF sky layer: 120 stratified samples per ray through an 8x256 NeRF MLP, alpha-composited.
//
// Replaces models.py:326-337 (call), :852-904 (render_rays), :822-850 (raw2outputs) and
// :743-820 (NeRF, D=8, W=256, skips=[4], multires_view=4) of /root/reference/nerf/internal/models.py.
// 562,688 MAC per sample x 120 samples: the largest FLOP term of the path when model_sky is on, so
// it runs on the same register-chained fp32 MFMA engine as the field MLP (mfma_chain.h): a wave
// owns 32 samples, the eight 256-wide layers ping-pong between two 128-register tile sets, the
// 3-d input layers and the 1/3-wide heads stay on the VALU, weights stream once per workgroup
// through LDS.  The 27-d view encoding is per RAY (it encodes cam_dirs, models.py:331,866): its
// product with views_linears is folded into a per-ray bias.
//
// Reference quirks kept: z = near(1-t) + t/far with near = batch.far and
// far = 1.5*near[0], i.e. z DEcreases; the last interval is 1e10; 1e-10 is added inside the
// transmittance product.
//
// Weight stream: pts_linears.1..7 ([ot<8][it<8][r4] each; layer 5 reads columns 3..258),
// feature_linear (same shape), views_linears.0 ([ot<4][it<8][r4], columns 0..255).
#include "mfma_chain.h"

namespace {

constexpr int kSkySamples = 120;
constexpr uint64_t kSkyStreamGroups = 7 * 256 + 256 + 128;             // 2176 = 68 chunks
constexpr uint64_t kOffIn0 = kSkyStreamGroups * 256;                    // {w0,w1,w2,b} of pts_linears.0
constexpr uint64_t kOffIn5 = kOffIn0 + 1024;                            // {w0,w1,w2,b} of pts_linears.5[:, :3]
constexpr uint64_t kOffAlpha = kOffIn5 + 1024;                          // alpha_linear, 256 floats
constexpr uint64_t kOffRgb = kOffAlpha + 256;                           // rgb_linear, 128 x float4
constexpr uint64_t kSkyPackedFloats = kOffRgb + 512;

struct SkyArgs {
    const float *packed;
    const float *b_pts[8];
    const float *b_alpha, *b_feat, *b_view, *b_rgb;
    const float *view_bias;      // [N,128]
    const float *origins, *dirs, *far_, *t_vals;
    float inv_sky_far;
    uint32_t N;
    float *raw;                  // [N*120, 4] = rgb(3), sigma
};

// acc[t][r] = w0*x + w1*y + w2*z + b with {w0,w1,w2,b} float4 per accumulator slot (VALU)
__device__ __forceinline__ void input3(f32x16 (&acc)[8], const float4 *__restrict__ p, float x, float y, float z, int h) {
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float4 w = p[(t * 16 + r) * 2 + h];
            acc[t][r] = ((w.x * x + w.y * y) + w.z * z) + w.w;
        }
}

__global__ __launch_bounds__(256) void k_sky_mlp(SkyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint64_t B = (uint64_t)a.N * kSkySamples;
    const uint64_t b0 = ((uint64_t)blockIdx.x * 4u + wave) * 32u;
    const bool live = b0 + j < B;
    const uint64_t b = live ? b0 + j : B - 1;
    const uint32_t ray = (uint32_t)(b / kSkySamples), s = (uint32_t)(b - (uint64_t)ray * kSkySamples);

    WeightStream ws{a.packed, s_w, lane, wave, (uint32_t)(kSkyStreamGroups / kChunkGroups)};
    ws.issue(0);

    const float tv = a.t_vals[s];
    const float z = a.far_[ray] * (1.0f - tv) + a.inv_sky_far * tv;                    // models.py:872
    const float px = a.origins[ray * 3 + 0] + a.dirs[ray * 3 + 0] * z;
    const float py = a.origins[ray * 3 + 1] + a.dirs[ray * 3 + 1] * z;
    const float pz = a.origins[ray * 3 + 2] + a.dirs[ray * 3 + 2] * z;

    f32x16 hA[8], hB[8];
    input3(hA, reinterpret_cast<const float4 *>(a.packed + kOffIn0), px, py, pz, h);   // layer 0
    relu_tiles<8>(hA);
    init_bias<8>(hB, a.b_pts[1], nullptr, h); chain<8, 8>(0 * 256, hB, hA, ws); relu_tiles<8>(hB);
    init_bias<8>(hA, a.b_pts[2], nullptr, h); chain<8, 8>(1 * 256, hA, hB, ws); relu_tiles<8>(hA);
    init_bias<8>(hB, a.b_pts[3], nullptr, h); chain<8, 8>(2 * 256, hB, hA, ws); relu_tiles<8>(hB);
    init_bias<8>(hA, a.b_pts[4], nullptr, h); chain<8, 8>(3 * 256, hA, hB, ws); relu_tiles<8>(hA);
    // layer 5 reads cat[pts, h] (skip at 4): pts columns on the VALU (bias folded), h via MFMA
    input3(hB, reinterpret_cast<const float4 *>(a.packed + kOffIn5), px, py, pz, h);
    chain<8, 8>(4 * 256, hB, hA, ws); relu_tiles<8>(hB);
    init_bias<8>(hA, a.b_pts[6], nullptr, h); chain<8, 8>(5 * 256, hA, hB, ws); relu_tiles<8>(hA);
    init_bias<8>(hB, a.b_pts[7], nullptr, h); chain<8, 8>(6 * 256, hB, hA, ws); relu_tiles<8>(hB);
    // alpha head 256 -> 1 (VALU)
    const float *pa = a.packed + kOffAlpha;
    float sig = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) sig = fmaf(hB[t][r], pa[(t * 16 + r) * 2 + h], sig);
    sig = (sig + __shfl_xor(sig, 32, 64)) + a.b_alpha[0];
    // feature_linear 256 -> 256 (no activation)
    init_bias<8>(hA, a.b_feat, nullptr, h); chain<8, 8>(7 * 256, hA, hB, ws);
    // views_linears.0: [feature, enc(cam_dir)] -> 128, ReLU; the encoding part is a per-ray bias
    f32x16 v[4];
    init_bias<4>(v, a.b_view, a.view_bias + (size_t)ray * 128, h);
    chain<4, 8>(8 * 256, v, hA, ws);
    const float4 *pr = reinterpret_cast<const float4 *>(a.packed + kOffRgb);
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float4 w = pr[(t * 16 + r) * 2 + h];
            const float x = fmaxf(v[t][r], 0.0f);
            c0 = fmaf(x, w.x, c0); c1 = fmaf(x, w.y, c1); c2 = fmaf(x, w.z, c2);
        }
    c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
    if (live && h == 0)
        *reinterpret_cast<float4 *>(a.raw + b * 4) = make_float4(c0 + a.b_rgb[0], c1 + a.b_rgb[1], c2 + a.b_rgb[2], sig);
}

// per-ray view bias: W_view[:, 256:283] . embed(cam_dir), embed = [x, sin(f x), cos(f x) for f in 1,2,4,8]
__global__ __launch_bounds__(128) void k_sky_view_bias(const float *__restrict__ cam, const float *__restrict__ w_view,
                                                       uint32_t N, float *__restrict__ out) {
    __shared__ float enc[27];
    const uint32_t ray = blockIdx.x;
    if (threadIdx.x < 27) {
        const uint32_t k = threadIdx.x;
        float v;
        if (k < 3) v = cam[ray * 3 + k];
        else {
            const uint32_t kk = k - 3, a = kk % 3, fn = (kk / 3) & 1u, fi = kk / 6;
            const float x = cam[ray * 3 + a] * (float)(1u << fi);
            v = fn ? cosf(x) : sinf(x);
        }
        enc[k] = v;
    }
    __syncthreads();
    const float *row = w_view + (size_t)threadIdx.x * 283 + 256;
    float s = 0.0f;
    for (int k = 0; k < 27; k++) s = fmaf(row[k], enc[k], s);
    out[(size_t)ray * 128 + threadIdx.x] = s;
}

// raw2outputs (models.py:822-850): one thread per ray, 120 sequential samples
__global__ __launch_bounds__(256) void k_sky_composite(const float *__restrict__ raw, const float *__restrict__ dirs,
                                                       const float *__restrict__ far_, const float *__restrict__ t_vals,
                                                       float inv_sky_far, uint32_t N, float *__restrict__ out) {
    const uint32_t ray = blockIdx.x * 256u + threadIdx.x;
    if (ray >= N) return;
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float nr = far_[ray];
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    float z = nr * (1.0f - t_vals[0]) + inv_sky_far * t_vals[0];
    for (int s = 0; s < kSkySamples; s++) {
        float dist;
        float zn = z;
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
        b += w * (1.0f / (1.0f + expf(-v.z)));
        T = T * ((1.0f - alpha) + 1e-10f);
        z = zn;
    }
    out[ray * 3 + 0] = r; out[ray * 3 + 1] = g; out[ray * 3 + 2] = b;
}

}  // namespace

extern "C" uint64_t ucn_sky_packed_floats(void) { return kSkyPackedFloats; }

extern "C" int ucn_sky_pack(const ucn_sky_t *s, ucn_stream_t stream) {
    UCN_REQUIRE(s && s->packed, "sky_pack: null descriptor / packed buffer");
    for (int i = 0; i < 8; i++) UCN_REQUIRE(s->w_pts[i] && s->b_pts[i], "sky_pack: pts_linears.%d missing", i);
    UCN_REQUIRE(s->w_alpha && s->b_alpha && s->w_feat && s->b_feat && s->w_view && s->b_view && s->w_rgb && s->b_rgb,
                "sky_pack: head weights missing");
    hipStream_t st = (hipStream_t)stream;
    uint64_t off = 0;
    auto chainpack = [&](const float *W, uint32_t ld, uint32_t col0, uint32_t nto, uint32_t nti) {
        hipLaunchKernelGGL(k_pack_chain, dim3(ucn_div_up((uint64_t)nto * nti * 1024, 256)), dim3(256), 0, st, W, ld, col0,
                           0u, nto, nti, s->packed + off);
        off += (uint64_t)nto * nti * 1024;
    };
    for (int i = 1; i < 8; i++) chainpack(s->w_pts[i], i == 5 ? 259 : 256, i == 5 ? 3 : 0, 8, 8);
    chainpack(s->w_feat, 256, 0, 8, 8);
    chainpack(s->w_view, 283, 0, 4, 8);
    hipLaunchKernelGGL(k_pack_in3, dim3(4), dim3(256), 0, st, s->w_pts[0], 3u, s->b_pts[0], 256u, s->packed + kOffIn0);
    hipLaunchKernelGGL(k_pack_in3, dim3(4), dim3(256), 0, st, s->w_pts[5], 259u, s->b_pts[5], 256u, s->packed + kOffIn5);
    hipLaunchKernelGGL(k_pack_head, dim3(1), dim3(256), 0, st, s->w_alpha, 256u, 0u, 256u, 1u, 1u, s->packed + kOffAlpha);
    hipLaunchKernelGGL(k_pack_head, dim3(2), dim3(256), 0, st, s->w_rgb, 128u, 0u, 128u, 3u, 4u, s->packed + kOffRgb);
    UCN_LAUNCH_CHECK("sky_pack");
    return 0;
}

extern "C" uint64_t ucn_sky_workspace_floats(uint32_t N) { return (uint64_t)N * (128 + kSkySamples * 4); }

extern "C" int ucn_sky_render(const ucn_sky_t *s, const float *origins, const float *directions,
                                 const float *cam_dirs, const float *far_, float far0_times_1p5,
                                 const float *t_vals /*DEVICE [120] = linspace(0,1,120)*/, uint32_t N,
                                 float *workspace /*DEVICE N*(128+480) floats*/, float *sky_rgb_out,
                                 ucn_stream_t stream) {
    UCN_REQUIRE(s && s->packed && origins && directions && cam_dirs && far_ && t_vals && workspace && sky_rgb_out,
                "sky_render: null pointer argument");
    if (N == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    float *view_bias = workspace;
    float *raw = workspace + (size_t)N * 128;
    hipLaunchKernelGGL(k_sky_view_bias, dim3(N), dim3(128), 0, st, cam_dirs, s->w_view, N, view_bias);
    SkyArgs a;
    a.packed = s->packed;
    for (int i = 0; i < 8; i++) a.b_pts[i] = s->b_pts[i];
    a.b_alpha = s->b_alpha; a.b_feat = s->b_feat; a.b_view = s->b_view; a.b_rgb = s->b_rgb;
    a.view_bias = view_bias; a.origins = origins; a.dirs = directions; a.far_ = far_; a.t_vals = t_vals;
    a.inv_sky_far = 1.0f / far0_times_1p5;
    a.N = N; a.raw = raw;
    const uint64_t B = (uint64_t)N * kSkySamples;
    hipLaunchKernelGGL(k_sky_mlp, dim3(ucn_div_up(B, 128)), dim3(256), 2 * kChunkGroups * 256 * sizeof(float), st, a);
    hipLaunchKernelGGL(k_sky_composite, dim3(ucn_div_up(N, 256)), dim3(256), 0, st, raw, directions, far_, t_vals,
                       a.inv_sky_far, N, sky_rgb_out);
    UCN_LAUNCH_CHECK("sky_render");
    return 0;
}
