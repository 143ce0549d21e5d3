// Training-side reductions and the optimiser step over the hash tables.
//
//   k_adam_step       <- torch.optim.Adam as built by train_utils.py:347-366 (create_optimizer), applied to the large
//                        fp32 tables, with the `param.grad.nan_to_num_()` of train_utils.py:335-344 folded in
//   k_distortion_fwd / _bwd <- stepfun.py:297-307 `lossfun_distortion` (train_utils.py:273-279), O(S) per ray instead
//                        of the reference's [N, S, S] matrix
//
// The reference's optimiser step over a 7.1 M x 2 table is ~12 elementwise passes (nan_to_num, lerp, mul, addcmul,
// sqrt, div, add, addcdiv); here every element is read and written once: 5 x 4 B per parameter.
#include "ucn_common.h"

namespace {

__device__ __forceinline__ float nan_to_num0(float v) {
    // torch.nan_to_num(): NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
    if (v != v) return 0.0f;
    if (v == INFINITY) return 3.4028234663852886e38f;
    if (v == -INFINITY) return -3.4028234663852886e38f;
    return v;
}

// torch/optim/adam.py (_single_tensor_adam / _multi_tensor_adam, amsgrad = False, weight_decay = 0, maximize = False):
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
__global__ __launch_bounds__(256) void k_adam_step(float *__restrict__ param, float *__restrict__ grad, float *__restrict__ m,
                                                   float *__restrict__ v, uint64_t n4, uint64_t n, float w1, float beta2,
                                                   float w2, float bc2_sqrt, float eps, float neg_step, int sanitize) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n4) {
        float4 p = reinterpret_cast<float4 *>(param)[i], g = reinterpret_cast<float4 *>(grad)[i];
        float4 a = reinterpret_cast<float4 *>(m)[i], b = reinterpret_cast<float4 *>(v)[i];
        float *pp = &p.x, *gp = &g.x, *ap = &a.x, *bp = &b.x;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float gk = sanitize ? nan_to_num0(gp[k]) : gp[k];
            gp[k] = gk;
            ap[k] = fmaf(w1, gk - ap[k], ap[k]);
            bp[k] = fmaf(w2 * gk, gk, bp[k] * beta2);
            pp[k] = fmaf(neg_step, ap[k] / (sqrtf(bp[k]) / bc2_sqrt + eps), pp[k]);
        }
        reinterpret_cast<float4 *>(param)[i] = p;
        reinterpret_cast<float4 *>(m)[i] = a;
        reinterpret_cast<float4 *>(v)[i] = b;
        if (sanitize) reinterpret_cast<float4 *>(grad)[i] = g;            // the reference sanitises the stored gradient
    }
    if (i == 0) {                                                         // up to three tail elements
        for (uint64_t j = n4 * 4; j < n; j++) {
            const float gk = sanitize ? nan_to_num0(grad[j]) : grad[j];
            if (sanitize) grad[j] = gk;
            m[j] = fmaf(w1, gk - m[j], m[j]);
            v[j] = fmaf(w2 * gk, gk, v[j] * beta2);
            param[j] = fmaf(neg_step, m[j] / (sqrtf(v[j]) / bc2_sqrt + eps), param[j]);
        }
    }
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wscan(float v, int lane) {        // inclusive
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// One wave per ray, lane owns CH consecutive intervals.  loss = sum_i w_i^2 d_i / 3 + sum_ij w_i w_j |u_i - u_j| with
// u = interval midpoints (sorted, so the double sum is 2 sum_i w_i (u_i W_i - M_i), W / M exclusive prefix sums of w and
// w u); d loss / d w_k = 2 w_k d_k / 3 + 2 [u_k (W_<k - W_>k) - (M_<k - M_>k)].  t carries no gradient (models.py:204).
template <int CH, bool BWD>
__global__ __launch_bounds__(256) void k_distortion(const float *__restrict__ t, const float *__restrict__ w, uint32_t N, uint32_t S,
                                                    const float *__restrict__ g_loss, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float *tr = t + (size_t)ray * (S + 1), *wr = w + (size_t)ray * S;
    float u[CH], d[CH], wv[CH], lw = 0.0f, lm = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        u[c] = d[c] = wv[c] = 0.0f;
        if (i < S) {
            const float t0 = tr[i], t1 = tr[i + 1];
            u[c] = (t1 + t0) / 2.0f;
            d[c] = t1 - t0;
            wv[c] = wr[i];
        }
        lw += wv[c];
        lm += wv[c] * u[c];
    }
    const float iw = wscan(lw, lane), im = wscan(lm, lane);
    float W = iw - lw, M = im - lm;                                    // exclusive prefixes at the lane's first interval
    if (!BWD) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            acc += 2.0f * (wv[c] * (u[c] * W - M)) + wv[c] * wv[c] * d[c] / 3.0f;
            W += wv[c];
            M += wv[c] * u[c];
        }
        acc = wsum(acc);
        if (lane == 0) out[ray] = acc;
    } else {
        const float Wt = __shfl(iw, 63, 64), Mt = __shfl(im, 63, 64), g = g_loss[ray];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t i = lane * CH + c;
            const float Wgt = Wt - W - wv[c], Mgt = Mt - M - wv[c] * u[c];      // sums over j > i
            if (i < S) out[(size_t)ray * S + i] = g * (2.0f * wv[c] * d[c] / 3.0f + 2.0f * (u[c] * (W - Wgt) - (M - Mgt)));
            W += wv[c];
            M += wv[c] * u[c];
        }
    }
}

}  // namespace

extern "C" int ucn_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, uint64_t n, float lr, float beta1,
                             float beta2, float eps, uint32_t step, int sanitize_grad, ucn_stream_t stream) {
    if (n == 0) return 0;
    UCN_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer argument");
    UCN_REQUIRE(step >= 1, "adam_step: step counts from 1");
    UCN_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                "adam_step: tensors must be 16-byte aligned");
    // bias corrections in double like the Python side of torch.optim.Adam, handed to the kernel as fp32 scalars
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const uint64_t n4 = n / 4, threads = n4 ? n4 : 1;
    hipLaunchKernelGGL(k_adam_step, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n4, n, 1.0f - beta1, beta2, 1.0f - beta2, (float)sqrt(bc2), eps, (float)(-(double)lr / bc1),
                       sanitize_grad);
    UCN_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int ucn_distortion_loss(const float *t, const float *w, uint32_t N, uint32_t S, const float *g_loss, float *out,
                                   ucn_stream_t stream) {
    if (N == 0) return 0;
    UCN_REQUIRE(t && w && out, "distortion_loss: null pointer argument");
    UCN_REQUIRE(S >= 1 && S <= 512, "distortion_loss: samples per ray must be in [1,512], got %u", S);
    const dim3 grid(ucn_div_up(N, 4));
    hipStream_t st = (hipStream_t)stream;
#define UCN_DL(CH)                                                                                                   \
    do {                                                                                                             \
        if (g_loss) hipLaunchKernelGGL((k_distortion<CH, true>), grid, dim3(256), 0, st, t, w, N, S, g_loss, out);   \
        else hipLaunchKernelGGL((k_distortion<CH, false>), grid, dim3(256), 0, st, t, w, N, S, g_loss, out);         \
    } while (0)
    if (S <= 64) UCN_DL(1);
    else if (S <= 128) UCN_DL(2);
    else if (S <= 256) UCN_DL(4);
    else UCN_DL(8);
#undef UCN_DL
    UCN_LAUNCH_CHECK("distortion_loss");
    return 0;
}
