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
#include "wave_dpp.h"

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

__device__ __forceinline__ float wsum(float v) { return wave_sum_dpp<float>(v); }
__device__ __forceinline__ float wscan(float v, int) { return wave_scan_dpp<float>(v); }        // inclusive (wave_dpp.h)

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
        const float Wt = wave_last<float>(iw), Mt = wave_last<float>(im), g = g_loss[ray];
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

// Anti-aliased interlevel loss of Zip-NeRF for one proposal level (train_utils.py:247-270): the NeRF level's
// histogram (c, w) is blurred with a box of half-width r (stepfun.py:395-403 blur_stepfun: a piecewise-linear pdf on
// the merged knots c -+ r), integrated (math.py:110-133 sorted_interp_quad) at the proposal fenceposts cp, and the
// proposal weights wp are pushed up to the resampled ones: sum_j max(w_s - wp, 0)^2 / (wp + 1e-5).  The NeRF level is
// detached, so only d/d wp exists.  One wave per ray; the reference's sort of 2(S+1) knots is a two-list merge rank,
// its O(n m) interpolation masks a binary search, its cumulative sums wave scans (accumulated in double like
// torch-CPU's cumsum; the reference's device cumsum is fp32 and agrees to ~3e-5).
__device__ __forceinline__ double wscan_d(double v, int) { return wave_scan_dpp<double>(v); }      // inclusive; torch-CPU's cumsum accumulates in double

__global__ __launch_bounds__(256) void k_interlevel(const float *__restrict__ c_, const float *__restrict__ w_, uint32_t S1,
                                                    const float *__restrict__ cp_, const float *__restrict__ wp_, uint32_t Sp, float r,
                                                    uint32_t N, float *__restrict__ loss_ray, float *__restrict__ dterm) {
    extern __shared__ float s_il[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ray_raw = blockIdx.x * 4u + wv;
    const bool live = ray_raw < N;
    const uint32_t ray = live ? ray_raw : N - 1;
    const uint32_t n1 = S1 + 1, m = 2 * n1;
    float *c = s_il + (size_t)wv * (n1 + S1 + 4 * m + (Sp + 1));
    float *pdf = c + n1, *kn = pdf + S1, *ds = kn + m, *vals = ds + m, *cdf = vals + m, *q = cdf + m;
    for (uint32_t i = lane; i < n1; i += 64) c[i] = c_[(size_t)ray * n1 + i];
    __syncthreads();
    for (uint32_t i = lane; i < S1; i += 64) pdf[i] = w_[(size_t)ray * S1 + i] / (c[i + 1] - c[i]);
    __syncthreads();
    // merged knots of (c - r) and (c + r) with the slope jumps +-y1 attached (blur_stepfun)
    for (uint32_t e = lane; e < m; e += 64) {
        const bool second = e >= n1;
        const uint32_t i = second ? e - n1 : e;
        const float y1 = ((i < S1 ? pdf[i] : 0.0f) - (i > 0 ? pdf[i - 1] : 0.0f)) / (2.0f * r);
        const float v = second ? c[i] + r : c[i] - r;
        uint32_t lo = 0, hi = n1;                          // #other-list elements ordered before v
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const float o = second ? c[mid] - r : c[mid] + r;
            if (second ? (o <= v) : (o < v)) lo = mid + 1;
            else hi = mid;
        }
        kn[i + lo] = v;
        ds[i + lo] = second ? -y1 : y1;
    }
    __syncthreads();
    // vals = [0, clamp_min(cumsum(dx * cumsum(dslope[:-1])), 0)], cdf = [0, cumsum(trapezoids)]: lane owns CH knots
    const uint32_t CH = (m + 63) / 64;
    double part = 0.0;
    for (uint32_t k = 0; k < CH; k++) {
        const uint32_t i = lane * CH + k;
        if (i + 1 < m) part += (double)ds[i];
    }
    double run = wscan_d(part, lane) - part;
    double part2 = 0.0;
    for (uint32_t k = 0; k < CH; k++) {
        const uint32_t i = lane * CH + k;
        if (i + 1 < m) {
            run += (double)ds[i];
            const float inc = (kn[i + 1] - kn[i]) * (float)run;       // cumsum(dslope) is a float tensor in the reference
            cdf[i] = inc;                                  // parked: the increments of the second cumsum
            part2 += (double)inc;
        }
    }
    __syncthreads();
    double run2 = wscan_d(part2, lane) - part2;
    for (uint32_t k = 0; k < CH; k++) {
        const uint32_t i = lane * CH + k;
        if (i + 1 < m) {
            run2 += (double)cdf[i];
            vals[i + 1] = fmaxf((float)run2, 0.0f);
        }
    }
    if (lane == 0) vals[0] = 0.0f;
    __syncthreads();
    double part3 = 0.0;
    for (uint32_t k = 0; k < CH; k++) {
        const uint32_t i = lane * CH + k;
        if (i + 1 < m) part3 += (double)(0.5f * (vals[i + 1] + vals[i]) * (kn[i + 1] - kn[i]));
    }
    double run3 = wscan_d(part3, lane) - part3;
    for (uint32_t k = 0; k < CH; k++) {
        const uint32_t i = lane * CH + k;
        if (i + 1 < m) {
            run3 += (double)(0.5f * (vals[i + 1] + vals[i]) * (kn[i + 1] - kn[i]));
            cdf[i + 1] = (float)run3;
        }
    }
    __syncthreads();
    if (lane == 0) cdf[0] = 0.0f;
    __syncthreads();
    // integrate the blurred pdf up to every proposal fencepost (sorted_interp_quad)
    for (uint32_t j = lane; j <= Sp; j += 64) {
        const float x = cp_[(size_t)ray * (Sp + 1) + j];
        uint32_t lo = 0, hi = m;                           // #(knots <= x)
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (kn[mid] <= x) lo = mid + 1;
            else hi = mid;
        }
        const uint32_t i0 = lo > 0 ? lo - 1 : 0, i1 = lo < m - 1 ? lo : m - 1;
        const float xp0 = kn[i0], xp1 = kn[i1], p0 = vals[i0], p1 = vals[i1];
        float off = (x - xp0) / (xp1 - xp0);
        if (off != off) off = 0.0f;                        // nan_to_num(.., 0).clamp(0, 1)
        else if (off == INFINITY) off = 3.4028234663852886e38f;
        else if (off == -INFINITY) off = -3.4028234663852886e38f;
        off = fminf(fmaxf(off, 0.0f), 1.0f);
        q[j] = cdf[i0] + (x - xp0) * (p0 + p1 * off + p0 * (1.0f - off)) / 2.0f;
    }
    __syncthreads();
    float acc = 0.0f;
    for (uint32_t j = lane; j < Sp; j += 64) {
        const float ws = q[j + 1] - q[j], wp = wp_[(size_t)ray * Sp + j];
        const float ex = fmaxf(ws - wp, 0.0f), den = wp + 1e-5f;
        acc += ex * ex / den;
        if (live) dterm[(size_t)ray * Sp + j] = -2.0f * ex / den - ex * ex / (den * den);
    }
    acc = wsum(acc);
    if (lane == 0 && live) loss_ray[ray] = acc;
}

// ---- elementwise halves of the colour MLP's hidden layers in training (models.py:615-640 under autograd).
// The reference concatenates the per-ray direction encoding to every sample; here it enters as a per-RAY row
// (direction block of the weight times the encoding, plus the bias) that is broadcast over the ray's S samples:
//   forward   h = relu(pre + per_ray[ray])                      (in place over the GEMM output)
//   backward  d_pre = gy * [h > 0],  d_per_ray[ray] = sum_s d_pre   (the second feeds the direction-weight and bias
//             gradients: a reduction over rays instead of samples)
// T = float (no autocast) or bf16 (the reference's accelerator.autocast()): arithmetic in fp32, one rounding per store,
// which is what torch's bf16 elementwise kernels do.
struct Bf16 { uint16_t v; };
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(Bf16 x) { return __uint_as_float((uint32_t)x.v << 16); }
__device__ __forceinline__ void from_f32(float f, float &o) { o = f; }
__device__ __forceinline__ void from_f32(float f, Bf16 &o) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);                       // round to nearest even (finite inputs)
    o.v = (uint16_t)(u >> 16);
}

// 8 consecutive elements as fp32 (16-byte vector accesses; the pointers are 16-byte aligned: W % 8 == 0)
__device__ __forceinline__ void load8(const float *p, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const Bf16 *p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[2 * k] = __uint_as_float(w[k] << 16);
        v[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void store8(float *p, const float (&v)[8]) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(Bf16 *p, const float (&v)[8]) {       // v already rounded to bf16 values
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = (__float_as_uint(v[2 * k]) >> 16) | (__float_as_uint(v[2 * k + 1]) & 0xFFFF0000u);
    *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
// x rounded to the storage type, as fp32
__device__ __forceinline__ float rounded(float x, const float *) { return x; }
__device__ __forceinline__ float rounded(float x, const Bf16 *) {
    Bf16 o;
    from_f32(x, o);
    return to_f32(o);
}

template <typename T>
__global__ __launch_bounds__(256) void k_bias_relu(T *__restrict__ pre, const T *__restrict__ per_ray, uint64_t total, uint32_t W,
                                                   uint32_t S) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 8u;          // 8 consecutive columns of one row
    if (i0 >= total) return;
    const uint64_t row = i0 / W;
    const uint32_t col = (uint32_t)(i0 - row * W);
    float a[8], b[8];
    load8(pre + i0, a);
    load8(per_ray + (row / S) * W + col, b);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = rounded(fmaxf(a[k] + b[k], 0.0f), pre);
    store8(pre + i0, a);
}

// One workgroup per ray: thread = (column group of 8, one of SL sample lanes); partial column sums meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_relu_bwd_reduce(const T *__restrict__ gy, const T *__restrict__ h, T *__restrict__ d_pre,
                                                         T *__restrict__ d_per_ray, uint32_t S, uint32_t W) {
    __shared__ float s_part[256 * 8];
    const uint32_t ray = blockIdx.x, groups = W / 8u, lanes = 256u / groups;   // W = 256 -> 32 groups x 8 sample lanes
    const uint32_t cg = threadIdx.x % groups, sl = threadIdx.x / groups;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sl < lanes) {
        for (uint32_t s = sl; s < S; s += lanes) {
            const uint64_t i0 = ((uint64_t)ray * S + s) * W + cg * 8u;
            float g[8], a[8];
            load8(gy + i0, g);
            load8(h + i0, a);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                g[k] = a[k] > 0.0f ? rounded(g[k], d_pre) : 0.0f;
                acc[k] += g[k];                             // the sum torch forms reads the ROUNDED d_pre
            }
            store8(d_pre + i0, g);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) s_part[threadIdx.x * 8 + k] = acc[k];
    __syncthreads();
    if (sl == 0) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            t[k] = 0.0f;
            for (uint32_t q = 0; q < lanes; q++) t[k] += s_part[(q * groups + cg) * 8 + k];
            t[k] = rounded(t[k], d_per_ray);
        }
        store8(d_per_ray + (uint64_t)ray * W + cg * 8u, t);
    }
}

// Hash decay (models.py:297-306): mean over levels and channels of the per-level mean of embeddings^2 = sum over rows of
// w_level * sum_c e^2 with w_level = 1 / (rows_of_level * L * C).  One pass over the table forward (block partials, added
// in a fixed order by k_decay_finish), one pass backward (grad = 2 g w_level e).  As torch ops: pow + per-row sum + dot
// over a 57 MB table and a [rows] weight vector, 110 us forward + 40 us backward per table and step.
constexpr int kDecayBlocks = 1024, kDecayLevels = 32;
struct DecayLevels {
    uint32_t n;
    uint32_t off[kDecayLevels + 1];      // first row of each level, off[n] = rows
    float w[kDecayLevels];
};
template <bool BWD>
__global__ __launch_bounds__(256) void k_hash_decay(const float *__restrict__ emb, uint32_t C, DecayLevels lv, const float *__restrict__ g,
                                                    float *__restrict__ out) {
    const uint64_t total = (uint64_t)lv.off[lv.n] * C;                     // floats
    const uint64_t per = ((total + kDecayBlocks - 1) / kDecayBlocks + 3) & ~3ull;
    const uint64_t e0 = (uint64_t)blockIdx.x * per, e1 = e0 + per < total ? e0 + per : total;
    float acc = 0.0f;
    const float g2 = BWD ? 2.0f * g[0] : 0.0f;
    for (uint32_t l = 0; l < lv.n; l++) {                                  // wave-uniform: the levels this block's range meets
        const uint64_t lo = (uint64_t)lv.off[l] * C > e0 ? (uint64_t)lv.off[l] * C : e0;
        const uint64_t hi = (uint64_t)lv.off[l + 1] * C < e1 ? (uint64_t)lv.off[l + 1] * C : e1;
        if (lo >= hi) continue;
        const float w = lv.w[l];
        float part = 0.0f;
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 256u) {
            const float e = emb[i];
            if (BWD) out[i] = e * (g2 * w);
            else part = fmaf(e, e, part);
        }
        acc = fmaf(w, part, acc);
    }
    if (BWD) return;
    __shared__ float s_p[256];
    s_p[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_p[threadIdx.x] += s_p[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_p[0];
}
__global__ __launch_bounds__(256) void k_decay_finish(const float *__restrict__ partial, float *__restrict__ out) {
    __shared__ float s_p[256];
    float a = 0.0f;
    for (int i = threadIdx.x; i < kDecayBlocks; i += 256) a += partial[i];
    s_p[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_p[threadIdx.x] += s_p[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s_p[0];
}

// nan_to_num_ (nan -> 0, +-inf -> +-FLT_MAX) on up to kManyMax small fp32 tensors in one launch: blockIdx.y = tensor
constexpr int kManyMax = 48;
struct ManyTensors {
    float *ptr[kManyMax];
    uint32_t n[kManyMax];
};
__global__ __launch_bounds__(256) void k_nan_to_num_many(ManyTensors t) {
    float *p = t.ptr[blockIdx.y];
    const uint32_t n = t.n[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float v = p[i];
        if (!(fabsf(v) <= 3.4028234663852886e38f)) p[i] = v != v ? 0.0f : (v > 0.0f ? 3.4028234663852886e38f : -3.4028234663852886e38f);
    }
}

// The same Adam step (k_adam_step's arithmetic, element by element) on up to kAdamMany SMALL tensors of one parameter
// group in one launch: blockIdx.y = tensor.  torch's foreach path takes seven launches (~105 us) for the 16 dense-layer
// parameters of the two fields.
constexpr int kAdamMany = 24;
struct AdamMany {
    float *p[kAdamMany], *g[kAdamMany], *m[kAdamMany], *v[kAdamMany];
    uint32_t n[kAdamMany];
};
__global__ __launch_bounds__(256) void k_adam_step_many(AdamMany t, float w1, float beta2, float w2, float bc2_sqrt, float eps, float neg_step,
                                                        int sanitize) {
    float *__restrict__ p = t.p[blockIdx.y], *__restrict__ g = t.g[blockIdx.y], *__restrict__ m = t.m[blockIdx.y],
                        *__restrict__ v = t.v[blockIdx.y];
    const uint32_t n = t.n[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float gk = sanitize ? nan_to_num0(g[i]) : g[i];
        if (sanitize) g[i] = gk;
        const float a = fmaf(w1, gk - m[i], m[i]);
        const float b = fmaf(w2 * gk, gk, v[i] * beta2);
        m[i] = a;
        v[i] = b;
        p[i] = fmaf(neg_step, a / (sqrtf(b) / bc2_sqrt + eps), p[i]);
    }
}

}  // namespace

extern "C" int ucn_adam_step_many(float *const *params_host, float *const *grads_host, float *const *exp_avg_host,
                                  float *const *exp_avg_sq_host, const uint64_t *numel_host, uint32_t count, float lr, float beta1,
                                  float beta2, float eps, uint32_t step, int sanitize_grad, ucn_stream_t stream) {
    UCN_REQUIRE(count == 0 || (params_host && grads_host && exp_avg_host && exp_avg_sq_host && numel_host),
                "adam_step_many: null pointer argument");
    UCN_REQUIRE(step >= 1, "adam_step_many: step counts from 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (uint32_t base = 0; base < count; base += kAdamMany) {
        AdamMany t{};
        const uint32_t k = count - base < (uint32_t)kAdamMany ? count - base : (uint32_t)kAdamMany;
        uint64_t biggest = 0;
        for (uint32_t i = 0; i < k; i++) {
            UCN_REQUIRE(numel_host[base + i] < 0xFFFFFFFFull, "adam_step_many: tensor %u too large", base + i);
            t.p[i] = params_host[base + i]; t.g[i] = grads_host[base + i];
            t.m[i] = exp_avg_host[base + i]; t.v[i] = exp_avg_sq_host[base + i];
            t.n[i] = (uint32_t)numel_host[base + i];
            biggest = numel_host[base + i] > biggest ? numel_host[base + i] : biggest;
        }
        if (biggest == 0) continue;
        const uint32_t bx = (uint32_t)(ucn_div_up(biggest, 256) < 512 ? ucn_div_up(biggest, 256) : 512);
        hipLaunchKernelGGL(k_adam_step_many, dim3(bx, k), dim3(256), 0, (hipStream_t)stream, t, 1.0f - beta1, beta2, 1.0f - beta2,
                           (float)sqrt(bc2), eps, (float)(-(double)lr / bc1), sanitize_grad);
        UCN_LAUNCH_CHECK("adam_step_many");
    }
    return 0;
}

extern "C" int ucn_hash_decay(const float *embeddings, const int32_t *offsets_host, uint32_t L, uint32_t C, const float *g_dev,
                              float *out, float *workspace, ucn_stream_t stream) {
    UCN_REQUIRE(embeddings && offsets_host && out, "hash_decay: null pointer argument");
    UCN_REQUIRE(L >= 1 && L <= (uint32_t)kDecayLevels && C >= 1, "hash_decay: 1..32 levels, got %u", L);
    UCN_REQUIRE(g_dev || workspace, "hash_decay: the forward needs a workspace of 1024 floats");
    DecayLevels lv{};
    lv.n = L;
    for (uint32_t l = 0; l <= L; l++) lv.off[l] = (uint32_t)offsets_host[l];
    for (uint32_t l = 0; l < L; l++) {
        UCN_REQUIRE(offsets_host[l + 1] > offsets_host[l], "hash_decay: level %u is empty", l);
        lv.w[l] = (float)(1.0 / ((double)(offsets_host[l + 1] - offsets_host[l]) * (double)L * (double)C));
    }
    if (g_dev) {
        hipLaunchKernelGGL(k_hash_decay<true>, dim3(kDecayBlocks), dim3(256), 0, (hipStream_t)stream, embeddings, C, lv, g_dev, out);
        UCN_LAUNCH_CHECK("hash_decay (backward)");
        return 0;
    }
    hipLaunchKernelGGL(k_hash_decay<false>, dim3(kDecayBlocks), dim3(256), 0, (hipStream_t)stream, embeddings, C, lv, (const float *)nullptr,
                       workspace);
    hipLaunchKernelGGL(k_decay_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, out);
    UCN_LAUNCH_CHECK("hash_decay");
    return 0;
}

extern "C" int ucn_nan_to_num_many(float *const *tensors_host, const uint64_t *numel_host, uint32_t count, ucn_stream_t stream) {
    UCN_REQUIRE(count == 0 || (tensors_host && numel_host), "nan_to_num_many: null pointer argument");
    for (uint32_t base = 0; base < count; base += kManyMax) {
        ManyTensors t{};
        const uint32_t m = count - base < (uint32_t)kManyMax ? count - base : (uint32_t)kManyMax;
        uint64_t biggest = 0;
        for (uint32_t i = 0; i < m; i++) {
            UCN_REQUIRE(numel_host[base + i] < 0xFFFFFFFFull, "nan_to_num_many: tensor %u too large", base + i);
            t.ptr[i] = tensors_host[base + i];
            t.n[i] = (uint32_t)numel_host[base + i];
            biggest = numel_host[base + i] > biggest ? numel_host[base + i] : biggest;
        }
        if (biggest == 0) continue;
        const uint32_t bx = (uint32_t)(ucn_div_up(biggest, 1024) < 2048 ? ucn_div_up(biggest, 1024) : 2048);
        hipLaunchKernelGGL(k_nan_to_num_many, dim3(bx ? bx : 1, m), dim3(256), 0, (hipStream_t)stream, t);
        UCN_LAUNCH_CHECK("nan_to_num_many");
    }
    return 0;
}

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

extern "C" int ucn_interlevel_loss(const float *c, const float *w, uint32_t S_nerf, const float *cp, const float *wp, uint32_t S_prop,
                                   float pulse_width, uint32_t N, float *loss_ray, float *dterm, ucn_stream_t stream) {
    if (N == 0) return 0;
    UCN_REQUIRE(c && w && cp && wp && loss_ray && dterm, "interlevel_loss: null pointer argument");
    UCN_REQUIRE(S_nerf >= 1 && S_nerf <= 512 && S_prop >= 1 && S_prop <= 1024, "interlevel_loss: unsupported sample counts %u / %u",
                S_nerf, S_prop);
    UCN_REQUIRE(pulse_width > 0.0f, "interlevel_loss: pulse width must be positive");
    const uint32_t n1 = S_nerf + 1, m = 2 * n1;
    const size_t lds = 4 * sizeof(float) * ((size_t)n1 + S_nerf + 4 * m + (S_prop + 1));
    hipLaunchKernelGGL(k_interlevel, dim3(ucn_div_up(N, 4)), dim3(256), lds, (hipStream_t)stream, c, w, S_nerf, cp, wp, S_prop, pulse_width,
                       N, loss_ray, dterm);
    UCN_LAUNCH_CHECK("interlevel_loss");
    return 0;
}

extern "C" int ucn_bias_relu(void *pre_inout, const void *per_ray, uint32_t N, uint32_t S, uint32_t W, int dtype, ucn_stream_t stream) {
    if ((uint64_t)N * S * W == 0) return 0;
    UCN_REQUIRE(pre_inout && per_ray, "bias_relu: null pointer argument");
    UCN_REQUIRE(W % 8 == 0, "bias_relu: width must be a multiple of 8, got %u", W);
    UCN_REQUIRE(dtype == 0 || dtype == 2, "bias_relu: dtype must be 0 (float32) or 2 (bfloat16)");
    const uint64_t total = (uint64_t)N * S * W;
    const dim3 grid((uint32_t)((total / 8 + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL(k_bias_relu<float>, grid, dim3(256), 0, (hipStream_t)stream, (float *)pre_inout, (const float *)per_ray, total, W, S);
    else hipLaunchKernelGGL(k_bias_relu<Bf16>, grid, dim3(256), 0, (hipStream_t)stream, (Bf16 *)pre_inout, (const Bf16 *)per_ray, total, W, S);
    UCN_LAUNCH_CHECK("bias_relu");
    return 0;
}

extern "C" int ucn_relu_backward_reduce(const void *gy, const void *h, void *d_pre, void *d_per_ray, uint32_t N, uint32_t S, uint32_t W,
                                        int dtype, ucn_stream_t stream) {
    if ((uint64_t)N * S * W == 0) return 0;
    UCN_REQUIRE(gy && h && d_pre && d_per_ray, "relu_backward_reduce: null pointer argument");
    UCN_REQUIRE(W % 8 == 0 && W / 8 <= 256 && 256 % (W / 8) == 0, "relu_backward_reduce: unsupported width %u", W);
    UCN_REQUIRE(dtype == 0 || dtype == 2, "relu_backward_reduce: dtype must be 0 (float32) or 2 (bfloat16)");
    if (dtype == 0)
        hipLaunchKernelGGL(k_relu_bwd_reduce<float>, dim3(N), dim3(256), 0, (hipStream_t)stream, (const float *)gy, (const float *)h,
                           (float *)d_pre, (float *)d_per_ray, S, W);
    else
        hipLaunchKernelGGL(k_relu_bwd_reduce<Bf16>, dim3(N), dim3(256), 0, (hipStream_t)stream, (const Bf16 *)gy, (const Bf16 *)h,
                           (Bf16 *)d_pre, (Bf16 *)d_per_ray, S, W);
    UCN_LAUNCH_CHECK("relu_backward_reduce");
    return 0;
}
