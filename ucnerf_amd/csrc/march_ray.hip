// Per-ray step-function kernels of the ray-march: resampling, cone basis, alpha compositing.
//
//   k_resample   <- stepfun.py:75-105 (max_dilate_weights), models.py:168-191 (trim, anneal,
//                   logits), stepfun.py:154-218,251-294 (softmax -> CDF -> inverse-CDF -> fenceposts)
//   k_cone_basis <- render.py:139-146
//   k_composite  <- render.py:155-174 (alpha weights), :177-244 (volumetric_rendering),
//                   stepfun.py:329-339 (weighted percentiles)
//
// The reference does the dilation and the inverse-CDF lookup with O(n*m) broadcast masks
// ([N,3n,n] and [N,n,m] temporaries in HBM).  Here a ray's step function never leaves the chip: k_resample and
// k_composite give one wave64 to each ray (step function in LDS, ranks by binary search, sums as wave scans): transmittance is a wave prefix-sum (DPP-free shuffles), the percentile
// lookups are ballots over the CDF held in LDS.
#include "ucn_common.h"
#include "wave_dpp.h"

namespace {

// ------------------------------------------------------------------ resample
// One wave64 per ray, 4 rays per workgroup; the ray's step function lives in LDS:
//   t[n+1], p[n]      previous fenceposts and pdf = weight / width                      (weight_to_pdf)
//   kn[m], wt[m-1]    the max-dilated step function, m = 3n + 1                         (max_dilate_weights)
//   cdf[nw+1], c[S]   CDF of the annealed, trimmed weights and the S inverse-CDF samples (sample_intervals)
// Every sequential walk of the reference (3-way merge of sorted lists, sliding-window maximum, cumulative sums,
// monotone inverse-CDF pointer) becomes a rank computation by binary search or a wave scan:
//   * rank of an element in the merge = own index + elements of the other two lists ordered before it, with the
//     sequential tie rule (shifted-down copy first, then the fencepost, then the shifted-up copy); only VALUES
//     reach the next stage, so ties cannot change the result;
//   * window bounds jlo / jhi = counts of shifted fenceposts <= the knot (both lists are sorted);
//   * sums accumulate in double like torch-CPU's cumsum (acc_type) -- a float sum over ~200-400 terms would drift by
//     ~1e-5 in the CDF, i.e. whole samples in t; the wave scan adds the same addends in a different order, which
//     is invisible after the cast back to float.
// (The first version ran one THREAD per ray with its arrays in scratch memory: 6.4 ms per 2.46 M rays; the stores
//  are now coalesced across the lanes of the ray's wave as well.)
#ifdef UCN_WAVE_SHFL          // the first form: butterflies / Hillis-Steele steps over __shfl (ds_bpermute_b32)
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_scan_d(double v, int lane) {      // inclusive
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#else
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_dpp<double>(v); }
__device__ __forceinline__ double wave_scan_d(double v, int) { return wave_scan_dpp<double>(v); }      // inclusive
__device__ __forceinline__ float wave_max_f(float v) { return wave_max_dpp(v); }
#endif
// #{j < len : f(j) <= x} / #{j < len : f(j) < x} for a non-decreasing f
template <bool STRICT, class F>
__device__ __forceinline__ uint32_t count_before(F f, uint32_t len, float x) {
    uint32_t lo = 0, hi = len;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const float v = f(mid);
        if (STRICT ? (v < x) : (v <= x)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The same count, started from a guess: gallop away from `hint` in the direction the predicate at the guess points to,
// then bisect the bracket.  The ranks of the 3-way merge and the window bounds of the max-pool are known to within a
// few positions from the element's own index (the dilation is of the order of one interval), so 2-3 probes replace
// the 7 of a cold binary search over 64 entries; the result is the exact count either way.
template <bool STRICT, class F>
__device__ __forceinline__ uint32_t count_from(F f, uint32_t len, float x, uint32_t hint) {
    auto before = [&](uint32_t j) { const float v = f(j); return STRICT ? (v < x) : (v <= x); };
    uint32_t lo = 0, hi = len;
    uint32_t c = hint < len ? hint : len;
    if (c < len && before(c)) {                  // the count exceeds c
        lo = c + 1;
        uint32_t step = 1;
        while (lo < hi) {
            const uint32_t probe = lo + step - 1 < hi - 1 ? lo + step - 1 : hi - 1;
            if (before(probe)) { lo = probe + 1; step <<= 1; }
            else { hi = probe; break; }
        }
    } else {                                     // the count is at most c
        hi = c;
        uint32_t step = 1;
        while (lo < hi) {
            const uint32_t probe = hi - lo > step ? hi - step : lo;
            if (!before(probe)) { hi = probe; step <<= 1; }
            else { lo = probe + 1; break; }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (before(mid)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_resample(const float *__restrict__ sd_prev, const float *__restrict__ w_prev,
                                                  uint32_t n_prev, float dilation, float anneal, float pad,
                                                  const float *__restrict__ u_table, const float *__restrict__ jitter,
                                                  uint32_t jcols, float max_jitter, uint32_t N, uint32_t S,
                                                  float *__restrict__ sd_out) {
    extern __shared__ float s_mem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ray_raw = blockIdx.x * 4u + wv;
    if (ray_raw >= N) return;                            // wave-uniform; every hand-off below is inside the wave
    const bool live = true;
    const uint32_t ray = ray_raw;
    const uint32_t n = n_prev, m = 3 * n + 1;            // n >= 1: the first level is k_resample_first
    float *t = s_mem + (size_t)wv * ((n + 1) + n + (m + 1) + m + (m + 1) + S);
    float *p = t + (n + 1), *kn = p + n, *wt = kn + (m + 1), *cdf = wt + m, *c = cdf + (m + 1);
    const float *sd;                                     // fenceposts actually resampled (after the [1:-1] trim)
    const float *w;
    uint32_t nw;
    if (!(dilation > 0.0f)) {
        // models.py:167-168 use_dilation == False (both dilation knobs 0): the previous level's fenceposts and weights
        // are resampled as they are -- no envelope, no [1:-1] trim, no renormalisation
        for (uint32_t i = lane; i <= n; i += 64) t[i] = sd_prev[(size_t)ray * (n + 1) + i];
        for (uint32_t i = lane; i < n; i += 64) wt[i] = w_prev[(size_t)ray * n + i];
        sd = t; w = wt; nw = n;
        wave_lds_handoff();
    } else {
        for (uint32_t i = lane; i <= n; i += 64) t[i] = sd_prev[(size_t)ray * (n + 1) + i];
        wave_lds_handoff();
        for (uint32_t i = lane; i < n; i += 64)
            p[i] = w_prev[(size_t)ray * n + i] / fmaxf(t[i + 1] - t[i], UCN_EPS);       // weight_to_pdf
        // sort(cat[t, t0-d, t1+d]) == ranks in the 3-way merge of three sorted lists, then clip to [0,1]
        auto A = [&](uint32_t j) { return t[j]; };
        auto Bq = [&](uint32_t j) { return t[j] - dilation; };
        auto Cq = [&](uint32_t j) { return t[j + 1] + dilation; };
        uint32_t *src = reinterpret_cast<uint32_t *>(cdf);     // which t[] index a knot came from (cdf[] is free until the softmax)
        for (uint32_t e = lane; e < m; e += 64) {
            float v;
            uint32_t pos, near;
            if (e <= n) {
                v = A(e);
                near = e;
                pos = e + count_from<false>(Bq, n, v, e + 1) + count_from<true>(Cq, n, v, e >= 1 ? e - 1 : 0);
            } else if (e <= 2 * n) {
                const uint32_t i = e - (n + 1);
                v = Bq(i);
                near = i;
                pos = i + count_from<true>(A, n + 1, v, i) + count_from<true>(Cq, n, v, i >= 1 ? i - 1 : 0);
            } else {
                const uint32_t i = e - (2 * n + 1);
                v = Cq(i);
                near = i + 1;
                pos = i + count_from<false>(A, n + 1, v, i + 2) + count_from<false>(Bq, n, v, i + 2);
            }
            kn[pos] = fminf(fmaxf(v, 0.0f), 1.0f);
            src[pos] = near;
        }
        wave_lds_handoff();
        // max-pool the pdf over the dilated intervals covering each knot interval, times its width
        double part = 0.0;
        for (uint32_t i = lane; i + 1 < m; i += 64) {
            const float k = kn[i];
            const uint32_t near = src[i];
            const uint32_t jhi = count_from<false>(Bq, n, k, near + 1);                        // intervals with t0 - d <= k
            const uint32_t jlo = count_from<false>(Cq, n, k, near >= 1 ? near - 1 : 0);        // intervals with t1 + d <= k end before k
            float env = 0.0f;
            for (uint32_t j = jlo; j < jhi; j++) env = fmaxf(env, p[j]);
            const float wv_ = env * (kn[i + 1] - kn[i]);                 // pdf_to_weight
            wt[i] = wv_;
            part += (double)wv_;
        }
        const float norm = fmaxf((float)wave_sum_d(part), UCN_EPS);
        for (uint32_t i = lane; i + 1 < m; i += 64) wt[i] = wt[i] / norm;
        sd = kn + 1; w = wt + 1; nw = m - 3;                             // models.py:175-176
        wave_lds_handoff();
    }
    // logits -> softmax -> CDF; lane owns CH consecutive intervals so that the prefix sum is a wave scan.  The logit
    // and then the exponential of an interval are parked in the lane's own cdf[] slots (one logf / expf each).
    const uint32_t CH = (nw + 63) / 64;
    float mx = -INFINITY;
    for (uint32_t q = 0; q < CH; q++) {
        const uint32_t i = lane * CH + q;
        if (i < nw) {
            const float lg = (sd[i + 1] > sd[i]) ? anneal * logf(w[i] + pad) : -INFINITY;
            cdf[i] = lg;
            mx = fmaxf(mx, lg);
        }
    }
    mx = wave_max_f(mx);
    double zpart = 0.0;
    for (uint32_t q = 0; q < CH; q++) {
        const uint32_t i = lane * CH + q;
        if (i < nw) {
            const float e = expf(cdf[i] - mx);
            cdf[i] = e;
            zpart += (double)e;
        }
    }
    const float z = (float)wave_sum_d(zpart);
    double run = 0.0;
    for (uint32_t q = 0; q < CH; q++) {
        const uint32_t i = lane * CH + q;
        if (i < nw) run += (double)(cdf[i] / z);
    }
    run = wave_scan_d(run, lane) - run;                                  // sum of pw over the lanes before this one
    // cdf[0] = 0, cdf[i] = min(1, sum_{k<i} pw_k) for i < nw, cdf[nw] = 1  (stepfun.py:123-127)
    for (uint32_t q = 0; q < CH; q++) {
        const uint32_t i = lane * CH + q;
        if (i < nw) {
            const float pw = cdf[i] / z;
            cdf[i] = fminf((float)run, 1.0f);                            // run == 0 exactly for i == 0
            run += (double)pw;
        }
    }
    if (lane == 0) cdf[nw] = 1.0f;
    wave_lds_handoff();
    // inverse CDF at the sorted u's (stepfun.py:283-293): idx = #{1 <= j <= nw : cdf[j] <= u}
    const float jit = jitter ? jitter[(size_t)ray * jcols] : 0.0f;
    auto Cdf1 = [&](uint32_t j) { return cdf[j + 1]; };
    for (uint32_t k = lane; k < S; k += 64) {
        float u = u_table[k];
        if (jitter) u = u + (jcols > 1 ? jitter[(size_t)ray * jcols + k] : jit) * max_jitter;
        const uint32_t idx = count_before<false>(Cdf1, nw, u);
        const uint32_t i1 = idx + 1 <= nw ? idx + 1 : nw;
        const float x0 = cdf[idx], x1 = cdf[i1];
        float fr = (u - x0) / (x1 - x0);
        if (fr != fr) fr = 0.0f;                                         // nan_to_num(.., 0); +-inf clip below
        fr = fminf(fmaxf(fr, 0.0f), 1.0f);
        const float f0 = sd[idx], f1 = sd[i1];
        c[k] = f0 + fr * (f1 - f0);
    }
    wave_lds_handoff();
    // midpoints + reflected / clamped ends; the S + 1 outputs of the ray leave as coalesced stores
    if (!live) return;
    float *out = sd_out + (size_t)ray * (S + 1);
    for (uint32_t k = lane; k <= S; k += 64) {
        float v;
        if (k == 0) v = fmaxf(2.0f * c[0] - (c[1] + c[0]) / 2.0f, 0.0f);
        else if (k == S) v = fminf(2.0f * c[S - 1] - (c[S - 1] + c[S - 2]) / 2.0f, 1.0f);
        else v = (c[k] + c[k - 1]) / 2.0f;
        out[k] = v;
    }
}

// First level (n_prev == 0): the step function is the single interval [0, 1] with weight 1, so cdf = [0, 1], the
// softmax is exactly 1 and every sample is c_k = clamp(u_k, 0, 1) (stepfun.py:283-293 with fp = xp = [0, 1]: the
// interpolation 0 + fr * (1 - 0) is exact).  One thread per output fencepost.
__global__ __launch_bounds__(256) void k_resample_first(const float *__restrict__ u_table, const float *__restrict__ jitter,
                                                        uint32_t jcols, float max_jitter, uint32_t N, uint32_t S,
                                                        float *__restrict__ sd_out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)N * (S + 1)) return;
    const uint32_t ray = (uint32_t)(i / (S + 1)), k = (uint32_t)(i - (uint64_t)ray * (S + 1));
    auto cu = [&](uint32_t q) {
        float u = u_table[q];
        if (jitter) u = u + jitter[(size_t)ray * jcols + (jcols > 1 ? q : 0u)] * max_jitter;
        return fminf(fmaxf(u, 0.0f), 1.0f);
    };
    float v;
    if (k == 0) v = fmaxf(2.0f * cu(0) - (cu(1) + cu(0)) / 2.0f, 0.0f);
    else if (k == S) v = fminf(2.0f * cu(S - 1) - (cu(S - 1) + cu(S - 2)) / 2.0f, 1.0f);
    else v = (cu(k) + cu(k - 1)) / 2.0f;
    sd_out[i] = v;
}

// ------------------------------------------------------------------ cone basis
__global__ __launch_bounds__(256) void k_cone_basis(const float *__restrict__ cam, const float *__restrict__ rnd,
                                                    uint32_t N, float *__restrict__ basis) {
    const uint32_t ray = blockIdx.x * 256u + threadIdx.x;
    if (ray >= N) return;
    const float a0 = cam[ray * 3 + 0], a1 = cam[ray * 3 + 1], a2 = cam[ray * 3 + 2];
    const float b0 = rnd[ray * 3 + 0], b1 = rnd[ray * 3 + 1], b2 = rnd[ray * 3 + 2];
    // ortho1 = normalize(cam x rand)  (F.normalize: v / max(|v|, 1e-12))
    float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
    float n = fmaxf(sqrtf((c0 * c0 + c1 * c1) + c2 * c2), 1e-12f);
    c0 /= n; c1 /= n; c2 /= n;
    // ortho2 = normalize(cam x ortho1)
    float d0 = a1 * c2 - a2 * c1, d1 = a2 * c0 - a0 * c2, d2 = a0 * c1 - a1 * c0;
    n = fmaxf(sqrtf((d0 * d0 + d1 * d1) + d2 * d2), 1e-12f);
    d0 /= n; d1 /= n; d2 /= n;
    float *o = basis + (size_t)ray * 6;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = d0; o[4] = d1; o[5] = d2;
}

// ------------------------------------------------------------------ composite
#ifdef UCN_WAVE_SHFL
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_excl(float incl, int lane) {      // inclusive -> exclusive (0 in lane 0)
    const float u = __shfl_up(incl, 1, 64);
    return lane == 0 ? 0.0f : u;
}
__device__ __forceinline__ uint32_t wave_count(uint32_t cnt) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    return cnt;
}
#else
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp<float>(v); }
__device__ __forceinline__ float wave_scan(float v, int) { return wave_scan_dpp<float>(v); }            // inclusive
__device__ __forceinline__ float wave_excl(float incl, int) { return wave_shift_up1<float>(incl); }    // -> exclusive
__device__ __forceinline__ uint32_t wave_count(uint32_t cnt) { return wave_sum_dpp<uint32_t>(cnt); }
#endif
__device__ __forceinline__ float nan_to_num_inf(float v) {
    // torch.nan_to_num(x, nan=inf): NaN -> +inf (as given), +inf -> FLT_MAX, -inf -> -FLT_MAX
    if (v != v) return INFINITY;
    if (v == INFINITY) return 3.4028234663852886e38f;
    if (v == -INFINITY) return -3.4028234663852886e38f;
    return v;
}

// One wave64 per ray, 4 rays per workgroup.  Lane owns CH consecutive samples.
template <int CH>
__global__ __launch_bounds__(256) void k_composite(const float *__restrict__ density, const float *__restrict__ rgbs,
                                                   const float *__restrict__ sdist, const float *__restrict__ near_,
                                                   const float *__restrict__ far_, const float *__restrict__ dirs,
                                                   float bg, int opaque, uint32_t N, uint32_t S,
                                                   float *__restrict__ weights_out, float *__restrict__ out_main,
                                                   float *__restrict__ out_extras) {
    __shared__ float s_cdf[4][64 * CH + 2];
    __shared__ float s_t[4][64 * CH + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ray_raw = blockIdx.x * 4u + wv;
    const bool live = ray_raw < N;                       // wave-uniform; dead waves still reach the barrier
    const uint32_t ray = live ? ray_raw : N - 1;
    const float nr = near_[ray], fr = far_[ray];
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float *sd = sdist + (size_t)ray * (S + 1);
    float tlo[CH], thi[CH], tau[CH];
    float lane_tau = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        if (i < S) {
            const float s0 = sd[i], s1 = sd[i + 1];
            tlo[c] = s0 * fr + (1.0f - s0) * nr;          // coord.py:176 with fn = identity
            thi[c] = s1 * fr + (1.0f - s1) * nr;
            float td = density[(size_t)ray * S + i] * ((thi[c] - tlo[c]) * dnorm);
            if (opaque && i == S - 1) td = INFINITY;
            tau[c] = td;
        } else {
            tlo[c] = thi[c] = 0.0f;
            tau[c] = 0.0f;
        }
        lane_tau += tau[c];
    }
    // exclusive prefix of tau over the ray = transmittance exponent
    // (shifted inclusive scan, not `inclusive - own`: with opaque_background the last tau is +inf)
    float before = wave_excl(wave_scan(lane_tau, lane), lane);
    float acc = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f, dnum = 0.0f, lnum = 0.0f;
    float wloc[CH];
    float lane_w = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        float w = 0.0f;
        if (i < S) {
            const float alpha = 1.0f - expf(-tau[c]);
            const float trans = expf(-before);
            w = alpha * trans;
            if (live) weights_out[(size_t)ray * S + i] = w;
            const float tm = 0.5f * (tlo[c] + thi[c]);
            if (rgbs) {
                const float *col = rgbs + ((size_t)ray * S + i) * 3;
                r += w * col[0]; g += w * col[1]; b += w * col[2];
            }
            dnum += w * tm;
            lnum += w * logf(tm);
            before += tau[c];
        }
        wloc[c] = w;
        lane_w += w;
    }
    acc = wave_sum(lane_w);
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
    dnum = wave_sum(dnum);
    lnum = wave_sum(lnum);
    const float t_first = sd[0] * fr + (1.0f - sd[0]) * nr;
    const float t_last = sd[S] * fr + (1.0f - sd[S]) * nr;
    const float bg_w = fmaxf(1.0f - acc, 0.0f);
    const float denom = fmaxf(acc, UCN_EPS);
    float depth = fminf(fmaxf(nan_to_num_inf(dnum / denom), t_first), t_last);
    if (acc < 0.6f) depth = 300.0f;                                  // render.py:208-213
    if (lane == 0 && live) {
        float *o = out_main + (size_t)ray * 5;
        o[0] = r + bg_w * bg; o[1] = g + bg_w * bg; o[2] = b + bg_w * bg;
        o[3] = depth; o[4] = acc;
    }
    if (!out_extras) return;
    // CDF of [w_0..w_{S-1}, bg_w] at the S+2 fenceposts [t_0..t_S, far]  (render.py:234-238)
    float incl = wave_excl(wave_scan(lane_w, lane), lane);
    float *cdf = s_cdf[wv], *tt = s_t[wv];
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        if (i < S) {
            incl += wloc[c];
            cdf[i + 1] = fminf(incl, 1.0f);
            tt[i] = tlo[c];
        }
    }
    if (lane == 0) { cdf[0] = 0.0f; cdf[S + 1] = 1.0f; tt[S] = t_last; tt[S + 1] = fr; }
    wave_lds_handoff();
    float pct[3];
    const float ps[3] = {0.05f, 0.5f, 0.95f};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        // #{i in [0,S+1] : cdf[i] <= p}; the CDF is non-decreasing so this is a prefix
        uint32_t cnt = 0;
        for (uint32_t i = lane; i <= S + 1; i += 64) cnt += (cdf[i] <= ps[q]) ? 1u : 0u;
        cnt = wave_count(cnt);
        const uint32_t i0 = cnt > 0 ? cnt - 1 : 0;
        const uint32_t i1 = cnt <= S + 1 ? cnt : S + 1;
        const float x0 = cdf[i0], x1 = cdf[i1];
        float f = (ps[q] - x0) / (x1 - x0);
        if (f != f) f = 0.0f;
        f = fminf(fmaxf(f, 0.0f), 1.0f);
        pct[q] = tt[i0] + f * (tt[i1] - tt[i0]);
    }
    if (lane == 0 && live) {
        float *e = out_extras + (size_t)ray * 4;
        e[0] = fminf(fmaxf(nan_to_num_inf(expf(lnum / denom)), t_first), t_last);
        e[1] = pct[0]; e[2] = pct[1]; e[3] = pct[2];
    }
}

// Backward of k_composite's differentiable outputs (weights, rgb, depth, acc) w.r.t. density and rgbs: what autograd
// derives from render.py:155-174 + :203-216, as one wave per ray.  With tau_i = density_i * delta_i,
// w_i = (1 - e^{-tau_i}) T_i, T_i = exp(-sum_{k<i} tau_k):
//     dL/dtau_j = G_j e^{-tau_j} T_j - sum_{i>j} G_i w_i,      G_i = total gradient arriving at w_i
// (the suffix sum is a wave scan).  G collects the direct weight gradient (the losses on `weights`), rgb
// (sum_i w_i c_i + bg * clamp_min(1 - acc, 0)), acc and depth (clip(nan_to_num(sum_i w_i tm_i / max(acc, eps))), 300
// where acc < 0.6 -- constant there).
template <int CH>
__global__ __launch_bounds__(256) void k_composite_bwd(const float *__restrict__ density, const float *__restrict__ rgbs,
                                                       const float *__restrict__ sdist, const float *__restrict__ near_,
                                                       const float *__restrict__ far_, const float *__restrict__ dirs,
                                                       float bg, int opaque, uint32_t N, uint32_t S,
                                                       const float *__restrict__ g_weights, const float *__restrict__ g_main,
                                                       float *__restrict__ g_density, float *__restrict__ g_rgbs) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t ray = blockIdx.x * 4u + wv;
    if (ray >= N) return;                                 // wave-uniform, no barriers below
    const float nr = near_[ray], fr = far_[ray];
    const float dx = dirs[ray * 3 + 0], dy = dirs[ray * 3 + 1], dz = dirs[ray * 3 + 2];
    const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float *sd = sdist + (size_t)ray * (S + 1);
    const float *gm = g_main + (size_t)ray * 5;
    const float gr = gm[0], gg = gm[1], gb = gm[2], gdepth = gm[3], gacc = gm[4];
    float tm[CH], scale[CH], tau[CH], w[CH], att[CH];
    float lane_tau = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        tm[c] = scale[c] = tau[c] = 0.0f;
        if (i < S) {
            const float s0 = sd[i], s1 = sd[i + 1];
            const float tlo = s0 * fr + (1.0f - s0) * nr, thi = s1 * fr + (1.0f - s1) * nr;
            tm[c] = 0.5f * (tlo + thi);
            scale[c] = (thi - tlo) * dnorm;
            float td = density[(size_t)ray * S + i] * scale[c];
            if (opaque && i == S - 1) { td = INFINITY; scale[c] = 0.0f; }       // a constant: no gradient to that density
            tau[c] = td;
        }
        lane_tau += tau[c];
    }
    float before = wave_excl(wave_scan(lane_tau, lane), lane);
    float lane_w = 0.0f, dnum = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        w[c] = att[c] = 0.0f;
        if (i < S) {
            const float e = expf(-tau[c]), trans = expf(-before);
            w[c] = (1.0f - e) * trans;
            att[c] = e * trans;                                               // dw_i / dtau_i
            dnum += w[c] * tm[c];
            before += tau[c];
        }
        lane_w += w[c];
    }
    const float acc = wave_sum(lane_w);
    dnum = wave_sum(dnum);
    // depth = clip(nan_to_num(dnum / max(acc, eps)), t_0, t_S), overwritten by 300 where acc < 0.6
    const float t_first = sd[0] * fr + (1.0f - sd[0]) * nr, t_last = sd[S] * fr + (1.0f - sd[S]) * nr;
    const float denom = fmaxf(acc, UCN_EPS);
    const float draw = dnum / denom;
    const bool dlive = gdepth != 0.0f && !(acc < 0.6f) && draw == draw && fabsf(draw) != INFINITY && draw >= t_first &&
                       draw <= t_last;
    const float gd_num = dlive ? gdepth / denom : 0.0f;                                        // d depth / d dnum
    const float gd_acc = dlive && acc >= UCN_EPS ? -gdepth * draw / denom : 0.0f;              // through max(acc, eps)
    const float g_bgw = (1.0f - acc >= 0.0f) ? -(gr + gg + gb) * bg : 0.0f;                    // clamp_min(1 - acc, 0)
    float G[CH], lane_p = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        G[c] = 0.0f;
        if (i < S) {
            float g = gacc + g_bgw + gd_acc + gd_num * tm[c];
            if (g_weights) g += g_weights[(size_t)ray * S + i];
            if (rgbs) {
                const float *col = rgbs + ((size_t)ray * S + i) * 3;
                g += (gr * col[0] + gg * col[1]) + gb * col[2];
                float *o = g_rgbs + ((size_t)ray * S + i) * 3;
                o[0] = w[c] * gr; o[1] = w[c] * gg; o[2] = w[c] * gb;
            }
            G[c] = g;
            lane_p += g * w[c];
        }
    }
    // sum_{i > j} G_i w_i = total - inclusive prefix
    const float incl_lane = wave_scan(lane_p, lane);
    const float total = __shfl(incl_lane, 63, 64);
    float incl = incl_lane - lane_p;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        const uint32_t i = lane * CH + c;
        if (i < S) {
            incl += G[c] * w[c];
            const float gtau = G[c] * att[c] - (total - incl);
            g_density[(size_t)ray * S + i] = gtau * scale[c];
        }
    }
}

}  // namespace

extern "C" int ucn_resample(const float *sdist_prev, const float *weights_prev, uint32_t n_prev, float dilation,
                            float anneal, float resample_padding, const float *u_table, const float *jitter,
                            uint32_t jitter_cols, float max_jitter, uint32_t N, uint32_t S, float *sdist_out,
                            ucn_stream_t stream) {
    UCN_REQUIRE(S > 1, "num_samples must be > 1, is %u.", S);                      // stepfun.py:271-272
    UCN_REQUIRE(N == 0 || (u_table && sdist_out), "resample: null pointer argument");
    UCN_REQUIRE(N == 0 || n_prev == 0 || (sdist_prev && weights_prev), "resample: previous level missing");
    UCN_REQUIRE(n_prev <= 256, "resample: at most 256 intervals per level are supported, got %u", n_prev);
    UCN_REQUIRE(!jitter || jitter_cols == 1 || jitter_cols == S, "resample: jitter must be [N,1] or [N,S]");
    if (N == 0) return 0;
    UCN_REQUIRE(S <= 1024, "resample: at most 1024 samples per ray are supported, got %u", S);
    if (n_prev == 0) {
        hipLaunchKernelGGL(k_resample_first, dim3((uint32_t)(((uint64_t)N * (S + 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           u_table, jitter, jitter_cols, max_jitter, N, S, sdist_out);
        UCN_LAUNCH_CHECK("resample (first level)");
        return 0;
    }
    const uint32_t m = 3 * n_prev + 1;
    const size_t lds = 4 * sizeof(float) * ((size_t)(n_prev + 1) + n_prev + (m + 1) + m + (m + 1) + S);
    hipLaunchKernelGGL(k_resample, dim3(ucn_div_up(N, 4)), dim3(256), lds, (hipStream_t)stream, sdist_prev, weights_prev, n_prev,
                       dilation, anneal, resample_padding, u_table, jitter, jitter_cols, max_jitter, N, S, sdist_out);
    UCN_LAUNCH_CHECK("resample");
    return 0;
}

extern "C" int ucn_cone_basis(const float *cam_dirs, const float *rand_vec, uint32_t N, float *basis_out,
                              ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (cam_dirs && rand_vec && basis_out), "cone_basis: null pointer argument");
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_cone_basis, dim3(ucn_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, cam_dirs, rand_vec, N, basis_out);
    UCN_LAUNCH_CHECK("cone_basis");
    return 0;
}

extern "C" int ucn_composite(const float *density, const float *rgbs, const float *sdist, const float *near_,
                             const float *far_, const float *directions, float bg_intensity, int opaque_background,
                             uint32_t N, uint32_t S, float *weights_out, float *out_main, float *out_extras,
                             ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (density && sdist && near_ && far_ && directions && weights_out && out_main), "composite: null pointer argument");
    UCN_REQUIRE(S >= 1 && S <= 512, "composite: samples per ray must be in [1,512], got %u", S);
    if (N == 0) return 0;
    const dim3 grid(ucn_div_up(N, 4));
    hipStream_t st = (hipStream_t)stream;
#define UCN_CP(CH)                                                                                               \
    hipLaunchKernelGGL(k_composite<CH>, grid, dim3(256), 0, st, density, rgbs, sdist, near_, far_, directions,   \
                       bg_intensity, opaque_background, N, S, weights_out, out_main, out_extras)
    if (S <= 64) UCN_CP(1);
    else if (S <= 128) UCN_CP(2);
    else if (S <= 256) UCN_CP(4);
    else UCN_CP(8);
#undef UCN_CP
    UCN_LAUNCH_CHECK("composite");
    return 0;
}

namespace {
// alive list of a pass.  The features of a pass are indexed b = s * N + ray (rays_fastest: the lanes of an MLP wave are neighbouring
// rays at one sample) or b = ray * S + s; the compositing weights lie [N][S].  A workgroup takes a TILE of 64 rays x 64 samples:
// the weights are read in memory order (a wave = one ray's 64 samples: coalesced) and balloted into one 64-bit mask per ray in
// LDS; the list is then written in b order inside the tile (a wave = 64 rays at one sample: ballot + popcount give the rank) so that
// consecutive list items are consecutive feature rows for the colour kernel's gathers; ONE atomic per tile reserves its span
// (r03: one returning atomic per WAVE on a single counter -- 20 480 same-address atomics per pass, 236 us; r04: 320, ~10 us).
// The order of the tiles in the list is unspecified.
__global__ __launch_bounds__(256) void k_compact_alive(const float *__restrict__ weights, uint32_t N, uint32_t S, int rays_fastest,
                                                       float min_weight, uint32_t *__restrict__ idx, uint32_t *__restrict__ count) {
    __shared__ unsigned long long s_mask[64];
    __shared__ uint32_t s_cnt[4], s_base;
    const uint32_t s_tiles = (S + 63u) / 64u;
    const uint32_t ray0 = (blockIdx.x / s_tiles) * 64u, s0 = (blockIdx.x % s_tiles) * 64u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t r = wave; r < 64u; r += 4u) {                        // wave-uniform r: this wave's rays of the tile
        const uint32_t ray = ray0 + r, sm = s0 + lane;
        bool alive = false;
        if (ray < N && sm < S) alive = weights[(size_t)ray * S + sm] >= min_weight;       // NaN weights are not alive
        const unsigned long long m = __ballot(alive);
        if (lane == 0) s_mask[r] = m;
    }
    __syncthreads();
    // b order inside the tile: rays_fastest -> sample-major (a wave step = the 64 rays at one sample, i.e. bit `s` of the 64 ray masks);
    // ray-major -> the masks as they are (a wave step = one ray).  16 steps per wave; pass 1 counts, pass 2 writes.
    const unsigned long long mine = s_mask[lane];                        // rays_fastest: lane = ray; else only used via s_mask[step]
    uint32_t total = 0;
    for (uint32_t step = wave; step < 64u; step += 4u) {
        const unsigned long long m = rays_fastest ? __ballot((mine >> step) & 1ull) : s_mask[step];
        total += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_cnt[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = t ? atomicAdd(count, t) : 0u;
    }
    __syncthreads();
    uint32_t pos = s_base;
    for (uint32_t w = 0; w < wave; w++) pos += s_cnt[w];
    for (uint32_t step = wave; step < 64u; step += 4u) {
        const unsigned long long m = rays_fastest ? __ballot((mine >> step) & 1ull) : s_mask[step];
        if ((m >> lane) & 1ull) {
            // rays_fastest: lane = ray, step = sample;  ray-major: step = ray, lane = sample
            const uint32_t ray = ray0 + (rays_fastest ? lane : step), sm = s0 + (rays_fastest ? step : lane);
            const uint32_t b = rays_fastest ? sm * N + ray : ray * S + sm;
            idx[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = b;
        }
        pos += (uint32_t)__popcll(m);
    }
}
__global__ void k_zero_u32(uint32_t *p) { *p = 0u; }
}  // namespace

extern "C" int ucn_compact_alive(const float *weights, uint32_t N, uint32_t S, int rays_fastest, float min_weight,
                                 uint32_t *idx_out, uint32_t *count, ucn_stream_t stream) {
    UCN_REQUIRE(count && (N == 0 || (weights && idx_out)), "compact_alive: null pointer argument");
    UCN_REQUIRE((uint64_t)N * S <= 0xFFFFFF00ull, "compact_alive: too many samples in one call");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_zero_u32, dim3(1), dim3(1), 0, st, count);
    if (N)
        hipLaunchKernelGGL(k_compact_alive, dim3(ucn_div_up(N, 64) * ucn_div_up(S, 64)), dim3(256), 0, st, weights, N, S, rays_fastest,
                           min_weight, idx_out, count);
    UCN_LAUNCH_CHECK("compact_alive");
    return 0;
}

extern "C" int ucn_composite_backward(const float *density, const float *rgbs, const float *sdist, const float *near_,
                                      const float *far_, const float *directions, float bg_intensity, int opaque_background,
                                      uint32_t N, uint32_t S, const float *g_weights, const float *g_main,
                                      float *g_density, float *g_rgbs, ucn_stream_t stream) {
    UCN_REQUIRE(N == 0 || (density && sdist && near_ && far_ && directions && g_main && g_density), "composite_backward: null pointer argument");
    UCN_REQUIRE((rgbs == nullptr) == (g_rgbs == nullptr), "composite_backward: rgbs and g_rgbs come together");
    UCN_REQUIRE(S >= 1 && S <= 512, "composite_backward: samples per ray must be in [1,512], got %u", S);
    if (N == 0) return 0;
    const dim3 grid(ucn_div_up(N, 4));
    hipStream_t st = (hipStream_t)stream;
#define UCN_CB(CH)                                                                                                \
    hipLaunchKernelGGL(k_composite_bwd<CH>, grid, dim3(256), 0, st, density, rgbs, sdist, near_, far_, directions, \
                       bg_intensity, opaque_background, N, S, g_weights, g_main, g_density, g_rgbs)
    if (S <= 64) UCN_CB(1);
    else if (S <= 128) UCN_CB(2);
    else if (S <= 256) UCN_CB(4);
    else UCN_CB(8);
#undef UCN_CB
    UCN_LAUNCH_CHECK("composite_backward");
    return 0;
}
