// Multi-resolution hash-grid operator for gfx950 -- the `_gridencoder` ABI (include/ucnerf_march.h).
//
// Replaces /root/reference/nerf/gridencoder/src/gridencoder.cu (kernel_grid :87-245,
// kernel_grid_backward :248-340, kernel_input_backward :343-369, kernel_grad_tv :506-610).
// Written for CDNA4: 256-thread workgroups (4 wave64), level-major grid so one level's table
// slice is what the XCD L2s hold at a time, per-level constants in kernel arguments (SGPRs)
// instead of per-thread exp2f/ceil, power-of-two row masks instead of integer modulo.
//
// Numerics: the arithmetic spelling (explicit fmaf where nvcc -fmad contracts, nothing else
// fused; built with -ffp-contract=off) is the one oracle/grid_oracle.c documents, so the forward
// pass is bit-identical to the oracle.
#include <stdarg.h>

#include "ucn_common.h"

thread_local char g_ucn_err[512] = {0};

int ucn_fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ucn_err, sizeof(g_ucn_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char *ucn_last_error(void) { return g_ucn_err; }
extern "C" uint32_t ucn_abi_version(void) { return 26; }

int ucn_build_levels(UcnLevels *out, const int32_t *offsets, const int32_t *grid_sizes, uint32_t L, uint32_t C,
                     uint32_t D, float S, uint32_t H, uint32_t gridtype, int align_corners) {
    UCN_REQUIRE(L >= 1 && L <= UCN_MAX_LEVELS, "GridEncoding: num_levels must be in [1,%d], got %u", UCN_MAX_LEVELS, L);
    UCN_REQUIRE(D >= 2 && D <= 5, "GridEncoding: D must be 2, 3, 4 or 5.");
    memset(out, 0, sizeof(*out));
    out->L = L;
    out->C = C;
    for (uint32_t l = 0; l < L; l++) {
        UcnLevel &lv = out->lv[l];
        UCN_REQUIRE(offsets[l + 1] > offsets[l], "GridEncoding: offsets must be increasing");
        lv.first_row = (uint32_t)offsets[l];
        lv.rows = (uint32_t)(offsets[l + 1] - offsets[l]);
        lv.scale = exp2f((float)l * S) * (float)H - 1.0f;   // gridencoder.cu:138
        lv.resolution = (uint32_t)ceilf(lv.scale) + 1u;      // gridencoder.cu:139
        lv.mask = (lv.rows & (lv.rows - 1)) == 0 ? lv.rows - 1 : 0;
        // gridencoder.cu:68-81, evaluated once per level with the same uint32 arithmetic
        const uint32_t side = align_corners ? lv.resolution : lv.resolution + 1u;
        uint32_t stride = 1;
        for (uint32_t d = 0; d < D && stride <= lv.rows; d++) {
            lv.stride[d] = stride;
            stride *= side;
        }
        lv.hashed = (gridtype == 0u && stride > lv.rows) ? 1u : 0u;
        if (grid_sizes) {
            const int32_t g = grid_sizes[l];
            lv.gs2 = (float)(int32_t)((uint32_t)g * (uint32_t)g);   // int32 wrap, models.py:495
            lv.inv_gs = 1.0f / sqrtf(lv.gs2);                       // NaN if the wrap went negative, like torch
        }
    }
    return 0;
}

namespace {

__constant__ const uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                          2097192037u, 1434869437u, 2165219737u};

template <uint32_t D>
__device__ __forceinline__ uint32_t row_of(const UcnLevel &lv, const uint32_t (&cell)[D]) {
    uint32_t idx = 0;
    if (lv.hashed) {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) idx ^= cell[d] * kPrimes[d];
    } else {
#pragma unroll
        for (uint32_t d = 0; d < D; d++) idx += cell[d] * lv.stride[d];
    }
    return lv.mask ? (idx & lv.mask) : (idx % lv.rows);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }

// acc += a * b in the reference's scalar_t arithmetic.  float: nvcc contracts to one fmaf.  at::Half: c10::Half has
// no mixed accumulate -- the float product is rounded to half, the sum is formed in float and rounded again
// (oracle/grid_oracle.c, "fp16 tables"; pinned against torch/headeronly/util/Half.h by tests/test_oracle_grid.py).
// The product must be ROUNDED TO float before it is converted: the backend otherwise selects v_fma_mixlo_f16 for
// half(a * float(half)), which rounds the exact product once (1 half ulp off the reference's float multiply +
// conversion in ~2e-4 of the cases; seen with C = 1).  An empty asm on the value keeps the two roundings apart.
__device__ __forceinline__ float f32_product(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
template <typename T>
__device__ __forceinline__ T acc_mul(float a, float b, T acc) {
    if constexpr (sizeof(T) == 4) return fmaf(a, b, acc);
    else return from_f<T>(to_f<T>(acc) + to_f<T>(from_f<T>(f32_product(a, b))));
}

template <uint32_t D>
__device__ __forceinline__ bool outside(const float *x) {
    bool o = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) o |= (x[d] < 0.0f) | (x[d] > 1.0f);
    return o;
}

template <uint32_t D>
__device__ __forceinline__ void locate(const float *x, float scale, bool align, uint32_t interp,
                                       uint32_t (&cell)[D], float (&frac)[D], float (&dfrac)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        float p = fmaf(x[d], scale, align ? 0.0f : 0.5f);
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1u) {
            dfrac[d] = 6.0f * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            dfrac[d] = 1.0f;
        }
        frac[d] = p;
    }
}

// ------------------------------------------------------------------ forward
// One thread = one (point, level); grid = (ceil(B/256), L).  Output [L,B,C].
template <typename T, uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_forward(const float *__restrict__ inputs,
                                                      const T *__restrict__ table, T *__restrict__ outputs,
                                                      T *__restrict__ dy_dx, uint32_t B, UcnLevels lvls,
                                                      bool align, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const UcnLevel lv = lvls.lv[level];
    const uint32_t L = lvls.L;
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    T *out = outputs + ((size_t)level * B + b) * C;
    T *jac = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : nullptr;
    if (outside<D>(x)) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) out[c] = from_f<T>(0.0f);
        if (jac) {
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) jac[i] = from_f<T>(0.0f);
        }
        return;
    }
    const T *tab = table + (size_t)lv.first_row * C;
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, lv.scale, align, interp, cell, frac, dfrac);

    // The reference accumulates in scalar_t (gridencoder.cu:164,187): for float that is an fmaf
    // chain in corner order; for half every partial sum is rounded to half.
    T acc[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) acc[c] = from_f<T>(0.0f);
#pragma unroll
    for (uint32_t k = 0; k < (1u << D); k++) {
        float w = 1.0f;
        uint32_t corner[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (k & (1u << d)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
            else               { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
        }
        const T *row = tab + (size_t)row_of<D>(lv, corner) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) acc[c] = acc_mul<T>(w, to_f<T>(row[c]), acc[c]);
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++) out[c] = acc[c];

    if (jac) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            T g[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) g[c] = from_f<T>(0.0f);
#pragma unroll
            for (uint32_t k = 0; k < (1u << (D - 1)); k++) {
                float w = lv.scale;
                uint32_t corner[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                    if (k & (1u << nd)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
                    else                { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
                }
                corner[gd] = cell[gd];
                const T *lo = tab + (size_t)row_of<D>(lv, corner) * C;
                corner[gd] = cell[gd] + 1u;
                const T *hi = tab + (size_t)row_of<D>(lv, corner) * C;
#pragma unroll
                for (uint32_t c = 0; c < C; c++)   // (Half - Half) is itself rounded to half (gridencoder.cu:235)
                    g[c] = acc_mul<T>(w * to_f<T>(from_f<T>(to_f<T>(hi[c]) - to_f<T>(lo[c]))), dfrac[gd], g[c]);
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) jac[gd * C + c] = g[c];
        }
    }
}

// ------------------------------------------------------------------ backward (table)
// One thread = one (point, level) and ALL C channels of it: a lane's C atomics hit one
// 4C-byte row, and the wave's 64 lanes issue them together.  (The reference splits channels
// over threads in pairs, gridencoder.cu:260-264; the sum is the same.)  fp16 tables use
// packed half2 atomics like the reference (:325-331).
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_backward_f32(const float *__restrict__ grad,
                                                           const float *__restrict__ inputs,
                                                           float *__restrict__ grad_table, uint32_t B,
                                                           UcnLevels lvls, bool align, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const UcnLevel lv = lvls.lv[level];
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    if (outside<D>(x)) return;
    float g[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) g[c] = grad[((size_t)level * B + b) * C + c];
    float *gtab = grad_table + (size_t)lv.first_row * C;
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, lv.scale, align, interp, cell, frac, dfrac);
#pragma unroll
    for (uint32_t k = 0; k < (1u << D); k++) {
        float w = 1.0f;
        uint32_t corner[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (k & (1u << d)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
            else               { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
        }
        float *row = gtab + (size_t)row_of<D>(lv, corner) * C;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) atomicAdd(row + c, w * g[c]);
    }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_backward_f16(const __half *__restrict__ grad,
                                                           const float *__restrict__ inputs,
                                                           __half *__restrict__ grad_table, uint32_t B,
                                                           UcnLevels lvls, bool align, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const UcnLevel lv = lvls.lv[level];
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    if (outside<D>(x)) return;
    float g[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) g[c] = __half2float(grad[((size_t)level * B + b) * C + c]);
    __half *gtab = grad_table + (size_t)lv.first_row * C;
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, lv.scale, align, interp, cell, frac, dfrac);
#pragma unroll
    for (uint32_t k = 0; k < (1u << D); k++) {
        float w = 1.0f;
        uint32_t corner[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (k & (1u << d)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
            else               { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
        }
        __half *row = gtab + (size_t)row_of<D>(lv, corner) * C;
        if constexpr (C % 2 == 0) {
#pragma unroll
            for (uint32_t c = 0; c < C; c += 2)
                unsafeAtomicAdd(reinterpret_cast<__half2 *>(row + c),
                                __halves2half2(__float2half(f32_product(w, g[c])), __float2half(f32_product(w, g[c + 1]))));
        } else {
            // C == 1: CAS loop on the containing 32-bit word
            uint32_t *word = reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(row) & ~uintptr_t(3));
            const bool hi = (reinterpret_cast<uintptr_t>(row) & 2) != 0;
            uint32_t old = *word, assumed;
            do {
                assumed = old;
                __half2 cur = *reinterpret_cast<__half2 *>(&assumed);
                // gpuAtomicAdd(at::Half*, at::Half(w * grad)): the addend is rounded to half first
                float v = __half2float(hi ? __high2half(cur) : __low2half(cur)) + __half2float(__float2half(f32_product(w, g[0])));
                __half2 nw = hi ? __halves2half2(__low2half(cur), __float2half(v))
                                : __halves2half2(__float2half(v), __high2half(cur));
                old = atomicCAS(word, assumed, *reinterpret_cast<uint32_t *>(&nw));
            } while (old != assumed);
        }
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (gridencoder.cu:343-369)
template <typename T>
__global__ __launch_bounds__(256) void k_input_backward(const T *__restrict__ grad, const T *__restrict__ dy_dx,
                                                        T *__restrict__ grad_inputs, uint32_t B, uint32_t D,
                                                        uint32_t C, uint32_t L) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *jac = dy_dx + (size_t)b * L * D * C;
    T r = from_f<T>(0.0f);
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t c = 0; c < C; c++)
            r = acc_mul<T>(to_f<T>(grad[((size_t)l * B + b) * C + c]), to_f<T>(jac[(size_t)l * D * C + d * C + c]), r);
    grad_inputs[t] = r;
}

// ------------------------------------------------------------------ total variation
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grad_tv(const float *__restrict__ inputs, const float *__restrict__ table,
                                                 float *__restrict__ grad_table, float weight, uint32_t B,
                                                 UcnLevels lvls, bool align) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const UcnLevel lv = lvls.lv[blockIdx.y];
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    if (outside<D>(x)) return;
    const float *tab = table + (size_t)lv.first_row * C;
    float *gtab = grad_table + (size_t)lv.first_row * C;
    uint32_t cell[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) cell[d] = (uint32_t)floorf(fmaf(x[d], lv.scale, align ? 0.0f : 0.5f));
    float sum[C], sq[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) sum[c] = sq[c] = 0.0f;
    const uint32_t centre = row_of<D>(lv, cell);
    const float wgt = weight / (float)(2u * D);
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = cell[d];
        if (cur < lv.resolution) {
            cell[d] = cur + 1u;
            const uint32_t nb = row_of<D>(lv, cell);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float dv = tab[(size_t)centre * C + c] - tab[(size_t)nb * C + c];
                sum[c] += dv;
                sq[c] = fmaf(dv, dv, sq[c]);
            }
        }
        if (cur > 0u) {
            cell[d] = cur - 1u;
            const uint32_t nb = row_of<D>(lv, cell);
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float dv = tab[(size_t)centre * C + c] - tab[(size_t)nb * C + c];
                sum[c] += dv;
                sq[c] = fmaf(dv, dv, sq[c]);
            }
        }
        cell[d] = cur;
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++)
        atomicAdd(gtab + (size_t)centre * C + c, wgt * sum[c] * (1.0f / sqrtf(sq[c] + 1e-9f)));
}

// ------------------------------------------------------------------ dispatch
template <typename T, uint32_t D>
int launch_forward_c(uint32_t C, dim3 grid, hipStream_t st, const float *in, const T *tab, T *out, T *jac,
                     uint32_t B, const UcnLevels &lv, bool align, uint32_t interp) {
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_forward<T, D, 1>), grid, dim3(256), 0, st, in, tab, out, jac, B, lv, align, interp); break;
        case 2: hipLaunchKernelGGL((k_grid_forward<T, D, 2>), grid, dim3(256), 0, st, in, tab, out, jac, B, lv, align, interp); break;
        case 4: hipLaunchKernelGGL((k_grid_forward<T, D, 4>), grid, dim3(256), 0, st, in, tab, out, jac, B, lv, align, interp); break;
        case 8: hipLaunchKernelGGL((k_grid_forward<T, D, 8>), grid, dim3(256), 0, st, in, tab, out, jac, B, lv, align, interp); break;
        default: return ucn_fail("GridEncoding: C must be 1, 2, 4, or 8.");
    }
    return 0;
}

template <typename T>
int launch_forward(uint32_t D, uint32_t C, dim3 grid, hipStream_t st, const float *in, const T *tab, T *out,
                   T *jac, uint32_t B, const UcnLevels &lv, bool align, uint32_t interp) {
    switch (D) {
        case 2: return launch_forward_c<T, 2>(C, grid, st, in, tab, out, jac, B, lv, align, interp);
        case 3: return launch_forward_c<T, 3>(C, grid, st, in, tab, out, jac, B, lv, align, interp);
        case 4: return launch_forward_c<T, 4>(C, grid, st, in, tab, out, jac, B, lv, align, interp);
        case 5: return launch_forward_c<T, 5>(C, grid, st, in, tab, out, jac, B, lv, align, interp);
        default: return ucn_fail("GridEncoding: D must be 2, 3, 4 or 5.");
    }
}

#define UCN_BWD_CASE(DD, CC)                                                                                     \
    if (D == DD && C == CC) {                                                                                    \
        if (f16)                                                                                                 \
            hipLaunchKernelGGL((k_grid_backward_f16<DD, CC>), grid, dim3(256), 0, st, (const __half *)grad, in,  \
                               (__half *)gtab, B, lv, align, interp);                                            \
        else                                                                                                     \
            hipLaunchKernelGGL((k_grid_backward_f32<DD, CC>), grid, dim3(256), 0, st, (const float *)grad, in,   \
                               (float *)gtab, B, lv, align, interp);                                             \
        return 0;                                                                                                \
    }

int launch_backward(uint32_t D, uint32_t C, bool f16, dim3 grid, hipStream_t st, const void *grad, const float *in,
                    void *gtab, uint32_t B, const UcnLevels &lv, bool align, uint32_t interp) {
    UCN_BWD_CASE(2, 1) UCN_BWD_CASE(2, 2) UCN_BWD_CASE(2, 4) UCN_BWD_CASE(2, 8)
    UCN_BWD_CASE(3, 1) UCN_BWD_CASE(3, 2) UCN_BWD_CASE(3, 4) UCN_BWD_CASE(3, 8)
    UCN_BWD_CASE(4, 1) UCN_BWD_CASE(4, 2) UCN_BWD_CASE(4, 4) UCN_BWD_CASE(4, 8)
    UCN_BWD_CASE(5, 1) UCN_BWD_CASE(5, 2) UCN_BWD_CASE(5, 4) UCN_BWD_CASE(5, 8)
    return ucn_fail("GridEncoding: unsupported D=%u / C=%u (D in 2..5, C in 1,2,4,8)", D, C);
}

#define UCN_TV_CASE(DD, CC)                                                                                 \
    if (D == DD && C == CC) {                                                                               \
        hipLaunchKernelGGL((k_grad_tv<DD, CC>), grid, dim3(256), 0, st, in, tab, gtab, weight, B, lv, align); \
        return 0;                                                                                           \
    }

int launch_tv(uint32_t D, uint32_t C, dim3 grid, hipStream_t st, const float *in, const float *tab, float *gtab,
              float weight, uint32_t B, const UcnLevels &lv, bool align) {
    UCN_TV_CASE(2, 1) UCN_TV_CASE(2, 2) UCN_TV_CASE(2, 4) UCN_TV_CASE(2, 8)
    UCN_TV_CASE(3, 1) UCN_TV_CASE(3, 2) UCN_TV_CASE(3, 4) UCN_TV_CASE(3, 8)
    UCN_TV_CASE(4, 1) UCN_TV_CASE(4, 2) UCN_TV_CASE(4, 4) UCN_TV_CASE(4, 8)
    UCN_TV_CASE(5, 1) UCN_TV_CASE(5, 2) UCN_TV_CASE(5, 4) UCN_TV_CASE(5, 8)
    return ucn_fail("GridEncoding: unsupported D=%u / C=%u (D in 2..5, C in 1,2,4,8)", D, C);
}

}  // namespace

extern "C" int ucn_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host,
                                       void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                       uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners,
                                       uint32_t interp, int emb_dtype, ucn_stream_t stream) {
    UCN_REQUIRE(inputs && embeddings && offsets_host && outputs, "grid_encode_forward: null pointer argument");
    UCN_REQUIRE(emb_dtype == UCN_DTYPE_F32 || emb_dtype == UCN_DTYPE_F16, "embeddings must be a floating tensor (float32 or float16)");
    if (B == 0) return 0;
    UcnLevels lv;
    if (int rc = ucn_build_levels(&lv, offsets_host, nullptr, L, C, D, S, H, gridtype, align_corners)) return rc;
    const dim3 grid(ucn_div_up(B, 256), L);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (emb_dtype == UCN_DTYPE_F32)
        rc = launch_forward<float>(D, C, grid, st, inputs, (const float *)embeddings, (float *)outputs, (float *)dy_dx, B, lv, align_corners != 0, interp);
    else
        rc = launch_forward<__half>(D, C, grid, st, inputs, (const __half *)embeddings, (__half *)outputs, (__half *)dy_dx, B, lv, align_corners != 0, interp);
    if (rc) return rc;
    UCN_LAUNCH_CHECK("grid_encode_forward");
    return 0;
}

extern "C" int ucn_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                                        const int32_t *offsets_host, void *grad_embeddings, uint32_t B, uint32_t D,
                                        uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx,
                                        void *grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp,
                                        int emb_dtype, ucn_stream_t stream) {
    (void)embeddings;
    UCN_REQUIRE(grad && inputs && offsets_host && grad_embeddings, "grid_encode_backward: null pointer argument");
    UCN_REQUIRE(emb_dtype == UCN_DTYPE_F32 || emb_dtype == UCN_DTYPE_F16, "grad must be a floating tensor (float32 or float16)");
    if (B == 0) return 0;
    UcnLevels lv;
    if (int rc = ucn_build_levels(&lv, offsets_host, nullptr, L, C, D, S, H, gridtype, align_corners)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_backward(D, C, emb_dtype == UCN_DTYPE_F16, dim3(ucn_div_up(B, 256), L), st, grad, inputs,
                                 grad_embeddings, B, lv, align_corners != 0, interp))
        return rc;
    UCN_LAUNCH_CHECK("grid_encode_backward");
    if (dy_dx && grad_inputs) {
        const dim3 g2(ucn_div_up((uint64_t)B * D, 256));
        if (emb_dtype == UCN_DTYPE_F32)
            hipLaunchKernelGGL(k_input_backward<float>, g2, dim3(256), 0, st, (const float *)grad, (const float *)dy_dx, (float *)grad_inputs, B, D, C, L);
        else
            hipLaunchKernelGGL(k_input_backward<__half>, g2, dim3(256), 0, st, (const __half *)grad, (const __half *)dy_dx, (__half *)grad_inputs, B, D, C, L);
        UCN_LAUNCH_CHECK("grid_input_backward");
    }
    return 0;
}

extern "C" int ucn_grad_total_variation(const float *inputs, const float *embeddings, float *grad,
                                        const int32_t *offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                        ucn_stream_t stream) {
    UCN_REQUIRE(inputs && embeddings && offsets_host && grad, "grad_total_variation: null pointer argument");
    if (B == 0) return 0;
    UcnLevels lv;
    if (int rc = ucn_build_levels(&lv, offsets_host, nullptr, L, C, D, S, H, gridtype, align_corners)) return rc;
    if (int rc = launch_tv(D, C, dim3(ucn_div_up(B, 256), L), (hipStream_t)stream, inputs, embeddings, grad, weight, B, lv, align_corners != 0))
        return rc;
    UCN_LAUNCH_CHECK("grad_total_variation");
    return 0;
}

// ------------------------------------------------------------------ bandwidth probe
namespace {
__global__ __launch_bounds__(256) void k_copy4(const float4 *__restrict__ s, float4 *__restrict__ d, uint64_t n4) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u) d[i] = s[i];
}
// four independent 16-byte loads in flight per thread before the first store (UCN_PROBE_COPY_UNROLL=1; read per call)
__global__ __launch_bounds__(256) void k_copy4x4(const float4 *__restrict__ s, float4 *__restrict__ d, uint64_t n4) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + 3u * stride < n4; i += 4u * stride) {
        const float4 a = s[i], b = s[i + stride], c = s[i + 2u * stride], e = s[i + 3u * stride];
        d[i] = a; d[i + stride] = b; d[i + 2u * stride] = c; d[i + 3u * stride] = e;
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
}  // namespace
extern "C" int ucn_probe_copy(const float *src, float *dst, uint64_t n_floats, ucn_stream_t stream) {
    UCN_REQUIRE((n_floats & 3) == 0, "probe_copy: n_floats must be a multiple of 4");
    const char *u = getenv("UCN_PROBE_COPY_UNROLL");
    if (u && atoi(u) == 1)
        hipLaunchKernelGGL(k_copy4x4, dim3(4096), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n_floats / 4);
    else
    hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n_floats / 4);
    UCN_LAUNCH_CHECK("probe_copy");
    return 0;
}
