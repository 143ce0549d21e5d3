/*
 * grid_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (gcc) CPU restatement of the reference's multi-resolution hash-grid
 * operator, whose only implementation is CUDA:
 *     /root/reference/nerf/gridencoder/src/gridencoder.cu
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  The shipped HIP path never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" by the reference -- gridencoder.cu has no CPU
 * path (CHECK_CUDA, gridencoder.cu:449-452), no unit tests and no golden
 * vectors anywhere under /root/reference, and it cannot be compiled here (needs
 * nvcc + CUDA ATen headers).  It is pinned instead by (i) an independent
 * vectorised numpy restatement (oracle/grid_numpy.py) that must agree with this
 * file bit-for-bit on random, boundary and out-of-range inputs, and (ii) the
 * reference's own Python (grid.py / models.py) imported in the authoring
 * container with this file standing in for the `_gridencoder` backend, whose
 * outputs are committed as tests/golden/ (npz files).
 *
 * Arithmetic notes (what "the reference computes" means for a CUDA source):
 * nvcc's default -fmad=true contracts `a*b+c` into one fused multiply-add, so
 *   pos = inputs*scale + 0.5f            (gridencoder.cu:148)  is fmaf(x,scale,.5f)
 *   results[ch] += w * grid[index+ch]    (gridencoder.cu:187)  is fmaf(w,g,acc)
 *   results_grad += w*(r-l)*pos_deriv    (gridencoder.cu:235)  is fmaf(w*(r-l),pd,acc)
 *   result += grad * dy_dx               (gridencoder.cu:364)  is fmaf(g,d,acc)
 * This file spells those fmaf() calls out and is built with -ffp-contract=off
 * so nothing else is fused.  The HIP kernels use the same spelling.
 *
 * Build:  make -C oracle      (-> oracle/_build/libgrid_oracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define GO_MAX_D 5
#define GO_MAX_C 8

/* gridencoder.cu:54 -- per-dimension multipliers of the spatial hash */
static const uint32_t GO_PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                      2097192037u, 1434869437u, 2165219737u};

/* one level's constants; gridencoder.cu:137-139 */
typedef struct {
    uint32_t first_row;   /* offsets[level]                       */
    uint32_t rows;        /* hashmap_size = offsets[l+1]-offsets[l] */
    float scale;          /* exp2f(level*S)*H - 1                  */
    uint32_t resolution;  /* ceil(scale)+1                         */
} go_level_t;

static go_level_t go_level(const int32_t *offsets, uint32_t level, float S, uint32_t H) {
    go_level_t lv;
    lv.first_row = (uint32_t)offsets[level];
    lv.rows = (uint32_t)(offsets[level + 1] - offsets[level]);
    /* `level * S` is uint32*float -> float, exp2f in float, then *H (uint32->float) */
    lv.scale = exp2f((float)level * S) * (float)H - 1.0f;
    lv.resolution = (uint32_t)ceilf(lv.scale) + 1u;
    return lv;
}

/* gridencoder.cu:66-84 -- row of the table for an integer lattice point.
 * Dense (row-major, x fastest) addressing while the running stride still fits in
 * the level's row budget; otherwise (hash gridtype) the xor-of-products hash.
 * The dense walk stops at the first dimension whose stride overflows, exactly
 * like the `d < D && stride <= hashmap_size` loop condition. */
static uint32_t go_row(uint32_t D, uint32_t gridtype, int align_corners, uint32_t rows,
                       uint32_t resolution, const uint32_t *cell) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= rows; d++) {
        index += cell[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1u);
    }
    if (gridtype == 0u && stride > rows) {
        uint32_t h = 0;
        for (uint32_t d = 0; d < D; d++) h ^= cell[d] * GO_PRIMES[d];
        index = h;
    }
    return index % rows;
}

static int go_outside(const float *x, uint32_t D) {
    /* gridencoder.cu:113-117; NaN compares false on both sides => treated as inside */
    for (uint32_t d = 0; d < D; d++)
        if (x[d] < 0.0f || x[d] > 1.0f) return 1;
    return 0;
}

/* gridencoder.cu:146-159: lattice cell + (optionally smooth-stepped) fraction */
static void go_locate(const float *x, uint32_t D, float scale, int align_corners, uint32_t interp,
                      uint32_t *cell, float *frac, float *dfrac) {
    for (uint32_t d = 0; d < D; d++) {
        float p = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1u) {
            dfrac[d] = 6.0f * p * (1.0f - p);               /* :45-47 */
            p = p * p * (3.0f - 2.0f * p);                  /* :40-42 */
        } else {
            dfrac[d] = 1.0f;
        }
        frac[d] = p;
    }
}

/* ---------------------------------------------------------------- forward */
/* gridencoder.cu:87-245 (kernel_grid) + :448-471 (host entry).
 * inputs [B,D] in [0,1]; emb [rows_total,C]; offsets [L+1]; outputs [L,B,C];
 * dy_dx [B, L*D*C] or NULL. */
void grid_oracle_forward(const float *inputs, const float *emb, const int32_t *offsets,
                         float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                         uint32_t H, float *dy_dx, uint32_t gridtype, int align_corners,
                         uint32_t interp) {
    for (uint32_t level = 0; level < L; level++) {
        const go_level_t lv = go_level(offsets, level, S, H);
        const float *tab = emb + (size_t)lv.first_row * C;
#pragma omp parallel for schedule(static)
        for (int64_t bb = 0; bb < (int64_t)B; bb++) {
            const uint32_t b = (uint32_t)bb;
            const float *x = inputs + (size_t)b * D;
            float *out = outputs + ((size_t)level * B + b) * C;
            float *jac = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : 0;
            if (go_outside(x, D)) {
                for (uint32_t c = 0; c < C; c++) out[c] = 0.0f;
                if (jac) memset(jac, 0, sizeof(float) * D * C);
                continue;
            }
            uint32_t cell[GO_MAX_D], corner[GO_MAX_D];
            float frac[GO_MAX_D], dfrac[GO_MAX_D], acc[GO_MAX_C];
            go_locate(x, D, lv.scale, align_corners, interp, cell, frac, dfrac);
            for (uint32_t c = 0; c < C; c++) acc[c] = 0.0f;
            for (uint32_t k = 0; k < (1u << D); k++) {       /* :167-191 */
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (k & (1u << d)) { w *= frac[d];          corner[d] = cell[d] + 1u; }
                    else               { w *= 1.0f - frac[d];   corner[d] = cell[d]; }
                }
                const float *row = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                        lv.resolution, corner) * C;
                for (uint32_t c = 0; c < C; c++) acc[c] = fmaf(w, row[c], acc[c]);
            }
            for (uint32_t c = 0; c < C; c++) out[c] = acc[c];
            if (jac) {                                       /* :201-244 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float g[GO_MAX_C];
                    for (uint32_t c = 0; c < C; c++) g[c] = 0.0f;
                    for (uint32_t k = 0; k < (1u << (D - 1)); k++) {
                        float w = lv.scale;
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                            if (k & (1u << nd)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
                            else                { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
                        }
                        corner[gd] = cell[gd];
                        const float *lo = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                               lv.resolution, corner) * C;
                        corner[gd] = cell[gd] + 1u;
                        const float *hi = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                               lv.resolution, corner) * C;
                        for (uint32_t c = 0; c < C; c++)
                            g[c] = fmaf(w * (hi[c] - lo[c]), dfrac[gd], g[c]);
                    }
                    for (uint32_t c = 0; c < C; c++) jac[gd * C + c] = g[c];
                }
            }
        }
    }
}

/* --------------------------------------------------------------- backward */
/* gridencoder.cu:248-340 (kernel_grid_backward), :343-369 (kernel_input_backward),
 * :473-503 (host entry).  grad [L,B,C]; grad_emb [rows_total,C] pre-zeroed by the
 * caller (grid.py:77) and accumulated into.  The CUDA kernel's float atomics make
 * its summation order nondeterministic; this restatement adds in increasing b,
 * then corner order, i.e. ONE of the orders the reference can produce.  Levels own
 * disjoint row ranges, so they are run in parallel without changing that order. */
void grid_oracle_backward(const float *grad, const float *inputs, const float *emb,
                          const int32_t *offsets, float *grad_emb, uint32_t B, uint32_t D,
                          uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx,
                          float *grad_inputs, uint32_t gridtype, int align_corners,
                          uint32_t interp) {
    (void)emb;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ll = 0; ll < (int64_t)L; ll++) {
        const uint32_t level = (uint32_t)ll;
        const go_level_t lv = go_level(offsets, level, S, H);
        float *gtab = grad_emb + (size_t)lv.first_row * C;
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            if (go_outside(x, D)) continue;                  /* :276-281 */
            const float *g = grad + ((size_t)level * B + b) * C;
            uint32_t cell[GO_MAX_D], corner[GO_MAX_D];
            float frac[GO_MAX_D], dfrac[GO_MAX_D];
            go_locate(x, D, lv.scale, align_corners, interp, cell, frac, dfrac);
            for (uint32_t k = 0; k < (1u << D); k++) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (k & (1u << d)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
                    else               { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
                }
                float *row = gtab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                   lv.resolution, corner) * C;
                for (uint32_t c = 0; c < C; c++) row[c] += w * g[c];   /* atomicAdd(w*grad) */
            }
        }
    }
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t % D);
            const float *jac = dy_dx + (size_t)b * L * D * C;
            float r = 0.0f;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t c = 0; c < C; c++)
                    r = fmaf(grad[((size_t)l * B + b) * C + c], jac[(size_t)l * D * C + d * C + c], r);
            grad_inputs[t] = r;
        }
    }
}

/* -------------------------------------------------------- total variation */
/* gridencoder.cu:506-610 (kernel_grad_tv) + :639-645.  grad_emb += w/(2D) * sum of
 * neighbour differences, normalised per channel by rsqrt(sum of squares + 1e-9).
 * Note the reference never smooth-steps or fractions here: only the cell matters.
 * (rsqrtf is 1/sqrtf on the CPU; CUDA's rsqrtf may differ in the last 1-2 ulp.) */
void grid_oracle_total_variation(const float *inputs, const float *emb, float *grad_emb,
                                 const int32_t *offsets, float weight, uint32_t B, uint32_t D,
                                 uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                 int align_corners) {
    const float w = weight / (float)(2u * D);
    for (uint32_t level = 0; level < L; level++) {
        const go_level_t lv = go_level(offsets, level, S, H);
        const float *tab = emb + (size_t)lv.first_row * C;
        float *gtab = grad_emb + (size_t)lv.first_row * C;
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            if (go_outside(x, D)) continue;
            uint32_t cell[GO_MAX_D];
            for (uint32_t d = 0; d < D; d++)
                cell[d] = (uint32_t)floorf(fmaf(x[d], lv.scale, align_corners ? 0.0f : 0.5f));
            float sum[GO_MAX_C], sq[GO_MAX_C];
            for (uint32_t c = 0; c < C; c++) sum[c] = sq[c] = 0.0f;
            const uint32_t centre = go_row(D, gridtype, align_corners, lv.rows, lv.resolution, cell);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = cell[d];
                if (cur < lv.resolution) {                   /* right neighbour :572-583 */
                    cell[d] = cur + 1u;
                    const uint32_t nb = go_row(D, gridtype, align_corners, lv.rows, lv.resolution, cell);
                    for (uint32_t c = 0; c < C; c++) {
                        const float dv = tab[(size_t)centre * C + c] - tab[(size_t)nb * C + c];
                        sum[c] += dv;
                        sq[c] = fmaf(dv, dv, sq[c]);
                    }
                }
                if (cur > 0u) {                              /* left neighbour :586-597 */
                    cell[d] = cur - 1u;
                    const uint32_t nb = go_row(D, gridtype, align_corners, lv.rows, lv.resolution, cell);
                    for (uint32_t c = 0; c < C; c++) {
                        const float dv = tab[(size_t)centre * C + c] - tab[(size_t)nb * C + c];
                        sum[c] += dv;
                        sq[c] = fmaf(dv, dv, sq[c]);
                    }
                }
                cell[d] = cur;
            }
            for (uint32_t c = 0; c < C; c++)
                gtab[(size_t)centre * C + c] += w * sum[c] * (1.0f / sqrtf(sq[c] + 1e-9f));
        }
    }
}

/* ------------------------------------------------------- fp16 tables (autocast) */
/* grid.py:43-44 casts the table to torch.half under autocast (C even), so kernel_grid runs with
 * scalar_t = at::Half.  What the reference's expressions then compute follows from c10::Half's operator
 * set (torch/headeronly/util/Half.h; tests/test_oracle_grid.py compiles the expressions against that very
 * header and checks this file's spelling of them):
 *   float * Half -> float;  Half - Half -> Half (float subtract, rounded);  Half * Half -> Half;
 *   Half += <float>  converts the float to Half FIRST (round to nearest even), then adds in float and rounds:
 *       results[ch] += w * grid[i]         (gridencoder.cu:187)  acc = h(f(acc) + f(h(w * f(g))))
 *       results_grad[ch] += w*(r-l)*pd     (gridencoder.cu:235)  acc = h(f(acc) + f(h((w * f(h(f(r)-f(l)))) * pd)))
 *       result += grad * dy_dx             (gridencoder.cu:364)  acc = h(f(acc) + f(h(f(g) * f(d))))
 *   (__half)(w * grad_cur[c]) + atomicAdd(__half2)  (gridencoder.cu:325-331)  row = h(f(row) + f(h(w * f(g))))
 * None of these contains a float multiply feeding a float add directly, so nvcc's fmad contraction has
 * nothing to fuse.  gcc 11 has no _Float16 on x86-64: binary16 <-> binary32 is done in software below
 * (round to nearest even, subnormals and infinities included) and checked against numpy in the tests. */
typedef uint16_t go_half;

float grid_oracle_h2f(go_half h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                           /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

go_half grid_oracle_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (go_half)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));   /* NaN / inf */
    if (x >= 0x477FF000u) return (go_half)(sign | 0x7C00u);                /* >= 65520 rounds to inf */
    if (x < 0x33000001u) return sign;                                      /* <= 2^-25: rounds to zero */
    const int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7FFFFFu) | 0x800000u;                            /* 24-bit significand */
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                            /* bits dropped */
    uint32_t q = man >> shift, rem = man & ((1u << shift) - 1u), half_ulp = 1u << (shift - 1);
    if (rem > half_ulp || (rem == half_ulp && (q & 1u))) q++;
    /* q carries the hidden bit for normals (bit 10); the exponent field absorbs a mantissa carry */
    const uint32_t out = e >= -14 ? (((uint32_t)(e + 15 - 1) << 10) + q) : q;
    return (go_half)(sign | out);
}

#define H2F grid_oracle_h2f
#define F2H grid_oracle_f2h

void grid_oracle_forward_h(const float *inputs, const go_half *emb, const int32_t *offsets,
                           go_half *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                           uint32_t H, go_half *dy_dx, uint32_t gridtype, int align_corners,
                           uint32_t interp) {
    for (uint32_t level = 0; level < L; level++) {
        const go_level_t lv = go_level(offsets, level, S, H);
        const go_half *tab = emb + (size_t)lv.first_row * C;
#pragma omp parallel for schedule(static)
        for (int64_t bb = 0; bb < (int64_t)B; bb++) {
            const uint32_t b = (uint32_t)bb;
            const float *x = inputs + (size_t)b * D;
            go_half *out = outputs + ((size_t)level * B + b) * C;
            go_half *jac = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : 0;
            if (go_outside(x, D)) {
                for (uint32_t c = 0; c < C; c++) out[c] = 0;
                if (jac) memset(jac, 0, sizeof(go_half) * D * C);
                continue;
            }
            uint32_t cell[GO_MAX_D], corner[GO_MAX_D];
            float frac[GO_MAX_D], dfrac[GO_MAX_D];
            go_half acc[GO_MAX_C];
            go_locate(x, D, lv.scale, align_corners, interp, cell, frac, dfrac);
            for (uint32_t c = 0; c < C; c++) acc[c] = 0;
            for (uint32_t k = 0; k < (1u << D); k++) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (k & (1u << d)) { w *= frac[d];          corner[d] = cell[d] + 1u; }
                    else               { w *= 1.0f - frac[d];   corner[d] = cell[d]; }
                }
                const go_half *row = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                          lv.resolution, corner) * C;
                for (uint32_t c = 0; c < C; c++)
                    acc[c] = F2H(H2F(acc[c]) + H2F(F2H(w * H2F(row[c]))));
            }
            for (uint32_t c = 0; c < C; c++) out[c] = acc[c];
            if (jac) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    go_half g[GO_MAX_C];
                    for (uint32_t c = 0; c < C; c++) g[c] = 0;
                    for (uint32_t k = 0; k < (1u << (D - 1)); k++) {
                        float w = lv.scale;
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                            if (k & (1u << nd)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
                            else                { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
                        }
                        corner[gd] = cell[gd];
                        const go_half *lo = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                                 lv.resolution, corner) * C;
                        corner[gd] = cell[gd] + 1u;
                        const go_half *hi = tab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                                 lv.resolution, corner) * C;
                        for (uint32_t c = 0; c < C; c++) {
                            const float diff = H2F(F2H(H2F(hi[c]) - H2F(lo[c])));
                            g[c] = F2H(H2F(g[c]) + H2F(F2H((w * diff) * dfrac[gd])));
                        }
                    }
                    for (uint32_t c = 0; c < C; c++) jac[gd * C + c] = g[c];
                }
            }
        }
    }
}

/* grad [L,B,C] half, grad_emb [rows,C] half (pre-zeroed), dy_dx / grad_inputs half or NULL.  Adds in
 * increasing b then corner order: one of the orders the half atomics can produce. */
void grid_oracle_backward_h(const go_half *grad, const float *inputs, const int32_t *offsets,
                            go_half *grad_emb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                            uint32_t H, const go_half *dy_dx, go_half *grad_inputs, uint32_t gridtype,
                            int align_corners, uint32_t interp) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t ll = 0; ll < (int64_t)L; ll++) {
        const uint32_t level = (uint32_t)ll;
        const go_level_t lv = go_level(offsets, level, S, H);
        go_half *gtab = grad_emb + (size_t)lv.first_row * C;
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            if (go_outside(x, D)) continue;
            const go_half *g = grad + ((size_t)level * B + b) * C;
            uint32_t cell[GO_MAX_D], corner[GO_MAX_D];
            float frac[GO_MAX_D], dfrac[GO_MAX_D];
            go_locate(x, D, lv.scale, align_corners, interp, cell, frac, dfrac);
            for (uint32_t k = 0; k < (1u << D); k++) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; d++) {
                    if (k & (1u << d)) { w *= frac[d];        corner[d] = cell[d] + 1u; }
                    else               { w *= 1.0f - frac[d]; corner[d] = cell[d]; }
                }
                go_half *row = gtab + (size_t)go_row(D, gridtype, align_corners, lv.rows,
                                                     lv.resolution, corner) * C;
                for (uint32_t c = 0; c < C; c++)
                    row[c] = F2H(H2F(row[c]) + H2F(F2H(w * H2F(g[c]))));
            }
        }
    }
    if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t % D);
            const go_half *jac = dy_dx + (size_t)b * L * D * C;
            go_half r = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t c = 0; c < C; c++)
                    r = F2H(H2F(r) + H2F(F2H(H2F(grad[((size_t)l * B + b) * C + c]) *
                                             H2F(jac[(size_t)l * D * C + d * C + c]))));
            grad_inputs[t] = r;
        }
    }
}

/* exported for tests: the per-level constants the kernels must agree on */
void grid_oracle_level_constants(const int32_t *offsets, uint32_t L, float S, uint32_t H,
                                 float *scale_out, uint32_t *resolution_out) {
    for (uint32_t l = 0; l < L; l++) {
        go_level_t lv = go_level(offsets, l, S, H);
        scale_out[l] = lv.scale;
        resolution_out[l] = lv.resolution;
    }
}
