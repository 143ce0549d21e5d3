// Fused sample featurisation: cone cast -> contraction -> hash-grid gather -> erf damping -> mean of 6.
//
// Replaces, for one sampling level, the chain
//   render.cast_rays            (/root/reference/nerf/internal/render.py:94-152)
//   coord.track_linearize       (coord.py:60-116, 'contract')
//   GridEncoder.forward         (gridencoder/grid.py:158-174 -> gridencoder.cu:87-199)
//   erf down-weighting + mean   (models.py:494-496)
// The reference materialises [N*S*6,3] points, the [L,N*S*6,C] gather result, its permuted copy
// and the erf weights in HBM (~1.5 GB per 15000-ray chunk at 128 samples); here the six
// multisamples of a sample live in registers and only the [L][N*S][C] mean feature is written.
//
// Mapping (CDNA4): one thread = one sample x `levels_per_block` consecutive levels; blockIdx.y is
// the level group, so the grid is level-major in dispatch order and an XCD's L2 (4 MiB) sees one
// 4 MiB hashed level slice at a time; lanes of a wave are consecutive samples of a ray -> the
// [L][B][C] store is a contiguous 64*C*4-byte run per wave, and the 48 corner gathers of a
// thread-level are independent loads the memory pipe can keep in flight.  The bound is the L2 /
// HBM random-sector rate, not flops: keep VGPRs low (occupancy) and never re-read the table.
//
// Arithmetic follows the reference op-by-op in fp32 (built with -ffp-contract=off); the grid
// interpolation uses the fmaf spelling of oracle/grid_oracle.c and is bit-identical to it.
#include "ucn_common.h"

namespace {

struct HexPattern {
    float cs[2][6];   // cos of the deterministic angles for even / odd samples (render.py:126-131)
    float sn[2][6];
    float ang[6];     // pi/3 * [0,2,4,3,5,1]   (render.py:119)
    float cj[6];      // 3/sqrt(7) * (2j/5 - 1)  (render.py:116)
};

struct RayInputs {
    const float *sdist, *near_, *far_, *origins, *dirs, *basis, *radii, *flip, *spin;
};

// trilinear lookup of one point in one level; gridencoder.cu:146-191 for D = 3, linear, no align
template <uint32_t C>
__device__ __forceinline__ void level_lookup(const UcnLevel &lv, const float *__restrict__ tab, float px, float py,
                                             float pz, float (&out)[C]) {
#pragma unroll
    for (uint32_t c = 0; c < C; c++) out[c] = 0.0f;
    if (px < 0.0f || px > 1.0f || py < 0.0f || py > 1.0f || pz < 0.0f || pz > 1.0f) return;
    float fx = fmaf(px, lv.scale, 0.5f), fy = fmaf(py, lv.scale, 0.5f), fz = fmaf(pz, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    uint32_t rows[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t x = x0 + (k & 1u), y = y0 + ((k >> 1) & 1u), z = z0 + ((k >> 2) & 1u);
        uint32_t idx = lv.hashed ? ucn_hash3(x, y, z) : x * lv.stride[0] + y * lv.stride[1] + z * lv.stride[2];
        rows[k] = lv.mask ? (idx & lv.mask) : (idx < lv.rows ? idx : idx % lv.rows);
    }
    // issue all eight row loads before the first use
    float v[8][C];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const float *r = tab + (size_t)rows[k] * C;
        if constexpr (C == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(r);
            v[k][0] = t.x; v[k][1] = t.y;
        } else if constexpr (C == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(r);
            v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) v[k][c] = r[c];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        // w = ((1 * wx) * wy) * wz in the reference's multiplication order (gridencoder.cu:168-180)
        float w = 1.0f;
        w *= (k & 1u) ? fx : gx;
        w *= (k & 2u) ? fy : gy;
        w *= (k & 4u) ? fz : gz;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) out[c] = fmaf(w, v[k][c], out[c]);
    }
}

// Lattice cell of a point in one level (same arithmetic as level_lookup); false when the point is
// outside [0,1]^3 (zero feature, gridencoder.cu:110-135).
__device__ __forceinline__ bool cell_of(const UcnLevel &lv, float px, float py, float pz, uint32_t &x0, uint32_t &y0,
                                        uint32_t &z0) {
    if (px < 0.0f || px > 1.0f || py < 0.0f || py > 1.0f || pz < 0.0f || pz > 1.0f) return false;
    x0 = (uint32_t)floorf(fmaf(px, lv.scale, 0.5f));
    y0 = (uint32_t)floorf(fmaf(py, lv.scale, 0.5f));
    z0 = (uint32_t)floorf(fmaf(pz, lv.scale, 0.5f));
    return true;
}

__device__ __forceinline__ void cell_rows(const UcnLevel &lv, uint32_t x0, uint32_t y0, uint32_t z0, uint32_t (&rows)[8]) {
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t x = x0 + (k & 1u), y = y0 + ((k >> 1) & 1u), z = z0 + ((k >> 2) & 1u);
        uint32_t idx = lv.hashed ? ucn_hash3(x, y, z) : x * lv.stride[0] + y * lv.stride[1] + z * lv.stride[2];
        rows[k] = lv.mask ? (idx & lv.mask) : (idx < lv.rows ? idx : idx % lv.rows);
    }
}

// The six multisamples of a sample sit inside a cone of radius ~3e-4*t: on the coarse levels (cell
// >> cone) they share one lattice cell, so its 8 corner rows are fetched ONCE and interpolated six
// times from registers -- 8 lane-requests instead of 48 on those levels.  The arithmetic per point is
// the same fmaf chain as level_lookup, so the result is bit-identical.
template <uint32_t C>
__device__ __forceinline__ bool shared_cell(const UcnLevel &lv, const float (&u)[6][3], uint32_t G, uint32_t &cx,
                                            uint32_t &cy, uint32_t &cz) {
    bool same = cell_of(lv, u[0][0], u[0][1], u[0][2], cx, cy, cz);
#pragma unroll
    for (uint32_t j = 1; j < 6; j++) {
        if (j < G) {
            uint32_t ax = 0, ay = 0, az = 0;
            const bool in = cell_of(lv, u[j][0], u[j][1], u[j][2], ax, ay, az);
            same = same && in && ax == cx && ay == cy && az == cz;
        }
    }
    return same;
}

// coord.py:60-72 followed by the /2 of models.py:491-493; returns the [0,1] grid coordinate
__device__ __forceinline__ void contract_to_unit(float x, float y, float z, float sd, bool warp, float &u0, float &u1,
                                                 float &u2, float &sd_out, float &c0, float &c1, float &c2) {
    if (warp) {
        const float m = fmaxf((x * x + y * y) + z * z, UCN_EPS);
        if (!(m <= 1.0f)) {
            const float root = sqrtf(m);
            const float k = (2.0f * root - 1.0f) / m;
            x = k * x; y = k * y; z = k * z;
            float sh = powf(2.0f * root - 1.0f, 0.3333333432674408f) / root;
            sd = (sh * sh) * sd;
        }
        x = x / 2.0f; y = y / 2.0f; z = z / 2.0f;
        sd = sd / 2.0f;
    }
    c0 = x; c1 = y; c2 = z;
    u0 = (x + 1.0f) / 2.0f; u1 = (y + 1.0f) / 2.0f; u2 = (z + 1.0f) / 2.0f;    // grid.py:162, bound = 1
    sd_out = sd;
}

// Backward of level_lookup w.r.t. the table: grad_table[row_k] += w_k * g (gridencoder.cu:304-339).
template <uint32_t C>
__device__ __forceinline__ void level_scatter(const UcnLevel &lv, float *__restrict__ gtab, float px, float py,
                                              float pz, const float (&g)[C]) {
    if (px < 0.0f || px > 1.0f || py < 0.0f || py > 1.0f || pz < 0.0f || pz > 1.0f) return;
    float fx = fmaf(px, lv.scale, 0.5f), fy = fmaf(py, lv.scale, 0.5f), fz = fmaf(pz, lv.scale, 0.5f);
    const uint32_t x0 = (uint32_t)floorf(fx), y0 = (uint32_t)floorf(fy), z0 = (uint32_t)floorf(fz);
    fx -= (float)x0; fy -= (float)y0; fz -= (float)z0;
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t x = x0 + (k & 1u), y = y0 + ((k >> 1) & 1u), z = z0 + ((k >> 2) & 1u);
        uint32_t idx = lv.hashed ? ucn_hash3(x, y, z) : x * lv.stride[0] + y * lv.stride[1] + z * lv.stride[2];
        idx = lv.mask ? (idx & lv.mask) : (idx < lv.rows ? idx : idx % lv.rows);
        float w = 1.0f;
        w *= (k & 1u) ? fx : gx;
        w *= (k & 2u) ? fy : gy;
        w *= (k & 4u) ? fz : gz;
        float *r = gtab + (size_t)idx * C;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) atomicAdd(r + c, w * g[c]);
    }
}

template <uint32_t C>
__device__ __forceinline__ void featurise_bwd(const UcnLevels &lvls, float *__restrict__ grad_table, uint32_t lvl0,
                                              uint32_t lvl1, const float (&u)[6][3], const float (&sd)[6], uint32_t G,
                                              size_t B, size_t b, const float *__restrict__ grad, bool sample_major) {
    const uint32_t F = lvls.L * C;
    for (uint32_t lvl = lvl0; lvl < lvl1; lvl++) {
        const UcnLevel lv = lvls.lv[lvl];
        float *gtab = grad_table + (size_t)lv.first_row * C;
        const float *gp = sample_major ? grad + b * F + (size_t)lvl * C : grad + ((size_t)lvl * B + b) * C;
        float gout[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) gout[c] = gp[c] / (float)G;          // d(mean over G)
        uint32_t cx = 0, cy = 0, cz = 0;
        if (shared_cell<C>(lv, u, G, cx, cy, cz)) {
            // all multisamples in one cell: sum their corner weights first, 8*C atomics instead of 48*C
            float wsum[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) wsum[k] = 0.0f;
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) {
                if (j < G) {
                    const float fx = fmaf(u[j][0], lv.scale, 0.5f) - (float)cx;
                    const float fy = fmaf(u[j][1], lv.scale, 0.5f) - (float)cy;
                    const float fz = fmaf(u[j][2], lv.scale, 0.5f) - (float)cz;
                    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
                    const float damp = erff(1.0f / sqrtf((8.0f * (sd[j] * sd[j])) * lv.gs2));
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        float w = 1.0f;
                        w *= (k & 1u) ? fx : gx;
                        w *= (k & 2u) ? fy : gy;
                        w *= (k & 4u) ? fz : gz;
                        wsum[k] += w * damp;
                    }
                }
            }
            uint32_t rows[8];
            cell_rows(lv, cx, cy, cz, rows);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                float *r = gtab + (size_t)rows[k] * C;
#pragma unroll
                for (uint32_t c = 0; c < C; c++) atomicAdd(r + c, wsum[k] * gout[c]);
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) {
                if (j < G) {
                    const float damp = erff(1.0f / sqrtf((8.0f * (sd[j] * sd[j])) * lv.gs2));
                    float g[C];
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) g[c] = gout[c] * damp;
                    level_scatter<C>(lv, gtab, u[j][0], u[j][1], u[j][2], g);
                }
            }
        }
    }
}

template <uint32_t C>
__device__ __forceinline__ void featurise(const UcnLevels &lvls, const float *__restrict__ table, uint32_t lvl0,
                                          uint32_t lvl1, const float (&u)[6][3], const float (&sd)[6], uint32_t G,
                                          size_t B, size_t b, float *__restrict__ out, bool sample_major = false) {
    const uint32_t F_out = lvls.L * C;
    for (uint32_t lvl = lvl0; lvl < lvl1; lvl++) {
        const UcnLevel lv = lvls.lv[lvl];
        const float *tab = table + (size_t)lv.first_row * C;
        float acc[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) acc[c] = 0.0f;
        uint32_t cx = 0, cy = 0, cz = 0;
        if (shared_cell<C>(lv, u, G, cx, cy, cz)) {
            uint32_t rows[8];
            cell_rows(lv, cx, cy, cz, rows);
            float v[8][C];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const float *r = tab + (size_t)rows[k] * C;
                if constexpr (C == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(r);
                    v[k][0] = t.x; v[k][1] = t.y;
                } else if constexpr (C == 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(r);
                    v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
                } else {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) v[k][c] = r[c];
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) {
                if (j < G) {
                    const float fx = fmaf(u[j][0], lv.scale, 0.5f) - (float)cx;
                    const float fy = fmaf(u[j][1], lv.scale, 0.5f) - (float)cy;
                    const float fz = fmaf(u[j][2], lv.scale, 0.5f) - (float)cz;
                    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
                    float f[C];
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) f[c] = 0.0f;
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        float w = 1.0f;
                        w *= (k & 1u) ? fx : gx;
                        w *= (k & 2u) ? fy : gy;
                        w *= (k & 4u) ? fz : gz;
#pragma unroll
                        for (uint32_t c = 0; c < C; c++) f[c] = fmaf(w, v[k][c], f[c]);
                    }
                    const float damp = erff(1.0f / sqrtf((8.0f * (sd[j] * sd[j])) * lv.gs2));
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) acc[c] += f[c] * damp;
                }
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 6; j++) {
                if (j < G) {
                    float f[C];
                    level_lookup<C>(lv, tab, u[j][0], u[j][1], u[j][2], f);
                    // models.py:495: erf(1 / sqrt(8 * std^2 * grid_sizes^2)), grid_sizes^2 in wrapped int32
                    const float damp = erff(1.0f / sqrtf((8.0f * (sd[j] * sd[j])) * lv.gs2));
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) acc[c] += f[c] * damp;
                }
            }
        }
        float *o = sample_major ? out + b * F_out + (size_t)lvl * C : out + ((size_t)lvl * B + b) * C;
        const float inv = (float)G;
        if constexpr (C == 2) {
            *reinterpret_cast<float2 *>(o) = make_float2(acc[0] / inv, acc[1] / inv);
        } else if constexpr (C == 4) {
            *reinterpret_cast<float4 *>(o) = make_float4(acc[0] / inv, acc[1] / inv, acc[2] / inv, acc[3] / inv);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) o[c] = acc[c] / inv;
        }
    }
}

// The six multisample Gaussians of sample (ray, s): render.py:108-152 then coord.py:60-72 and the
// /2 + [0,1] mapping.  Shared by the forward and the backward kernel (the backward recomputes it:
// 200 flops instead of reading back 6x4 floats per sample).
__device__ __forceinline__ void cast_sample(const RayInputs &in, const HexPattern &hx, float std_scale, uint32_t ray,
                                            uint32_t s, uint32_t S, float (&u)[6][3], float (&sd)[6],
                                            float (&csum)[3], float &tsum) {
    const float nr = in.near_[ray], fr = in.far_[ray];
    const float s0 = in.sdist[(size_t)ray * (S + 1) + s], s1 = in.sdist[(size_t)ray * (S + 1) + s + 1];
    const float t0 = s0 * fr + (1.0f - s0) * nr, t1 = s1 * fr + (1.0f - s1) * nr;
    const float rad = in.radii[ray];
    const float *bp = in.basis + (size_t)ray * 6;
    const float e1x = bp[0], e1y = bp[1], e1z = bp[2], e2x = bp[3], e2y = bp[4], e2z = bp[5];
    const float dx = in.dirs[ray * 3 + 0], dy = in.dirs[ray * 3 + 1], dz = in.dirs[ray * 3 + 2];
    const float ox = in.origins[ray * 3 + 0], oy = in.origins[ray * 3 + 1], oz = in.origins[ray * 3 + 2];
    // render.py:112-117
    const float t_m = (t0 + t1) / 2.0f, t_d = (t1 - t0) / 2.0f;
    const float td2 = t_d * t_d, tm2 = t_m * t_m;
    const float a_ = t_d / (td2 + 3.0f * tm2);
    const float inner = td2 - tm2;
    const float root = sqrtf(inner * inner + 4.0f * (tm2 * tm2));
    const float base = t1 * t1 + 2.0f * tm2;
    // angles: deterministic hexagon (rotated 30 deg + mirrored on odd samples) or random spin/flip
    const bool rnd = in.flip != nullptr;
    float spin2pi = 0.0f;
    bool keep = true;
    if (rnd) {
        keep = in.flip[(size_t)ray * S + s] > 0.5f;
        spin2pi = 6.2831854820251465f * in.spin[(size_t)ray * S + s];
    }
    csum[0] = csum[1] = csum[2] = 0.0f;
    tsum = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        const float t = t0 + a_ * (base + hx.cj[j] * root);
        float cs, sn;
        if (rnd) {
            float ang = hx.ang[j] + spin2pi;
            if (!keep) ang = 5.235987663269043f - ang;
            cs = cosf(ang); sn = sinf(ang);
        } else {
            cs = hx.cs[s & 1u][j]; sn = hx.sn[s & 1u][j];
        }
        const float rt = rad * t;
        const float l0 = (rt * cs) / 1.4142135381698608f, l1 = (rt * sn) / 1.4142135381698608f;
        const float sdev = ((std_scale * rad) * t) / 1.4142135381698608f;
        // math.matmul with basis^T (render.py:146-148): sum_k local_k * axis_k, then + origin
        const float wx = ((l0 * e1x + l1 * e2x) + t * dx) + ox;
        const float wy = ((l0 * e1y + l1 * e2y) + t * dy) + oy;
        const float wz = ((l0 * e1z + l1 * e2z) + t * dz) + oz;
        float c0, c1, c2;
        contract_to_unit(wx, wy, wz, sdev, true, u[j][0], u[j][1], u[j][2], sd[j], c0, c1, c2);
        csum[0] += c0; csum[1] += c1; csum[2] += c2; tsum += t;
    }
}

template <uint32_t C>
__global__ __launch_bounds__(256) void k_march_features(UcnLevels lvls, const float *__restrict__ table, RayInputs in,
                                                        HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                        uint32_t lpb, int sample_major, float *__restrict__ features,
                                                        float *__restrict__ coord_out, float *__restrict__ tmean_out) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
    float u[6][3], sd[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, sd, csum, tsum);
    const uint32_t lvl0 = blockIdx.y * lpb;
    const uint32_t lvl1 = lvl0 + lpb < lvls.L ? lvl0 + lpb : lvls.L;
    featurise<C>(lvls, table, lvl0, lvl1, u, sd, 6, B, b, features, sample_major != 0);
    if (blockIdx.y == 0) {
        if (coord_out) {
            coord_out[b * 3 + 0] = csum[0] / 6.0f; coord_out[b * 3 + 1] = csum[1] / 6.0f; coord_out[b * 3 + 2] = csum[2] / 6.0f;
        }
        if (tmean_out) tmean_out[b] = tsum / 6.0f;
    }
}

// d(loss)/d(table) of k_march_features: grad_table[rows of the 6x8 corners] += w_corner * damp_j * g / 6.
// (means/stds carry no gradient: coord.track_linearize is @torch.no_grad, coord.py:75, and sdist is
//  detached, models.py:204-205.)  fp32 atomics in L2, like kernel_grid_backward (gridencoder.cu:336).
template <uint32_t C>
__global__ __launch_bounds__(256) void k_march_features_bwd(UcnLevels lvls, float *__restrict__ grad_table, RayInputs in,
                                                            HexPattern hx, float std_scale, uint32_t N, uint32_t S,
                                                            uint32_t lpb, int sample_major,
                                                            const float *__restrict__ grad_features) {
    const size_t B = (size_t)N * S;
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= B) return;
    const uint32_t ray = (uint32_t)(b / S), s = (uint32_t)(b - (size_t)ray * S);
    float u[6][3], sd[6], csum[3], tsum;
    cast_sample(in, hx, std_scale, ray, s, S, u, sd, csum, tsum);
    const uint32_t lvl0 = blockIdx.y * lpb;
    const uint32_t lvl1 = lvl0 + lpb < lvls.L ? lvl0 + lpb : lvls.L;
    featurise_bwd<C>(lvls, grad_table, lvl0, lvl1, u, sd, 6, B, b, grad_features, sample_major != 0);
}

// predict_density's featurisation for caller-supplied Gaussians (extract.py / API parity)
template <uint32_t C>
__global__ __launch_bounds__(256) void k_points_features(UcnLevels lvls, const float *__restrict__ table,
                                                         const float *__restrict__ means, const float *__restrict__ stds,
                                                         uint32_t Bn, uint32_t G, int warp, uint32_t lpb,
                                                         float *__restrict__ features, float *__restrict__ coord_out) {
    const size_t b = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= Bn) return;
    float u[6][3], sd[6];
    float cs0 = 0.0f, cs1 = 0.0f, cs2 = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 6; j++) {
        if (j < G) {
            const float *m = means + (b * G + j) * 3;
            float c0, c1, c2;
            contract_to_unit(m[0], m[1], m[2], stds[b * G + j], warp != 0, u[j][0], u[j][1], u[j][2], sd[j], c0, c1, c2);
            cs0 += c0; cs1 += c1; cs2 += c2;
        } else {
            u[j][0] = u[j][1] = u[j][2] = 0.0f; sd[j] = 1.0f;
        }
    }
    const uint32_t lvl0 = blockIdx.y * lpb;
    const uint32_t lvl1 = lvl0 + lpb < lvls.L ? lvl0 + lpb : lvls.L;
    featurise<C>(lvls, table, lvl0, lvl1, u, sd, G, Bn, b, features);
    if (blockIdx.y == 0 && coord_out) {
        coord_out[b * 3 + 0] = cs0 / (float)G; coord_out[b * 3 + 1] = cs1 / (float)G; coord_out[b * 3 + 2] = cs2 / (float)G;
    }
}

HexPattern make_hex() {
    HexPattern hx;
    const int order[6] = {0, 2, 4, 3, 5, 1};
    const float third = (float)(M_PI / 3.0), sixth = (float)(M_PI / 6.0), fivethirds = (float)(M_PI * 5.0 / 3.0);
    for (int j = 0; j < 6; j++) {
        const float a = third * (float)order[j];
        hx.ang[j] = a;
        hx.cs[0][j] = cosf(a);
        hx.sn[0][j] = sinf(a);
        const float o = fivethirds - (a + sixth);
        hx.cs[1][j] = cosf(o);
        hx.sn[1][j] = sinf(o);
        hx.cj[j] = (float)(3.0 / sqrt(7.0)) * ((float)(2 * j) / 5.0f - 1.0f);
    }
    return hx;
}

int field_levels(const ucn_field_t *f, UcnLevels *lv) {
    UCN_REQUIRE(f && f->embeddings && f->offsets_host && f->grid_sizes_host, "field: grid pointers missing");
    UCN_REQUIRE(f->level_dim == 1 || f->level_dim == 2 || f->level_dim == 4 || f->level_dim == 8,
                "GridEncoding: C must be 1, 2, 4, or 8.");
    return ucn_build_levels(lv, f->offsets_host, f->grid_sizes_host, f->num_levels, f->level_dim, 3,
                            f->log2_per_level_scale, f->base_resolution, 0, 0);
}

}  // namespace

extern "C" int ucn_march_features(const ucn_field_t *f, const float *sdist, const float *near_, const float *far_,
                                  const float *origins, const float *directions, const float *basis,
                                  const float *radii, const float *flip, const float *spin, float std_scale,
                                  uint32_t N, uint32_t S, uint32_t levels_per_block, int sample_major,
                                  float *features_out, float *coord_out, float *tmean_out, ucn_stream_t stream) {
    UCN_REQUIRE(sdist && near_ && far_ && origins && directions && basis && radii && features_out,
                "march_features: null pointer argument");
    UCN_REQUIRE((flip == nullptr) == (spin == nullptr), "march_features: flip and spin come together");
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (N == 0) return 0;
    if (levels_per_block == 0) levels_per_block = 1;
    const size_t B = (size_t)N * S;
    UCN_REQUIRE(B <= 0xFFFFFF00ull, "march_features: too many samples in one call (%zu)", B);
    const dim3 grid(ucn_div_up(B, 256), ucn_div_up(lv.L, levels_per_block));
    const RayInputs in{sdist, near_, far_, origins, directions, basis, radii, flip, spin};
    const HexPattern hx = make_hex();
    hipStream_t st = (hipStream_t)stream;
#define UCN_MF(CC)                                                                                              \
    hipLaunchKernelGGL(k_march_features<CC>, grid, dim3(256), 0, st, lv, f->embeddings, in, hx, std_scale, N, S, \
                       levels_per_block, sample_major, features_out, coord_out, tmean_out)
    switch (lv.C) {
        case 1: UCN_MF(1); break;
        case 2: UCN_MF(2); break;
        case 4: UCN_MF(4); break;
        case 8: UCN_MF(8); break;
    }
#undef UCN_MF
    UCN_LAUNCH_CHECK("march_features");
    return 0;
}

extern "C" int ucn_points_features(const ucn_field_t *f, const float *means, const float *stds, uint32_t B, uint32_t G,
                                   int warp, uint32_t levels_per_block, float *features_out, float *coord_out,
                                   ucn_stream_t stream) {
    UCN_REQUIRE(means && stds && features_out, "points_features: null pointer argument");
    UCN_REQUIRE(G >= 1 && G <= 6, "points_features: 1..6 Gaussians per feature, got %u", G);
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (B == 0) return 0;
    if (levels_per_block == 0) levels_per_block = 1;
    const dim3 grid(ucn_div_up(B, 256), ucn_div_up(lv.L, levels_per_block));
    hipStream_t st = (hipStream_t)stream;
#define UCN_PF(CC)                                                                                          \
    hipLaunchKernelGGL(k_points_features<CC>, grid, dim3(256), 0, st, lv, f->embeddings, means, stds, B, G, \
                       warp, levels_per_block, features_out, coord_out)
    switch (lv.C) {
        case 1: UCN_PF(1); break;
        case 2: UCN_PF(2); break;
        case 4: UCN_PF(4); break;
        case 8: UCN_PF(8); break;
    }
#undef UCN_PF
    UCN_LAUNCH_CHECK("points_features");
    return 0;
}

extern "C" int ucn_march_features_backward(const ucn_field_t *f, const float *sdist, const float *near_, const float *far_,
                                           const float *origins, const float *directions, const float *basis,
                                           const float *radii, const float *flip, const float *spin, float std_scale,
                                           uint32_t N, uint32_t S, uint32_t levels_per_block, int sample_major,
                                           const float *grad_features, float *grad_embeddings, ucn_stream_t stream) {
    UCN_REQUIRE(sdist && near_ && far_ && origins && directions && basis && radii && grad_features && grad_embeddings,
                "march_features_backward: null pointer argument");
    UCN_REQUIRE((flip == nullptr) == (spin == nullptr), "march_features_backward: flip and spin come together");
    UcnLevels lv;
    if (int rc = field_levels(f, &lv)) return rc;
    if (N == 0) return 0;
    if (levels_per_block == 0) levels_per_block = 1;
    const size_t B = (size_t)N * S;
    UCN_REQUIRE(B <= 0xFFFFFF00ull, "march_features_backward: too many samples in one call (%zu)", B);
    const dim3 grid(ucn_div_up(B, 256), ucn_div_up(lv.L, levels_per_block));
    const RayInputs in{sdist, near_, far_, origins, directions, basis, radii, flip, spin};
    const HexPattern hx = make_hex();
    hipStream_t st = (hipStream_t)stream;
#define UCN_MB(CC)                                                                                                  \
    hipLaunchKernelGGL(k_march_features_bwd<CC>, grid, dim3(256), 0, st, lv, grad_embeddings, in, hx, std_scale, N, \
                       S, levels_per_block, sample_major, grad_features)
    switch (lv.C) {
        case 1: UCN_MB(1); break;
        case 2: UCN_MB(2); break;
        case 4: UCN_MB(4); break;
        case 8: UCN_MB(8); break;
    }
#undef UCN_MB
    UCN_LAUNCH_CHECK("march_features_backward");
    return 0;
}
