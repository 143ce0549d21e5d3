// Shared host/device helpers for libucnerf_march.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ucnerf_march.h"

// ---------------------------------------------------------------- error plumbing
extern thread_local char g_ucn_err[512];
int ucn_fail(const char *fmt, ...);
#define UCN_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) return ucn_fail(__VA_ARGS__);  \
    } while (0)
#define UCN_LAUNCH_CHECK(name)                                                             \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) return ucn_fail("%s launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------- per-level constants
// One level of the multi-resolution table.  Derived on the host (ref gridencoder.cu:137-139,
// models.py:495) and passed to kernels by value so that the device, the host and the CPU
// oracle share identical float constants.
struct UcnLevel {
    float scale;          // exp2f(level*S)*H - 1
    uint32_t resolution;  // ceil(scale)+1
    uint32_t rows;        // hashmap_size of the level
    uint32_t first_row;   // offsets[level]
    float gs2;            // float(int32 wrap of grid_sizes[level]^2)   (models.py:495 quirk)
    uint32_t hashed;      // 1: xor-prime hash, 0: strided ("dense") addressing
    uint32_t mask;        // rows-1 if rows is a power of two else 0
    // Strides of the dense walk of gridencoder.cu:71-75, INCLUDING its uint32 wrap-around and its
    // early exit: stride[d] = 0 for dimensions the walk never reaches.  (With side = 65537 the
    // product 65537^2 wraps to 131073 <= rows, so that level is *not* hashed in the reference.)
    uint32_t stride[5];
    float inv_gs;         // 1/sqrt(gs2): erf argument = inv_gs / sqrt(8 std^2)   (models.py:495)
};
struct UcnLevels {
    UcnLevel lv[UCN_MAX_LEVELS];
    uint32_t L;
    uint32_t C;
};

int ucn_build_levels(UcnLevels *out, const int32_t *offsets_host, const int32_t *grid_sizes_host,
                     uint32_t L, uint32_t C, uint32_t D, float S, uint32_t H, uint32_t gridtype,
                     int align_corners);

// ---------------------------------------------------------------- device helpers
#define UCN_EPS 1.1920928955078125e-07f  // torch.finfo(float32).eps

__device__ __forceinline__ uint32_t ucn_hash3(uint32_t x, uint32_t y, uint32_t z) {
    // gridencoder.cu:50-63 with primes {1, 2654435761, 805459861}
    return x ^ (y * 2654435761u) ^ (z * 805459861u);
}

// Row of an integer lattice point inside one level, D = 3, align_corners = false
// (gridencoder.cu:66-84).  `side` = resolution + 1.
__device__ __forceinline__ uint32_t ucn_row3(const UcnLevel &lv, uint32_t side, uint32_t x, uint32_t y,
                                             uint32_t z) {
    (void)side;
    uint32_t idx = lv.hashed ? ucn_hash3(x, y, z) : x * lv.stride[0] + y * lv.stride[1] + z * lv.stride[2];
    return lv.mask ? (idx & lv.mask) : (idx % lv.rows);
}

static inline uint32_t ucn_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
