// Weight gradients of the dense layers of a training step (the reference gets them from
// autograd through nn.Linear: models.py:438-483 / :743-820 under train.py:165-171's bf16 autocast):
//
//     C[KA][KB] (fp32) = A^T B,   A = pre-activation gradients [M, KA] (bf16),  B = layer inputs [M, KB] (bf16),  M ~ 1e6 samples
//
// -- a GEMM whose reduction runs over the SAMPLES, i.e. over the slow axis of both row-major operands, so neither operand
// is an MFMA fragment as it lies in memory (a fragment wants 8 consecutive k per lane).  The kernels that produce the
// operands (field_train.hip, sky_train.hip) write plain rows (one 64-byte sector per lane); here a workgroup stages 32-row
// slabs of both operands in LDS as they are (coalesced 16-byte loads, ds_write_b128) and reads them back TRANSPOSED with
// gfx950's ds_read_b64_tr_b16: within a group of 16 lanes the four 16-bit elements each lane receives are one column of the
// 4 x 16 block the group's 8-byte chunks form (tools/tr_probe.hip prints it), so lane (j, g) addressing row 8 g + (j & 15) / 4
// (+ 4 for the second read), column chunk 4 (j & 3) + 16 ((j >> 4) & 1) gets k = 8 g .. 8 g + 7 of column j -- the operand
// layout of v_mfma_f32_32x32x16_bf16 -- with no shuffles.  Row stride 576 bytes (= 64 mod 256): the four rows of a group and
// the two groups of a 32-lane half fall on disjoint banks.
//
// Decomposition: 8 waves per workgroup, wave w owns A columns [32 w, 32 w + 32) against ALL B tiles (<= 9 x 32 columns:
// 144 accumulator registers), so every operand element is read from HBM exactly once per launch; split-K over the samples
// in `n_slabs` slabs (about one per CU) whose fp32 partials are summed by a second small kernel in a fixed order
// (deterministic, unlike atomics).  B may come as TWO column blocks of different buffers (a layer's hidden inputs and the
// per-sample auxiliary tile that carries the constant 1 of the bias), so that block need not be copied behind every layer.
// HBM-bound by construction: 36 KiB of operands per 18 MFMAs per wave.
#include "bf_tiles.h"

namespace {

constexpr int kWgRows = 32;                 // rows (samples) per stage = two k-steps of 16
constexpr int kWgStride = 576;              // LDS row stride in bytes of both images (A: 256 columns + 64 pad, B: 288 columns)
constexpr int kWgImage = kWgRows * kWgStride;       // 18 KiB
constexpr int kWgThreads = 512;

struct WgradArgs {
    const uint16_t *A;          // [M, lda], columns a0 .. a0 + KA
    const uint16_t *B1, *B2;    // column blocks of B: [M, ldb1] kb1 columns, then [M, ldb2] kb2 columns (kb2 may be 0)
    uint32_t lda, ldb1, ldb2, KA, kb1, kb2;
    uint32_t M, rows_per_slab;
    float *partial;             // [n_slabs][KA][KB]
};

__device__ __forceinline__ uint64_t tr_read(uint32_t lds_byte) {
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte) : "memory");
    return v;
}

template <int NTB>
__global__ __launch_bounds__(kWgThreads, 1) void k_wgrad_bf16(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_img[];       // [2 stages][A image | B image]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t KB = a.kb1 + a.kb2;
    const uint32_t row0 = blockIdx.x * a.rows_per_slab;
    const uint32_t row1 = row0 + a.rows_per_slab < a.M ? row0 + a.rows_per_slab : a.M;
    const uint32_t n_stage = row1 > row0 ? (row1 - row0 + kWgRows - 1) / kWgRows : 0;
    // ---- this thread's 16-byte chunks of a stage: chunk q of the A image = (row q / ca, column chunk q % ca), then B's
    const uint32_t ca = a.KA / 8, cb = KB / 8, cb1 = a.kb1 / 8;
    const uint32_t n_chunks = kWgRows * (ca + cb);
    constexpr int kPer = (kWgRows * (32 + 36) + kWgThreads - 1) / kWgThreads;       // 5: at most 256 + 288 columns
    uint4 stage[kPer];
    auto fetch = [&](uint32_t st) {
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const uint32_t q = tid + i * kWgThreads;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < n_chunks) {
                const bool isA = q < kWgRows * ca;
                const uint32_t qq = isA ? q : q - kWgRows * ca;
                const uint32_t per = isA ? ca : cb;
                const uint32_t r = qq / per, c = qq - r * per;
                const uint32_t m = row0 + st * kWgRows + r;
                if (m < row1) {
                    const uint16_t *p = isA ? a.A + (size_t)m * a.lda + 8u * c
                                            : (c < cb1 ? a.B1 + (size_t)m * a.ldb1 + 8u * c : a.B2 + (size_t)m * a.ldb2 + 8u * (c - cb1));
                    v = *reinterpret_cast<const uint4 *>(p);
                }
            }
            stage[i] = v;
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kPer; i++) {
            const uint32_t q = tid + i * kWgThreads;
            if (q < n_chunks) {
                const bool isA = q < kWgRows * ca;
                const uint32_t qq = isA ? q : q - kWgRows * ca;
                const uint32_t per = isA ? ca : cb;
                const uint32_t r = qq / per, c = qq - r * per;
                *reinterpret_cast<uint4 *>(s_img + buf * 2 * kWgImage + (isA ? 0 : kWgImage) + r * kWgStride + 16u * c) = stage[i];
            }
        }
    };
    f32x16 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; t++) zero_acc(acc[t]);
    const bool active = 32u * wave < a.KA;                          // waves beyond the A columns only help with the loads
    // lane part of the transposed reads: row 8 g + (j & 15) / 4, column chunk (4 (j & 3) + 16 ((j >> 4) & 1)) elements
    const uint32_t j = lane & 31, g = lane >> 5;
    const uint32_t lane_off = (8u * g + ((j & 15u) >> 2)) * kWgStride + (4u * (j & 3u) + 16u * ((j >> 4) & 1u)) * 2u;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)s_img;
    if (n_stage) {
        fetch(0);
        deposit(0);
    }
    __syncthreads();
    for (uint32_t st = 0; st < n_stage; st++) {
        const int buf = st & 1;
        if (st + 1 < n_stage) fetch(st + 1);                         // next stage's global loads fly under this stage's MFMAs
        if (active) {
            const uint32_t abase = lds0 + buf * 2 * kWgImage + lane_off + 64u * wave;       // A columns 32 w ..
            const uint32_t bbase = lds0 + buf * 2 * kWgImage + kWgImage + lane_off;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint32_t ro = ks * 16 * kWgStride;
                const uint64_t a_lo = tr_read(abase + ro), a_hi = tr_read(abase + ro + 4 * kWgStride);
                uint64_t b_lo[NTB], b_hi[NTB];
#pragma unroll
                for (int t = 0; t < NTB; t++) {
                    b_lo[t] = tr_read(bbase + ro + 64u * t);
                    b_hi[t] = tr_read(bbase + ro + 64u * t + 4 * kWgStride);
                }
                // the reads above are inline asm: the compiler does not count them.  One wait, then every value is passed through
                // an (ordered) empty asm so that no MFMA can be scheduled in front of the wait.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                uint64_t a0 = a_lo, a1 = a_hi;
                asm volatile("" : "+v"(a0), "+v"(a1));
#pragma unroll
                for (int t = 0; t < NTB; t++) asm volatile("" : "+v"(b_lo[t]), "+v"(b_hi[t]));
                typedef uint64_t u2 __attribute__((ext_vector_type(2)));
                const u2 av = {a0, a1};
                const bf8 af = __builtin_bit_cast(bf8, av);
#pragma unroll
                for (int t = 0; t < NTB; t++) {
                    const u2 bv = {b_lo[t], b_hi[t]};
                    acc[t] = mfma_bf(af, __builtin_bit_cast(bf8, bv), acc[t]);
                }
            }
        }
        // buffer buf ^ 1 was last read in the previous iteration, in front of its barrier: it can be refilled right away;
        // the barrier below publishes it and ends everybody's reads of `buf`
        if (st + 1 < n_stage) deposit(buf ^ 1);
        __syncthreads();
    }
    // ---- partial C of this slab: acc[t] register r of lane (j, g) = C[32 w + (r & 3) + 8 (r >> 2) + 4 g][32 t + j]
    if (active) {
        float *out = a.partial + (size_t)blockIdx.x * a.KA * KB;
#pragma unroll
        for (int t = 0; t < NTB; t++)
            if (32u * t < KB) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t i = 32u * wave + (r & 3) + 8 * (r >> 2) + 4 * g;
                    out[(size_t)i * KB + 32u * t + j] = acc[t][r];
                }
            }
    }
}

// out[i] = sum over the slabs in a fixed order (bit-reproducible): eight interleaved chains per element keep eight loads in
// flight per thread (one chain: 61 us per 75 MB of partials, 1.2 TB/s), added together at the end
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, uint32_t n_slabs, uint32_t n,
                                                      float *__restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t k = 0;
    for (; k + 8 <= n_slabs; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) s[u] += partial[(size_t)(k + u) * n + i];
    }
    for (; k < n_slabs; k++) s[k & 7u] += partial[(size_t)k * n + i];
    out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

uint32_t wgrad_slabs(uint64_t M) {
    const uint64_t stages = (M + kWgRows - 1) / kWgRows;
    return (uint32_t)(stages < 256 ? (stages ? stages : 1) : 256);   // one workgroup per CU when there is enough work
}

}  // namespace

extern "C" uint64_t ucn_wgrad_ws_floats(uint32_t KA, uint32_t KB, uint64_t M) { return (uint64_t)wgrad_slabs(M) * KA * KB; }

extern "C" int ucn_wgrad_bf16(const void *A, uint32_t lda, uint32_t KA, const void *B1, uint32_t ldb1, uint32_t kb1, const void *B2,
                              uint32_t ldb2, uint32_t kb2, uint64_t M, float *workspace, float *out, ucn_stream_t stream) {
    UCN_REQUIRE(A && B1 && workspace && out, "wgrad: null pointer argument");
    UCN_REQUIRE(KA >= 32 && KA <= 256 && KA % 32 == 0, "wgrad: KA = %u (a multiple of 32 up to 256)", KA);
    UCN_REQUIRE(kb1 % 32 == 0 && kb2 % 32 == 0 && kb1 >= 32 && kb1 + kb2 <= 288, "wgrad: B columns %u + %u (multiples of 32, together <= 288)", kb1, kb2);
    UCN_REQUIRE(kb2 == 0 || B2, "wgrad: second B block missing");
    UCN_REQUIRE(lda % 8 == 0 && ldb1 % 8 == 0 && (kb2 == 0 || ldb2 % 8 == 0), "wgrad: row strides must be multiples of 8 elements (16-byte loads)");
    UCN_REQUIRE(((uintptr_t)A | (uintptr_t)B1 | (uintptr_t)B2) % 16 == 0, "wgrad: operands must be 16-byte aligned");
    UCN_REQUIRE(M < 0xFFFFFF00ull, "wgrad: too many rows");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t KB = kb1 + kb2, n = KA * KB;
    if (M == 0) {
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(ucn_div_up(n, 256)), dim3(256), 0, st, workspace, 0u, n, out);
        UCN_LAUNCH_CHECK("wgrad");
        return 0;
    }
    const uint32_t slabs = wgrad_slabs(M);
    const uint32_t stages = (uint32_t)((M + kWgRows - 1) / kWgRows);
    const uint32_t rows_per_slab = (stages + slabs - 1) / slabs * kWgRows;
    const uint32_t used = (uint32_t)((M + rows_per_slab - 1) / rows_per_slab);
    WgradArgs a{(const uint16_t *)A, (const uint16_t *)B1, (const uint16_t *)B2, lda, ldb1, ldb2, KA, kb1, kb2, (uint32_t)M, rows_per_slab, workspace};
    const size_t lds = 2 * 2 * kWgImage;
    switch (KB / 32) {
#define UCN_WG(N) case N: hipLaunchKernelGGL(k_wgrad_bf16<N>, dim3(used), dim3(kWgThreads), lds, st, a); break;
        UCN_WG(1) UCN_WG(2) UCN_WG(3) UCN_WG(4) UCN_WG(5) UCN_WG(6) UCN_WG(7) UCN_WG(8) UCN_WG(9)
#undef UCN_WG
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(ucn_div_up(n, 256)), dim3(256), 0, st, workspace, used, n, out);
    UCN_LAUNCH_CHECK("wgrad");
    return 0;
}
