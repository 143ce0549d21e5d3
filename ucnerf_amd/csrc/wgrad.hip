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

// r05: how a stage's 35 KB reach LDS.  Before, all of it travelled through 5 x 16 bytes of registers per thread and only ONE stage
// could be in flight: 35 KB per CU against ~2 us of loaded HBM latency is 8.3 bytes per clock and CU, 57 % of the HBM roofline, which is
// what the kernel reached (233 us per 983 k x (256 + 288) pass); a second register set does not fit beside the 144 accumulator
// registers.  Now the B image arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write), three stages ahead in a
// four-buffer ring, and the A image keeps the register path with two stages in flight.  (Everything by DMA was measured first: that
// path lands ~1 KB per 100 cycles and CU whatever the instruction carries -- 401 us with one row per instruction.)  A DMA instruction
// fills 1 KB of CONSECUTIVE image bytes, each lane fetching whatever chunk belongs there (pad bytes: the row's first chunk again).
// All loads are inline assembly and waited for by hand: vmcnt retires in issue order, an iteration issues [A(st + 2), B(st + 3)], so
// "A(st + 1) and everything older has landed" is "at most |B(st + 2)| + |A(st + 2)| + |B(st + 3)| of my operations outstanding".
// Every wave always issues its full share (rows outside the slab are clamped into the matrix and zeroed after landing), so the
// counts hold in the first and last stages too.
constexpr int kWgBStages = 4;               // B image ring: one being read, three in flight
constexpr int wg_stride_b(int ntb) { return ntb == 1 ? 64 : (ntb <= 5 ? 320 : 576); }   // >= 64 ntb and = 64 mod 256 (bank spread)
// The A image's staging registers are v240 .. v255 BY NAME (set S, chunk i: v[240 + 8 S + 4 i ..+3]): between the load's issue and
// its wait a whole stage passes, and a register the compiler knows as an asm OUTPUT is one it may copy or spill in that window --
// before the data is there.  The kernel's allocatable registers end at v239 (amdgpu_num_vgpr), the asm statements name the rest.
template <int S, int I>
__device__ __forceinline__ void wg_load_a(const uint16_t *p) {
    static_assert(S >= 0 && S < 2 && I >= 0 && I < 2, "two sets of two chunks");
#define UCN_WG_LD(R0, R1, R2, R3) asm volatile("global_load_dwordx4 v[" #R0 ":" #R3 "], %0, off" : : "v"(p) : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3)
    if constexpr (S == 0 && I == 0) UCN_WG_LD(240, 241, 242, 243);
    if constexpr (S == 0 && I == 1) UCN_WG_LD(244, 245, 246, 247);
    if constexpr (S == 1 && I == 0) UCN_WG_LD(248, 249, 250, 251);
    if constexpr (S == 1 && I == 1) UCN_WG_LD(252, 253, 254, 255);
#undef UCN_WG_LD
}
template <int S, int I>
__device__ __forceinline__ void wg_store_a(uint32_t lds_byte) {
#define UCN_WG_ST(R0, R3) asm volatile("ds_write_b128 %0, v[" #R0 ":" #R3 "]" : : "v"(lds_byte) : "memory")
    if constexpr (S == 0 && I == 0) UCN_WG_ST(240, 243);
    if constexpr (S == 0 && I == 1) UCN_WG_ST(244, 247);
    if constexpr (S == 1 && I == 0) UCN_WG_ST(248, 251);
    if constexpr (S == 1 && I == 1) UCN_WG_ST(252, 255);
#undef UCN_WG_ST
}
template <int N>
__device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NTB>
__global__ __launch_bounds__(kWgThreads, 1) __attribute__((amdgpu_num_vgpr(240))) void k_wgrad_bf16(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_img[];       // [2][A image] [kWgBStages][B image]
    constexpr int SB = wg_stride_b(NTB), kBImage = kWgRows * SB, kBInstr = kBImage / 1024;      // 2 / 10 / 18 DMA instructions per stage
    constexpr int kBOff = 2 * kWgImage;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t KB = a.kb1 + a.kb2;
    const uint32_t row0 = blockIdx.x * a.rows_per_slab;
    const uint32_t row1 = row0 + a.rows_per_slab < a.M ? row0 + a.rows_per_slab : a.M;
    const uint32_t n_stage = row1 > row0 ? (row1 - row0 + kWgRows - 1) / kWgRows : 0;
    const uint32_t ca = a.KA / 8, cb = KB / 8, cb1 = a.kb1 / 8;            // 16-byte chunks per row of A / of B (first block)
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)s_img;
    // ---- A: this thread's two chunks of a stage (chunk q = row q / ca, column chunk q % ca; a thread without one repeats the last)
    const uint16_t *acol[2];
    uint32_t arow[2], alds[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        uint32_t q = tid + i * kWgThreads;
        q = q < kWgRows * ca ? q : kWgRows * ca - 1;
        const uint32_t r = q / ca, c = q - r * ca;
        arow[i] = r;
        acol[i] = a.A + 8u * c;
        alds[i] = r * kWgStride + 16u * c;
    }
    // ---- B: this wave's DMA instructions i = wave, wave + 8, ... < kBInstr; lane l of instruction i fills image bytes 1024 i + 16 l
    constexpr int kBMine = (kBInstr + 7) / 8;                        // most instructions a wave issues per stage
    const int nb = (kBInstr - wave + 7) / 8;                          // this wave's (uniform): kBMine or kBMine - 1
    const uint16_t *bcol[kBMine];
    uint32_t brow[kBMine], bld[kBMine];
#pragma unroll
    for (int k = 0; k < kBMine; k++) {
        const uint32_t pbyte = 1024u * (wave + 8 * k) + 16u * lane;
        const uint32_t r = (pbyte / SB) & 31u, c = (pbyte % SB) / 16u;
        const uint32_t cc = c < cb ? c : 0u;                          // pad bytes of the row: its first chunk again
        brow[k] = r;
        bcol[k] = cc < cb1 ? a.B1 + 8u * cc : a.B2 + 8u * (cc - cb1);
        bld[k] = cc < cb1 ? a.ldb1 : a.ldb2;
    }
    auto fetch_a = [&](auto setc, uint32_t st) {
        constexpr int S = decltype(setc)::value;
        sfor<2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint32_t m = row0 + st * kWgRows + arow[i];
            wg_load_a<S, i>(acol[i] + (size_t)(m < a.M ? m : a.M - 1) * a.lda);
        });
    };
    auto deposit_a = [&](auto setc, uint32_t st) {           // behind the hand-placed wait
        constexpr int S = decltype(setc)::value;
        sfor<2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint32_t l = (st & 1u) * kWgImage + alds[i];
            wg_store_a<S, i>(lds0 + l);
            // a row outside the slab (its last stage only): zeros over what was just written (same lane, LDS writes stay in order)
            if (!(row0 + st * kWgRows + arow[i] < row1)) *reinterpret_cast<uint4 *>(s_img + l) = make_uint4(0, 0, 0, 0);
        });
    };
    auto issue_b = [&](uint32_t st) {
        const uint32_t img = lds0 + kBOff + (st % kWgBStages) * kBImage;
#pragma unroll
        for (int k = 0; k < kBMine; k++) {
            if (k < nb) {
                const uint32_t m = row0 + st * kWgRows + brow[k];
                const uint16_t *g = bcol[k] + (size_t)(m < a.M ? m : a.M - 1) * bld[k];
                const uint32_t l = img + 1024u * (wave + 8 * k);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(l), "v"(g) : "memory");
            }
        }
    };
    // rows of B(st) outside the slab (a slab's last stage only): zeroed by the lane that loaded them, after they have landed
    auto zero_b = [&](uint32_t st) {
        const uint32_t valid = row1 - (row0 + st * kWgRows);
        if (valid >= kWgRows) return;
        uint8_t *img = s_img + kBOff + (st % kWgBStages) * kBImage;
#pragma unroll
        for (int k = 0; k < kBMine; k++)
            if (k < nb && brow[k] >= valid) *reinterpret_cast<uint4 *>(img + 1024u * (wave + 8 * k) + 16u * lane) = make_uint4(0, 0, 0, 0);
    };
    // at most (B(st + 2), A(st + 2), B(st + 3)) of this wave outstanding = everything up to A(st + 1) has landed
    auto wait_stage = [&]() {
        if (nb == kBMine) wg_wait_vm<2 * kBMine + 2>();
        else wg_wait_vm<2 * (kBMine - 1) + 2>();
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    f32x16 acc[NTB];
#pragma unroll
    for (int t = 0; t < NTB; t++) zero_acc(acc[t]);
    const bool active = 32u * wave < a.KA;                          // waves beyond the A columns only help with the loads
    // lane part of the transposed reads: row 8 g + (j & 15) / 4, column chunk (4 (j & 3) + 16 ((j >> 4) & 1)) elements
    const uint32_t j = lane & 31, g = lane >> 5;
    const uint32_t lrow = 8u * g + ((j & 15u) >> 2), lcol = (4u * (j & 3u) + 16u * ((j >> 4) & 1u)) * 2u;
    const uint32_t lane_a = lrow * kWgStride + lcol, lane_b = lrow * SB + lcol;
    auto compute = [&](uint32_t st) {
        if (active) {
            const uint32_t abase = lds0 + (st & 1u) * kWgImage + lane_a + 64u * wave;       // A columns 32 w ..
            const uint32_t bbase = lds0 + kBOff + (st % kWgBStages) * kBImage + lane_b;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint64_t a_lo = tr_read(abase + ks * 16 * kWgStride), a_hi = tr_read(abase + ks * 16 * kWgStride + 4 * kWgStride);
                uint64_t b_lo[NTB], b_hi[NTB];
#pragma unroll
                for (int t = 0; t < NTB; t++) {
                    b_lo[t] = tr_read(bbase + ks * 16 * SB + 64u * t);
                    b_hi[t] = tr_read(bbase + ks * 16 * SB + 64u * t + 4 * SB);
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
    };
    // one stage: [issue A(st + 2), B(st + 3)] [MFMAs of st] [A(st + 1) -> LDS, B(st + 1) checked] barrier.  The barrier publishes
    // stage st + 1 and ends everybody's reads of stage st, whose buffers (A: st & 1, B: st % 4) the next iteration's issues refill.
    auto step = [&](auto cur, auto nxt, uint32_t st) {
        fetch_a(cur, st + 2);                   // set `cur` held A(st), deposited one iteration ago
        issue_b(st + 3);
        compute(st);                            // (nothing may sit between its operand reads and their wait: the reads' targets are
        //                                         registers the compiler believes written, and under pressure it would move them)
        wait_stage();
        deposit_a(nxt, st + 1);
        zero_b(st + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    if (n_stage) {
        issue_b(0);
        fetch_a(S0{}, 0);
        issue_b(1);
        fetch_a(S1{}, 1);
        issue_b(2);
        // A(0) and B(0): everything but (B(1), A(1), B(2)) has landed
        if (nb == kBMine) wg_wait_vm<2 * kBMine + 2>();
        else wg_wait_vm<2 * (kBMine - 1) + 2>();
        deposit_a(S0{}, 0);
        zero_b(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (uint32_t st = 0; st < n_stage; st += 2) {
            step(S0{}, S1{}, st);
            if (st + 1 < n_stage) step(S1{}, S0{}, st + 1);
        }
        wg_wait_vm<0>();                        // nothing of mine may land in LDS after I am gone
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
    const size_t lds = 2 * kWgImage + (size_t)kWgBStages * kWgRows * wg_stride_b((int)(KB / 32));
    switch (KB / 32) {
#define UCN_WG(N) case N: hipLaunchKernelGGL(k_wgrad_bf16<N>, dim3(used), dim3(kWgThreads), lds, st, a); break;
        UCN_WG(1) UCN_WG(2) UCN_WG(3) UCN_WG(4) UCN_WG(5) UCN_WG(6) UCN_WG(7) UCN_WG(8) UCN_WG(9)
#undef UCN_WG
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(ucn_div_up(n, 256)), dim3(256), 0, st, workspace, used, n, out);
    UCN_LAUNCH_CHECK("wgrad");
    return 0;
}
