// bf16 MFMA tile helpers shared by the training kernels (field_train.hip) and the mixed-precision sky layer (sky.hip):
// a wave owns 32 samples (the N of v_mfma_f32_32x32x16_bf16); activations live in registers as B operands (two k-steps of
// 8 bf16 per 32-wide tile), weights arrive as 1 KiB A-fragments through the LDS-DMA ring of mlp_ring.h.
#pragma once
#include <utility>

#include "mlp_ring.h"
#include "wave_dpp.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf(bf8 a, bf8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// fp32 -> bf16, round to nearest even: the C cast is v_cvt_pk_bf16_f32 on gfx950 (two values per instruction)
// (written as two-element vector conversions: element by element -- these files are built without the SLP vectoriser --
// the compiler emitted one conversion per VALUE plus a v_perm_b32 per pair, twice the instructions)
__device__ __forceinline__ bf8 pack8(const float (&v)[8]) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
    bf8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f2v t = {v[e], v[e + 1]};
        const bf2v r = __builtin_convertvector(t, bf2v);
        o[e] = r[0];
        o[e + 1] = r[1];
    }
    return o;
}
// 8 accumulator registers (k-step s of a tile) -> the B operand of the next layer
__device__ __forceinline__ bf8 to_b(const f32x16 &a, int s, bool relu) {
#ifdef UCN_EXP_NOCVT        // experiment builds (tools/build_variant.sh): timing only, results are garbage
    return __builtin_bit_cast(bf8, make_uint4(__float_as_uint(a[8 * s]), __float_as_uint(a[8 * s + 1]), __float_as_uint(a[8 * s + 2]),
                                              __float_as_uint(a[8 * s + 3])));
#endif
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = a[8 * s + e];
    const bf8 o = pack8(v);
    if (!relu) return o;
    // ReLU on the packed result: a negative bf16 is a negative int16 (v_pk_max_i16, two values per instruction; -0 -> +0)
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf8, __builtin_elementwise_max(__builtin_bit_cast(s8v, o), z));
}
__device__ __forceinline__ void zero_acc(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = 0.0f;
}
__device__ __forceinline__ void load_acc(const float *__restrict__ p, f32x16 &acc) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = reinterpret_cast<const float4 *>(p)[q];
        acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
    }
}

// ring geometry: 4 x 16 KiB chunks filled two chunks ahead = 64 KiB, two workgroups per CU
constexpr int kTChunk = 16, kTSlots = 4, kTLead = 2;

template <int... Is, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

// P output tiles (a PAIR, or one) from NT_IN input tiles: acc[o2] += A(frag) . in[it][s], fragments [it][s][o2] from
// stream position G0.  The two tiles of a pair alternate (two MFMAs into the same accumulator do not issue back to
// back); only the pair's 32 accumulator registers are live, the caller converts / stores it before the next pair.
template <int P, int NT_IN, int G0, class RING>
__device__ __forceinline__ void tile_pair(RING &ring, f32x16 (&acc)[P], const bf8 (&in)[NT_IN][2]) {
    sfor<NT_IN * 2 * P>([&](auto i) {
        constexpr int I = i.value, G = G0 + I;
        constexpr int o2 = I % P, s = (I / P) % 2, it = I / (2 * P);
        constexpr int CH = RING::kChunk, NW = RING::kWaves;     // a wave issues one DMA piece per NW fragments
        if constexpr (G % CH == 0 && G / CH >= 1) ring.template boundary<G / CH>();
        if constexpr (G % NW == 0) ring.template piece<G / CH + RING::kLeadChunks, (G % CH) / NW>();
        acc[o2] = mfma_bf(__builtin_bit_cast(bf8, ring.template group<G>()), in[it][s], acc[o2]);
        // one operand read per MFMA: left alone, the scheduler hoists a chunk's sixteen reads (64 registers) to its start
#ifndef UCN_EXP_NOSGB
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#endif
    });
}
// ---- the same pair for TWO sample tiles of the wave (64 samples; r05) ------------------------------------------------------------------
// Every weight fragment read from LDS feeds two MFMAs, so the LDS-DMA stream (what the training kernels are bound by: ~10 B per clock
// and CU land) and the LDS operand reads are paid once per 64 samples.  Both tiles' in AND out activations are 256 registers plus 64
// of accumulators: the kernel needs the whole 512-entry file and its two halves used on purpose -- the compiler's MFMA takes its A / B
// operands from arch VGPRs only and copied every operand parked in the accumulation half back per use (4600 v_accvgpr moves and 41
// registers of scratch in the first build, each scratch reload a vmcnt(0) that drains the stream's look-ahead).  Here the MFMA is inline
// assembly with the classes written out: accumulators and ONE of the two activation buffers live in AGPRs (the hardware reads B
// operands from either half), the other buffer and everything the VALU touches in VGPRs.  What the compiler no longer knows it cannot
// guard: the caller puts the MFMA -> VALU read wait states behind a chain (pair_settle) itself.
template <bool INA>
__device__ __forceinline__ void mfma_bf_cls(f32x16 &acc, const bf8 &w, const bf8 &in) {
    if constexpr (INA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "a"(in));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(in));
}
template <bool INA>
__device__ __forceinline__ void mfma_bf_cls_first(f32x16 &acc, const bf8 &w, const bf8 &in) {     // acc = A . B (+ 0)
    if constexpr (INA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "a"(in));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(in));
}
// an 8-pass MFMA's result is readable by the VALU 11 wait states after issue (the hazard the compiler inserts s_nop for when it knows
// the instruction); 16 + 8 covers the last four of the chain
__device__ __forceinline__ void pair_settle(f32x16 (&a0)[2], f32x16 (&a1)[2]) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a0[0]), "+a"(a0[1]), "+a"(a1[0]), "+a"(a1[1]));
}
// ... and a VALU write (v_accvgpr_mov / _write of a bias start, the conversions that produce B operands) must be 2 wait states old when
// an MFMA reads the register (seen: element 0 of every biased pair's accumulator, copied right in front of the chain, read stale)
__device__ __forceinline__ void pair_ready(f32x16 (&a0)[2], f32x16 (&a1)[2]) {
    asm volatile("s_nop 7" : "+a"(a0[0]), "+a"(a0[1]), "+a"(a1[0]), "+a"(a1[1]));
}
__device__ __forceinline__ bf8 to_agpr(bf8 v) {
    asm volatile("" : "+a"(v));
    return v;
}
#ifndef UCN_W_AHEAD
#define UCN_W_AHEAD 3
#endif
constexpr int kWAhead = UCN_W_AHEAD, kWSlots = UCN_W_AHEAD + 1;   // weight fragments requested ahead of their MFMAs; register slots of the pipe
// request fragment GF: the ring's housekeeping rides on the requests (as pipe_fetch of mlp_ring.h)
template <int GF, int NG, class RING>
__device__ __forceinline__ void frag_fetch(RING &ring, bf8 (&wp)[kWSlots]) {
    if constexpr (GF < NG) {
        constexpr int CH = RING::kChunk, NW = RING::kWaves;
        if constexpr (GF % CH == 0 && GF / CH >= 1) ring.template boundary<GF / CH>();
        if constexpr (GF % NW == 0) ring.template piece<GF / CH + RING::kLeadChunks, (GF % CH) / NW>();
        wp[GF % kWSlots] = __builtin_bit_cast(bf8, ring.template group<GF>());
    }
}
// tile_pair with the fragment pipe (r05): the plain form above issues "one operand read, one MFMA" and the register allocator gives every
// read the SAME four registers -- read, wait for it, MFMA: with one wave per SIMD the LDS latency stands in front of every MFMA (a pair's
// 32 MFMAs took ~1900 cycles instead of 1024).  Here fragment G + kWAhead is requested in front of the MFMA of G (program order pinned
// by a scheduling barrier per step), the pipe lives across pairs and layers.
template <int P, int NT_IN, int G0, int NG, class RING>
__device__ __forceinline__ void tile_pair_pf(RING &ring, bf8 (&wp)[kWSlots], f32x16 (&acc)[P], const bf8 (&in)[NT_IN][2]) {
    sfor<NT_IN * 2 * P>([&](auto i) {
        constexpr int I = i.value, G = G0 + I;
        constexpr int o2 = I % P, s = (I / P) % 2, it = I / (2 * P);
        frag_fetch<G + kWAhead, NG>(ring, wp);
        acc[o2] = mfma_bf(wp[G % kWSlots], in[it][s], acc[o2]);
        __builtin_amdgcn_sched_barrier(0);
    });
}
// acc{0,1}[o2] (+)= A(frag) . in{0,1}: tiles 0..7 of the input from in (class INA), tile 8 (NT_IN = 9) = the per-ray tile, always AGPRs
template <int NT_IN, int G0, int NG, bool INA, bool ZERO, class RING>
__device__ __forceinline__ void tile_pair2(RING &ring, bf8 (&wp)[kWSlots], f32x16 (&acc0)[2], f32x16 (&acc1)[2], const bf8 (&in0)[8][2],
                                           const bf8 (&in1)[8][2], const bf8 (&aux0)[2], const bf8 (&aux1)[2]) {
    sfor<NT_IN * 4>([&](auto i) {
        constexpr int I = i.value, G = G0 + I;
        constexpr int o2 = I % 2, s = (I / 2) % 2, it = I / 4;
        frag_fetch<G + kWAhead, NG>(ring, wp);
        if constexpr (it < 8) {
            if constexpr (ZERO && I < 2) {
                mfma_bf_cls_first<INA>(acc0[o2], wp[G % kWSlots], in0[it][s]);
                mfma_bf_cls_first<INA>(acc1[o2], wp[G % kWSlots], in1[it][s]);
            } else {
                mfma_bf_cls<INA>(acc0[o2], wp[G % kWSlots], in0[it][s]);
                mfma_bf_cls<INA>(acc1[o2], wp[G % kWSlots], in1[it][s]);
            }
        } else {
            mfma_bf_cls<true>(acc0[o2], wp[G % kWSlots], aux0[s]);
            mfma_bf_cls<true>(acc1[o2], wp[G % kWSlots], aux1[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}
// bit r = accumulator register r is positive (the ReLU mask of the lane's 16 features of a tile).  One v_alignbit_b32 per
// register shifts its SIGN bit into the word ((m << 1) | sign), registers taken from 15 down to 0, then one inversion: 17 VALU
// instructions per tile where compare + select + or took 48 (the training kernels' masks were 2700 of the sky forward's 5600
// VALU instructions per wave).  A pre-activation of exactly +0 counts as positive (torch: gradient 0 there): a set of
// measure zero -- MFMA accumulation from a +0 / bias start cannot produce -0, and a bias that cancels a dot product exactly
// does not occur in practice.
__device__ __forceinline__ uint32_t mask16(const f32x16 &a) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 15; r >= 0; r--) m = __builtin_amdgcn_alignbit(m, __float_as_uint(a[r]), 31);
    return ~m & 0xFFFFu;
}
// the masked (ReLU') half tile as a B operand: v_bfe_i32 spreads the register's mask bit over a word, one v_and_b32 applies it
__device__ __forceinline__ bf8 to_b_masked(const f32x16 &a, int s, uint32_t bits) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++)
        v[e] = __uint_as_float(__float_as_uint(a[8 * s + e]) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, 8 * s + e, 1));
    return pack8(v);
}
// store a tile of activations: lane (j, h) holds features 32t + (r&3) + 8(r>>2) + 4h of sample j -- four 8-byte pieces
// {0-3, 8-11, 16-19, 24-27} + 4h.  The two lanes of a sample first trade pieces (v_permlane32_swap: lane j's pieces 2, 3
// against lane j + 32's pieces 0, 1), after which lane (j, 0) holds features 0-15 and lane (j, 1) features 16-31 of the
// tile contiguously: two 16-byte stores per lane instead of four 8-byte ones -- the kernels' 1.8 GB of activations leave
// in half as many write transactions.  (Transposing whole rows through LDS: 1.13 -> 1.06 ms, not worth 66 KiB of LDS.)
__device__ __forceinline__ void store_tile(uint16_t *__restrict__ dst, uint32_t width, uint32_t sample, int t, int h, const bf8 (&b)[2],
                                           bool live) {
    const uint4 lo = __builtin_bit_cast(uint4, b[0]), hi = __builtin_bit_cast(uint4, b[1]);
    // (a, b) -> a keeps lanes 0-31 and takes b's lanes 0-31 into its lanes 32-63; b takes a's lanes 32-63 into its lanes 0-31
    const auto s0 = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(lo.z, hi.z, false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap(lo.w, hi.w, false, false);
    if (!live) return;
    uint4 *p = reinterpret_cast<uint4 *>(dst + (size_t)sample * width + 32 * t + 16 * h);
    p[0] = make_uint4(s0[0], s1[0], s0[1], s1[1]);      // h = 0: features 0-3 (own), 4-7 (partner);  h = 1: 16-19, 20-23
    p[1] = make_uint4(s2[0], s3[0], s2[1], s3[1]);      // h = 0: features 8-11, 12-15;                h = 1: 24-27, 28-31
}
// ... and for the two tiles of an output PAIR (64 adjacent features = 128 bytes of the row): the two lanes of a sample
// trade whole tiles' worth of pieces, lane (j, 0) ends up with all 32 features of tile tp, lane (j, 1) with those of tile
// tp + 1 -- every lane writes one full 64-byte sector (4 x 16 bytes), same instruction count as two store_tile calls.
__device__ __forceinline__ void store_pair(uint16_t *__restrict__ dst, uint32_t width, uint32_t sample, int tp, int h,
                                           const bf8 (&t0)[2], const bf8 (&t1)[2], bool live) {
    const uint4 a0 = __builtin_bit_cast(uint4, t0[0]), a1 = __builtin_bit_cast(uint4, t0[1]);   // tile tp:     pieces 0,1 | 2,3
    const uint4 b0 = __builtin_bit_cast(uint4, t1[0]), b1 = __builtin_bit_cast(uint4, t1[1]);   // tile tp + 1
    // swap(a, b): lanes 0-31 end with (a, partner's a) = both halves of tile tp's piece, lanes 32-63 with (own b's partner, b)
    const auto p0x = __builtin_amdgcn_permlane32_swap(a0.x, b0.x, false, false), p0y = __builtin_amdgcn_permlane32_swap(a0.y, b0.y, false, false);
    const auto p1x = __builtin_amdgcn_permlane32_swap(a0.z, b0.z, false, false), p1y = __builtin_amdgcn_permlane32_swap(a0.w, b0.w, false, false);
    const auto p2x = __builtin_amdgcn_permlane32_swap(a1.x, b1.x, false, false), p2y = __builtin_amdgcn_permlane32_swap(a1.y, b1.y, false, false);
    const auto p3x = __builtin_amdgcn_permlane32_swap(a1.z, b1.z, false, false), p3y = __builtin_amdgcn_permlane32_swap(a1.w, b1.w, false, false);
    if (!live) return;
#ifdef UCN_EXP_NOSTORE      // timing-only experiment build: what do the activation stores (and the waits they widen) cost?
    return;
#endif
    uint4 *p = reinterpret_cast<uint4 *>(dst + (size_t)sample * width + 32 * (tp + h));
    p[0] = make_uint4(p0x[0], p0y[0], p0x[1], p0y[1]);      // features 0-3 (lane j's piece 0), 4-7 (lane j + 32's piece 0)
    p[1] = make_uint4(p1x[0], p1y[0], p1x[1], p1y[1]);      // 8-11, 12-15
    p[2] = make_uint4(p2x[0], p2y[0], p2x[1], p2y[1]);      // 16-19, 20-23
    p[3] = make_uint4(p3x[0], p3y[0], p3x[1], p3y[1]);      // 24-27, 28-31
}
// ... and the same pair through the wave's own LDS tile (r05; kernels that run ONE workgroup per CU have the room): the per-lane form
// above is 4 store instructions of 64 separate 16-byte requests each (a lane = a row), 256 texture-address cycles per pair and wave --
// with four waves per CU as long as the pair's 32 MFMAs take, i.e. the training kernels were bound by the address unit, not by bytes
// (without the stores the sky forward kernel runs 1.29 ms instead of 1.89).  Here a lane writes its eight 8-byte pieces to row j of a
// [32][128 B + 16] tile (no lane swaps), the wave reads the tile back row-contiguous and stores 8 rows x one whole 128-byte line per
// instruction: 8 requests instead of 64.  `sample0` = the wave's first sample, `n_rows` = how many of its 32 rows exist.
constexpr int kStageRow = 144;                                       // bytes
constexpr int kStageTile = 32 * kStageRow;                           // 4.5 KiB per wave
__device__ __forceinline__ void store_pair_staged(uint8_t *__restrict__ tile, uint16_t *__restrict__ dst, uint32_t width, uint32_t sample0,
                                                  uint32_t n_rows, int tp, int lane, const bf8 (&t0)[2], const bf8 (&t1)[2]) {
    const int j = lane & 31, h = lane >> 5;
    wave_lds_handoff();                                               // the previous pair's reads are done
    // ds_write_b64 is served 16 consecutive lanes at a time on 32 banks; with a 36-dword row stride rows j and j + 8 start on the same
    // bank (2-way conflicts on every write: 5.0e7 conflict cycles per launch, SQ_LDS_BANK_CONFLICT, where the kernels had none before
    // the staging).  Rows with bit 3 set keep the two 8-byte halves of each 16-byte unit swapped; the read below (row >> 3 == i) undoes
    // it for free by naming the dwords in the other order.
    uint8_t *row = tile + j * kStageRow + 8 * (h ^ ((j >> 3) & 1));
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint4 lo = __builtin_bit_cast(uint4, t == 0 ? t0[0] : t1[0]), hi = __builtin_bit_cast(uint4, t == 0 ? t0[1] : t1[1]);
        // bf8 [0] = features {0-3, 8-11} + 4h, [1] = {16-19, 24-27} + 4h of the tile: 8-byte pieces at byte 2 x feature
        *reinterpret_cast<uint2 *>(row + 64 * t + 0) = make_uint2(lo.x, lo.y);
        *reinterpret_cast<uint2 *>(row + 64 * t + 16) = make_uint2(lo.z, lo.w);
        *reinterpret_cast<uint2 *>(row + 64 * t + 32) = make_uint2(hi.x, hi.y);
        *reinterpret_cast<uint2 *>(row + 64 * t + 48) = make_uint2(hi.z, hi.w);
    }
    wave_lds_handoff();
    // a full tile (wave-uniform; all but a launch's last tiles) stores without per-row predicates: each predicate is an exec save, a
    // branch and a restore around ONE store -- 700 of the sky forward kernel's 10 k instructions per wave, at one wave per SIMD
    const bool full = __builtin_amdgcn_readfirstlane(n_rows) == 32u;
    const uint32_t r0 = (uint32_t)lane >> 3, c = (uint32_t)lane & 7u;
    const uint8_t *src = tile + r0 * kStageRow + 16u * c;
    uint16_t *out = dst + (size_t)(sample0 + r0) * width + 32 * tp + 8u * c;
    const size_t step = (size_t)8 * width;                            // 8 rows further per instruction
#ifndef UCN_EXP_NOSTORE
    if (full) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 u = *reinterpret_cast<const uint4 *>(src + i * 8 * kStageRow);
            *reinterpret_cast<uint4 *>(out + i * step) = (i & 1) ? make_uint4(u.z, u.w, u.x, u.y) : u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 u = *reinterpret_cast<const uint4 *>(src + i * 8 * kStageRow);
            if (8u * i + r0 < n_rows) *reinterpret_cast<uint4 *>(out + i * step) = (i & 1) ? make_uint4(u.z, u.w, u.x, u.y) : u;
        }
    }
#else
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint4 u = *reinterpret_cast<const uint4 *>(src + i * 8 * kStageRow);
        asm volatile("" ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w));
    }
#endif
}
template <bool PAIR>
__device__ __forceinline__ void store_two(uint16_t *__restrict__ dst, uint32_t width, uint32_t sample, int tp, int h, const bf8 (&t0)[2],
                                          const bf8 (&t1)[2], bool live) {
    if constexpr (PAIR) {
        store_pair(dst, width, sample, tp, h, t0, t1, live);
    } else {
        store_tile(dst, width, sample, tp, h, t0, live);
        store_tile(dst, width, sample, tp + 1, h, t1, live);
    }
}

template <class RING>
__device__ __forceinline__ void ring_start(RING &ring) {
    rstatic_for<RING::kLeadChunks>([&](auto c) { ring.template issue_chunk<c.value>(); });
}

}  // namespace
