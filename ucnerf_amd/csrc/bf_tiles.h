// bf16 MFMA tile helpers shared by the training kernels (field_train.hip) and the mixed-precision sky layer (sky.hip):
// a wave owns 32 samples (the N of v_mfma_f32_32x32x16_bf16); activations live in registers as B operands (two k-steps of
// 8 bf16 per 32-wide tile), weights arrive as 1 KiB A-fragments through the LDS-DMA ring of mlp_ring.h.
#pragma once
#include <utility>

#include "mlp_ring.h"

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
template <class RING>
__device__ __forceinline__ void ring_start(RING &ring) {
    rstatic_for<RING::kLeadChunks>([&](auto c) { ring.template issue_chunk<c.value>(); });
}

}  // namespace
