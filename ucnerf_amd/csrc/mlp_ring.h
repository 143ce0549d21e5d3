// Weight ring + operand pipe of the split-f16 MLP kernels (field_mlp_h.hip), gfx950 / wave64.
//
// The network's weights are one linear stream of 1 KiB "groups" (64 lanes x 8 halfs: one A operand of
// v_mfma_f32_32x32x16_f16), in exactly the order the kernel consumes them.  A workgroup's four waves (one per
// SIMD) run the same program on 32 samples each and share the stream through LDS:
//
//   L2 --global_load_lds (16 B / lane, no staging registers)--> LDS ring of kRingSlots x kRingChunk KiB
//      --ds_read_b128, kPipeDepth operand pairs ahead--> A operands in VGPRs --> MFMA
//
// Round 1's engine (mfma_chain_h.h: two 64 KiB buffers, the next buffer's DMA spread over the current one) waited
// at every buffer boundary for DMA pieces issued a few hundred cycles earlier: the whole L2 -> LDS latency
// (~1.3 us) stood in front of the MFMAs eight times per tile.  Here the DMA runs kLead whole chunks ahead of the reads:
//   * chunk c lives in slot c % kRingSlots; while chunk c is being read, the pieces of chunk c + kLead are issued, one
//     per four groups consumed (a burst of DMA instructions costs the issuing wave 100-185 cycles each);
//   * at the first read of chunk c a wave waits until at most (kLead - 1) chunks' worth of its DMA instructions are
//     outstanding (vmcnt counts in order: those are the later chunks'), then the workgroup barrier publishes chunk c;
//   * the slot being refilled (chunk c + kLead -> slot of chunk c - 2) was last read a whole chunk ago: no read of it
//     can still be in flight, which the two-buffer engine could not guarantee.
// A small side table (bias tiles, VALU head weights, scale factors) is loaded once, in front of the ring.
#pragma once
#include <utility>

#include "mfma_chain.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kRingChunk = 16;                       // groups (KiB) per chunk of the default ring
constexpr int kRingChunkSmall = 8;                   // ... of the 64 KiB ring used beside a co-resident featurisation workgroup
constexpr int kRingPad = 32;                         // the packed stream is padded to a multiple of this many groups
constexpr int kRingSlots = 8;
constexpr int kLead = 6;                             // the DMA runs this many chunks ahead of the reads: while chunk c is read
//                                                      the pieces of chunk c + 6 are issued, so the LAST piece of a chunk has
//                                                      5 chunks = 20 double steps (~1.6 us) to land; measured L2 -> LDS latency
//                                                      ~1.3 us.  (kLead 2 of 4 x 32 KiB left the last piece one chunk = 0.64 us:
//                                                      the boundaries of colour layer 1 ran at 420 cycles per double step)
constexpr int kSideGroups = 8;                       // side table in front of the ring (8 KiB)
constexpr int ring_lds_bytes(int chunk) { return (kSideGroups + kRingSlots * chunk) * 1024; }
#ifdef UCN_EXP_PIPEDEPTH
constexpr int kPipeDepth = UCN_EXP_PIPEDEPTH;
#else
constexpr int kPipeDepth = 4;                        // A-operand pairs in flight per wave (8 VGPRs each): the next double
//                                                      step's operands are requested one whole step ahead
#endif

template <int... Is, class F>
__device__ __forceinline__ void rstatic_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void rstatic_for(F &&f) {
    rstatic_for_impl(std::make_integer_sequence<int, N>{}, f);
}

struct HPair {   // one 32-neuron activation tile as B operands: [k-step s]
    h8 hi[2], lo[2];
};
struct OpPipe {  // ring of A operands, slot = (pair index) % kPipeDepth
    h8 hi[kPipeDepth], lo[kPipeDepth];
};

__device__ __forceinline__ f32x16 mfma_h(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi = f16(v), lo = f16(v - hi): |v - hi - lo| <= 2^-22 |v| while lo is a normal f16 (|v| >= 2^-3); the packers'
// power-of-two scaling (field_mlp_h.hip) keeps the operands in that window and below the f16 maximum.
#ifndef UCN_SPLIT_ASM
// r06: the operand split with COMPILER-VISIBLE instructions (gemm_h3.hip's form: -1.0f made opaque so that the fma is not folded into a
// subtraction and hipcc itself selects v_fma_mix{lo,hi}_f16): 5 instructions per pair instead of 3 (measured on the NeRF-level MLP alone:
// 4.899 against 4.878 ms per 8.4 M samples, profiles/r06/mlp_waves_ab.txt), but the hazard recogniser sees them.
// The inline-asm form below (-DUCN_SPLIT_ASM: the r02-r05 form, kept for the A/B) can be handed, as its output, the register an MFMA issued just before still reads as its A operand -- for an
// asm statement no wait states are inserted (tools/isa_asm_hazard.py lists the places; r06 found it the hard way in k_gemm_h3<4, 16>).
__device__ __forceinline__ float rsplit_neg1() {
    float v = -1.0f;
    asm("" : "+s"(v));
    return v;
}
__device__ __forceinline__ void rsplit8(const float (&v)[8], h8 &hi, h8 &lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float neg1 = rsplit_neg1();
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        h2 hp, lp;
        hp[0] = (_Float16)v[2 * p];
        hp[1] = (_Float16)v[2 * p + 1];
        lp[0] = (_Float16)__builtin_fmaf((float)hp[0], neg1, v[2 * p]);
        lp[1] = (_Float16)__builtin_fmaf((float)hp[1], neg1, v[2 * p + 1]);
        hw[p] = __builtin_bit_cast(uint32_t, hp);
        lw[p] = __builtin_bit_cast(uint32_t, lp);
    }
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}
#else
__device__ __forceinline__ void rsplit8(const float (&v)[8], h8 &hi, h8 &lo) {
    // three VALU instructions per PAIR of values instead of six: hi pair = v_cvt_pk_f16_f32; each lo = f16(v - hi) is
    // ONE v_fma_mix{lo,hi}_f16 (f16 source hi, f32 constant -1, f32 source v: the exact difference rounded once, into its
    // half of the pair) -- no conversion of hi back to f32, no separate subtraction, no second pack
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        h2 hp;
        hp[0] = (_Float16)v[2 * p];
        hp[1] = (_Float16)v[2 * p + 1];
        hw[p] = __builtin_bit_cast(uint32_t, hp);
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lw[p]) : "v"(hw[p]), "v"(v[2 * p]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw[p]) : "v"(hw[p]), "v"(v[2 * p + 1]));
    }
    hi = __builtin_bit_cast(h8, hw);
    lo = __builtin_bit_cast(h8, lw);
}
#endif
// ReLU as ONE v_max_i32 on the bit pattern: negative floats (and -0) are negative integers.  fmaxf costs two VALU
// instructions here (IEEE mode canonicalises the MFMA result first).  A negative NaN becomes 0, a positive one stays.
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// (ReLU +) split of the 8 accumulator registers that form k-step s of a tile
template <bool RELU>
__device__ __forceinline__ void split_half(const f32x16 &a, const int s, HPair &t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = RELU ? relu_bits(a[8 * s + e]) : a[8 * s + e];
    rsplit8(v, t.hi[s], t.lo[s]);
}

template <int N>
__device__ __forceinline__ void ring_wait_lds() {     // lgkmcnt only
#ifdef UCN_EXP_NOHINT
    return;
#endif
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
}
template <int N>
__device__ __forceinline__ void ring_wait_vm() {      // vmcnt only (6 bits: [3:0] and [15:14])
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NGROUPS = length of the stream in groups (a multiple of 4; the packed stream is padded to whole chunks)
// STAGE = 0: the stream reaches LDS by LDS-DMA (global_load_lds).  STAGE = n > 0: by plain 16-byte global loads into n
// staging registers per lane and ds_write_b128 n pieces later.  Measured on this kernel (profiles/r02*/mlp_timeline.txt):
// the DMA path lands ~10 B per clock and CU and stretches the latency of every ds_read issued meanwhile; with one
// wave per SIMD the registers for the classic path are there.
template <int NGROUPS, int CHUNK, int NWAVES = 4, int SLOTS = kRingSlots, int LEAD = kLead, int STAGE = 0>
struct Ring {
    static constexpr int kStage = STAGE;
    static constexpr int kExtraLds = STAGE > 0 ? 1 : 0;      // LDS operations per double step besides the operand reads
    f4v stage[STAGE > 0 ? STAGE : 1];
    float *ldsw[3];         // staged mode: this lane's WRITE pointers (ring base + wave KiB + lane * 16 + 0 / 60 / 120 KiB)
    static constexpr int kChunk = CHUNK;
    static constexpr int kSlots = SLOTS, kLeadChunks = LEAD;
    static constexpr int kWaves = NWAVES;                    // waves of the workgroup that share the stream
    static constexpr int kPiecesPerChunk = CHUNK / NWAVES;  // DMA instructions per wave and chunk
    static constexpr int kChunks = (NGROUPS + CHUNK - 1) / CHUNK;
    const float *wsrc;      // this wave's share of the stream in global memory (group 0 + wave)
    const float *ldsb[3];   // this lane's read pointers at ring base + 0 / 64 / 128 KiB: ds_read takes a 16-bit immediate
    //                         offset, and WITHOUT explicit bases the compiler materialises (and parks in AGPRs) one
    //                         address register per 64 KiB-crossing constant: 67 extra registers in the first build
    uint32_t wlds;          // LDS byte address of ring base + wave * 1 KiB
    uint32_t voff;          // lane * 16
    int lane;

    __device__ __forceinline__ Ring(const float *stream, float *lds_ring, int lane_, int wave)
        : wsrc(stream + wave * 256),
          wlds((uint32_t)(size_t)(__attribute__((address_space(3))) float *)lds_ring + (uint32_t)wave * 1024u),
          voff((uint32_t)lane_ * 16u), lane(lane_) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t a = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)lds_ring + (uint32_t)lane_ * 16u + (uint32_t)k * 61440u;
            asm volatile("" : "+v"(a));                    // opaque: keeps the three bases apart
            ldsb[k] = (const float *)(__attribute__((address_space(3))) const float *)(size_t)a;
            uint32_t w = a + (uint32_t)wave * 1024u;
            asm volatile("" : "+v"(w));
            ldsw[k] = (float *)(__attribute__((address_space(3))) float *)(size_t)w;
        }
    }

    // this wave's I-th DMA instruction of chunk C: group C * kRingChunk + 4 I + wave -> slot C % kRingSlots.
    // Inline asm on purpose: the waitcnt pass must not see the DMA (it would put vmcnt(0) behind every issue);
    // boundary() waits for it explicitly.
    template <int C, int I>
    __device__ __forceinline__ void piece() {
#ifdef UCN_EXP_NODMA          // experiment builds (tools/build_variant.sh): timing only, results are garbage
        if constexpr (C >= LEAD) return;
#endif
        if constexpr (STAGE > 0) {
            constexpr int q = C * kPiecesPerChunk + I;               // this wave's piece counter
            if constexpr (q >= STAGE) {                              // piece q - STAGE has had STAGE double steps to arrive
                constexpr int qw = q - STAGE, Cw = qw / kPiecesPerChunk, Iw = qw % kPiecesPerChunk;
                if constexpr (Cw < kChunks) {
                    constexpr int off = ((Cw % SLOTS) * CHUNK + Iw * NWAVES) * 1024;
                    constexpr int k = off / 61440, rem = off % 61440;
                    *reinterpret_cast<f4v *>(ldsw[k] + rem / 4) = stage[qw % STAGE];
                }
            }
            if constexpr (C < kChunks)
                stage[q % STAGE] = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(wsrc + (size_t)(C * CHUNK + I * NWAVES) * 256) + lane);
            return;
        }
        if constexpr (C < kChunks) {
            const float *g = wsrc + (size_t)(C * CHUNK + I * NWAVES) * 256;
            const uint32_t l = wlds + (uint32_t)(((C % SLOTS) * CHUNK + I * NWAVES) * 1024);
#ifdef UCN_EXP_DMA_DWORD      // experiment: the same number of DMA instructions moving a quarter of the bytes (results garbage)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" : : "s"(l), "v"(voff), "s"(g) : "memory");
#else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(l), "v"(voff), "s"(g) : "memory");
#endif
        }
    }
    template <int C>
    __device__ __forceinline__ void issue_chunk() {
        rstatic_for<kPiecesPerChunk>([&](auto i) { piece<C, i.value>(); });
    }
    // first read of chunk C: everything up to and including chunk C has landed for every wave.
    // (r05 tried counting the wave's activation STORES that are younger than chunk C into the allowed number -- vmcnt counts stores too --
    // so that the wait would not drain them.  Once the fragment pipe was in (bf_tiles.h tile_pair_pf) it bought nothing (sky forward 1.31
    // against 1.29 ms without), and it is NOT safe: it needs stores to retire in issue order with the LDS-DMA loads, and the two-tile sky
    // kernel -- 8 stores per pair, the last chunks awaited with no later pieces as slack -- read fragments that had not landed
    // (non-reproducible outputs, profiles/r05/sky_train_variants.txt).  The wait below counts this ring's DMA instructions only.)
    template <int C>
    __device__ __forceinline__ void boundary() {
        // DMA instructions of this wave that may still be in flight: those of the chunks behind C that have been
        // issued so far, i.e. chunks C+1 .. C+kLead-1 (chunk C+kLead is issued while C is read)
        constexpr int later = (C + LEAD - 1 < kChunks ? LEAD - 1 : (kChunks - 1 - C > 0 ? kChunks - 1 - C : 0));
#ifdef UCN_EXP_NOBAR
        return;
#endif
        if constexpr (STAGE > 0) {
            // this wave's ds_writes of chunk C were issued >= (LEAD - 1) * kPiecesPerChunk - STAGE double steps ago and
            // every double step waits its LDS queue down to a handful of operations: they are done.  (Prologue: drain.)
            static_assert(STAGE <= (LEAD - 1) * kPiecesPerChunk, "staged pieces would be written after they are needed");
            if constexpr (C == 0) ring_wait_lds<0>();
        } else {
            ring_wait_vm<later * kPiecesPerChunk>();
        }
        // bare barrier: __syncthreads() adds a fence whose lgkmcnt(0) would drain the operand pipe.  LDS is coherent
        // within the CU and every wave has waited for its own DMA; the slot being refilled was last read a chunk ago.
        asm volatile("s_barrier" ::: "memory");
    }
    template <int G>
    __device__ __forceinline__ h8 group() const {
        constexpr int off = (((G / CHUNK) % SLOTS) * CHUNK + G % CHUNK) * 1024;   // bytes from the ring base
        constexpr int k = off / 61440, rem = off % 61440;                                             // rem + 15 < 65536
        return __builtin_bit_cast(h8, *reinterpret_cast<const float4 *>(ldsb[k] + rem / 4));
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// Request pair G (even group index) into its pipe slot; the ring's housekeeping rides on the requests.
template <int G, class RING>
__device__ __forceinline__ void pipe_fetch(OpPipe &p, RING &ring) {
    constexpr int CH = RING::kChunk;
    if constexpr (G % CH == 0 && G / CH >= 1) ring.template boundary<G / CH>();   // chunk 0: prologue
    if constexpr (G % RING::kWaves == 0) ring.template piece<G / CH + RING::kLeadChunks, (G % CH) / RING::kWaves>();
#ifdef UCN_EXP_NOLDS
    if constexpr (G >= 2 * kPipeDepth) return;
#endif
    p.hi[(G / 2) % kPipeDepth] = ring.template group<G>();
#ifdef UCN_EXP_HALFLDS
    p.lo[(G / 2) % kPipeDepth] = p.hi[(G / 2) % kPipeDepth];
#else
    p.lo[(G / 2) % kPipeDepth] = ring.template group<G + 1>();
#endif
}
constexpr int rmin(int a, int b) { return a < b ? a : b; }
// reads still in flight that are YOUNGER than the four operands of the double step at G, once everything below
// `fetched` (group index, exclusive) has been requested
template <int G, int FETCHED>
constexpr int ring_younger() { return FETCHED - (G + 4) > 0 ? FETCHED - (G + 4) : 0; }

template <int NG, class RING>
__device__ __forceinline__ void pipe_prime(OpPipe &p, RING &ring) {
    rstatic_for<kPipeDepth>([&](auto d) {
        if constexpr (2 * d.value < NG) pipe_fetch<2 * d.value>(p, ring);
    });
    ring_wait_lds<ring_younger<0, rmin(NG, 2 * kPipeDepth)>()>();
}

// One double step: two OUTPUT tiles (A pairs G and G+2) against the same 16 k's of the input,
//   acc0 += A(G) . B,  acc1 += A(G+2) . B,
// six MFMAs alternating between the two accumulators (an MFMA that accumulates into the block the previous one is
// still writing does not issue back to back), then the two pairs kPipeDepth ahead are requested into the slots just
// consumed, then ONE wait for the next double step's four operands.  `shadow` is independent VALU / LDS work that
// rides under the six MFMAs (sched_group_barriers: 1 MFMA, then up to VALU_PER_MFMA VALU instructions, six times);
// EXTRA_LDS = LDS reads it issues (they share the counter of the operand reads).
#ifndef UCN_EXP_REFILL
#define UCN_EXP_REFILL 0
#endif
template <int G, int NG, int VALU_PER_MFMA, int EXTRA_LDS, class RING, class F>
__device__ __forceinline__ void dstep(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, OpPipe &p, RING &ring,
                                      F &&shadow) {
    constexpr int s0 = (G / 2) % kPipeDepth, s1 = (G / 2 + 1) % kPipeDepth;
    acc0 = mfma_h(p.hi[s0], bhi, acc0);
    acc1 = mfma_h(p.hi[s1], bhi, acc1);
    acc0 = mfma_h(p.hi[s0], blo, acc0);
    acc1 = mfma_h(p.hi[s1], blo, acc1);
    acc0 = mfma_h(p.lo[s0], bhi, acc0);
    acc1 = mfma_h(p.lo[s1], bhi, acc1);
#if UCN_EXP_REFILL == 2
    // refill the slots the PREVIOUS double step consumed (its MFMAs have long read them), two steps ahead of their use
    constexpr int GF = G - 4 + 2 * kPipeDepth;
    if constexpr (G >= 4 && GF < NG) pipe_fetch<GF>(p, ring);
    if constexpr (G >= 4 && GF + 2 < NG) pipe_fetch<GF + 2>(p, ring);
    if constexpr (VALU_PER_MFMA > 0) shadow();
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    if constexpr (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, 2 * VALU_PER_MFMA, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
    // keep the operands this step consumed allocated until here: the register allocator otherwise hands their
    // registers to the reads issued above, and a ds_read cannot issue while an MFMA in flight still reads its target
    asm volatile("" ::"v"(p.hi[s0]), "v"(p.hi[s1]), "v"(p.lo[s0]), "v"(p.lo[s1]));
    if constexpr (G + 4 < NG)
        ring_wait_lds<rmin(15, ring_younger<G + 4, rmin(NG, G >= 4 ? GF + 4 : 2 * kPipeDepth)>() + EXTRA_LDS)>();
#else
#ifdef UCN_EXP_NOSHADOW
    if constexpr (false) {
#else
    if constexpr (VALU_PER_MFMA > 0) {
#endif
        shadow();
#pragma unroll
        for (int i = 0; i < 6; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);   // VALU
        }
    }
    if constexpr (G + 2 * kPipeDepth < NG) pipe_fetch<G + 2 * kPipeDepth>(p, ring);
    if constexpr (G + 2 * kPipeDepth + 2 < NG) pipe_fetch<G + 2 * kPipeDepth + 2>(p, ring);
#if UCN_EXP_REFILL == 1
    if constexpr (VALU_PER_MFMA == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
#endif
    if constexpr (G + 4 < NG)
        ring_wait_lds<rmin(15, ring_younger<G + 4, rmin(NG, G + 2 * kPipeDepth + 4)>() + EXTRA_LDS + RING::kExtraLds)>();
#endif
#ifndef UCN_EXP_NOSCHEDBAR
    __builtin_amdgcn_sched_barrier(0);   // keep each step's MFMAs and its requests together, in program order
#endif
}
template <int G, int NG, class RING>
__device__ __forceinline__ void dstep(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, OpPipe &p, RING &ring) {
    dstep<G, NG, 0, 0>(acc0, acc1, bhi, blo, p, ring, [] {});
}

// ---- plain double step with a DEPTH-pair operand pipe (the 8-wave kernel: two waves per SIMD hide each other's
//      LDS latency and VALU phases, so the pipe can be short and nothing rides in MFMA shadows)
template <int DEPTH>
struct OpPipeD {
    h8 hi[DEPTH], lo[DEPTH];
};
template <int G, int DEPTH, class RING>
__device__ __forceinline__ void pipe_fetch_d(OpPipeD<DEPTH> &p, RING &ring) {
    constexpr int CH = RING::kChunk;
    if constexpr (G % CH == 0 && G / CH >= 1) ring.template boundary<G / CH>();
    if constexpr (G % RING::kWaves == 0) ring.template piece<G / CH + RING::kLeadChunks, (G % CH) / RING::kWaves>();
    p.hi[(G / 2) % DEPTH] = ring.template group<G>();
    p.lo[(G / 2) % DEPTH] = ring.template group<G + 1>();
}
template <int NG, int DEPTH, class RING>
__device__ __forceinline__ void pipe_prime_d(OpPipeD<DEPTH> &p, RING &ring) {
    rstatic_for<DEPTH>([&](auto d) {
        if constexpr (2 * d.value < NG) pipe_fetch_d<2 * d.value>(p, ring);
    });
    ring_wait_lds<ring_younger<0, rmin(NG, 2 * DEPTH)>()>();
}
template <int G, int NG, int DEPTH, class RING>
__device__ __forceinline__ void dstep_d(f32x16 &acc0, f32x16 &acc1, const h8 bhi, const h8 blo, OpPipeD<DEPTH> &p, RING &ring) {
    constexpr int s0 = (G / 2) % DEPTH, s1 = (G / 2 + 1) % DEPTH;
    acc0 = mfma_h(p.hi[s0], bhi, acc0);
    acc1 = mfma_h(p.hi[s1], bhi, acc1);
    acc0 = mfma_h(p.hi[s0], blo, acc0);
    acc1 = mfma_h(p.hi[s1], blo, acc1);
    acc0 = mfma_h(p.lo[s0], bhi, acc0);
    acc1 = mfma_h(p.lo[s1], bhi, acc1);
    if constexpr (G + 2 * DEPTH < NG) pipe_fetch_d<G + 2 * DEPTH>(p, ring);
    if constexpr (G + 2 * DEPTH + 2 < NG) pipe_fetch_d<G + 2 * DEPTH + 2>(p, ring);
    if constexpr (G + 4 < NG) ring_wait_lds<ring_younger<G + 4, rmin(NG, G + 2 * DEPTH + 4)>()>();
    __builtin_amdgcn_sched_barrier(0);
}
