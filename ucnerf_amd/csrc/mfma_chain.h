// Register-chained fp32 MFMA engine shared by field_mlp.hip and sky.hip (gfx950, wave64).
//
// A dense layer is computed TRANSPOSED, OUT^T[neuron, sample] = W . IN^T, with
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; no xf32 on gfx950):
//   A operand = weights (32 output neurons per tile), pre-packed into MFMA-fragment order as ONE
//               linear stream of 1 KiB "groups" (64 lanes x float4 = the A operands of four
//               consecutive MFMAs) in exactly the order the kernel consumes them; DMA'd
//               global -> LDS with global_load_lds (16 B/lane, no staging registers) in 32 KiB
//               chunks, double buffered, shared by the workgroup's 4 waves, read with ds_read_b128;
//   B operand = activations; a wave owns 32 samples (columns) for the whole network.
// The C/D layout of the 32x32 MFMA (lane = column + 32*half; register r = row
// (r&3) + 8(r>>2) + 4*half) is exactly a B operand for the k-pair {row(r,0), row(r,1)}, so
// accumulator register r of one layer feeds the next layer's MFMA directly: activations never
// touch LDS or HBM and there is no barrier on their account.
#pragma once
#include "ucn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunkGroups = 64;   // 64 KiB per LDS buffer, two buffers (128 of the CU's 160 KiB)

// neuron held by accumulator register r of tile `tile` in wave-half h
__host__ __device__ __forceinline__ uint32_t acc_row(uint32_t tile, uint32_t r, uint32_t h) {
    return 32u * tile + (r & 3u) + 8u * (r >> 2) + 4u * h;
}

// ---------------------------------------------------------------- packing kernels
// dst[((ot*n_in + it)*4 + r4)*64 + lane][e] =
//     W[32(row_tile0+ot) + (lane&31)][col0 + 32it + 8r4 + 4(lane>>5) + e]
static __global__ __launch_bounds__(256) void k_pack_chain(const float *__restrict__ W, uint32_t ld, uint32_t col0,
                                                           uint32_t row_tile0, uint32_t nt_out, uint32_t nt_in,
                                                           float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = nt_out * nt_in * 1024u;
    if (i >= total) return;
    const uint32_t e = i & 3u, lane = (i >> 2) & 63u, g = i >> 8;
    const uint32_t r4 = g & 3u, it = (g >> 2) % nt_in, ot = (g >> 2) / nt_in;
    const uint32_t row = 32u * (row_tile0 + ot) + (lane & 31u);
    const uint32_t col = col0 + 32u * it + 8u * r4 + 4u * (lane >> 5) + e;
    dst[i] = W[(size_t)row * ld + col];
}
static __global__ __launch_bounds__(256) void k_fill_zero(float *__restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = 0.0f;
}
// VALU heads: dst[((it*16 + r)*2 + h)*nc_pad + c] = W[c][col0 + acc_row(it,r,h)]   (W is [nc, ld])
static __global__ __launch_bounds__(256) void k_pack_head(const float *__restrict__ W, uint32_t ld, uint32_t col0,
                                                          uint32_t K, uint32_t nc, uint32_t nc_pad,
                                                          float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t total = (K / 32u) * 32u * nc_pad;
    if (i >= total) return;
    const uint32_t c = i % nc_pad, slot = i / nc_pad;
    const uint32_t h = slot & 1u, r = (slot >> 1) & 15u, it = slot >> 5;
    dst[i] = c < nc ? W[(size_t)c * ld + col0 + acc_row(it, r, h)] : 0.0f;
}
// VALU input layers (K <= 3): dst[((t*16 + r)*2 + h)*4 + k] = k < K ? W[acc_row(t,r,h)][k] : bias[..]
// i.e. {w0, w1, w2, b} per accumulator slot.
static __global__ __launch_bounds__(256) void k_pack_in3(const float *__restrict__ W, uint32_t ld,
                                                         const float *__restrict__ bias, uint32_t n_out,
                                                         float *__restrict__ dst) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_out * 4u) return;
    const uint32_t k = i & 3u, slot = i >> 2;
    const uint32_t h = slot & 1u, r = (slot >> 1) & 15u, t = slot >> 5;
    const uint32_t n = acc_row(t, r, h);
    dst[i] = k < 3u ? W[(size_t)n * ld + k] : bias[n];
}

#ifdef __HIP_DEVICE_COMPILE__
#define UCN_DEVICE_ONLY 1
#endif

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Double-buffered global -> LDS weight stream shared by the workgroup's 4 waves.
// Chunk c lives in buffer c&1.  boundary(c) is called by every wave right before it consumes the
// first group of chunk c: the barrier both publishes chunk c (its DMA was issued a whole chunk
// of MFMA time ago; __syncthreads drains vmcnt) and retires every wave's reads of chunk c-1,
// whose buffer is then refilled with chunk c+1.  EVERY wave of the workgroup must run the whole
// program (barrier-uniform), including waves whose samples are past the end of the batch.
struct WeightStream {
    const float *src;          // packed stream in global memory
    float *lds;                // 2 * kChunkGroups * 256 floats
    int lane, wave;
    uint32_t n_chunks;

    // The DMA is written as inline asm on purpose: with __builtin_amdgcn_global_load_lds the compiler
    // cannot tell that the DMA's LDS writes (buffer c+1) never alias the ds_reads that follow (buffer c)
    // and puts s_waitcnt vmcnt(0) right behind the issue -- the whole L2 -> LDS latency (~1.3 us per
    // chunk, measured) then sits in front of the MFMAs instead of under them.  Here the DMA is invisible
    // to the waitcnt pass, and boundary() waits for it explicitly before the publishing barrier.
    // this wave's share of the stream: global and LDS byte positions of its group 0 (wave-uniform)
    const float *wsrc = src + wave * 256;
    uint32_t wlds = (uint32_t)(size_t)(__attribute__((address_space(3))) float *)lds + wave * 1024u;
    uint32_t voff = lane * 16u;

    // this wave's i-th piece (one 1 KiB group per wave-instruction, i < kChunkGroups/4) of chunk c;
    // the caller guarantees c < n_chunks.  With compile-time (c, i): 2 SALU + s_mov m0 + the DMA.
    __device__ __forceinline__ void piece_unchecked(uint32_t c, int i) {
        const float *g = wsrc + ((size_t)c * kChunkGroups + i * 4) * 256;
        const uint32_t l = wlds + ((c & 1u) * kChunkGroups + i * 4) * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :
                     : "s"(l), "v"(voff), "s"(g)
                     : "memory");           // m0 is a reserved register: the compiler never keeps a value in it
    }
    __device__ __forceinline__ void piece(uint32_t c, int i) {
        if (c < n_chunks) piece_unchecked(c, i);
    }
    __device__ __forceinline__ void issue(uint32_t c) {
#pragma unroll
        for (int i = 0; i < kChunkGroups / 4; i++) piece(c, i);
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void sync() {   // publish the chunk whose DMA is in flight, retire reads of the other
        drain();
        __syncthreads();
    }
    __device__ __forceinline__ void boundary(uint32_t c) {
        sync();
        issue(c + 1);
    }
    __device__ __forceinline__ const float *group_ptr(int g) const {   // g = index in the whole stream
        return lds + ((g / kChunkGroups) & 1) * kChunkGroups * 256 + (g % kChunkGroups) * 256;
    }
    __device__ __forceinline__ float4 group(int g) const {
        const float *b0 = lds + lane * 4, *b1 = lds + kChunkGroups * 256 + lane * 4;
        const float *l = ((g / kChunkGroups) & 1) ? b1 : b0;
        return *reinterpret_cast<const float4 *>(l + (g % kChunkGroups) * 256);
    }
};

// One output tile from NT_IN input tiles; groups [it][r4] start at stream position G0.  G0 and the
// loop indices are compile-time constants after inlining + full unrolling (every unrolled loop is
// kept at <= 32 iterations: hipcc only partially unrolls larger bodies with barriers inside, which
// turns accumulator indices into run-time values and sends the tiles to scratch).
template <int NT_IN>
__device__ __forceinline__ void chain_one(const int G0, f32x16 &acc, const f32x16 (&in)[NT_IN], WeightStream &ws) {
#pragma unroll
    for (int i = 0; i < NT_IN * 4; i++) {
        const int g = G0 + i;
        if (g % kChunkGroups == 0) ws.boundary(g / kChunkGroups);
        const float4 a = ws.group(g);
        const int it = i / 4, r4 = i % 4;
        acc = mfma32(a.x, in[it][4 * r4 + 0], acc);
        acc = mfma32(a.y, in[it][4 * r4 + 1], acc);
        acc = mfma32(a.z, in[it][4 * r4 + 2], acc);
        acc = mfma32(a.w, in[it][4 * r4 + 3], acc);
    }
}
// NT_OUT output tiles from ONE input tile (order [ot][r4])
template <int NT_OUT>
__device__ __forceinline__ void chain_from_one(const int G0, f32x16 (&acc)[NT_OUT], const f32x16 &in, WeightStream &ws) {
#pragma unroll
    for (int i = 0; i < NT_OUT * 4; i++) {
        const int g = G0 + i;
        if (g % kChunkGroups == 0) ws.boundary(g / kChunkGroups);
        const float4 a = ws.group(g);
        const int ot = i / 4, r4 = i % 4;
        acc[ot] = mfma32(a.x, in[4 * r4 + 0], acc[ot]);
        acc[ot] = mfma32(a.y, in[4 * r4 + 1], acc[ot]);
        acc[ot] = mfma32(a.z, in[4 * r4 + 2], acc[ot]);
        acc[ot] = mfma32(a.w, in[4 * r4 + 3], acc[ot]);
    }
}
// acc[ot] += W . in for all NT_OUT x NT_IN tiles, order [ot][it][r4] (recursion over ot)
template <int OT, int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain_rec(const int G0, f32x16 (&acc)[NT_OUT], const f32x16 (&in)[NT_IN], WeightStream &ws) {
    chain_one<NT_IN>(G0 + OT * NT_IN * 4, acc[OT], in, ws);
    if constexpr (OT + 1 < NT_OUT) chain_rec<OT + 1, NT_OUT, NT_IN>(G0, acc, in, ws);
}
template <int NT_OUT, int NT_IN>
__device__ __forceinline__ void chain(const int G0, f32x16 (&acc)[NT_OUT], const f32x16 (&in)[NT_IN], WeightStream &ws) {
    chain_rec<0, NT_OUT, NT_IN>(G0, acc, in, ws);
}

// acc[r] = bias[acc_row(tile, r, h)] (+ extra[...]) : float4 loads, 16-byte aligned
__device__ __forceinline__ void init_tile(f32x16 &acc, int tile, const float *__restrict__ bias,
                                          const float *__restrict__ extra, int h) {
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
        const int n = 32 * tile + 8 * r4 + 4 * h;
        float4 v = *reinterpret_cast<const float4 *>(bias + n);
        if (extra) {
            const float4 x = *reinterpret_cast<const float4 *>(extra + n);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        acc[4 * r4 + 0] = v.x; acc[4 * r4 + 1] = v.y; acc[4 * r4 + 2] = v.z; acc[4 * r4 + 3] = v.w;
    }
}
template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NT], const float *__restrict__ bias,
                                          const float *__restrict__ extra, int h) {
#pragma unroll
    for (int ot = 0; ot < NT; ot++) init_tile(acc[ot], ot, bias, extra, h);
}
__device__ __forceinline__ void relu_tile(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; r++) a[r] = fmaxf(a[r], 0.0f);
}
template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++) relu_tile(a[t]);
}
