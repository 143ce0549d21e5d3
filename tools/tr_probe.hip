// What does ds_read_b64_tr_b16 hand each lane?  LDS holds u16 element i at byte 2 i; lane l supplies byte address
// `addr[l]` (8-byte aligned); the four 16-bit elements it receives are printed.  hipcc --offload-arch=gfx950 tools/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *addr, uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t *)lds + addr[threadIdx.x];
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = (uint16_t)(v >> (16 * e));
}
int main() {
    uint32_t h_addr[64], *d_addr; uint16_t h_out[256], *d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; pat++) {
        // pattern 0: lane l -> byte 8 l (the wave's chunks contiguous).  pattern 1: a row-major [rows][64 elements] image
        // (row stride 128 B): lane l -> row (l & 15) / 4 + 4 (l >> 5)... chosen as the candidate MFMA-operand addressing
        for (int l = 0; l < 64; l++)
            h_addr[l] = pat == 0 ? 8u * l : (uint32_t)((8 * (l >> 5) + ((l & 15) >> 2)) * 128 + (16 * ((l >> 4) & 1) + 4 * (l & 3)) * 2);
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; l++) {
            printf("lane %2d addr %4u:", l, h_addr[l]);
            for (int e = 0; e < 4; e++) {
                if (pat == 0) printf(" %4u", h_out[l * 4 + e]);
                else printf(" (r%u,c%u)", h_out[l * 4 + e] / 64, h_out[l * 4 + e] % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
