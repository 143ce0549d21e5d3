// CU hog for partition experiments: k workgroups that each reserve (almost) a whole CU's LDS and sleep-spin for a fixed wall time, so
// that kernels launched meanwhile on another stream can only use the remaining CUs (a workgroup that needs LDS does not fit beside a
// hog).  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hog.hip -o tools/_exp/libhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void k_hog(uint64_t ticks, uint32_t *where) {
    extern __shared__ uint32_t s[];
    s[threadIdx.x] = 0;
    const uint64_t t0 = wall_clock64();                       // 100 MHz
    if (threadIdx.x == 0 && where) where[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3u << 11)) & 7u;    // XCC id
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int hog_launch(uint32_t k, uint32_t lds_bytes, double ms, uint32_t *where, void *stream) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_hog, dim3(k), dim3(64), lds_bytes, (hipStream_t)stream, (uint64_t)(ms * 1e5), where);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
