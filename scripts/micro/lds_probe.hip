// A trivial kernel with a chosen dynamic-LDS size and block count, for scripts/contention_probe.py: does a small kernel's LDS request
// decide whether it can run beside the side stream's weight-gradient blocks?
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/lds_probe.hip -o scripts/micro/liblds_probe.so
#include <hip/hip_runtime.h>
template <int PRIO>
__global__ __launch_bounds__(256) void lds_probe_kernel(float* out, int n) {
    extern __shared__ float sm[];
    if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float v = sm[(threadIdx.x + 1) & 255];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;      // ~n x 4 cycles
    if (v == 12345.f) out[0] = v;
}
extern "C" int lds_probe_launch(float* out, int lds_bytes, int blocks, int n, void* stream, int prio) {
    static int set = 0;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_probe_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_probe_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        set = 1;
    }
    if (prio) hipLaunchKernelGGL(lds_probe_kernel<3>, dim3(blocks), dim3(256), lds_bytes < 1024 ? 1024 : lds_bytes, (hipStream_t)stream, out, n);
    else hipLaunchKernelGGL(lds_probe_kernel<0>, dim3(blocks), dim3(256), lds_bytes < 1024 ? 1024 : lds_bytes, (hipStream_t)stream, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
