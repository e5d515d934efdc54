// dev probe: v_permlane32_swap_b32 a, b  (gfx950): which halves are exchanged?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
    int a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
    auto r = __builtin_amdgcn_permlane32_swap(threadIdx.x, 100 + threadIdx.x, false, false);
    out[128 + threadIdx.x] = r[0]; out[192 + threadIdx.x] = r[1];
#else
    out[128 + threadIdx.x] = -1; out[192 + threadIdx.x] = -1;
#endif
}
int main() {
    int* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    int h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int k = 0; k < 4; ++k) printf("%d: lane0 %d lane1 %d lane31 %d lane32 %d lane33 %d lane63 %d\n", k, h[64*k+0], h[64*k+1], h[64*k+31], h[64*k+32], h[64*k+33], h[64*k+63]);
    return 0;
}
