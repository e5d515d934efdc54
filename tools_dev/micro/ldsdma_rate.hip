// dev micro-benchmark: what does the L2 -> LDS path (buffer_load ... lds, 1 KB per wave instruction) sustain per CU when nothing else runs?
// NW waves per workgroup, one workgroup per CU (LDS-limited), each wave streams 8-KB tiles of an L2-resident plane set through a
// private 2-slot ring with a counted vmcnt, like attn_strip_kernel's key stream.   hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) unsigned char* lds_bptr;
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const unsigned short* src, int bytes_per_wg_plane, int iters, int planes, float* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned short ring[16 * 4096];   // 128 KB whatever NW (NW * DEPTH <= 16): one workgroup per CU
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int img = blockIdx.x % planes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long long)img * (bytes_per_wg_plane / 2)), 0, bytes_per_wg_plane, 0x00020000);
    const unsigned base = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)ring + wave * DEPTH * 8192;
    const int voff = lane * 16;
    const int ntile = bytes_per_wg_plane / 8192;
    int t = wave % ntile;
    for (int it = 0; it < iters; ++it) {
        unsigned dst = base + (it % DEPTH) * 8192;
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_bptr)(unsigned long long)(dst + i * 1024), 16, voff, t * 8192 + i * 1024, 0, 0);
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t += NW; if (t >= ntile) t -= ntile;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && ring[lane] == 0x1234) sink[0] = 1.f;
}
template <int NW, int DEPTH> void run(const unsigned short* d, int plane_bytes, int planes, int wgs, float* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<NW, DEPTH>), dim3(wgs), dim3(NW * 64), 0, 0, d, plane_bytes, iters, planes, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * NW * iters * 8192.0;
    printf("waves %d depth %d wgs %d: %.3f ms  %.2f TB/s  %.1f GB/s per CU  (%.1f B/clk/CU at 2.1 GHz)\n", NW, DEPTH, wgs, ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
}
int main() {
    const int plane_bytes = 200 * 1024, planes = 32;
    unsigned short* d; float* sink;
    hipMalloc(&d, (size_t)plane_bytes * planes); hipMemset(d, 0, (size_t)plane_bytes * planes); hipMalloc(&sink, 4);
    run<8, 2>(d, plane_bytes, planes, 256, sink);
    run<7, 2>(d, plane_bytes, planes, 256, sink);
    run<4, 2>(d, plane_bytes, planes, 256, sink);
    run<2, 2>(d, plane_bytes, planes, 256, sink);
    run<1, 2>(d, plane_bytes, planes, 256, sink);
    run<8, 1>(d, plane_bytes, planes, 256, sink);
    run<5, 3>(d, plane_bytes, planes, 256, sink);
    run<8, 2>(d, plane_bytes, 1, 256, sink);
    run<8, 2>(d, plane_bytes, planes, 128, sink);
    return 0;
}
