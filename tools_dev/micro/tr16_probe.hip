// dev probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value = its own element index; every lane of a 16-lane group supplies
// the address of "its" 8-byte chunk of a 4 x 16 row-major block (chunk i = row i/4, columns 4(i%4)..+3), row stride RS elements.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4_t __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int RS) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    const int el = g * 1024 + (i >> 2) * RS + (i & 3) * 4;        // element index of this lane's chunk
    short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(lds + el));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
    for (int RS : {16, 64, 128}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, RS);
        unsigned short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements\n", RS);
        for (int l = 0; l < 64; ++l) { if (l % 16 == 0 || l % 16 == 1 || l % 16 == 5 || l % 16 == 15) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    }
    return 0;
}
