#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out) {
    __shared__ short lds[64 * 4 * 2];
    const int l = threadIdx.x;
    for (int i = 0; i < 8; ++i) lds[l * 8 + i] = (short)(l * 8 + i);   // element id = its index
    __syncthreads();
    // each lane reads the 8-byte word at index `l` (elements 4l .. 4l+3)
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + l * 4));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    probe<<<1, 64>>>(d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
