// Developer probe: semantics of ds_read_b64_tr_b16 on gfx950 (hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const short* addr = lds + (g * 4 + q) * 64 + 4 * p;      // lane 4q+p of a group: row q of the block, columns 4p..4p+3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 512); short h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l < 20 || l % 16 == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
