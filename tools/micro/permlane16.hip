// micro test: semantics of v_permlane16_swap_b32 on gfx950 (used by the layer kernel's epilogue)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x * 10 + 0, y = threadIdx.x * 10 + 1;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 16, 17, 32, 48, 63}) printf("lane %2d: vdst %4u src %4u\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
