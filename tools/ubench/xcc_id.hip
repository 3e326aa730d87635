// Which XCD does workgroup i of a grid of one workgroup per CU run on?  Prints the hardware's XCC_ID per workgroup
// next to blockIdx.x % 8.   hipcc --offload-arch=gfx950 -O2 xcc_id.hip -o xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) probe(unsigned* out) {
    __shared__ char big[150 * 1024];  // one workgroup per CU, as the pooled kernel
    big[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    unsigned id = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID, bits 3:0
    if (threadIdx.x == 0) out[blockIdx.x] = id + (big[5] == 99 ? 1000u : 0u);
}
int main() {
    unsigned* d; unsigned h[512];
    hipMalloc(&d, sizeof h);
    for (int grid : {256, 300}) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 0, 0, d);
        hipMemcpy(h, d, sizeof(unsigned) * grid, hipMemcpyDeviceToHost);
        int bad = 0, per[16] = {0};
        for (int i = 0; i < grid; ++i) { bad += (h[i] != (unsigned)(i % 8)); per[h[i] & 15]++; }
        printf("grid %d: workgroups whose XCC_ID != blockIdx %% 8: %d; per XCC:", grid, bad);
        for (int k = 0; k < 8; ++k) printf(" %d", per[k]);
        printf("\nfirst 24:");
        for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
        printf("\n");
    }
    return 0;
}
