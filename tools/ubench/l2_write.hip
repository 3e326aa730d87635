// Do 16-byte stores that are overwritten again and again within an L2-sized region reach the memory side?
// Each workgroup owns a private region of `slots` 64-byte records and rewrites 16 or 48 bytes of random
// records `reps` times (what the pooled kernel's cold slot records see).  Run under
//   rocprofv3 --pmc WRITE_SIZE -- ./l2_write <KB per workgroup> <bytes per record write: 16|48|64> [nt]
// and compare WRITE_SIZE with the bytes stored (printed) and with the region's size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) rewrite(float4* base, int slots, int reps, int parts, int nt) {
    float4* mine = base + (size_t)blockIdx.x * slots * 4;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int r = 0; r < reps; ++r) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        float4* rec = mine + (size_t)(s % (unsigned)slots) * 4;
        const float4 v = make_float4((float)r, 1.0f, 2.0f, 3.0f);
        for (int p = 0; p < parts; ++p) {
            if (nt) __builtin_nontemporal_store(v.x, &rec[p].x), __builtin_nontemporal_store(v.y, &rec[p].y), __builtin_nontemporal_store(v.z, &rec[p].z), __builtin_nontemporal_store(v.w, &rec[p].w);
            else rec[p] = v;
        }
    }
}
int main(int argc, char** argv) {
    const int kb = argc > 1 ? atoi(argv[1]) : 88, bytes = argc > 2 ? atoi(argv[2]) : 16, nt = argc > 3;
    const int slots = kb * 1024 / 64, reps = 2000, grid = 256, parts = bytes / 16;
    float4* buf;
    hipMalloc((void**)&buf, (size_t)grid * slots * 64);
    hipMemset(buf, 0, (size_t)grid * slots * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int k = 0; k < 3; ++k) {
        hipEventRecord(a);
        hipLaunchKernelGGL(rewrite, dim3(grid), dim3(1024), 0, 0, buf, slots, reps, parts, nt);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%d KB per workgroup (%.1f MB in all, %.2f MB per XCD), %d B per write%s: %.3f GB stored per launch, %.3f ms\n", kb, grid * kb / 1024.0, grid * kb / 1024.0 / 8,
           bytes, nt ? " nontemporal" : "", (double)grid * 1024 * reps * bytes / 1e9, ms);
    return 0;
}
