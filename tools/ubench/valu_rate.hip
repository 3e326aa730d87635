// VALU issue-rate probe: independent v_fma_f32 / v_rcp_f32 / division streams at a chosen number of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = c / a0 + m; a1 = c / a1 + m; a2 = c / a2 + m; a3 = c / a3 + m; a4 = c / a4 + m; a5 = c / a5 + m; a6 = c / a6 + m; a7 = c / a7 + m;
            }
        } else if (MODE == 4) {  // min/max chain (the slab test)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = fminf(a0, a1); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4); a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0);
            }
        } else if (MODE == 5) {  // integer add / and / shift
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3), u4 = __float_as_uint(a4), u5 = __float_as_uint(a5), u6 = __float_as_uint(a6), u7 = __float_as_uint(a7);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u0 = u0 + u7; u1 = u1 ^ u0; u2 = (u2 << 1) | 1u; u3 = u3 + u2; u4 = u4 ^ u3; u5 = (u5 >> 1) ^ 5u; u6 = u6 + u5; u7 = u7 ^ u6;
            }
            a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3); a4 = __uint_as_float(u4); a5 = __uint_as_float(u5); a6 = __uint_as_float(u6); a7 = __uint_as_float(u7);
        } else if (MODE == 6) {  // compare + select
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = a0 < a1 ? a2 : a3; a1 = a1 < a2 ? a3 : a4; a2 = a2 < a3 ? a4 : a5; a3 = a3 < a4 ? a5 : a6; a4 = a4 < a5 ? a6 : a7; a5 = a5 < a6 ? a7 : a0; a6 = a6 < a7 ? a0 : a1; a7 = a7 < a0 ? a1 : a2;
            }
        } else if (MODE == 7) {  // plain mul / add (no fma)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = a0 * m; a1 = a1 + c; a2 = a2 * m; a3 = a3 + c; a4 = a4 * m; a5 = a5 + c; a6 = a6 * m; a7 = a7 + c;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0 = __builtin_sqrtf(a0) + m; a1 = __builtin_sqrtf(a1) + m; a2 = __builtin_sqrtf(a2) + m; a3 = __builtin_sqrtf(a3) + m;
                a4 = __builtin_sqrtf(a4) + m; a5 = __builtin_sqrtf(a5) + m; a6 = __builtin_sqrtf(a6) + m; a7 = __builtin_sqrtf(a7) + m;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char* name, int wps, float* d) {
    int blocks = 256 * wps;  // 256-thread blocks = 4 waves = 1 wave/SIMD each
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 4 * iters * 64.0;  // wave-instructions of the probed kind
    printf("%-6s waves/SIMD %d: %.3f ms  %.1f G wave-ops/s  => %.2f cycles@2.4GHz per op per SIMD\n", name, wps, ms, ops / ms / 1e6,
           1024.0 * 2.4e9 / (ops / (ms * 1e-3)));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) { run<0>("fma", w, d); }
    for (int w : {1, 2, 4}) { run<1>("rcp", w, d); }
    for (int w : {1, 2, 4}) { run<2>("div", w, d); }
    for (int w : {1, 2, 4}) { run<3>("sqrt", w, d); }
    for (int w : {1, 2, 3, 4}) { run<4>("minmax", w, d); }
    for (int w : {1, 2, 3, 4}) { run<5>("int", w, d); }
    for (int w : {1, 2, 3, 4}) { run<6>("cmpsel", w, d); }
    for (int w : {1, 2, 3, 4}) { run<7>("muladd", w, d); }
    return 0;
}
