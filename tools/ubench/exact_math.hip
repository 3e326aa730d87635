// Exhaustive comparison of short reciprocal / square-root sequences with the IEEE operations:
// every binary32 bit pattern is run through each candidate and through 1.0f / x, sqrtf(x) and
// 1.0f / sqrtf(x) as the compiler expands them (correctly rounded), and mismatches are counted
// per biased exponent of the input (sign in bit 8 of the bin).  Two NaNs count as equal.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_math.hip -o exact_math && ./exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define DEV __device__ __forceinline__
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- reciprocal ---------------------------------------------------------------------------
DEV float rcp_n(float x, int rounds) {  // v_rcp_f32 + Newton rounds
    float r = __builtin_amdgcn_rcpf(x);
    for (int i = 0; i < rounds; ++i) {
        const float e = fma_(-x, r, 1.0f);
        r = fma_(e, r, r);
    }
    return r;
}
DEV float rcp_q(float x) {  // the compiler's division sequence for 1 / x without scaling and fix-up
    float r = __builtin_amdgcn_rcpf(x);
    float e = fma_(-x, r, 1.0f);
    r = fma_(e, r, r);
    float q = r;
    e = fma_(-x, q, 1.0f);
    q = fma_(e, r, q);
    e = fma_(-x, q, 1.0f);
    return fma_(e, r, q);
}
DEV float rcp_q2(float x) {  // the same with one residual round less
    float r = __builtin_amdgcn_rcpf(x);
    float e = fma_(-x, r, 1.0f);
    r = fma_(e, r, r);
    e = fma_(-x, r, 1.0f);
    return fma_(e, r, r);
}

// ---- square root --------------------------------------------------------------------------
DEV float sqrt_a(float x, float& h_out) {  // v_rsq_f32, coupled Newton round on g ~ sqrt, h ~ 1/(2 sqrt), residual round
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = fma_(-h, g, 0.5f);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
    const float d = fma_(-g, g, x);
    h_out = h;
    return fma_(d, h, g);
}
DEV float sqrt_b(float x, float& h_out) {  // the same without refining h
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = fma_(-h, g, 0.5f);
    g = fma_(g, e, g);
    const float d = fma_(-g, g, x);
    h_out = h;
    return fma_(d, h, g);
}
DEV float sqrt_c(float x, float& h_out) {  // v_sqrt_f32 + one residual round with h from v_rsq_f32
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = fma_(-g, g, x);
    h_out = h;
    return fma_(d, h, g);
}
DEV float sqrt_d(float x, float& h_out) {  // two residual rounds
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = fma_(-h, g, 0.5f);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
    float d = fma_(-g, g, x);
    g = fma_(d, h, g);
    d = fma_(-g, g, x);
    h_out = h;
    return fma_(d, h, g);
}

// ---- 1 / sqrt(x) as two roundings -----------------------------------------------------------
template <int SQ, int START, int ROUNDS>
DEV float inv_sqrt(float x) {
    float h;
    const float s = SQ == 0 ? sqrt_a(x, h) : SQ == 1 ? sqrt_b(x, h) : SQ == 2 ? sqrt_c(x, h) : sqrt_d(x, h);
    if (START == 2) return rcp_q(s);
    if (START == 3) return rcp_q2(s);
    float r = START == 0 ? h + h : __builtin_amdgcn_rcpf(s);
    for (int i = 0; i < ROUNDS; ++i) {
        const float e = fma_(-s, r, 1.0f);
        r = fma_(e, r, r);
    }
    return r;
}

constexpr int kCandidates = 16;
__global__ void __launch_bounds__(256) sweep(unsigned long long* bins /* [kCandidates][512] */) {
    const uint64_t total = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ref_r = 1.0f / x, ref_s = __builtin_sqrtf(x), ref_i = 1.0f / __builtin_sqrtf(x);
        float h;
        float c[kCandidates];
        c[0] = rcp_n(x, 1);
        c[1] = rcp_n(x, 2);
        c[2] = rcp_q(x);
        c[3] = rcp_q2(x);
        c[4] = sqrt_a(x, h);
        c[5] = sqrt_b(x, h);
        c[6] = sqrt_c(x, h);
        c[7] = sqrt_d(x, h);
        c[8] = inv_sqrt<0, 0, 1>(x);
        c[9] = inv_sqrt<0, 0, 2>(x);
        c[10] = inv_sqrt<0, 0, 3>(x);
        c[11] = inv_sqrt<0, 1, 1>(x);
        c[12] = inv_sqrt<0, 1, 2>(x);
        c[13] = inv_sqrt<0, 2, 0>(x);
        c[14] = inv_sqrt<0, 3, 0>(x);
        c[15] = inv_sqrt<2, 1, 2>(x);
        const int bin = (int)((bits >> 23) & 0xffu) | (int)((bits >> 31) << 8);
#pragma unroll
        for (int k = 0; k < kCandidates; ++k) {
            const float ref = k < 4 ? ref_r : k < 8 ? ref_s : ref_i;
            const bool same = (__float_as_uint(ref) == __float_as_uint(c[k])) || (ref != ref && c[k] != c[k]);
            if (!same) atomicAdd(&bins[k * 512 + bin], 1ull);
        }
    }
}

int main() {
    unsigned long long* d;
    if (hipMalloc(&d, sizeof(unsigned long long) * kCandidates * 512) != hipSuccess) return 1;
    hipMemset(d, 0, sizeof(unsigned long long) * kCandidates * 512);
    hipLaunchKernelGGL(sweep, dim3(256 * 8), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::vector<unsigned long long> h(kCandidates * 512);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[kCandidates] = {"rcp_n1", "rcp_n2", "rcp_q", "rcp_q2", "sqrt_a", "sqrt_b", "sqrt_c", "sqrt_d",
                                      "isq_a_h1", "isq_a_h2", "isq_a_h3", "isq_a_r1", "isq_a_r2", "isq_a_q", "isq_a_q2", "isq_c_r2"};
    for (int k = 0; k < kCandidates; ++k) {
        unsigned long long total = 0;
        for (int b = 0; b < 512; ++b) total += h[k * 512 + b];
        std::printf("%-9s mismatches %llu; exponent bins (sign:biased exp:count):", names[k], total);
        // print runs of bad bins compactly
        int run_start = -1;
        for (int b = 0; b <= 512; ++b) {
            const bool bad = b < 512 && h[k * 512 + b] != 0;
            if (bad && run_start < 0) run_start = b;
            if (!bad && run_start >= 0) {
                unsigned long long s = 0;
                for (int j = run_start; j < b; ++j) s += h[k * 512 + j];
                std::printf(" [%c %d..%d: %llu]", run_start >= 256 ? '-' : '+', run_start & 255, (b - 1) & 255, s);
                run_start = -1;
            }
            if (b == 255 && run_start >= 0) {  // do not merge runs across the sign boundary
                unsigned long long s = 0;
                for (int j = run_start; j <= b; ++j) s += h[k * 512 + j];
                std::printf(" [+ %d..%d: %llu]", run_start & 255, 255, s);
                run_start = -1;
            }
        }
        std::printf("\n");
    }
    hipFree(d);
    return 0;
}
