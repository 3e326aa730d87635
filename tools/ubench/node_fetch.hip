// L1/TA rate probe: every active lane fetches a random 64-B node (4 x 16-B loads) or a 48-B
// record (3 loads) from a small cache-resident array, with next to no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int LOADS>
__global__ void __launch_bounds__(768) k(const float4* __restrict__ nodes, int n_nodes, float* out, int iters, int active_mod) {
    unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    const bool active = (threadIdx.x % 64) % active_mod == 0;
    if (active) {
        for (int i = 0; i < iters; ++i) {
            const unsigned n = (idx >> 8) % (unsigned)n_nodes;
            const float4* p = nodes + (size_t)n * 4;
            float4 a = p[0];
            acc += a.x;
            if (LOADS > 1) { float4 b = p[1]; acc += b.y; }
            if (LOADS > 2) { float4 c = p[2]; acc += c.z; }
            if (LOADS > 3) { float4 d = p[3]; acc += d.w; }
            idx = idx * 1664525u + 1013904223u + (unsigned)(acc > 1e30f);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int LOADS>
void run(const float4* d_nodes, int n_nodes, float* d_out, int active_mod) {
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<LOADS><<<blocks, 768>>>(d_nodes, n_nodes, d_out, 10, active_mod);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<LOADS><<<blocks, 768>>>(d_nodes, n_nodes, d_out, iters, active_mod);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_fetches = (double)blocks * 12 * iters;  // per-wave fetch rounds
    printf("loads/lane %d  active 1/%d  nodes %7d: %.3f ms, %.1f cycles@2.4GHz per wave-round per CU (12 waves/CU)\n", LOADS, active_mod, n_nodes, ms,
           ms * 1e-3 * 2.4e9 / (wave_fetches / 256.0));
}
int main() {
    const int max_nodes = 1 << 20;
    float4* d_nodes; float* d_out;
    (void)hipMalloc(&d_nodes, (size_t)max_nodes * 64);
    (void)hipMemset(d_nodes, 0, (size_t)max_nodes * 64);
    (void)hipMalloc(&d_out, 256 * 768 * 4);
    for (int n : {64, 256, 1000, 3000}) {
        for (int am : {1, 2, 4}) { run<4>(d_nodes, n, d_out, am); run<3>(d_nodes, n, d_out, am); run<1>(d_nodes, n, d_out, am); }
    }
    return 0;
}
