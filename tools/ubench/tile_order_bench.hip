// Times the tile order kernel alone (1080p: 32400 tiles) on costs of a realistic spread.
// hipcc --offload-arch=gfx950 -O3 -I raytracer_amd/csrc/device tools/ubench/tile_order_bench.hip raytracer_amd/csrc/device/rt_kernels.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rt_kernels.h"
int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 32400u;
    std::vector<uint32_t> h(n);
    uint32_t* cost; uint32_t* order; uint32_t* aux; hipMalloc((void**)&aux, rtd::tile_order_aux_bytes()); hipMemset(aux, 0, 256); uint32_t* stamp; hipMalloc((void**)&stamp, 64); const uint32_t key[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    hipMalloc((void**)&cost, n * 4); hipMalloc((void**)&order, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0.0f;
    for (int rep = 0; rep < 20; ++rep) {
        for (uint32_t i = 0; i < n; ++i) h[i] = 40u + (uint32_t)(rand() % 200) + ((i / 240u) % 7u) * 30u;
        hipMemcpy(cost, h.data(), n * 4, hipMemcpyHostToDevice);
        hipEventRecord(a, s);
        rtd::launch_tile_order(cost, order, n, 16, aux, stamp, key, s);
        hipEventRecord(b, s);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    std::vector<uint32_t> o(n);
    hipMemcpy(o.data(), order, n * 4, hipMemcpyDeviceToHost);
    std::vector<int> seen(n, 0);
    int bad = 0;
    for (uint32_t i = 0; i < n; ++i) { if (o[i] >= n || seen[o[i]]++) ++bad; }
    uint32_t st[16]; hipMemcpy(st, stamp, 64, hipMemcpyDeviceToHost);
    // the order must list the classes dearest first and the tiles of a class in image order
    uint32_t mx = 0;
    for (uint32_t i = 0; i < n; ++i) mx = h[i] > mx ? h[i] : mx;
    const float per_cost = 16.0f / ((float)mx + 1.0f);
    auto cls = [&](uint32_t i) { int c = (int)((float)h[i] * per_cost); return 15 - (c > 15 ? 15 : c); };
    for (uint32_t k = 1; k < n && !bad; ++k) {
        const int a = cls(o[k - 1]), b = cls(o[k]);
        if (a > b || (a == b && o[k - 1] > o[k])) ++bad;
    }
    printf("n=%u best %.1f us mean %.1f us (event time incl. launch), permutation errors %d\n", n, best * 1e3f, sum / 18.0f * 1e3f, bad);
    return bad != 0;
}
