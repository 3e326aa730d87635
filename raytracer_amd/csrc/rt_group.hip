// rt_group.hip — several GPUs of one node behind the C ABI: the row-strip exchange (pack / unpack of a
// context's own rows) and RtGroup, one context per device driven from one host thread and combined with
// RCCL (single process, ncclCommInitAll) over xGMI.  The reference is a single-GPU program; what this
// replaces is its one drawFrame per frame (src/main.cpp:1308-1392), issued here once per device.
//
// RCCL is loaded on first use (dlopen), so a host that renders on one GPU never pays for it and the
// library has no load-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "device/rt_trace.h"
#include "host/guarded.h"
#include "host/scene_handle.h"
#include "rt_context.h"

using rtapi::fail;

namespace {

// ---- row-strip exchange ------------------------------------------------------------------------
// A rank's packed rows: its owned rows in ascending y, first all accumulated colours (float4 per
// pixel), then all sample counts (int32 per pixel); every rank's block has room for the largest share
// and is padded to a multiple of 16 bytes.

__host__ __device__ inline int strip_owned_rows(int height, int strip_rows, int strip_count, int strip_index) {
    const int n_strips = (height + strip_rows - 1) / strip_rows;
    int owned = 0;
    for (int s = strip_index; s < n_strips; s += strip_count) {
        const int rows = height - s * strip_rows;
        owned += rows < strip_rows ? rows : strip_rows;
    }
    return owned;
}

// owned row r of rank k -> image row
__device__ inline int strip_row_to_y(int r, int strip_rows, int strip_count, int k) {
    return ((r / strip_rows) * strip_count + k) * strip_rows + (r % strip_rows);
}

__global__ void __launch_bounds__(256) strip_pack_kernel(const float4* __restrict__ accum, const int32_t* __restrict__ counts, int width,
                                                         int height, int strip_rows, int strip_count, int strip_index, int max_rows,
                                                         float4* __restrict__ out_acc, int32_t* __restrict__ out_cnt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(i / width), x = (int)(i % width);
    if (r >= max_rows) return;
    const int y = strip_row_to_y(r, strip_rows, strip_count, strip_index);
    if (y >= height) return;
    out_acc[i] = accum[(long long)y * width + x];
    out_cnt[i] = counts[(long long)y * width + x];
}

// Writes the rows of every rank but `skip` (-1: none) into the framebuffer and rebuilds their BGRA8
// bytes from the colours and counts (raytracer.comp:992-995).
__global__ void __launch_bounds__(256) strip_unpack_kernel(const unsigned char* __restrict__ gathered, size_t rank_bytes, int width, int height,
                                                           int strip_rows, int strip_count, int skip, int max_rows,
                                                           float4* __restrict__ accum, int32_t* __restrict__ counts,
                                                           uint32_t* __restrict__ bgra) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = (int)blockIdx.y;
    if (k == skip) return;
    const int r = (int)(i / width), x = (int)(i % width);
    if (r >= max_rows) return;
    const int y = strip_row_to_y(r, strip_rows, strip_count, k);
    if (y >= height) return;
    const unsigned char* base = gathered + (size_t)k * rank_bytes;
    const float4 a = ((const float4*)base)[i];
    const int32_t n = ((const int32_t*)(base + (size_t)max_rows * width * 16))[i];
    const long long p = (long long)y * width + x;
    accum[p] = a;
    counts[p] = n;
    bgra[p] = rtd::resolve_pixel(a, n);
}

int check_partition(RtContext* c, int strip_rows, int strip_count) {
    if (!c->d_accum) return fail(c, RT_ERR_NO_FRAMEBUFFER, "no framebuffer");
    if (strip_rows <= 0 || strip_count < 1 || strip_count > 65535) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad strip partition");
    return RT_OK;
}

// ---- RCCL, loaded on first use ---------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

void load_rccl(Rccl& r);
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;  // groups may be created from several threads
    std::call_once(once, [] { load_rccl(r); });
    return r;
}
void load_rccl(Rccl& r) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        const char* why = dlerror();  // (a second call would return null: the message is consumed by the first)
        r.error = std::string("cannot load librccl.so: ") + (why ? why : "");
        return;
    }
    auto sym = [&](const char* n) {
        void* p = dlsym(r.handle, n);
        if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
        return p;
    };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
}

// total = first ? src : total + src, element-wise: the samples' reduction where the transport is device copies (fixed
// order 0, 1, ..., n - 1, so the sum is the same every time)
__global__ void __launch_bounds__(256) sum_into_kernel(float4* __restrict__ total_acc, int32_t* __restrict__ total_cnt,
                                                      const float4* __restrict__ acc, const int32_t* __restrict__ cnt, size_t n, int first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = acc[i];
    const int32_t c = cnt[i];
    if (first) {
        total_acc[i] = a;
        total_cnt[i] = c;
    } else {
        float4 t = total_acc[i];
        t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
        total_acc[i] = t;
        total_cnt[i] += c;
    }
}

}  // namespace

// How the group's contexts exchange data.  RCCL (ncclCommInitAll over xGMI) wants every rank on a device of its
// own; contexts that share a device - several groups' worth of work on one GPU, or a rehearsal of the n-device path
// on the one GPU there is - exchange through device-to-device copies instead, with everything else unchanged.
enum { RT_TRANSPORT_RCCL = 0, RT_TRANSPORT_MEMCPY = 1 };

struct RtGroup {
    std::vector<RtContext*> ctx;
    std::vector<ncclComm_t> comm;
    int transport = RT_TRANSPORT_RCCL;
    std::vector<void*> staging;   // per context: its packed rows
    std::vector<void*> gathered;  // per context: every context's packed rows
    // The exchange runs on a stream of its own per context, beside the next call's kernels: render stream = kernel(N),
    // unpack(N - 1), pack(N); exchange stream = gather(N) once pack(N) is through.  `pending`: a gathered frame still
    // waits to be unpacked (done by the next strip call, and by everything that reads or replaces the framebuffer).
    std::vector<hipStream_t> xfer;
    std::vector<hipEvent_t> packed, arrived;
    bool pending = false;
    int32_t pending_rows = 0;
    size_t staging_bytes = 0;
    int32_t stage_key[4] = {0, 0, 0, 0};  // width, height, strip_rows, contexts the staging buffers were sized for
    void* total_accum = nullptr;          // on context 0's device: sum of all contexts' accumulators (rt_group_reduce_samples)
    void* total_counts = nullptr;
    void* peer_accum = nullptr;           // copy transport, contexts on other devices: where their buffers land on context 0's device
    void* peer_counts = nullptr;
    size_t total_pixels = 0;
    std::string err;
};

namespace {
int gfail(RtGroup* g, int code, const std::string& msg) {
    if (g) g->err = msg;
    return code;
}
#define RT_GHIP(g, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return gfail((g), RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RT_GNCCL(g, call)                                                                      \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) return gfail((g), RT_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

void free_group_buffers(RtGroup* g) {
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        // both streams touch the buffers: the exchange writes them, the context's unpack and pack read and write them
        if (i < g->xfer.size() && g->xfer[i]) (void)hipStreamSynchronize(g->xfer[i]);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (i < g->staging.size() && g->staging[i]) (void)hipFree(g->staging[i]);
        if (i < g->gathered.size() && g->gathered[i]) (void)hipFree(g->gathered[i]);
    }
    g->staging.assign(g->ctx.size(), nullptr);
    g->gathered.assign(g->ctx.size(), nullptr);
    g->staging_bytes = 0;
    g->pending = false;
    memset(g->stage_key, 0, sizeof g->stage_key);
}

// The rows gathered by the last strip call, into every context's frame (after the kernels queued since: they wrote
// own rows only).
int flush_pending(RtGroup* g) {
    if (!g->pending) return RT_OK;
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; ++i) {
        RtContext* c = g->ctx[(size_t)i];
        RT_GHIP(g, hipSetDevice(c->device));
        RT_GHIP(g, hipStreamWaitEvent(c->stream, g->arrived[(size_t)i], 0));
        int rc = rt_strip_unpack(c, g->pending_rows, n, i, g->gathered[(size_t)i]);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
    }
    g->pending = false;
    return RT_OK;
}
}  // namespace

extern "C" {

size_t rt_strip_staging_bytes(int32_t width, int32_t height, int32_t strip_rows, int32_t strip_count) {
    if (width <= 0 || height <= 0 || strip_rows <= 0 || strip_count < 1) return 0;
    const int max_rows = strip_owned_rows(height, strip_rows, strip_count, 0);  // rank 0 owns the most rows
    return ((size_t)max_rows * (size_t)width * 20u + 15u) & ~(size_t)15u;  // blocks stay 16-byte aligned back to back
}

int rt_strip_pack(RtContext* c, int32_t strip_rows, int32_t strip_count, int32_t strip_index, void* d_staging) {
    if (!c || !d_staging) return RT_ERR_INVALID_ARGUMENT;
    int rc = check_partition(c, strip_rows, strip_count);
    if (rc != RT_OK) return rc;
    if (strip_index < 0 || strip_index >= strip_count || ((uintptr_t)d_staging & 15u))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "bad strip index or staging pointer (16-byte alignment)");
    RT_HIP(c, hipSetDevice(c->device));
    const int max_rows = strip_owned_rows(c->height, strip_rows, strip_count, 0);
    const long long n = (long long)max_rows * c->width;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(strip_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const float4*)c->d_accum,
                       (const int32_t*)c->d_counts, c->width, c->height, strip_rows, strip_count, strip_index, max_rows, (float4*)d_staging,
                       (int32_t*)((char*)d_staging + (size_t)n * 16));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int rt_strip_unpack(RtContext* c, int32_t strip_rows, int32_t strip_count, int32_t skip_index, const void* d_gathered) {
    if (!c || !d_gathered) return RT_ERR_INVALID_ARGUMENT;
    int rc = check_partition(c, strip_rows, strip_count);
    if (rc != RT_OK) return rc;
    if (((uintptr_t)d_gathered & 15u)) return fail(c, RT_ERR_INVALID_ARGUMENT, "gathered rows must be 16-byte aligned");
    RT_HIP(c, hipSetDevice(c->device));
    const int max_rows = strip_owned_rows(c->height, strip_rows, strip_count, 0);
    const long long n = (long long)max_rows * c->width;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(strip_unpack_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)strip_count), dim3(256), 0, c->stream,
                       (const unsigned char*)d_gathered, rt_strip_staging_bytes(c->width, c->height, strip_rows, strip_count), c->width, c->height, strip_rows, strip_count, skip_index, max_rows,
                       (float4*)c->d_accum, (int32_t*)c->d_counts, (uint32_t*)c->d_bgra);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

static int group_create(int n_devices, const int* device_ordinals, RtGroup** out) {
    RtGroup* g = new RtGroup();
    bool shared_device = false;
    for (int i = 0; i < n_devices; ++i) {
        for (int j = 0; j < i; ++j) shared_device = shared_device || device_ordinals[i] == device_ordinals[j];
        RtContext* c = nullptr;
        int rc = rt_create(device_ordinals[i], &c);
        if (rc != RT_OK) {
            rt_group_destroy(g);
            return rc;  // rt_last_error(NULL) has the text
        }
        g->ctx.push_back(c);
    }
    const size_t n = (size_t)n_devices;
    g->staging.assign(n, nullptr);
    g->gathered.assign(n, nullptr);
    g->xfer.assign(n, nullptr);
    g->packed.assign(n, nullptr);
    g->arrived.assign(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        hipError_t e = hipSetDevice(g->ctx[i]->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->xfer[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->packed[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->arrived[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)rtapi::fail(nullptr, RT_ERR_HIP, std::string("rt_group_create: ") + hipGetErrorString(e));
            rt_group_destroy(g);
            return RT_ERR_HIP;
        }
    }
    // RT_GROUP_TRANSPORT = memcpy | rccl overrides the choice (rccl with a shared device is refused by RCCL itself)
    const char* want = getenv("RT_GROUP_TRANSPORT");
    g->transport = shared_device ? RT_TRANSPORT_MEMCPY : RT_TRANSPORT_RCCL;
    if (want && strcmp(want, "memcpy") == 0) g->transport = RT_TRANSPORT_MEMCPY;
    if (want && strcmp(want, "rccl") == 0) g->transport = RT_TRANSPORT_RCCL;
    if (g->transport == RT_TRANSPORT_RCCL) {
        Rccl& r = rccl();
        if (!r.error.empty()) {
            (void)rtapi::fail(nullptr, RT_ERR_HIP, r.error);
            rt_group_destroy(g);
            return RT_ERR_HIP;
        }
        g->comm.assign(n, nullptr);
        ncclResult_t nr = r.CommInitAll(g->comm.data(), n_devices, device_ordinals);
        if (nr != ncclSuccess) {
            (void)rtapi::fail(nullptr, RT_ERR_HIP, std::string("ncclCommInitAll: ") + r.GetErrorString(nr));
            g->comm.clear();
            rt_group_destroy(g);
            return RT_ERR_HIP;
        }
    }
    *out = g;
    return RT_OK;
}

int rt_group_create(int n_devices, const int* device_ordinals, RtGroup** out) {
    if (!out || n_devices < 1 || n_devices > 64 || !device_ordinals) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    return rt::guarded([&]() -> int { return group_create(n_devices, device_ordinals, out); });
}

int rt_group_destroy(RtGroup* g) {
    if (!g) return RT_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (i < g->xfer.size() && g->xfer[i]) (void)hipStreamSynchronize(g->xfer[i]);
    }
    for (ncclComm_t cm : g->comm)
        if (cm) (void)rccl().CommDestroy(cm);
    free_group_buffers(g);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device);
        if (i < g->xfer.size() && g->xfer[i]) (void)hipStreamDestroy(g->xfer[i]);
        if (i < g->packed.size() && g->packed[i]) (void)hipEventDestroy(g->packed[i]);
        if (i < g->arrived.size() && g->arrived[i]) (void)hipEventDestroy(g->arrived[i]);
    }
    if (!g->ctx.empty()) {
        (void)hipSetDevice(g->ctx[0]->device);
        for (void* p : {g->total_accum, g->total_counts, g->peer_accum, g->peer_counts})
            if (p) (void)hipFree(p);
    }
    for (RtContext* c : g->ctx) (void)rt_destroy(c);
    delete g;
    return RT_OK;
}

int rt_group_size(const RtGroup* g) { return g ? (int)g->ctx.size() : 0; }

int rt_group_context(RtGroup* g, int index, RtContext** out) {
    if (!g || !out || index < 0 || index >= (int)g->ctx.size()) return RT_ERR_INVALID_ARGUMENT;
    *out = g->ctx[(size_t)index];
    return RT_OK;
}

const char* rt_group_last_error(RtGroup* g) { return g ? g->err.c_str() : ""; }

// The scene is replicated: checked and built once on the host (the tree of a million triangles takes seconds), the
// same arrays copied to every context's device.
int rt_group_upload_scene_from(RtGroup* g, const RtScene* s) {
    if (!g || !s) return RT_ERR_INVALID_ARGUMENT;
    return rt::guarded([&]() -> int {
        int rc = flush_pending(g);
        if (rc != RT_OK) return rc;
        const rt::Scene& sc = s->scene;
        rtapi::BuiltScene built;
        std::string err;
        rc = rtapi::build_scene(g->ctx[0]->opt, sc.sphereVec.data(), sc.total_spheres, sc.materialVec.data(), (int32_t)sc.materialVec.size(),
                                sc.lightsVec.data(), sc.total_lights, sc.triangleVec.data(), sc.total_triangles, sc.vertexVec.data(),
                                (int32_t)sc.vertexVec.size(), sc.indexVec.data(), (int32_t)sc.indexVec.size(), sc.meshVec.data(),
                                sc.total_meshes, built, err);
        if (rc != RT_OK) return gfail(g, rc, err);
        for (RtContext* c : g->ctx) {
            rc = rtapi::upload_built(c, built);
            if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
        }
        return RT_OK;
    });
}

int rt_group_resize(RtGroup* g, int32_t width, int32_t height) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    g->pending = false;  // rows gathered for the framebuffers that go away here
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        RT_GHIP(g, hipSetDevice(g->ctx[i]->device));
        RT_GHIP(g, hipStreamSynchronize(g->xfer[i]));
        int rc = rt_resize(g->ctx[i], width, height);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(g->ctx[i]));
    }
    return RT_OK;
}

// `n_frames` frames (pcs[0 .. n_frames - 1]) shared by row strips: every context renders its rows of all of them with
// global pixel coordinates (the bits of one-GPU frames) - in one rt_render_frames call -, unpacks what the previous
// call gathered, packs its accumulated rows, and one all-gather on the exchange streams completes the accumulated
// frame everywhere (unpacked by the next strip call, rt_group_sync, or whatever else needs the whole frame).
static int render_strips(RtGroup* g, const RtCamera* camera, const RtPushConstants* pcs, int n_frames, RtParams p, int32_t strip_rows) {
    const int n = (int)g->ctx.size();
    if (strip_rows <= 0) return gfail(g, RT_ERR_INVALID_ARGUMENT, "strip_rows must be positive");
    const int32_t w = g->ctx[0]->width, h = g->ctx[0]->height;
    if (w <= 0 || h <= 0) return gfail(g, RT_ERR_NO_FRAMEBUFFER, "rt_group_resize has not been called");
    const int32_t key[4] = {w, h, strip_rows, n};
    if (memcmp(key, g->stage_key, sizeof key) != 0) {
        int rc = flush_pending(g);  // rows gathered under the previous partition
        if (rc != RT_OK) return rc;
        free_group_buffers(g);
        g->staging_bytes = rt_strip_staging_bytes(w, h, strip_rows, n);
        for (int i = 0; i < n; ++i) {
            RT_GHIP(g, hipSetDevice(g->ctx[(size_t)i]->device));
            RT_GHIP(g, hipStreamSynchronize(g->ctx[(size_t)i]->stream));
            RT_GHIP(g, hipMalloc(&g->staging[(size_t)i], g->staging_bytes));
            RT_GHIP(g, hipMalloc(&g->gathered[(size_t)i], g->staging_bytes * (size_t)n));
            RT_GHIP(g, hipMemsetAsync(g->staging[(size_t)i], 0, g->staging_bytes, g->ctx[(size_t)i]->stream));
        }
        memcpy(g->stage_key, key, sizeof key);
    }
    for (int i = 0; i < n; ++i) {
        RtContext* c = g->ctx[(size_t)i];
        p.strip_rows = strip_rows; p.strip_count = n; p.strip_index = i;
        int rc = n_frames == 1 ? rt_render_frame(c, camera, pcs, &p) : rt_render_frames(c, camera, pcs, n_frames, &p);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
    }
    {
        int rc = flush_pending(g);  // the previous call's rows, now that this call's kernels are queued behind nothing of it
        if (rc != RT_OK) return rc;
    }
    for (int i = 0; i < n; ++i) {
        RtContext* c = g->ctx[(size_t)i];
        // (the previous gather read this staging buffer on the exchange streams: every context's unpack above waited for
        // its own gather, which copied from all of them, only where the transport is RCCL's all-gather; the copy
        // transport reads staging[j] on context i's exchange stream, so pack(N) waits for all of those)
        if (g->transport == RT_TRANSPORT_MEMCPY)
            for (int j = 0; j < n; ++j) RT_GHIP(g, hipStreamWaitEvent(c->stream, g->arrived[(size_t)j], 0));
        int rc = rt_strip_pack(c, strip_rows, n, i, g->staging[(size_t)i]);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
        RT_GHIP(g, hipSetDevice(c->device));
        RT_GHIP(g, hipEventRecord(g->packed[(size_t)i], c->stream));
    }
    if (g->transport == RT_TRANSPORT_RCCL) {
        Rccl& r = rccl();
        for (int i = 0; i < n; ++i) {
            RT_GHIP(g, hipSetDevice(g->ctx[(size_t)i]->device));
            RT_GHIP(g, hipStreamWaitEvent(g->xfer[(size_t)i], g->packed[(size_t)i], 0));
        }
        RT_GNCCL(g, r.GroupStart());
        for (int i = 0; i < n; ++i) {
            ncclResult_t nr = r.AllGather(g->staging[(size_t)i], g->gathered[(size_t)i], g->staging_bytes, ncclChar, g->comm[(size_t)i],
                                          g->xfer[(size_t)i]);
            if (nr != ncclSuccess) {
                (void)r.GroupEnd();
                return gfail(g, RT_ERR_HIP, std::string("ncclAllGather: ") + r.GetErrorString(nr));
            }
        }
        RT_GNCCL(g, r.GroupEnd());
    } else {
        // the all-gather as copies: context i's exchange stream fetches every context's packed rows once they are packed
        // (packed[i] first: it also lies behind context i's unpack of the previous call, which read gathered[i])
        for (int i = 0; i < n; ++i) {
            RT_GHIP(g, hipSetDevice(g->ctx[(size_t)i]->device));
            RT_GHIP(g, hipStreamWaitEvent(g->xfer[(size_t)i], g->packed[(size_t)i], 0));
            for (int j = 0; j < n; ++j) {
                RT_GHIP(g, hipStreamWaitEvent(g->xfer[(size_t)i], g->packed[(size_t)j], 0));
                RT_GHIP(g, hipMemcpyAsync((char*)g->gathered[(size_t)i] + (size_t)j * g->staging_bytes, g->staging[(size_t)j], g->staging_bytes,
                                          hipMemcpyDefault, g->xfer[(size_t)i]));
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        RT_GHIP(g, hipSetDevice(g->ctx[(size_t)i]->device));
        RT_GHIP(g, hipEventRecord(g->arrived[(size_t)i], g->xfer[(size_t)i]));
    }
    g->pending = true;
    g->pending_rows = strip_rows;
    return RT_OK;
}

static int group_render_frames(RtGroup* g, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, int mode,
                               int32_t strip_rows, int32_t frames) {
    const int n = (int)g->ctx.size();
    RtParams p;
    if (params) p = *params; else rt_default_params(&p);
    std::vector<RtPushConstants> pcs((size_t)frames);
    if (mode == RT_GROUP_STRIPS) {
        // frames frameCount, frameCount + 1, ...: every context its rows of each, one exchange at the end
        for (int k = 0; k < frames; ++k) {
            pcs[(size_t)k] = *pc;
            pcs[(size_t)k].frameCount = pc->frameCount + (uint32_t)k;
            if (k > 0) pcs[(size_t)k].reset_frame_accumulation = 0;
        }
        return render_strips(g, camera, pcs.data(), frames, p, strip_rows);
    }
    int rc = flush_pending(g);
    if (rc != RT_OK) return rc;
    p.strip_rows = 0; p.strip_count = 1; p.strip_index = 0;
    // context i renders the frames numbered frameCount + i, + i + n, ...: what `frames` calls of
    // rt_group_render_frame(RT_GROUP_SAMPLES) with frameCount advanced by n each time give, but every context gets
    // its frames in one rt_render_frames call (one launch where that applies); pc's reset applies to a context's first
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < frames; ++k) {
            pcs[(size_t)k] = *pc;
            pcs[(size_t)k].frameCount = pc->frameCount + (uint32_t)(i + k * n);
            if (k > 0) pcs[(size_t)k].reset_frame_accumulation = 0;
        }
        rc = rt_render_frames(g->ctx[(size_t)i], camera, pcs.data(), frames, &p);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(g->ctx[(size_t)i]));
    }
    return RT_OK;
}

int rt_group_render_frames(RtGroup* g, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, int mode,
                           int32_t strip_rows, int32_t frames) {
    if (!g || !camera || !pc) return RT_ERR_INVALID_ARGUMENT;
    if (mode != RT_GROUP_STRIPS && mode != RT_GROUP_SAMPLES) return gfail(g, RT_ERR_INVALID_ARGUMENT, "unknown group mode");
    if (frames < 1 || frames > 4096) return gfail(g, RT_ERR_INVALID_ARGUMENT, "frames out of range");
    return rt::guarded([&]() -> int { return group_render_frames(g, camera, pc, params, mode, strip_rows, frames); });
}

int rt_group_render_frame(RtGroup* g, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, int mode,
                          int32_t strip_rows) {
    return rt_group_render_frames(g, camera, pc, params, mode, strip_rows, 1);
}

int rt_group_reduce_samples(RtGroup* g) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    const int n = (int)g->ctx.size();
    RtContext* root = g->ctx[0];
    if (!root->d_accum) return gfail(g, RT_ERR_NO_FRAMEBUFFER, "rt_group_resize has not been called");
    int rc = flush_pending(g);
    if (rc != RT_OK) return rc;
    const size_t pixels = (size_t)root->width * (size_t)root->height;
    RT_GHIP(g, hipSetDevice(root->device));
    if (pixels != g->total_pixels) {
        RT_GHIP(g, hipStreamSynchronize(root->stream));
        for (void** p : {&g->total_accum, &g->total_counts, &g->peer_accum, &g->peer_counts}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        g->total_pixels = 0;
        RT_GHIP(g, hipMalloc(&g->total_accum, pixels * 16));
        RT_GHIP(g, hipMalloc(&g->total_counts, pixels * 4));
        g->total_pixels = pixels;
    }
    if (g->transport == RT_TRANSPORT_RCCL) {
        Rccl& r = rccl();
        RT_GNCCL(g, r.GroupStart());
        for (int i = 0; i < n; ++i) {
            RtContext* c = g->ctx[(size_t)i];
            ncclResult_t a = r.Reduce(c->d_accum, i == 0 ? g->total_accum : nullptr, pixels * 4, ncclFloat, ncclSum, 0, g->comm[(size_t)i], c->stream);
            ncclResult_t b = a == ncclSuccess
                                 ? r.Reduce(c->d_counts, i == 0 ? g->total_counts : nullptr, pixels, ncclInt32, ncclSum, 0, g->comm[(size_t)i], c->stream)
                                 : a;
            if (b != ncclSuccess) {
                (void)r.GroupEnd();
                return gfail(g, RT_ERR_HIP, std::string("ncclReduce: ") + r.GetErrorString(b));
            }
        }
        RT_GNCCL(g, r.GroupEnd());
    } else {
        // the reduction as a sum on context 0's stream, contexts in order: a context on the same device is read in
        // place once its stream has got this far, one on another device is copied over first
        for (int i = 0; i < n; ++i) {
            RtContext* c = g->ctx[(size_t)i];
            const void* acc = c->d_accum;
            const void* cnt = c->d_counts;
            if (i > 0) {
                RT_GHIP(g, hipSetDevice(c->device));
                RT_GHIP(g, hipEventRecord(g->packed[(size_t)i], c->stream));
                RT_GHIP(g, hipSetDevice(root->device));
                RT_GHIP(g, hipStreamWaitEvent(root->stream, g->packed[(size_t)i], 0));
                if (c->device != root->device) {
                    if (!g->peer_accum) {
                        RT_GHIP(g, hipMalloc(&g->peer_accum, pixels * 16));
                        RT_GHIP(g, hipMalloc(&g->peer_counts, pixels * 4));
                    }
                    RT_GHIP(g, hipMemcpyAsync(g->peer_accum, acc, pixels * 16, hipMemcpyDefault, root->stream));
                    RT_GHIP(g, hipMemcpyAsync(g->peer_counts, cnt, pixels * 4, hipMemcpyDefault, root->stream));
                    acc = g->peer_accum;
                    cnt = g->peer_counts;
                }
            }
            hipLaunchKernelGGL(sum_into_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, root->stream, (float4*)g->total_accum,
                               (int32_t*)g->total_counts, (const float4*)acc, (const int32_t*)cnt, pixels, i == 0 ? 1 : 0);
            RT_GHIP(g, hipGetLastError());
        }
        // the other contexts' next frames must not overwrite what the sum still reads
        RT_GHIP(g, hipEventRecord(g->arrived[0], root->stream));
        for (int i = 1; i < n; ++i) {
            RT_GHIP(g, hipSetDevice(g->ctx[(size_t)i]->device));
            RT_GHIP(g, hipStreamWaitEvent(g->ctx[(size_t)i]->stream, g->arrived[0], 0));
        }
    }
    // the image of the sum, on context 0's device: acc / max(1, count) as B,G,R,A bytes (raytracer.comp:992-995)
    RT_GHIP(g, hipSetDevice(root->device));
    RT_GHIP(g, rtd::launch_resolve((float4*)g->total_accum, (int32_t*)g->total_counts, (uint32_t*)root->d_bgra, (int)pixels, root->stream));
    return RT_OK;
}

int rt_group_read_total(RtGroup* g, float* accum, int32_t* counts) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (!g->total_accum) return gfail(g, RT_ERR_INVALID_ARGUMENT, "rt_group_reduce_samples has not been called");
    RtContext* root = g->ctx[0];
    RT_GHIP(g, hipSetDevice(root->device));
    RT_GHIP(g, hipStreamSynchronize(root->stream));
    if (accum) RT_GHIP(g, hipMemcpy(accum, g->total_accum, g->total_pixels * 16, hipMemcpyDeviceToHost));
    if (counts) RT_GHIP(g, hipMemcpy(counts, g->total_counts, g->total_pixels * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_group_sync(RtGroup* g) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    int rc = flush_pending(g);
    if (rc != RT_OK) return rc;
    for (RtContext* c : g->ctx) {
        rc = rt_sync(c);
        if (rc != RT_OK) return gfail(g, rc, rt_last_error(c));
    }
    return RT_OK;
}

}  // extern "C"
