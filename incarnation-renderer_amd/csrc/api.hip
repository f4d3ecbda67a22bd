// api.hip -- the C ABI of libincr_hip.so (include/incr.h): context, resource layer, scene upload and the
// per-pass entry points.  Host code only; kernels live in pathtrace.hip / raster.hip.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <chrono>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <new>
#include <algorithm>
#include <thread>
#include <chrono>
#include <vector>

#include "api_internal.h"
#include "test_hooks.h"

using namespace incr;

extern "C" hipError_t incr_launch_pathtrace_wf(const incr::PtParams* p, const incr::WfBuffers* pipes, int numPipes, const uint32_t* tileRow, hipStream_t* streams,
                                               hipEvent_t fork, hipEvent_t* joins, int numCUs, uint32_t stackEntries, hipEvent_t* events,
                                               hipStream_t* sideStreams, hipEvent_t* bounceFork, hipEvent_t* bounceJoin, int overlap, int accPlacement, unsigned long long* overflowTally);
extern "C" hipError_t incr_launch_bvh_update(uint32_t numTris, const incr::TriObject* src, const float* meshLocalToWorld, const incr::MeshInstanceInfo* meshInfo,
                                             incr::TriIsect* tris, incr::TriShade* shade, incr::BvhNode8* nodes, const uint32_t* dLevelNodes,
                                             const uint32_t* ranges, uint32_t numRanges, void* nodeBoxes, hipStream_t stream);
extern "C" hipError_t incr_launch_mip_box(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh, uint32_t format,
                                          const float* srgbLut, hipStream_t stream);
extern "C" hipError_t incr_launch_mip_tile(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_bc_decode(const uint8_t* blocks, uint32_t w, uint32_t h, uint32_t format, uint32_t* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_mip_untile(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_alpha_quads(const uint32_t* level, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_material_chain(const uint32_t* base, const uint32_t* normal, const uint32_t* mr, uint64_t words, void* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_material_texels(const uint32_t* base, const uint32_t* normal, const uint32_t* mr, uint32_t w, uint32_t h, void* dst, hipStream_t stream);
extern "C" hipError_t incr_launch_mip_adopt_vmf(const uint32_t* vmfN, const uint32_t* vmfMR, uint32_t* dstN, uint32_t* dstMR, uint64_t first, uint64_t count, hipStream_t stream);
extern "C" hipError_t incr_launch_trace_rays(const incr::PtParams* p, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax,
                                             int anyHit, uint32_t* ids, float* tuv, hipStream_t stream);
extern "C" hipError_t incr_launch_codec_selftest(unsigned long long* out4, hipStream_t stream);
extern "C" hipError_t incr_launch_rcp_selftest(unsigned long long* out2, hipStream_t stream);
extern "C" hipError_t incr_launch_tonemap(const uint32_t* hdr, uint32_t* ldr, size_t n, int ev100, int viewmode, const float* srgbSteps, hipStream_t stream);
extern "C" hipError_t incr_launch_unshard(uint8_t* dst, const uint8_t* src, uint32_t W, uint32_t H, uint32_t h, uint32_t numShards,
                                          uint32_t localRows, uint32_t texelBytes, hipStream_t stream);

// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
int incr_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
// ---------------------------------------------------------------------------------------------- profiler ranges (api_internal.h)
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* want = getenv("INCR_ROCTX");
        const bool on = want ? atoi(want) != 0 : getenv("ROCP_TOOL_LIBRARIES") != nullptr;      // rocprofv3 exports its tool library list to the profiled process
        if (!on) return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;                                                                          // no profiler SDK on this machine: ranges stay no-ops
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
}
void incr_roctx_push(const char* name) { const Roctx& r = roctx(); if (r.push) (void)r.push(name); }
void incr_roctx_pop() { const Roctx& r = roctx(); if (r.pop) (void)r.pop(); }

static constexpr int kProfEvents = kWfProfEvents;      // per pipeline of a recorded incr_pathtrace call (pt_params.h)
static constexpr int kMaxPipes = 8;

static void fill_scene_params(const incr_context* ctx, const incr_scene* sc, PtParams& P);

extern "C" {

const char* incr_last_error(void) { return g_err; }
const char* incr_version(void) { return "incarnation-renderer_amd 0.1 (gfx950)"; }

int incr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int incr_initialize(int device_index, incr_context** out_ctx) {
    if (!out_ctx) return fail(INCR_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    const int n = incr_device_count();
    if (n <= 0) return fail(INCR_ERR_NO_DEVICE, "no HIP device visible: the render path has no CPU fallback");
    if (device_index < 0 || device_index >= n) return fail(INCR_ERR_INVALID_ARGUMENT, "device_index %d out of range (0..%d)", device_index, n - 1);
    HIP_TRY(hipSetDevice(device_index));
    incr_context* ctx = new (std::nothrow) incr_context();
    if (!ctx) return fail(INCR_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device_index;
    HIP_TRY(hipGetDeviceProperties(&ctx->props, device_index));
    // D3D sRGB -> linear table (D3D11.3 functional spec, sRGB read conversion): exact formula in double, one rounding
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        const double c = i / 255.0;
        lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    HIP_TRY(hipMalloc((void**)&ctx->srgbLut, sizeof lut));
    HIP_TRY(hipMemcpy(ctx->srgbLut, lut, sizeof lut, hipMemcpyHostToDevice));
    // FLOAT -> UNORM_SRGB render-target encode as the values where it steps (pt_device.h linear_to_srgb8): steps[k] = the smallest float in [0, 1] whose code is >= k,
    // by bisection over the float patterns (non-negative floats order like their patterns; the encode is non-decreasing) with this host's powf
    {
        auto code = [](float c) {
            const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f;
            return (int)floorf(e * 255.f + 0.5f);
        };
        auto from_bits = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
        float steps[260];
        steps[0] = 0.f;
        for (int k = 1; k <= 255; ++k) {
            uint32_t lo = 0u, hi = 0x3f800000u;                   // code(lo) < k <= code(hi): code(0) = 0, code(1) = 255
            while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (code(from_bits(mid)) >= k) hi = mid; else lo = mid; }
            steps[k] = from_bits(hi);
        }
        steps[256] = steps[257] = steps[258] = steps[259] = INFINITY;
        HIP_TRY(hipMalloc((void**)&ctx->srgbSteps, sizeof steps));
        HIP_TRY(hipMemcpy(ctx->srgbSteps, steps, sizeof steps, hipMemcpyHostToDevice));
    }
    // 64 ray-count stripes, one 128-byte line each: the lane-per-pixel raster passes count their ray queries here (one
    // block-reduced atomic per workgroup, spread over 64 L2 lines) and a one-wave kernel folds them into the caller's stats
    HIP_TRY(hipMalloc((void**)&ctx->statStripes, 64 * 128));
    HIP_TRY(hipMemset(ctx->statStripes, 0, 64 * 128));
    const char* rc = test_hook_env("INCR_WF_RAYS_PER_PATH");      // test hook: a tiny queue forces the overflow (inline shadow) path
    ctx->raysPerPath = rc ? (uint32_t)std::max(0, atoi(rc)) + 0x80000000u : 0u;          // bit 31: forced; otherwise sized from the scene per call (wf_rays_per_path)
    ctx->wfRaysPerPathAllocated = 0; ctx->wfOverflowTally = nullptr;
    HIP_TRY(hipMalloc((void**)&ctx->wfOverflowTally, 8 * 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(ctx->wfOverflowTally, 0, 8 * 4 * sizeof(unsigned long long)));
    const char* se = test_hook_env("INCR_WF_STACK_ENTRIES");      // test hook: run the 16- / 24-entry traversal kernels on a shallow tree
    ctx->forceStackEntries = se ? (uint32_t)std::max(0, atoi(se)) : 0u;
    memset(&ctx->wf, 0, sizeof ctx->wf); ctx->wfBlock = nullptr; ctx->wfBytes = 0; ctx->wfStage = nullptr; ctx->wfStageBytes = 0;
    // pipelines of a path-traced frame (pathtrace_wf.hip): independent bands of tile rows on concurrent streams
    const char* pe = test_hook_env("INCR_WF_PIPELINES");
    ctx->wfPipesWanted = pe ? std::min(std::max(atoi(pe), 1), kMaxPipes) : INCR_WF_PIPELINES_DEFAULT;
    ctx->wfPipes = 1; memset(ctx->wfTileRow, 0, sizeof ctx->wfTileRow);
    memset(ctx->wfStreams, 0, sizeof ctx->wfStreams); memset(ctx->wfJoin, 0, sizeof ctx->wfJoin);
    memset(ctx->wfSideStreams, 0, sizeof ctx->wfSideStreams); memset(ctx->wfBounceFork, 0, sizeof ctx->wfBounceFork); memset(ctx->wfBounceJoin, 0, sizeof ctx->wfBounceJoin);
    { const char* ap = test_hook_env("INCR_WF_ACC_PLACEMENT"); ctx->wfAccPlacement = ap ? std::min(std::max(atoi(ap), -1), 1) : -1; }
    { const char* ov = test_hook_env("INCR_WF_OVERLAP"); ctx->wfOverlap = ov ? atoi(ov) != 0 : true; }          // bounce overlap (pathtrace_wf.hip launcher), incr_set_pathtrace_overlap
    HIP_TRY(hipEventCreateWithFlags(&ctx->wfFork, hipEventDisableTiming));      // the pipelines' own streams are created on first use
    ctx->profiling = false; ctx->profFrames = 0;
    ctx->rasterBlock = nullptr; ctx->rasterBytes = 0; ctx->rasterCounters = nullptr;
    *out_ctx = ctx;
    return INCR_OK;
}

int incr_teardown(incr_context* ctx) {
    if (!ctx) return INCR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->srgbLut) (void)hipFree(ctx->srgbLut);
    if (ctx->srgbSteps) (void)hipFree(ctx->srgbSteps);
    if (ctx->statStripes) (void)hipFree(ctx->statStripes);
    if (ctx->wfOverflowTally) (void)hipFree(ctx->wfOverflowTally);
    if (ctx->wfBlock) (void)hipFree(ctx->wfBlock);
    if (ctx->wfStage) (void)hipFree(ctx->wfStage);
    for (int i = 1; i < kMaxPipes; ++i) { if (ctx->wfStreams[i]) (void)hipStreamDestroy(ctx->wfStreams[i]); if (ctx->wfJoin[i]) (void)hipEventDestroy(ctx->wfJoin[i]); }
    for (int i = 0; i < kMaxPipes; ++i) if (ctx->wfSideStreams[i]) (void)hipStreamDestroy(ctx->wfSideStreams[i]);
    for (int i = 0; i < kMaxPipes * 4; ++i) { if (ctx->wfBounceFork[i]) (void)hipEventDestroy(ctx->wfBounceFork[i]); if (ctx->wfBounceJoin[i]) (void)hipEventDestroy(ctx->wfBounceJoin[i]); }
    if (ctx->wfFork) (void)hipEventDestroy(ctx->wfFork);
    if (ctx->rasterBlock) (void)hipFree(ctx->rasterBlock);
    for (hipEvent_t e : ctx->profEvents) (void)hipEventDestroy(e);
    delete ctx;
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- resources
int incr_buffer_create(incr_context* ctx, uint64_t bytes, void** out) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!out) return fail(INCR_ERR_INVALID_ARGUMENT, "out_device_ptr is NULL");
    *out = nullptr;
    HIP_TRY(hipMalloc(out, bytes ? bytes : 4));
    return INCR_OK;
}
int incr_buffer_destroy(incr_context* ctx, void* p) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (p) HIP_TRY(hipFree(p));
    return INCR_OK;
}
int incr_buffer_upload(incr_context* ctx, void* dst, const void* src, uint64_t bytes, incr_stream stream) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (bytes && (!dst || !src)) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return INCR_OK;
}
int incr_buffer_download(incr_context* ctx, void* dst, const void* src, uint64_t bytes, incr_stream stream) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (bytes && (!dst || !src)) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return INCR_OK;
}
int incr_buffer_clear(incr_context* ctx, void* dst, uint32_t value32, uint64_t bytes, incr_stream stream) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (bytes % 4) return fail(INCR_ERR_INVALID_ARGUMENT, "bytes must be a multiple of 4");
    if (bytes) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)dst, (int)value32, bytes / 4, (hipStream_t)stream));
    return INCR_OK;
}
int incr_stream_synchronize(incr_context* ctx, incr_stream stream) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return INCR_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------- pinned staging of the update path
int incr_staging_begin(incr_scene* sc, size_t bytes) {
    incr_scene::UpdateStaging& g = sc->staging;
    g.slot ^= 1; g.used = 0;
    const int k = g.slot;
    if (!g.done[k]) HIP_TRY(hipEventCreateWithFlags(&g.done[k], hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(g.done[k]));                  // the copies of the update before last: long done in any steady state
    bytes += 256;
    if (g.cap[k] < bytes) {
        if (g.host[k]) { HIP_TRY(hipHostFree(g.host[k])); g.host[k] = nullptr; g.cap[k] = 0; }
        HIP_TRY(hipHostMalloc((void**)&g.host[k], bytes * 2, hipHostMallocDefault));
        g.cap[k] = bytes * 2;
    }
    return INCR_OK;
}
int incr_staging_upload(incr_scene* sc, void* dst, const void* src, size_t bytes, hipStream_t stream) {
    incr_scene::UpdateStaging& g = sc->staging;
    if (!bytes) return INCR_OK;
    const size_t at = (g.used + 15) & ~(size_t)15;
    if (at + bytes > g.cap[g.slot]) return fail(INCR_ERR_INVALID_ARGUMENT, "update staging overflow (%zu + %zu of %zu bytes)", at, bytes, g.cap[g.slot]);
    memcpy(g.host[g.slot] + at, src, bytes);
    g.used = at + bytes;
    HIP_TRY(hipMemcpyAsync(dst, g.host[g.slot] + at, bytes, hipMemcpyHostToDevice, stream));
    return INCR_OK;
}
int incr_staging_end(incr_scene* sc, hipStream_t stream) {
    HIP_TRY(hipEventRecord(sc->staging.done[sc->staging.slot], stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- scene
// Light-grid arrays keep their allocation while the new lists fit (incr_update_bvh re-derives the grid on the stream).
static int upload_light_grid(incr_scene* sc, hipStream_t stream, bool staged) {
    if (!staged) build_light_grid(sc->host, sc->bvh, sc->grid);          // (the update path builds it before it sizes its staging)
    const size_t no = sc->grid.cellOffsets.size(), nl = sc->grid.cellLights.size();
    if (no > sc->gridOffsetsCap || !sc->dGridOffsets) {
        if (sc->dGridOffsets) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(sc->dGridOffsets)); sc->dGridOffsets = nullptr; }
        sc->gridOffsetsCap = no + 64; HIP_TRY(hipMalloc((void**)&sc->dGridOffsets, sc->gridOffsetsCap * 4));
    }
    if (nl > sc->gridLightsCap || !sc->dGridLights) {
        if (sc->dGridLights) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(sc->dGridLights)); sc->dGridLights = nullptr; }
        sc->gridLightsCap = nl * 2 + 64; HIP_TRY(hipMalloc((void**)&sc->dGridLights, sc->gridLightsCap * 4));
    }
    if (staged) {
        int rc = incr_staging_upload(sc, sc->dGridOffsets, sc->grid.cellOffsets.data(), no * 4, stream);
        if (!rc) rc = incr_staging_upload(sc, sc->dGridLights, sc->grid.cellLights.data(), nl * 4, stream);
        return rc;
    }
    HIP_TRY(hipMemcpy(sc->dGridOffsets, sc->grid.cellOffsets.data(), no * 4, hipMemcpyHostToDevice));          // scene creation
    if (nl) HIP_TRY(hipMemcpy(sc->dGridLights, sc->grid.cellLights.data(), nl * 4, hipMemcpyHostToDevice));
    return INCR_OK;
}

// load_light (pt_shade.h) evaluated once per light loop index on the host: what the shadow_gen kernel keeps in LDS
static int upload_resolved_lights(incr_scene* sc) {
    const HostScene& h = sc->host;
    const size_t n = h.globalLights.size();                 // PtParams::lightCount
    std::vector<float> rec(n * 20, 0.f);
    for (size_t li = 0; li < n; ++li) {
        const int lightId = li < h.lightIndices.size() ? h.lightIndices[li] : 0;
        incr_light L; memset(&L, 0, sizeof L);
        if (lightId >= 0 && (size_t)lightId < h.globalLights.size()) L = h.globalLights[(size_t)lightId];
        const bool haveT = lightId >= 0 && (size_t)lightId < h.lightIndices.size() && (size_t)lightId < h.lightTransforms.size();
        float m[16] = {0.f};
        if (haveT) memcpy(m, h.lightTransforms[(size_t)lightId].m, 64);
        float* r = &rec[li * 20];
        uint32_t typeBits = (uint32_t)L.type; memcpy(&r[0], &typeBits, 4); r[1] = L.range; r[2] = L.spotAngles[0]; r[3] = L.spotAngles[1];
        r[4] = L.color[0]; r[5] = L.color[1]; r[6] = L.color[2]; r[7] = L.intensity;
        r[8] = m[0]; r[9] = m[1]; r[10] = m[2]; r[11] = m[4]; r[12] = m[5]; r[13] = m[6]; r[14] = m[8]; r[15] = m[9];
        r[16] = m[10]; r[17] = m[12]; r[18] = m[13]; r[19] = m[14];
    }
    return upload_vec(&sc->dResolvedLights, rec.data(), rec.size());
}

static int upload_bvh(incr_scene* sc) {
    int rc;
    if ((rc = upload_vec(&sc->dNodes, sc->bvh.nodes8.data(), sc->bvh.nodes8.size()))) return rc;
    if ((rc = upload_vec(&sc->dTris, sc->bvh.tris.data(), sc->bvh.tris.size()))) return rc;
    if ((rc = upload_vec(&sc->dShade, sc->bvh.shade.data(), sc->bvh.shade.size()))) return rc;
    if ((rc = upload_vec(&sc->dMeshLocalToWorld, (const float*)sc->bvh.meshLocalToWorld.data(), sc->bvh.meshLocalToWorld.size() * 16))) return rc;
    // inputs of the device-side update (bvh_update.hip)
    if ((rc = upload_vec(&sc->dTriObj, sc->bvh.triObj.data(), sc->bvh.triObj.size()))) return rc;
    {   // refit order: the static part + room for the dynamic tail (mesh roots + top nodes: at most topCapacity entries)
        std::vector<uint32_t> order(sc->bvh.refitStaticNodes + sc->bvh.topCapacity, 0u);
        std::copy(sc->bvh.refitNodes.begin(), sc->bvh.refitNodes.end(), order.begin());
        if ((rc = upload_vec(&sc->dLevelNodes, order.data(), order.size()))) return rc;
    }
    std::vector<MeshInstanceInfo> info(sc->host.meshTransforms.size());
    if ((rc = upload_vec(&sc->dMeshInfo, info.data(), info.size()))) return rc;
    if (sc->dNodeBoxes) { (void)hipFree(sc->dNodeBoxes); sc->dNodeBoxes = nullptr; }
    HIP_TRY(hipMalloc(&sc->dNodeBoxes, (sc->bvh.nodes8.size() + 1) * 32));
    return upload_light_grid(sc, nullptr, false);
}

// pt_params.h FastDiv: magic number and shift for exact division of any 32-bit numerator by d >= 2 (d == 1 is passed through)
static FastDiv make_fastdiv(uint32_t d) {
    FastDiv f = {d, 0u, 0u};
    if (d <= 1u) return f;
    const uint32_t fl = 31u - (uint32_t)__builtin_clz(d);
    if ((d & (d - 1u)) == 0u) { f.more = fl - 1u; return f; }              // power of two: ((n - 0) >> 1) >> (log2 d - 1)
    const uint64_t num = 1ull << (32u + fl);
    uint32_t m = (uint32_t)(num / d); const uint32_t rem = (uint32_t)(num - (uint64_t)m * d);
    m += m; const uint32_t twice = rem + rem;
    if (twice >= d || twice < rem) m += 1u;
    f.magic = m + 1u; f.more = fl;
    return f;
}

static void upload_grid_params(const incr_scene* sc, PtParams& P) {
    for (int a = 0; a < 3; ++a) { P.gridLo[a] = sc->grid.lo[a]; P.gridInvCell[a] = sc->grid.invCell[a]; P.gridDim[a] = sc->grid.dim[a]; }
    P.gridCellOffsets = sc->dGridOffsets; P.gridCellLights = sc->dGridLights;
}

// Path-state workspace of the wavefront path tracer: one allocation, carved into SoA arrays (16-byte aligned) per pipeline.
// 656 B of path state (the vertex records of the four bounces side by side) + raysPerPath x 48 B of shadow-ray queue per pixel: 10.2 GB at 4K -- 3.5 % of the 288 GB.
// A pipeline owns a band of 8-pixel tile rows (equal shares; a frame of fewer tile rows than pipelines uses fewer).
// Shadow-ray queue slots per path: the four vertices of a path each ask for at most one ray per light of their light-grid cell plus the sky sample.  The mean over the
// OCCUPIED cells of the scene's grid, times four, plus four of slack, and never fewer than 12 (rounds 2-4 had a fixed 8 per bounce, then a fixed 12 per path: a light-dense
// scene overflowed into the inline path of wf_accumulate_kernel without anyone noticing).  48 B per slot: Sponza-stat (65 lights, 1.3 per occupied cell) takes 14 = 5.6 GB at 4K.
static uint32_t wf_rays_per_path(const incr_context* ctx, const incr_scene* sc) {
    if (ctx->raysPerPath & 0x80000000u) return ctx->raysPerPath & 0x7fffffffu;
    const std::vector<uint32_t>& off = sc->grid.cellOffsets;
    uint64_t lights = 0, occupied = 0;
    const size_t cells = off.size() >= 2 ? off.size() - 2 : 0;          // the last cell lists every light (the fallback for non-finite attenuation)
    for (size_t c = 0; c < cells; ++c) { const uint32_t k = off[c + 1] - off[c]; if (k) { lights += k; ++occupied; } }
    const double mean = occupied ? (double)lights / (double)occupied : 0.0;
    return (uint32_t)std::min(std::max(std::ceil(4.0 * (mean + 1.0)) + 4.0, 12.0), 96.0);
}
static int ensure_wf_workspace(incr_context* ctx, uint32_t W, uint32_t localRows, uint32_t numSamples, uint32_t raysPerPath, hipStream_t stream) {
    const uint32_t tilesX = (W + 7u) / 8u, tilesY = (localRows + 7u) / 8u * numSamples;       // samples in flight: the tile rows of one sample after the other
    // wfPipesWanted == 0: by size -- two pipelines for a frame, four for a framebuffer shard (a 1/8 shard of 4K: 4.80 -> 4.64 ms with
    // four, the full frame 23.2 -> 24.1 ms: the shorter the ray lists, the more of a launch is its longest ray, tools/shard_emulation.py)
    // With the any-hit launches on the side stream (deferred accumulation, pathtrace_wf.hip launcher) ONE pipeline is the fastest issue at every size
    // measured: 4K 17.90 ms with one pipeline, 18.24 with two (same box, profiles/r04_ab_deferred.txt); a 1/8 shard of 4K and the helmet at
    // 1080p are within 1 % either way.  More pipelines also mean more streams than the runtime has hardware queues (helmet 1080p: 1.10 ms with two
    // pipelines of two streams, 1.46 with three, 2.16 with four).  Without the side stream: two for a frame, four for a framebuffer shard, as in rounds 2-3.
    const uint32_t wanted = ctx->wfPipesWanted ? (uint32_t)ctx->wfPipesWanted : (ctx->wfOverlap ? 1u : ((uint64_t)W * localRows * numSamples >= 1500000ull ? 2u : 4u));
    const int pipes = (int)std::max(1u, std::min<uint32_t>(wanted, tilesY / 4u));      // >= 4 tile rows each
    ctx->wfPipes = pipes;
    for (int i = 1; i < pipes; ++i) if (!ctx->wfStreams[i]) {
        HIP_TRY(hipStreamCreateWithFlags(&ctx->wfStreams[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ctx->wfJoin[i], hipEventDisableTiming));
    }
    if (ctx->wfOverlap) for (int i = 0; i < pipes; ++i) if (!ctx->wfSideStreams[i]) {          // bounce overlap: a side stream + a fork / join event pair per bounce
        HIP_TRY(hipStreamCreateWithFlags(&ctx->wfSideStreams[i], hipStreamNonBlocking));
        for (int b = 0; b < 4; ++b) { HIP_TRY(hipEventCreateWithFlags(&ctx->wfBounceFork[i * 4 + b], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ctx->wfBounceJoin[i * 4 + b], hipEventDisableTiming)); }
    }
    uint64_t n[kMaxPipes], rayCap[kMaxPipes], need = 0;
    for (int i = 0; i <= pipes; ++i) ctx->wfTileRow[i] = (uint32_t)((uint64_t)tilesY * i / pipes);
    for (int i = 0; i < pipes; ++i) {
        const uint64_t paths = std::max<uint64_t>((uint64_t)(ctx->wfTileRow[i + 1] - ctx->wfTileRow[i]) * tilesX * 64u, 1);
        n[i] = (paths + 255) & ~255ull;
        rayCap[i] = std::min<uint64_t>(n[i] * raysPerPath, 0xfffffff0ull);
        need += n[i] * (32 * 4 + 16 + 4 * 128) + 65536 + (rayCap[i] + 1) * 48;          // vertex records of the four bounces side by side (deferred accumulation)
    }
    if (ctx->wfBytes < need) {
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 1; i < kMaxPipes; ++i) if (ctx->wfStreams[i]) HIP_TRY(hipStreamSynchronize(ctx->wfStreams[i]));
        for (int i = 0; i < kMaxPipes; ++i) if (ctx->wfSideStreams[i]) HIP_TRY(hipStreamSynchronize(ctx->wfSideStreams[i]));
        if (ctx->wfBlock) { HIP_TRY(hipFree(ctx->wfBlock)); ctx->wfBlock = nullptr; ctx->wfBytes = 0; }
        HIP_TRY(hipMalloc(&ctx->wfBlock, need));
        ctx->wfBytes = need;
    }
    uint8_t* p = (uint8_t*)ctx->wfBlock;
    for (int i = 0; i < pipes; ++i) {
        WfBuffers& w = ctx->wf[i];
        w.counters = (uint32_t*)p; p += 65536;          // 64 words of counters + the striped cursors of the persistent kernels, each on its own cache line
        w.extA = (float4*)p; p += n[i] * 32; w.extB = (float4*)p; p += n[i] * 32; w.stA = (float4*)p; p += n[i] * 32; w.stB = (float4*)p; p += n[i] * 32;
        w.hit = (uint4*)p; p += n[i] * 16; w.vertex = (float4*)p; p += n[i] * 48 * 4; w.vertexFull = (float4*)p; p += n[i] * 80 * 4;
        w.rayQ = (float4*)p; p += (rayCap[i] + 1) * 32; w.rayC = (float4*)p; p += (rayCap[i] + 1) * 16; w.rayCap = (uint32_t)rayCap[i];
        w.numPaths = (uint32_t)n[i];
    }
    return INCR_OK;
}

extern "C" {

int incr_scene_destroy(incr_context* ctx, incr_scene* sc) {
    if (!sc) return INCR_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    void* ptrs[] = {(void*)sc->dAlphaQuads, sc->dNodes, sc->dTris, sc->dShade, sc->dMaterials, sc->dTextures, sc->dMaterialTexels, sc->dMeshLocalToWorld, sc->dGlobalLights, sc->dLightIndices, sc->dLightTransforms,
                    sc->dGridOffsets, sc->dGridLights, sc->dResolvedLights, sc->dTriObj, sc->dLevelNodes, sc->dMeshInfo, sc->dNodeBoxes, sc->sunCover};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (void* p : sc->dTexels) if (p) (void)hipFree(p);
    if (sc->sunCoverBuilt) (void)hipEventDestroy(sc->sunCoverBuilt);
    for (int k = 0; k < 2; ++k) { if (sc->staging.done[k]) { (void)hipEventSynchronize(sc->staging.done[k]); (void)hipEventDestroy(sc->staging.done[k]); } if (sc->staging.host[k]) (void)hipHostFree(sc->staging.host[k]); }
    incr_raster_scene_free(sc);
    delete sc;
    return INCR_OK;
}

int incr_scene_create(incr_context* ctx, const incr_scene_desc* desc, incr_scene** out) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!desc || !out) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    incr_scene* sc = new (std::nothrow) incr_scene();
    if (!sc) return fail(INCR_ERR_OUT_OF_MEMORY, "host allocation failed");
    sc->ctx = ctx;
    struct Lap { const bool on = test_hook_env("INCR_SCENE_TIMING") != nullptr; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                 void operator()(const char* what) { if (!on) return; const auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  incr_scene_create %-26s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; } } lap;
    copy_scene_desc(*desc, sc->host);
    lap("copy of the desc");
    const char* why = "";
    if (!validate_scene(sc->host, &why)) { delete sc; return fail(INCR_ERR_INVALID_ARGUMENT, "scene desc rejected: %s", why); }
    for (uint32_t i = 0; i < desc->num_materials; ++i) {
        const incr_material& m = desc->materials[i];
        const int32_t idx[9] = {m.emissiveTextureIndex, m.baseColorTextureIndex, m.metallicRoughnessTextureIndex, m.normalTextureIndex, m.aoTextureIndex,
                                m.transmissionTextureIndex, m.clearcoatTextureIndex, m.clearcoatRoughnessTextureIndex, m.clearcoatNormalTextureIndex};
        for (int32_t t : idx) if (t < -1 || t >= (int32_t)desc->num_textures) { delete sc; return fail(INCR_ERR_INVALID_ARGUMENT, "material %u: texture index %d out of range", i, t); }
    }
    lap("validation");
    // The acceleration structure is built on the host's cores (scene_build.cpp) WHILE the device builds the texture chains and the raster side is baked: nothing below
    // reads sc->bvh before the join in front of upload_bvh (round 4: Bistro-scale scene creation 0.41 -> 0.31 s, Sponza-stat 0.29 -> 0.17 s; the stages used to run one after the other)
    std::chrono::steady_clock::time_point bvhDone;
    bool bvhFailed = false;          // an exception must not leave the thread (std::terminate): the build allocates ~1 GB of records for a Bistro-scale scene
    std::thread bvhThread([&] { try { build_world_bvh(sc->host, sc->bvh); } catch (...) { bvhFailed = true; } bvhDone = std::chrono::steady_clock::now(); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } bvhJoiner{bvhThread};          // every early return below waits for the builder first
    int rc = INCR_OK;
    if (!rc) rc = upload_vec(&sc->dMaterials, sc->host.materials.data(), sc->host.materials.size());
    if (!rc) rc = upload_vec(&sc->dGlobalLights, sc->host.globalLights.data(), sc->host.globalLights.size());
    if (!rc) rc = upload_vec(&sc->dLightIndices, sc->host.lightIndices.data(), sc->host.lightIndices.size());
    if (!rc) rc = upload_vec(&sc->dLightTransforms, (const float*)sc->host.lightTransforms.data(), sc->host.lightTransforms.size() * 16);
    if (!rc) rc = upload_resolved_lights(sc);
    // Textures: level 0 as handed over + the mip chain the raster passes sample (texture_mips.hip), all levels in one allocation
    std::vector<DeviceTexture> texs(desc->num_textures);
    auto mip_levels = [](uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 4 && h >= 4) { ++n; w >>= 1; h >>= 1; } return n ? n : 1u; };      // scene.cpp:1072-1081
    auto level_dim = [](uint32_t d, uint32_t l) { return std::max(d >> l, 1u); };
    std::vector<uint8_t> haveChain(desc->num_textures, 0);
    for (uint32_t i = 0; i < desc->num_textures && !rc; ++i) {
        const incr_texture_desc& t = desc->textures[i];
        if (!t.texels || !t.width || !t.height || t.format > INCR_TEX_BC5_SNORM) { rc = fail(INCR_ERR_INVALID_ARGUMENT, "texture %u invalid", i); break; }
        if (t.format >= INCR_TEX_BC3_SRGB) {
            // A block-compressed chain (what the reference uploads after DirectX::Compress): every level is decoded into the row-major
            // words of the chain buffer (texture_mips.hip bc_decode_kernel); nothing is generated or filtered for such a texture.
            uint32_t full = 1; while ((t.width >> full) || (t.height >> full)) ++full;            // levels down to 1 x 1
            if (!t.mip_levels || t.mip_levels > full) { rc = fail(INCR_ERR_INVALID_ARGUMENT, "texture %u: %u mip levels for a %u x %u block-compressed chain", i, t.mip_levels, t.width, t.height); break; }
            const size_t blockBytes = t.format == INCR_TEX_BC4_UNORM ? 8 : 16;
            size_t chainTexels = 0, chainBlocks = 0;
            for (uint32_t l = 0; l < t.mip_levels; ++l) {
                const uint32_t w = level_dim(t.width, l), h = level_dim(t.height, l);
                chainTexels += (size_t)w * h; chainBlocks += (size_t)((w + 3) / 4) * ((h + 3) / 4);
            }
            void* d = nullptr; void* blocks = nullptr;
            hipError_t e = hipMalloc(&d, chainTexels * 4);
            if (e == hipSuccess) e = hipMalloc(&blocks, chainBlocks * blockBytes);
            if (e == hipSuccess) e = hipMemcpy(blocks, t.texels, chainBlocks * blockBytes, hipMemcpyHostToDevice);
            const uint8_t* src = (const uint8_t*)blocks; uint32_t* dst = (uint32_t*)d;
            for (uint32_t l = 0; l < t.mip_levels && e == hipSuccess; ++l) {
                const uint32_t w = level_dim(t.width, l), h = level_dim(t.height, l);
                e = incr_launch_bc_decode(src, w, h, t.format, dst, nullptr);
                src += (size_t)((w + 3) / 4) * ((h + 3) / 4) * blockBytes; dst += (size_t)w * h;
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (blocks) (void)hipFree(blocks);
            if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "block-compressed texture decode failed: %s", hipGetErrorString(e)); break; }
            sc->dTexels.push_back(d);
            sc->textureBytes += chainTexels * 4;
            texs[i].texels = (const uint8_t*)d; texs[i].width = t.width; texs[i].height = t.height;
            texs[i].format = t.format == INCR_TEX_BC3_SRGB ? (uint32_t)INCR_TEX_RGBA8_SRGB : (t.format == INCR_TEX_BC5_SNORM ? (uint32_t)INCR_TEX_RG16S_OVER_4445 : (uint32_t)INCR_TEX_RG16_OVER_8925);
            texs[i].pad[0] = texs[i].pad[2] = 0; texs[i].pad[1] = t.mip_levels;
            if (t.format == INCR_TEX_BC3_SRGB) {
                // pad[0]: every alpha of EVERY level decodes to 255 (the raster passes then drop the alpha clip).  A handed-in chain's lower
                // levels are independent blocks, so all of them are inspected.  Palette of a BC4 alpha block (D3D11.3 functional spec 19.5.4):
                // index 0 / 1 = the endpoints; a0 > a1: six interpolants; a0 <= a1: four interpolants, index 6 = 0, index 7 = 255 -- so a block
                // with both endpoints 255 is in the second mode and index 6 still selects 0.
                size_t nBlocks = 0;
                for (uint32_t l = 0; l < t.mip_levels; ++l) nBlocks += (size_t)((level_dim(t.width, l) + 3) / 4) * ((level_dim(t.height, l) + 3) / 4);
                bool opaque = true;
                for (size_t b = 0; b < nBlocks && opaque; ++b) {
                    const uint8_t* blk = t.texels + b * 16;
                    const uint32_t a0 = blk[0], a1 = blk[1];
                    uint64_t bits = 0; for (int k = 0; k < 6; ++k) bits |= (uint64_t)blk[2 + k] << (8 * k);
                    for (int px = 0; px < 16 && opaque; ++px) {
                        const uint32_t idx = (uint32_t)(bits >> (3 * px)) & 7u;
                        bool is255;
                        if (idx == 0) is255 = a0 == 255;
                        else if (idx == 1) is255 = a1 == 255;
                        else if (a0 > a1) is255 = false;                                       // interpolants between 255 and something smaller
                        else is255 = idx == 7 || (idx < 6 && a0 == 255);                          // a0 <= a1: a0 == 255 means a1 == 255 too, interpolants 255; index 6 = 0
                        opaque = is255;
                    }
                }
                texs[i].pad[0] = opaque ? 1u : 0u;
            }
            haveChain[i] = 1;
            continue;
        }
        const size_t bytes = (size_t)t.width * t.height * 4;
        const uint32_t mips = mip_levels(t.width, t.height);
        size_t chainTexels = 0;
        for (uint32_t l = 0; l < mips; ++l) chainTexels += (size_t)level_dim(t.width, l) * level_dim(t.height, l);
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, chainTexels * 4);
        if (e == hipSuccess) e = hipMemcpy(d, t.texels, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "texture upload failed: %s", hipGetErrorString(e)); break; }
        sc->dTexels.push_back(d);
        sc->textureBytes += chainTexels * 4;
        texs[i].texels = (const uint8_t*)d; texs[i].width = t.width; texs[i].height = t.height; texs[i].format = t.format;
        texs[i].pad[0] = texs[i].pad[1] = texs[i].pad[2] = 0;
        // pad[0] = 1: every alpha texel is 255, so a bilinear alpha sample is 1 up to rounding and can never fail the
        // 0.5 alpha clip of the visibility pass (raster.hip skips the sample for such textures); the mips of such a texture are opaque too
        bool opaque = true;
        for (size_t k = 3; k < bytes && opaque; k += 4) opaque = t.texels[k] == 255;
        texs[i].pad[0] = opaque ? 1u : 0u;
        texs[i].pad[1] = mips;
    }
    // vMF-filtered chains for the normal + metallic-roughness pair of a material (scene.cpp:1157-1320), box chains for the rest.
    // The pairs are independent: their prefilters run on four streams (a pair's top levels are a few long serial sums,
    // content_pipeline.hip), one synchronisation at the end.
    struct VmfJob { void* tmpN; void* tmpM; };
    std::vector<VmfJob> vmfJobs;
    hipStream_t vmfStreams[4] = {nullptr, nullptr, nullptr, nullptr};
    for (hipStream_t& st : vmfStreams) if (!rc && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) rc = fail(INCR_ERR_HIP, "stream creation failed");
    if (!rc) { const hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) rc = fail(INCR_ERR_HIP, "texture upload failed: %s", hipGetErrorString(e)); }      // level 0 is in place before the side streams read it
    for (uint32_t mi = 0; mi < desc->num_materials && !rc; ++mi) {
        const int32_t nt = desc->materials[mi].normalTextureIndex, mt = desc->materials[mi].metallicRoughnessTextureIndex;
        if (nt < 0 || mt < 0 || nt == mt || haveChain[(size_t)nt] || haveChain[(size_t)mt]) continue;
        const DeviceTexture& N = texs[(size_t)nt]; const DeviceTexture& M = texs[(size_t)mt];
        if (N.width != M.width || N.height != M.height || N.pad[1] < 2 || N.format != INCR_TEX_RG_SNORM || M.format != INCR_TEX_BGRA8_SWIZZLE_RG) continue;
        if ((N.width >> (N.pad[1] - 1)) << (N.pad[1] - 1) != N.width || (N.height >> (N.pad[1] - 1)) << (N.pad[1] - 1) != N.height) continue;      // 2x2 footprints need even levels
        size_t chainTexels = 0;
        for (uint32_t l = 0; l < N.pad[1]; ++l) chainTexels += (size_t)(N.width >> l) * (N.height >> l);
        VmfJob job = {nullptr, nullptr};
        hipStream_t st = vmfStreams[vmfJobs.size() & 3u];
        hipError_t e = hipMalloc(&job.tmpN, chainTexels * 4);
        if (e == hipSuccess) e = hipMalloc(&job.tmpM, chainTexels * 4);
        vmfJobs.push_back(job);
        if (e == hipSuccess) {
            incr_prefilter_normal_roughness_desc pd = {};
            pd.normalmap = N.texels; pd.metallic_roughness = M.texels; pd.width = N.width; pd.height = N.height; pd.num_mips = N.pad[1];
            pd.out_normalmap = (uint8_t*)job.tmpN; pd.out_metallic_roughness = (uint8_t*)job.tmpM;
            pd.reserved = 1;        // exact summation order: the chain equals the oracle's bit for bit
            rc = incr_prefilter_normal_roughness(ctx, &pd, (incr_stream)st);
            const size_t first = (size_t)N.width * N.height;
            if (!rc) e = incr_launch_mip_adopt_vmf((const uint32_t*)job.tmpN, (const uint32_t*)job.tmpM, (uint32_t*)N.texels, (uint32_t*)M.texels, first, chainTexels - first, st);
        }
        if (!rc && e != hipSuccess) rc = fail(INCR_ERR_HIP, "vMF mip chain failed: %s", hipGetErrorString(e));
        haveChain[(size_t)nt] = haveChain[(size_t)mt] = 1;
    }
    { const hipError_t e = hipDeviceSynchronize(); if (!rc && e != hipSuccess) rc = fail(INCR_ERR_HIP, "vMF mip chain failed: %s", hipGetErrorString(e)); }
    for (const VmfJob& j : vmfJobs) { if (j.tmpN) (void)hipFree(j.tmpN); if (j.tmpM) (void)hipFree(j.tmpM); }
    for (hipStream_t st : vmfStreams) if (st) (void)hipStreamDestroy(st);
    for (uint32_t i = 0; i < desc->num_textures && !rc; ++i) {
        if (haveChain[i]) continue;
        const DeviceTexture& T = texs[i];
        uint32_t* level = (uint32_t*)T.texels;
        for (uint32_t l = 0; l + 1 < T.pad[1]; ++l) {
            const uint32_t sw = level_dim(T.width, l), sh = level_dim(T.height, l), dw = level_dim(T.width, l + 1), dh = level_dim(T.height, l + 1);
            uint32_t* next = level + (size_t)sw * sh;
            const hipError_t e = incr_launch_mip_box(level, sw, sh, next, dw, dh, T.format, ctx->srgbLut, nullptr);
            if (e != hipSuccess) { rc = fail(INCR_ERR_HIP, "mip generation failed: %s", hipGetErrorString(e)); break; }
            level = next;
        }
    }
    // Interleaved level-0 texels per material for the path tracer's chsMain (device_types.h MaterialTexels), from the row-major level 0
    std::vector<MaterialTexels> mtex(desc->num_materials, MaterialTexels{nullptr, 0u, 0u, 0u, 0u, nullptr});
    const bool fuseMaterials = !(test_hook_env("INCR_MATERIAL_TEXELS") && atoi(test_hook_env("INCR_MATERIAL_TEXELS")) == 0);      // A/B hook
    for (uint32_t mi = 0; mi < desc->num_materials && !rc && fuseMaterials; ++mi) {
        const int32_t bt = desc->materials[mi].baseColorTextureIndex, nt = desc->materials[mi].normalTextureIndex, mt = desc->materials[mi].metallicRoughnessTextureIndex;
        if (bt < 0 || nt < 0 || mt < 0) continue;
        const DeviceTexture& Bc = texs[(size_t)bt]; const DeviceTexture& N = texs[(size_t)nt]; const DeviceTexture& M = texs[(size_t)mt];
        if (Bc.width != N.width || Bc.width != M.width || Bc.height != N.height || Bc.height != M.height) continue;
        if (N.format != INCR_TEX_RG_SNORM || M.format != INCR_TEX_BGRA8_SWIZZLE_RG || (Bc.format != INCR_TEX_RGBA8_SRGB && Bc.format != INCR_TEX_RGBA8_UNORM)) continue;
        bool shared = false;          // materials that bind the same three textures share one interleaved image
        for (uint32_t mj = 0; mj < mi && !shared; ++mj)
            if (mtex[mj].texels && desc->materials[mj].baseColorTextureIndex == bt && desc->materials[mj].normalTextureIndex == nt && desc->materials[mj].metallicRoughnessTextureIndex == mt) { mtex[mi] = mtex[mj]; shared = true; }
        if (shared) continue;
        const uint64_t n = material_texel_count(Bc.width, Bc.height);
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, n * 8);
        if (e == hipSuccess) e = hipMemset(d, 0, n * 8);
        if (e == hipSuccess) e = incr_launch_material_texels((const uint32_t*)Bc.texels, (const uint32_t*)N.texels, (const uint32_t*)M.texels, Bc.width, Bc.height, d, nullptr);
        if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "material texel interleave failed: %s", hipGetErrorString(e)); break; }
        sc->dTexels.push_back(d);
        sc->textureBytes += n * 8;
        mtex[mi] = MaterialTexels{d, Bc.width, Bc.height, Bc.format, 0u, nullptr};
    }
    // The chains above are row-major (what the generators write); the samplers read a TILED copy.  The row-major chains are freed
    // afterwards (incr_scene_texture_download de-tiles on demand): a texture is resident once, not twice.
    std::vector<void*> linear(desc->num_textures, nullptr);
    for (uint32_t i = 0; i < desc->num_textures && i < texs.size(); ++i) linear[i] = (void*)texs[i].texels;
    for (uint32_t i = 0; i < desc->num_textures && !rc; ++i) {
        DeviceTexture& T = texs[i];
        size_t tiledWords = 0;
        for (uint32_t l = 0; l < T.pad[1]; ++l) tiledWords += texel_level_words(level_dim(T.width, l), level_dim(T.height, l));
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, tiledWords * 4);
        if (e == hipSuccess) e = hipMemset(d, 0, tiledWords * 4);
        if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "tiled texture allocation failed: %s", hipGetErrorString(e)); break; }
        sc->dTexels.push_back(d);
        sc->textureBytes += tiledWords * 4;
        const uint32_t* src = (const uint32_t*)T.texels; uint32_t* dst = (uint32_t*)d;
        for (uint32_t l = 0; l < T.pad[1] && e == hipSuccess; ++l) {
            const uint32_t w = level_dim(T.width, l), h = level_dim(T.height, l);
            e = incr_launch_mip_tile(src, w, h, dst, nullptr);
            src += (size_t)w * h; dst += texel_level_words(w, h);
        }
        if (e != hipSuccess) { rc = fail(INCR_ERR_HIP, "texture tiling failed: %s", hipGetErrorString(e)); break; }
        T.texels = (const uint8_t*)d;
    }
    // The same interleave for the whole TILED chain (MaterialTexels::chain): the raster passes' Sample of a material's three maps reads it
    const bool fuseChains = !(test_hook_env("INCR_MATERIAL_CHAINS") && atoi(test_hook_env("INCR_MATERIAL_CHAINS")) == 0);      // A/B hook
    for (uint32_t mi = 0; mi < desc->num_materials && !rc && fuseChains; ++mi) {
        if (!mtex[mi].texels || mtex[mi].baseFormat != INCR_TEX_RGBA8_SRGB) continue;
        const int32_t bt = desc->materials[mi].baseColorTextureIndex, nt = desc->materials[mi].normalTextureIndex, mt = desc->materials[mi].metallicRoughnessTextureIndex;
        const DeviceTexture& Bc = texs[(size_t)bt]; const DeviceTexture& N = texs[(size_t)nt]; const DeviceTexture& M = texs[(size_t)mt];
        if (Bc.pad[1] != N.pad[1] || Bc.pad[1] != M.pad[1] || !Bc.pad[1]) continue;
        bool shared = false;
        for (uint32_t mj = 0; mj < mi && !shared; ++mj)
            if (mtex[mj].chain && mtex[mj].texels == mtex[mi].texels) { mtex[mi].chain = mtex[mj].chain; shared = true; }
        if (shared) continue;
        uint64_t words = 0;
        for (uint32_t l = 0; l < Bc.pad[1]; ++l) words += texel_level_words(level_dim(Bc.width, l), level_dim(Bc.height, l));
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, words * 8);
        if (e == hipSuccess) e = incr_launch_material_chain((const uint32_t*)Bc.texels, (const uint32_t*)N.texels, (const uint32_t*)M.texels, words, d, nullptr);
        if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "material chain interleave failed: %s", hipGetErrorString(e)); break; }
        sc->dTexels.push_back(d);
        sc->textureBytes += words * 8;
        mtex[mi].chain = d;
    }
    if (!rc) rc = upload_vec(&sc->dMaterialTexels, mtex.data(), mtex.size());
    // Alpha quads of the textures the visibility scan alpha-tests against (RGBA8 with an alpha that is not 255 everywhere)
    std::vector<const uint32_t*> quads(desc->num_textures, nullptr);
    const bool alphaQuads = !(test_hook_env("INCR_ALPHA_QUADS") && atoi(test_hook_env("INCR_ALPHA_QUADS")) == 0);      // A/B hook
    for (uint32_t i = 0; i < desc->num_textures && !rc && alphaQuads; ++i) {
        const DeviceTexture& T = texs[i];
        if ((T.format != INCR_TEX_RGBA8_SRGB && T.format != INCR_TEX_RGBA8_UNORM) || T.pad[0] || !T.pad[1]) continue;
        size_t words = 0;
        for (uint32_t l = 0; l < T.pad[1]; ++l) words += texel_level_words(level_dim(T.width, l), level_dim(T.height, l));
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, words * 4);
        if (e == hipSuccess) e = hipMemset(d, 0, words * 4);
        const uint32_t* src = (const uint32_t*)T.texels; uint32_t* dst = (uint32_t*)d;
        for (uint32_t l = 0; l < T.pad[1] && e == hipSuccess; ++l) {
            const uint32_t w = level_dim(T.width, l), h = level_dim(T.height, l);
            e = incr_launch_alpha_quads(src, w, h, dst, nullptr);
            src += texel_level_words(w, h); dst += texel_level_words(w, h);
        }
        if (e != hipSuccess) { if (d) (void)hipFree(d); rc = fail(INCR_ERR_HIP, "alpha quad chain failed: %s", hipGetErrorString(e)); break; }
        sc->dTexels.push_back(d);
        sc->textureBytes += words * 4;
        quads[i] = (const uint32_t*)d;
    }
    if (!rc) rc = upload_vec(&sc->dAlphaQuads, quads.data(), quads.size());
    if (!rc) { const hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) rc = fail(INCR_ERR_HIP, "mip generation failed: %s", hipGetErrorString(e)); }
    if (!rc) for (void* lp : linear) {           // the tiled copies are complete: drop the row-major chains
        if (!lp) continue;
        for (void*& q : sc->dTexels) if (q == lp) { (void)hipFree(q); q = nullptr; break; }
    }
    if (!rc) for (uint32_t i = 0; i < desc->num_textures; ++i) {
        size_t chainTexels = 0;
        for (uint32_t l = 0; l < texs[i].pad[1]; ++l) chainTexels += (size_t)level_dim(texs[i].width, l) * level_dim(texs[i].height, l);
        sc->textureBytes -= chainTexels * 4;
    }
    sc->hostTextures = texs;
    if (!rc) rc = upload_vec(&sc->dTextures, texs.data(), texs.size());
    lap("textures");
    if (!rc) rc = incr_raster_scene_upload(sc);
    lap("raster-side bake + upload");
    bvhThread.join();
    if (lap.on) fprintf(stderr, "  incr_scene_create %-26s done %.1f ms before the join\n", "build_world_bvh (thread)", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - bvhDone).count());
    lap("wait for build_world_bvh");
    if (bvhFailed) { incr_scene_destroy(ctx, sc); return fail(INCR_ERR_OUT_OF_MEMORY, "acceleration-structure build failed (host allocation)"); }
    if (!rc && (sc->bvh.maxDepth > kMaxBvhDepth || sc->bvh.nodes8.size() > kMaxBvhNodes || sc->bvh.tris.size() > kMaxTriRecords)) {
        const uint32_t d = sc->bvh.maxDepth; const size_t nn = sc->bvh.nodes8.size(), nt = sc->bvh.tris.size();
        incr_scene_destroy(ctx, sc);
        return fail(INCR_ERR_INVALID_ARGUMENT, "acceleration structure outside the traversal limits: depth %u (max %u), %zu nodes (max %u), %zu triangle records (max %u)", d, kMaxBvhDepth, nn, kMaxBvhNodes, nt, kMaxTriRecords);
    }
    if (!rc) rc = upload_bvh(sc);
    lap("upload of the tree");
    if (rc) { incr_scene_destroy(ctx, sc); return rc; }
    *out = sc;
    return INCR_OK;
}

int incr_update_bvh(incr_context* ctx, incr_scene* sc, const incr_matrix* mesh_transforms, const uint32_t* mesh_visibility,
                    const incr_matrix* root_transform, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc) return fail(INCR_ERR_INVALID_ARGUMENT, "scene is NULL");
    bool dirty = false;
    const size_t nm = sc->host.meshTransforms.size();
    if (mesh_transforms)
        for (size_t i = 0; i < nm; ++i) if (memcmp(sc->host.meshTransforms[i].m, mesh_transforms[i].m, 64)) { sc->host.meshTransforms[i] = m44_from(mesh_transforms[i]); dirty = true; }
    if (mesh_visibility)
        for (size_t i = 0; i < nm; ++i) if (sc->host.meshVisibility[i] != mesh_visibility[i]) { sc->host.meshVisibility[i] = mesh_visibility[i]; dirty = true; }
    if (root_transform && memcmp(sc->host.rootTransform.m, root_transform->m, 64)) { sc->host.rootTransform = m44_from(*root_transform); dirty = true; }
    if (!dirty) return INCR_OK;          // the reference rebuilds its TLAS every frame; ours only when an input changed
    ++sc->geometryEpoch;                 // what is derived from the world-space triangles (the sun cover map) is stale
    // The device work below is ordered on the caller's stream: small uploads (per-mesh matrices and instance ids, the rebuilt top
    // tree, the light grid), then the re-transform and refit kernels of bvh_update.hip.  The CALL itself does host work -- the top
    // tree and the light grid are rebuilt on the host, and taking a staging slot may wait for the slot's previous use
    // (hipEventSynchronize) or grow it (hipHostMalloc) -- so it must not be issued on a capturing stream.
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<MeshInstanceInfo>& info = sc->meshInfo;
    refresh_mesh_instances(sc->host, sc->bvh, info);
    build_light_grid(sc->host, sc->bvh, sc->grid);
    // every host -> device byte of this update goes through one pinned, double-buffered staging slot (api_internal.h UpdateStaging)
    int rc = incr_staging_begin(sc, nm * (64 + sizeof(MeshInstanceInfo) + 4) + (sc->grid.cellOffsets.size() + sc->grid.cellLights.size()) * 4 + (size_t)sc->numDecals * 64
                                    + (size_t)sc->bvh.topCapacity * (sizeof(BvhNode8) + 4) + 1024);
    if (rc) return rc;
    // whatever happens below, the slot's event is recorded behind the copies already enqueued: a slot without it could be rewritten
    // by the next update while a hipMemcpyAsync still reads it
    struct StagingGuard { incr_scene* sc; hipStream_t st; bool armed = true; ~StagingGuard() { if (armed) (void)incr_staging_end(sc, st); } } guard{sc, st};
    if (nm) {
        if ((rc = incr_staging_upload(sc, sc->dMeshLocalToWorld, sc->bvh.meshLocalToWorld.data(), nm * 64, st))) return rc;
        if ((rc = incr_staging_upload(sc, sc->dMeshInfo, info.data(), nm * sizeof(MeshInstanceInfo), st))) return rc;
    }
    // the reference rebuilds its TLAS here (update-tlas.inl:24-89): the top tree over the meshes' new boxes (conservative: the eight
    // corners of every object-space box; the refit below tightens them), a fraction of a millisecond on the host for a thousand meshes
    {
        std::vector<float> meshBoxes(nm * 6, 0.f);
        for (size_t m = 0; m < nm; ++m) {
            float lo[3] = {1.f, 1.f, 1.f}, hi[3] = {-1.f, -1.f, -1.f};
            const float* b = sc->bvh.meshObjBox.size() == nm * 6 ? &sc->bvh.meshObjBox[m * 6] : nullptr;
            if (b && sc->host.meshVisibility[m] && b[0] <= b[3]) {
                for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
                for (int c = 0; c < 8; ++c) {
                    const V3 p = m44_transform_point(v3(b[(c & 1) ? 3 : 0], b[(c & 2) ? 4 : 1], b[(c & 4) ? 5 : 2]), sc->bvh.meshLocalToWorld[m]);
                    const float q[3] = {p.x, p.y, p.z};
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
                }
                for (int a = 0; a < 3; ++a) if (!(lo[a] >= -FLT_MAX && hi[a] <= FLT_MAX)) { lo[0] = lo[1] = lo[2] = 1.f; hi[0] = hi[1] = hi[2] = -1.f; break; }
            }
            for (int a = 0; a < 3; ++a) { meshBoxes[m * 6 + a] = lo[a]; meshBoxes[m * 6 + 3 + a] = hi[a]; }
        }
        rebuild_top_tree(sc->bvh, meshBoxes.data());
        if ((rc = incr_staging_upload(sc, sc->dNodes, sc->bvh.nodes8.data(), (size_t)sc->bvh.topCapacity * sizeof(BvhNode8), st))) return rc;
        const size_t dyn = sc->bvh.refitNodes.size() - sc->bvh.refitStaticNodes;
        if ((rc = incr_staging_upload(sc, sc->dLevelNodes + sc->bvh.refitStaticNodes, sc->bvh.refitNodes.data() + sc->bvh.refitStaticNodes, dyn * 4, st))) return rc;
    }
    HIP_TRY(incr_launch_bvh_update((uint32_t)sc->bvh.tris.size(), sc->dTriObj, sc->dMeshLocalToWorld, sc->dMeshInfo, sc->dTris, sc->dShade, sc->dNodes,
                                   sc->dLevelNodes, sc->bvh.refitRanges.data(), (uint32_t)sc->bvh.refitRanges.size() / 2u, sc->dNodeBoxes, st));
    rc = upload_light_grid(sc, st, true);
    if (!rc) rc = incr_raster_scene_update(sc, st);       // raster side: mesh visibility flags, decal transforms
    guard.armed = false;
    const int rc2 = incr_staging_end(sc, st);
    return rc ? rc : rc2;
}

int incr_scene_set_sun_dir(incr_context* ctx, incr_scene* sc, const float sun_dir[3]) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc || !sun_dir) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    memcpy(sc->host.sunDir, sun_dir, 12);
    return INCR_OK;
}

int incr_scene_set_max_anisotropy(incr_context* ctx, incr_scene* sc, uint32_t max_anisotropy) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (max_anisotropy > 16u) return fail(INCR_ERR_INVALID_ARGUMENT, "max_anisotropy %u: D3D12_REQ_MAXANISOTROPY is 16", max_anisotropy);
    sc->host.maxAnisotropy = max_anisotropy ? max_anisotropy : 16u;
    return INCR_OK;
}

int incr_scene_texture_info(incr_context* ctx, const incr_scene* sc, uint32_t ti, uint32_t* width, uint32_t* height, uint32_t* num_mips, uint64_t* chain_texels) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc || ti >= sc->hostTextures.size()) return fail(INCR_ERR_INVALID_ARGUMENT, "texture index out of range");
    const DeviceTexture& t = sc->hostTextures[ti];
    uint64_t n = 0; uint32_t w = t.width, h = t.height;
    for (uint32_t l = 0; l < t.pad[1]; ++l) { n += (uint64_t)w * h; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
    if (width) *width = t.width; if (height) *height = t.height; if (num_mips) *num_mips = t.pad[1]; if (chain_texels) *chain_texels = n;
    return INCR_OK;
}
// The row-major chain (level 0 first), de-tiled on demand from the storage the samplers read
int incr_scene_texture_download(incr_context* ctx, const incr_scene* sc, uint32_t ti, void* dst, uint64_t bytes) {
    uint64_t n = 0;
    int rc = incr_scene_texture_info(ctx, sc, ti, nullptr, nullptr, nullptr, &n);
    if (rc) return rc;
    if (!dst || bytes != n * 4) return fail(INCR_ERR_INVALID_ARGUMENT, "destination must hold the whole chain (%llu bytes)", (unsigned long long)(n * 4));
    const DeviceTexture& t = sc->hostTextures[ti];
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, n * 4));
    const uint32_t* src = (const uint32_t*)t.texels; uint32_t* out = (uint32_t*)tmp;
    uint32_t w = t.width, h = t.height;
    hipError_t e = hipSuccess;
    for (uint32_t l = 0; l < t.pad[1] && e == hipSuccess; ++l) {
        e = incr_launch_mip_untile(src, w, h, out, nullptr);
        src += texel_level_words(w, h); out += (size_t)w * h;
        w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
    }
    if (e == hipSuccess) e = hipMemcpy(dst, tmp, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(INCR_ERR_HIP, "texture download failed: %s", hipGetErrorString(e));
    return INCR_OK;
}

int incr_scene_get_info(incr_context* ctx, const incr_scene* sc, incr_scene_info* info) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc || !info) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    info->num_triangles = sc->bvh.numTriangles;
    info->num_bvh_nodes = sc->bvh.nodes8.size();
    info->bvh_bytes = sc->bvh.nodes8.size() * sizeof(BvhNode8);
    info->triangle_bytes = sc->bvh.tris.size() * sizeof(TriIsect);
    info->shading_bytes = sc->bvh.shade.size() * sizeof(TriShade);
    info->texture_bytes = sc->textureBytes;
    info->num_instances = sc->bvh.numInstances;
    info->bvh_max_depth = sc->bvh.maxDepth;
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- sharding
uint32_t incr_shard_local_rows(uint32_t res_y, const incr_shard* sh) {
    if (!sh || sh->row_block_height == 0) return res_y;
    const uint32_t nblk = (res_y + sh->row_block_height - 1) / sh->row_block_height;
    const uint32_t stride = sh->row_block_stride ? sh->row_block_stride : 1;
    const uint32_t mine = (nblk + stride - 1) / stride;     // same for every rank (equal-size all_gather); surplus rows are padding
    return mine * sh->row_block_height;
}

// ---------------------------------------------------------------------------------------------- passes
// Per-frame Preetham sun terms (preetham.hlsli:53-89): depend only on the sun direction and turbidity.
static float saturate_h(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
} // extern "C"
void incr_perez_from_turbidity(float t, incr_perez_distribution* p);
void incr_sky_sun_terms(const float s[3], float t, const incr_perez_distribution& pc, incr::SkySun* out);
static void perez_from_turbidity(float t, incr_perez_distribution* p) { incr_perez_from_turbidity(t, p); }
static void sky_sun_terms(const float s[3], float t, const incr_perez_distribution& pc, SkySun* out) { incr_sky_sun_terms(s, t, pc, out); }
void incr_perez_from_turbidity(float t, incr_perez_distribution* p) {
    // render-jobs/path-tracing.inl:134-140 (double literals * float t, rounded by the Vector4 constructor)
    const double A[3] = {0.1787 * t - 1.4630, -0.0193 * t - 0.2592, -0.0167 * t - 0.2608};
    const double B[3] = {-0.3554 * t + 0.4275, -0.0665 * t + 0.0008, -0.0950 * t + 0.0092};
    const double C[3] = {-0.0227 * t + 5.3251, -0.0004 * t + 0.2125, -0.0079 * t + 0.2102};
    const double D[3] = {0.1206 * t - 2.5771, -0.0641 * t - 0.8989, -0.0441 * t - 1.6537};
    const double E[3] = {-0.0670 * t + 0.3703, -0.0033 * t + 0.0452, -0.0109 * t + 0.0529};
    for (int i = 0; i < 3; ++i) { p->A[i] = (float)A[i]; p->B[i] = (float)B[i]; p->C[i] = (float)C[i]; p->D[i] = (float)D[i]; p->E[i] = (float)E[i]; }
    p->A[3] = p->B[3] = p->C[3] = p->D[3] = p->E[3] = 0.f;
}
void incr_sky_sun_terms(const float s[3], float t, const incr_perez_distribution& pc, SkySun* out) {
    const float cosS = saturate_h(s[1]);
    const float thetaS = acosf(cosS);
    const float k_Pi = 3.14159265358979323846f;
    const float chi = (4.f / 9.f - t / 120.f) * (k_Pi - 2.f * thetaS);
    const float Yz = (4.0453f * t - 4.9710f) * tanf(chi) - 0.2155f * t + 2.4192f;
    const float theta2 = thetaS * thetaS, theta3 = theta2 * thetaS, T = t, T2 = t * t;
    const float xz = (0.00165f * theta3 - 0.00375f * theta2 + 0.00209f * thetaS + 0.0f) * T2 +
                     (-0.02903f * theta3 + 0.06377f * theta2 - 0.03202f * thetaS + 0.00394f) * T +
                     (0.11693f * theta3 - 0.21196f * theta2 + 0.06052f * thetaS + 0.25886f);
    const float yz = (0.00275f * theta3 - 0.00610f * theta2 + 0.00317f * thetaS + 0.0f) * T2 +
                     (-0.04214f * theta3 + 0.08970f * theta2 - 0.04153f * thetaS + 0.00516f) * T +
                     (0.15346f * theta3 - 0.26756f * theta2 + 0.06670f * thetaS + 0.26688f);
    out->s[0] = s[0]; out->s[1] = s[1]; out->s[2] = s[2];
    out->YxyZenith[0] = Yz; out->YxyZenith[1] = xz; out->YxyZenith[2] = yz;
    // Perez(0, thetaS): cos(0) = 1, gamma = thetaS, cos(gamma) = cosS
    for (int i = 0; i < 3; ++i) {
        const float e1 = expf(pc.B[i] / 1.f), e2 = expf(pc.D[i] * thetaS);
        out->F0[i] = (1.0f + pc.A[i] * e1) * ((1.0f + pc.C[i] * e2) + (pc.E[i] * cosS) * cosS);
    }
    out->pad[0] = out->pad[1] = out->pad[2] = 0.f;
}
extern "C" {

int incr_pathtrace(incr_context* ctx, const incr_pathtrace_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    if (!d->res_x || !d->res_y) return fail(INCR_ERR_INVALID_ARGUMENT, "zero resolution");
    if ((uint64_t)d->res_x * d->res_y > 0xffffffffull) return fail(INCR_ERR_INVALID_ARGUMENT, "resolution overflows the 32-bit pixel index of the sampler");
    if (d->env_sky_mode != 0 && d->env_sky_mode != 1) return fail(INCR_ERR_INVALID_ARGUMENT, "env_sky_mode %d unsupported (0 = HDRI cube, 1 = dynamic sky)", d->env_sky_mode);
    if (d->env_sky_mode == 0 && !d->envmap_r11g11b10) return fail(INCR_ERR_INVALID_ARGUMENT, "env_sky_mode 0 (HDRI) needs envmap_r11g11b10: the miss shader samples the cube");
    // FConfig::Viewmode compiles pathtracing.hlsl with any VIEWMODE (path-tracing.inl:31-35): 1 lighting only, 2-5 material channels at the
    // first hit, 7 a mirror bounce at the first hit; 0, 6 (NaN check: a tonemap-side view) and 8-13 are named by no #if: the normal path
    if (d->viewmode < 0 || d->viewmode > 13) return fail(INCR_ERR_INVALID_ARGUMENT, "viewmode %d unsupported (0..13)", d->viewmode);
    if (d->max_sample_count < 1) return fail(INCR_ERR_INVALID_ARGUMENT, "max_sample_count must be >= 1");
    if (d->debug_hit_chain && d->num_samples > 1u) return fail(INCR_ERR_INVALID_ARGUMENT, "debug_hit_chain holds one chain per local pixel: num_samples must be <= 1 (got %u)", d->num_samples);
    if (d->shard.row_block_height && !d->shard.row_block_stride) return fail(INCR_ERR_INVALID_ARGUMENT, "row_block_stride must be >= 1");
    const incr_scene* sc = d->scene;
    PtParams P; fill_scene_params(ctx, sc, P);
    const M44 V = m44_from(d->view.view_transform), Pm = m44_from(d->view.projection_transform);
    const M44 p2w = m44_inverse(m44_mul(V, Pm)), cam = m44_inverse(V);              // :199, :201
    memcpy(P.projectionToWorld, p2w.m, 64); memcpy(P.cameraMatrix, cam.m, 64);
    P.W = d->res_x; P.H = d->res_y; P.numPixels = d->res_x * d->res_y;
    P.sampleIdx = d->current_sample_index;
    P.sqrtSampleCount = (uint32_t)std::sqrt((double)d->max_sample_count);             // :207
    P.divW = make_fastdiv(P.W); P.divSqrtN = make_fastdiv(P.sqrtSampleCount); P.divN = make_fastdiv(P.sqrtSampleCount * P.sqrtSampleCount);
    P.aperture = d->camera_aperture; P.focalLength = d->camera_focal_length; P.skyBrightness = d->sky_brightness;
    P.directLighting = d->direct_lighting; P.viewmode = d->viewmode;
    perez_from_turbidity(d->turbidity, &P.perez);
    // sun direction of the miss shader (:181-189): first directional scene light, else (1, 0.1, 1)
    V3 L = v3(1.f, 0.1f, 1.f);
    int sun = -1;
    for (size_t i = 0; i < sc->host.lightIndices.size(); ++i) {
        const int li = sc->host.lightIndices[i];
        if (sc->host.globalLights[(size_t)li].type == INCR_LIGHT_DIRECTIONAL) { sun = li; break; }
    }
    if (sun != -1 && (size_t)sun < sc->host.lightTransforms.size()) { const float* m = sc->host.lightTransforms[(size_t)sun].m; L = v3(-m[8], -m[9], -m[10]); }
    L = normalize3(L);
    const float Lm[3] = {L.x, L.y, L.z};
    sky_sun_terms(Lm, d->turbidity, P.perez, &P.sunMiss);
    sky_sun_terms(sc->host.sunDir, d->turbidity, P.perez, &P.sunEnv);
    if (d->envmap_r11g11b10) {
        if (!d->envmap_res || !d->envmap_mips || d->envmap_mips > 16) return fail(INCR_ERR_INVALID_ARGUMENT, "envmap given with zero resolution / level count");
        P.envmap = d->envmap_r11g11b10; P.envmapRes = d->envmap_res; P.envmapMips = d->envmap_mips;
    }
    P.envSkyMode = d->env_sky_mode;
    P.rowBlockHeight = d->shard.row_block_height; P.rowBlockFirst = d->shard.row_block_first; P.rowBlockStride = d->shard.row_block_stride ? d->shard.row_block_stride : 1;
    P.localRows = incr_shard_local_rows(d->res_y, &d->shard);
    P.target = (uint16_t*)d->target_rgba16f; P.history = (uint32_t*)d->history_r11g11b10;
    P.storageRounding = d->storage_rounding;
    P.stats = d->ray_stats;
    P.statStripes = ctx->statStripes;     // the wavefront kernels count through the striped counters
    upload_grid_params(sc, P);
    P.numSamples = std::max(d->num_samples, 1u);
    P.hitChain = d->debug_hit_chain;
    P.localPixels = P.localRows * P.W;
    P.divLocalPixels = make_fastdiv(std::max(P.localPixels, 1u));
    if ((uint64_t)P.localPixels * P.numSamples > 0x7fffffffull) return fail(INCR_ERR_INVALID_ARGUMENT, "num_samples %u x %u local pixels overflows the path index", P.numSamples, P.localPixels);
    int rc = ensure_wf_workspace(ctx, P.W, P.localRows, P.numSamples, wf_rays_per_path(ctx, sc), (hipStream_t)stream);
    if (rc) return rc;
    if (P.numSamples > 1u) {        // the samples' parking surface (RGBA16F bits per sample and local pixel), library-owned, grown on demand
        const uint64_t needStage = (uint64_t)P.localPixels * P.numSamples * 8u;
        if (ctx->wfStageBytes < needStage) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            if (ctx->wfStage) { HIP_TRY(hipFree(ctx->wfStage)); ctx->wfStage = nullptr; ctx->wfStageBytes = 0; }
            HIP_TRY(hipMalloc(&ctx->wfStage, needStage));
            ctx->wfStageBytes = needStage;
        }
        P.sampleStage = (uint2*)ctx->wfStage;
    }
    hipEvent_t* ev = nullptr;
    if (ctx->profiling) {       // one record of kMaxPipes x 22 events per call, whatever the pipeline count
        const size_t perCall = (size_t)kProfEvents * kMaxPipes;
        while (ctx->profEvents.size() < (ctx->profFrames + 1) * perCall) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ctx->profEvents.push_back(e); }
        ev = ctx->profEvents.data() + ctx->profFrames * perCall;
        ctx->profPipes.push_back(ctx->wfPipes | (ctx->wfOverlap ? 0x100 : 0));
        ctx->profFrames++;
    }
    // the scene's sun cover map: a shadow ray toward the first directional light that provably hits nothing is answered at its set-up (pathtrace_wf.hip shadow_gen_block)
    if (P.directLighting)
        for (size_t li = 0; li < sc->host.globalLights.size(); ++li)
            if (sc->host.globalLights[li].type == INCR_LIGHT_DIRECTIONAL) { if ((rc = incr_ensure_sun_cover(ctx, sc, (uint32_t)li, P, (hipStream_t)stream))) return rc; break; }
    const uint32_t stackEntries = std::max(sc->bvh.maxDepth > 1 ? sc->bvh.maxDepth - 1 : 1u, std::min(ctx->forceStackEntries, 24u));
    ctx->wfStreams[0] = (hipStream_t)stream;
    HIP_TRY(incr_launch_pathtrace_wf(&P, ctx->wf, ctx->wfPipes, ctx->wfTileRow, ctx->wfStreams, ctx->wfFork, ctx->wfJoin, ctx->props.multiProcessorCount, stackEntries, ev,
                                     ctx->wfSideStreams, ctx->wfBounceFork, ctx->wfBounceJoin, ctx->wfOverlap ? 1 : 0, ctx->wfAccPlacement, ctx->profiling ? ctx->wfOverflowTally : nullptr));
    return INCR_OK;
}

static void fill_scene_params(const incr_context* ctx, const incr_scene* sc, PtParams& P) {
    memset(&P, 0, sizeof P);
    P.nodes = sc->dNodes; P.tris = sc->dTris; P.shade = sc->dShade;
    P.materials = sc->dMaterials; P.textures = sc->dTextures; P.materialTexels = sc->dMaterialTexels; P.srgbLut = ctx->srgbLut;
    P.meshLocalToWorld = sc->dMeshLocalToWorld;
    P.globalLights = sc->dGlobalLights; P.lightIndices = sc->dLightIndices; P.lightTransforms = sc->dLightTransforms;
    P.resolvedLights = (const float4*)sc->dResolvedLights;
    P.numGlobalLights = (uint32_t)sc->host.globalLights.size(); P.numSceneLights = (uint32_t)sc->host.lightIndices.size();
    P.lightCount = (int32_t)sc->host.globalLights.size();                     // path-tracing.inl:198 m_globalLightList.size()
    P.numTris = (uint32_t)sc->bvh.tris.size(); P.numNodes = (uint32_t)sc->bvh.nodes8.size();
}

int incr_trace_rays(incr_context* ctx, const incr_scene* sc, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax,
                    int any_hit, uint32_t* out_ids, float* out_tuv, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!sc || (n && (!origins || !dirs || !out_ids || !out_tuv))) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    PtParams P; fill_scene_params(ctx, sc, P);
    HIP_TRY(incr_launch_trace_rays(&P, n, origins, dirs, tmin, tmax, any_hit, out_ids, out_tuv, (hipStream_t)stream));
    return INCR_OK;
}

int incr_scene_validate_host(const incr_scene_desc* desc, incr_scene_info* info) {
    if (!desc || !info) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    HostScene hs; HostBvh bvh;
    copy_scene_desc(*desc, hs);
    const char* why = "";
    if (!validate_scene(hs, &why)) return fail(INCR_ERR_INVALID_ARGUMENT, "scene desc rejected: %s", why);
    build_world_bvh(hs, bvh);
    const char* err = check_bvh(bvh);
    if (err) return fail(INCR_ERR_INVALID_ARGUMENT, "BVH invariant violated: %s", err);
    memset(info, 0, sizeof *info);
    info->num_triangles = bvh.numTriangles; info->num_bvh_nodes = bvh.nodes8.size();
    info->bvh_bytes = bvh.nodes8.size() * sizeof(BvhNode8); info->triangle_bytes = bvh.tris.size() * sizeof(TriIsect);
    info->shading_bytes = bvh.shade.size() * sizeof(TriShade);
    info->num_instances = bvh.numInstances; info->bvh_max_depth = bvh.maxDepth;
    return INCR_OK;
}

int incr_set_pathtrace_pipelines(incr_context* ctx, uint32_t pipelines) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (pipelines > (uint32_t)kMaxPipes) return fail(INCR_ERR_INVALID_ARGUMENT, "at most %d pipelines", kMaxPipes);
    ctx->wfPipesWanted = (int)pipelines;                                  // 0 = chosen per call from the number of paths
    return INCR_OK;
}

int incr_set_pathtrace_overlap(incr_context* ctx, int32_t enable) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    ctx->wfOverlap = enable != 0;
    return INCR_OK;
}

int incr_set_pathtrace_accumulate_placement(incr_context* ctx, int32_t placement) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (placement < -1 || placement > 1) return fail(INCR_ERR_INVALID_ARGUMENT, "placement %d: -1 (by size), 0 (end of the main stream) or 1 (behind the any-hit launches)", placement);
    ctx->wfAccPlacement = placement;
    return INCR_OK;
}

int incr_profile_begin(incr_context* ctx) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    ctx->profiling = true; ctx->profFrames = 0; ctx->profPipes.clear();
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ctx->wfOverflowTally, 0, 8 * 4 * sizeof(unsigned long long)));
    return INCR_OK;
}

int incr_profile_end(incr_context* ctx, incr_pathtrace_profile* out) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!out) return fail(INCR_ERR_INVALID_ARGUMENT, "out is NULL");
    memset(out, 0, sizeof *out);
    ctx->profiling = false;
    HIP_TRY(hipDeviceSynchronize());
    // every pipeline's kernels are timed on the pipeline's own stream; with more than one pipeline the intervals of different
    // pipelines overlap in time, so the per-kernel sums can exceed the frame's wall time (they price the kernels, not the frame)
    double* kinds[4] = {&out->extend_ms, &out->shade_ms, &out->shadow_ms, &out->accumulate_ms};
    const int launchSlot[4] = {0, 1, 3, 4};
    for (size_t f = 0; f < ctx->profFrames; ++f) for (int pipe = 0; pipe < (ctx->profPipes[f] & 0xff); ++pipe) {
        hipEvent_t* ev = ctx->profEvents.data() + (f * kMaxPipes + pipe) * kProfEvents;
        float ms = 0.f;
        for (int b = 0; b < 4; ++b) for (int k = 0; k < 4; ++k) {
            HIP_TRY(hipEventElapsedTime(&ms, ev[1 + (b * 4 + k) * 2], ev[1 + (b * 4 + k) * 2 + 1]));
            *kinds[k] += ms; if (ms > 0.f) out->launches[launchSlot[k]]++;
        }
    }
    out->frames = (uint32_t)ctx->profFrames;
    out->pipelines = ctx->profPipes.empty() ? 0u : (uint32_t)(ctx->profPipes.back() & 0xff);
    out->overlap = ctx->profPipes.empty() ? 0u : (uint32_t)((ctx->profPipes.back() >> 8) & 1);
    unsigned long long tally[32];
    HIP_TRY(hipMemcpy(tally, ctx->wfOverflowTally, sizeof tally, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) { out->shadow_rays_asked += tally[4 * i]; out->shadow_rays_inline += tally[4 * i + 1]; out->shadow_queue_overflows += (uint32_t)tally[4 * i + 2]; out->shadow_rays_answered += tally[4 * i + 3]; }
    ctx->profFrames = 0; ctx->profPipes.clear();
    return INCR_OK;
}

int incr_selftest_small_float(incr_context* ctx, uint64_t out_mismatches[4]) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!out_mismatches) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 32));
    hipError_t e = hipMemset(d, 0, 32);
    if (e == hipSuccess) e = incr_launch_codec_selftest(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_mismatches, d, 32, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(INCR_ERR_HIP, "small-float self-test failed: %s", hipGetErrorString(e));
    return INCR_OK;
}

int incr_selftest_reciprocal(incr_context* ctx, uint64_t out_mismatches[3]) {
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!out_mismatches) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 24));
    hipError_t e = hipMemset(d, 0, 24);
    if (e == hipSuccess) e = incr_launch_rcp_selftest(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_mismatches, d, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(INCR_ERR_HIP, "reciprocal self-test failed: %s", hipGetErrorString(e));
    return INCR_OK;
}

int incr_tonemap(incr_context* ctx, const incr_tonemap_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->hdr_r11g11b10 || !d->ldr_rgba8) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(incr_launch_tonemap((const uint32_t*)d->hdr_r11g11b10, (uint32_t*)d->ldr_rgba8, (size_t)d->res_x * d->rows, (int)d->exposure, d->viewmode, ctx->srgbSteps, (hipStream_t)stream));
    return INCR_OK;
}

int incr_unshard(incr_context* ctx, void* dst, const void* src, uint32_t res_x, uint32_t res_y, uint32_t row_block_height,
                 uint32_t num_shards, uint32_t texel_bytes, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!dst || !src || !row_block_height || !num_shards || texel_bytes % 4) return fail(INCR_ERR_INVALID_ARGUMENT, "bad unshard arguments");
    incr_shard sh = {row_block_height, 0, num_shards};
    const uint32_t localRows = incr_shard_local_rows(res_y, &sh);
    HIP_TRY(incr_launch_unshard((uint8_t*)dst, (const uint8_t*)src, res_x, res_y, row_block_height, num_shards, localRows, texel_bytes, (hipStream_t)stream));
    return INCR_OK;
}

} // extern "C"

void incr_fill_scene_params(const incr_context* ctx, const incr_scene* sc, PtParams& P) { fill_scene_params(ctx, sc, P); P.statStripes = ctx->statStripes; }
