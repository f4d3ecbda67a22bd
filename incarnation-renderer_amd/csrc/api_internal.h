// api_internal.h -- state shared by the host-side translation units of libincr_hip.so (api.hip, raster_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

#include "../../include/incr.h"
#include "device_types.h"
#include "host_math.h"
#include "pt_params.h"
#include "raster_params.h"
#include "scene_build.h"

// ---------------------------------------------------------------------------------------------- errors
int incr_fail(int code, const char* fmt, ...);      // stores the thread's incr_last_error() message, returns code
#define fail incr_fail
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? INCR_ERR_OUT_OF_MEMORY : INCR_ERR_HIP, \
                                 "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------------------------------------- profiler ranges
// The reference wraps every pass in a scoped PIX / Tracy event (inc/profiling.h:10-12 SCOPED_COMMAND_LIST_EVENT); here every C-ABI
// pass call is a roctx range named after its entry point: `rocprofv3 --marker-trace --kernel-trace` groups a frame's kernels by pass.
// The library does NOT link the profiler SDK: librocprofiler-sdk-roctx.so is looked up with dlopen on the first range, and only when a
// profiler is attached (rocprofv3 sets ROCP_TOOL_LIBRARIES) or INCR_ROCTX=1 asks for it; otherwise -- a runtime-only ROCm install, an
// ordinary run of the renderer or the bench -- a range is one predictable branch (api.hip incr_roctx_*).
void incr_roctx_push(const char* name);
void incr_roctx_pop();
struct IncrRange { explicit IncrRange(const char* name) { incr_roctx_push(name); } ~IncrRange() { incr_roctx_pop(); } };
#define INCR_RANGE() IncrRange incr_range_(__func__)

#ifndef INCR_WF_PIPELINES_DEFAULT
#define INCR_WF_PIPELINES_DEFAULT 0      /* 0 = by size (api.hip ensure_wf_workspace) */
#endif
struct incr_context {
    int device;
    hipDeviceProp_t props;
    float* srgbLut;           // 256 floats in HBM
    float* srgbSteps;         // the sRGB encode's 257 step positions (pt_device.h linear_to_srgb8)
    unsigned long long* statStripes;   // 64 x 128 B: striped ray counters of the lane-per-pixel raster passes (raster.hip count_rays)
    uint32_t forceStackEntries; // INCR_WF_STACK_ENTRIES (tests): lower bound for the traversal kernels' stack size
    uint32_t raysPerPath;     // INCR_WF_RAYS_PER_PATH (tests: a tiny queue forces the overflow path); 0 = sized from the scene (api.hip wf_rays_per_path)
    uint32_t wfRaysPerPathAllocated;     // what the workspace was carved for
    unsigned long long* wfOverflowTally; // device, 8 pipelines x 4 words: rays asked for, rays beyond the queue, calls that overflowed (profiled calls; wf_overflow_tally_kernel)
    incr::WfBuffers wf[8];    // path-state workspace of the wavefront path tracer, one set per pipeline (pathtrace_wf.hip), grown on demand
    uint32_t wfTileRow[9];    // tile rows [wfTileRow[i], wfTileRow[i + 1]) belong to pipeline i
    int wfPipes;              // pipelines of the current frame
    int wfPipesWanted;        // INCR_WF_PIPELINES / incr_set_pathtrace_pipelines; 0 = by size
    hipStream_t wfStreams[8]; // [0] is filled per call with the caller's stream; the others are library-owned
    hipEvent_t wfFork, wfJoin[8];
    bool wfOverlap;           // the any-hit launches (and, for small frames, the summation) on the pipeline's SIDE stream beside the main chain primary -> shade -> extend -> ... on the caller's
                              // (deferred accumulation, pathtrace_wf.hip launcher; INCR_WF_OVERLAP, incr_set_pathtrace_overlap)
    hipStream_t wfSideStreams[8];                     // library-owned, created on first use
    hipEvent_t wfBounceFork[32], wfBounceJoin[32];    // [pipeline * 4 + bounce]
    int wfAccPlacement;                               // where wf_accumulate_kernel runs: -1 = by size; 0 the end of the main stream, 1 behind the any-hit launches on the side stream (INCR_WF_ACC_PLACEMENT)
    void* wfBlock;            // one allocation behind wf
    uint64_t wfBytes;
    void* wfStage;            // incr_pathtrace_desc::num_samples > 1: RGBA16F bits per sample in flight and local pixel
    uint64_t wfStageBytes;
    bool profiling;                                   // incr_profile_begin .. incr_profile_end
    std::vector<hipEvent_t> profEvents;               // 22 per recorded incr_pathtrace call
    size_t profFrames;
    std::vector<int> profPipes;                       // pipelines of every recorded call
    // raster path workspace (raster_api.hip): 64-bit depth|id surface of the visibility pass + per-mesh frame constants
    void* rasterBlock;
    uint64_t rasterBytes;
    uint32_t* rasterCounters;  // queue counters of the last incr_visibility call (inside rasterBlock)
};

struct incr_scene {
    incr_context* ctx;
    incr::HostScene host;
    incr::HostBvh bvh;
    // device mirrors
    incr::BvhNode8* dNodes = nullptr; incr::TriIsect* dTris = nullptr; incr::TriShade* dShade = nullptr;
    const uint32_t** dAlphaQuads = nullptr;      // per texture: its alpha-quad chain (texture_mips.hip alpha_quads_kernel) or NULL (no alpha, or 255 everywhere)
    incr_material* dMaterials = nullptr; incr::DeviceTexture* dTextures = nullptr;
    float* dMeshLocalToWorld = nullptr;
    float* dResolvedLights = nullptr;
    incr_light* dGlobalLights = nullptr; int32_t* dLightIndices = nullptr; float* dLightTransforms = nullptr;
    // sun cover map of this scene (raster_params.h SunCover; 256-byte header + G x G words): built by the first lighting pass that has a sun, again after an
    // incr_update_bvh that moved something (geometryEpoch) or for another light index.  A cache behind const scene pointers.
    mutable void* sunCover = nullptr; mutable uint64_t sunCoverEpoch = 0; mutable uint32_t sunCoverIndex = 0;
    mutable hipEvent_t sunCoverBuilt = nullptr; mutable hipStream_t sunCoverStream = nullptr;      // a pass on another stream waits for the build
    mutable std::mutex sunCoverLock;                                                                   // host threads submitting passes of one scene at the same time
    uint64_t geometryEpoch = 1;
    std::vector<void*> dTexels;
    std::vector<incr::DeviceTexture> hostTextures;     // host copy of the descriptor table (device pointers of the TILED chains + mip counts)
    incr::MaterialTexels* dMaterialTexels = nullptr;   // per material: interleaved level-0 texels of its three main textures (device_types.h)
    uint64_t textureBytes = 0;
    incr::LightGrid grid;
    uint32_t* dGridOffsets = nullptr; uint32_t* dGridLights = nullptr;
    size_t gridOffsetsCap = 0, gridLightsCap = 0;
    // device-side acceleration-structure update (bvh_update.hip)
    incr::TriObject* dTriObj = nullptr; uint32_t* dLevelNodes = nullptr; incr::MeshInstanceInfo* dMeshInfo = nullptr; void* dNodeBoxes = nullptr;
    std::vector<incr::MeshInstanceInfo> meshInfo;
    // incr_update_bvh's host -> device uploads go through PINNED staging, double-buffered: an asynchronous copy out of pageable
    // memory either blocks the host until the stream drains or reads a source the next update already rewrites.  A slot is reused
    // only after the event recorded behind its copies has passed.
    struct UpdateStaging { uint8_t* host[2] = {nullptr, nullptr}; size_t cap[2] = {0, 0}, used = 0; hipEvent_t done[2] = {nullptr, nullptr}; int slot = 0; } staging;
    // raster path (meshlets resolved through the accessor indirection once, at scene creation)
    incr_gpu_meshlet* dMeshlets = nullptr; uint32_t* dPackedTris = nullptr; uint32_t* dMeshVisibility = nullptr;
    float4* dMvPos = nullptr; float2* dMvUv = nullptr; float4* dMvNormal = nullptr; float4* dMvTangent = nullptr;
    // primitive (non-meshlet) raster path: per-primitive draws, widened indices, vertex arrays, 64-triangle chunk table
    incr::PrimDraw* dPrimDraws = nullptr; uint32_t* dPrimIndices = nullptr;
    float4* dPvPos = nullptr; float2* dPvUv = nullptr; float4* dPvNormal = nullptr; float4* dPvTangent = nullptr;
    uint32_t* dChunkPrim = nullptr; uint32_t* dChunkFirstTri = nullptr;
    uint32_t numPrimChunks = 0;
    // decals (FScene::m_sceneMeshDecals): the same baked layout, one localToWorld (mesh transform * root transform) per decal draw
    incr::PrimDraw* dDecalDraws = nullptr; uint32_t* dDecalIndices = nullptr;
    float4* dDvPos = nullptr; float2* dDvUv = nullptr; float4* dDvNormal = nullptr; float4* dDvTangent = nullptr;
    uint32_t* dDecalChunkPrim = nullptr; uint32_t* dDecalChunkFirstTri = nullptr; float* dDecalLocalToWorld = nullptr;
    uint32_t numDecalChunks = 0, numDecals = 0;
};

// one zeroed element of slack: the traversal fetches 80 bytes per step, also when the step is a 48-byte triangle
template <class T> static int upload_vec(T** dptr, const T* src, size_t count) {
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    HIP_TRY(hipMalloc((void**)dptr, (count + 1) * sizeof(T)));
    HIP_TRY(hipMemset(*dptr + count, 0, sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dptr, src, count * sizeof(T), hipMemcpyHostToDevice));
    return INCR_OK;
}

// Preetham constants (api.hip): Perez coefficients from turbidity (path-tracing.inl:134-140) and the per-frame sun terms
void incr_perez_from_turbidity(float t, incr_perez_distribution* p);
void incr_sky_sun_terms(const float s[3], float t, const incr_perez_distribution& pc, incr::SkySun* out);
// scene part of the traversal / shading kernel arguments (api.hip)
void incr_fill_scene_params(const incr_context* ctx, const incr_scene* sc, incr::PtParams& P);
extern "C" int incr_ensure_sun_cover(incr_context* ctx, const incr_scene* sc, uint32_t sunIndex, incr::PtParams& S, hipStream_t stream);      // raster_api.hip
// raster_api.hip: bake + upload the per-meshlet-vertex attribute arrays; free them
int  incr_raster_scene_upload(incr_scene* sc);
void incr_raster_scene_free(incr_scene* sc);
// re-upload what the raster passes derive from mesh visibility / the root transform (stream-ordered)
int  incr_raster_scene_update(incr_scene* sc, hipStream_t stream);
// pinned staging of incr_update_bvh (api.hip): begin -> any number of uploads -> end (records the slot's event)
int  incr_staging_begin(incr_scene* sc, size_t bytes);
int  incr_staging_upload(incr_scene* sc, void* dst, const void* src, size_t bytes, hipStream_t stream);
int  incr_staging_end(incr_scene* sc, hipStream_t stream);
