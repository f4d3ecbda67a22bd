// render_jobs.h -- C++ host mirror of the reference's render-pass interface for the hot path:
//   namespace RenderJob::<Pass> { struct Desc { ... }; Result Execute(Sync*, const Desc&); }
// (reference: source/demo-dll/src/render-jobs/*.inl, job-sync.h).  Desc field names and meanings are the reference's;
// FShaderBuffer / FShaderSurface / FSystemBuffer carry HBM pointers instead of bindless descriptor indices, and every
// Execute forwards to one entry point of the C ABI (include/incr.h) on the Sync's HIP stream, in token order.
// The reference host is MSVC/Windows C++ (PPL tasks, D3D12 command lists, job-sync.h's token fence).  Here a pass is a
// closure over its Desc handed to Sync::Spawn with the token GetToken() gave it: with FConfig::RecordThreads == 0 the
// closure runs at once on the calling thread (Result::m_task already complete); with RecordThreads > 0 a pool of host
// threads runs the closures, each waiting for its token's turn before it submits to the stream (Sync::Execute there),
// and Result::m_task completes when the pass has been submitted.  m_status carries the error code the reference never
// had (it asserts: inc/common.h:113-148).  With FConfig::UseGraph the raster frame's submissions are captured into a
// HIP graph on an internal stream, the instantiated graph is updated in place each frame (hipGraphExecUpdate) and
// launched on the user's stream: one host submission per frame.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/incr.h"

// ---- math / config / view types the Descs use (SimpleMath names) ----
using Matrix = incr_matrix;
struct Vector2 { float x = 0.f, y = 0.f; };
struct Vector3 { float x = 0.f, y = 0.f, z = 0.f; };

// inc/common.h:6-48 -- the fields the hot path reads, reference defaults
struct FConfig {
    float Fov = 0.25f * 3.14159265358979323846f;
    float Exposure = 13.f;
    float CameraNearPlane = 1.f;
    int   Viewmode = 0;
    int   EnvSkyMode = 1;
    float SkyBrightness = 25000.f;
    bool  EnableDirectLighting = true;
    bool  EnableSkyLighting = true;
    bool  EnableDiffuseIBL = true;
    bool  EnableSpecularIBL = true;
    bool  PathTrace = false;
    bool  ForwardLighting = false;
    bool  FrustumCulling = true;
    bool  EnableTAA = true;
    bool  EnableHBAO = false;
    bool  UseBentNormals = EnableHBAO;
    bool  UseMeshlets = true;
    bool  FreezeCulling = false;
    int   LightClusterDimX = 16, LightClusterDimY = 9, LightClusterDimZ = 24;
    int   MaxLightsPerCluster = 64;
    float ClusterDepthExtent = 200.f;
    uint32_t MaxSampleCount = 256;
    float Pathtracing_CameraAperture = 0.01f;
    float Pathtracing_CameraFocalLength = 7.f;
    float Turbidity = 2.f;
    int   StorageRounding = 0;          // not in the reference: typed-UAV store rounding (incr_pathtrace_desc::storage_rounding)
    // not in the reference: how the frame is issued (results are bit-identical either way)
    bool  FusePasses = true;            // sky + sun + clustered lighting as one kernel, tonemap inside the TAA resolve
    int   RecordThreads = 0;            // > 0: passes are recorded by that many host threads and submitted in token order (PPL tasks there)
    bool  UseGraph = false;             // raster frame captured into a HIP graph, updated in place and replayed every frame
};

// inc/renderer.h FView
struct FView {
    Matrix m_viewTransform{}, m_projectionTransform{};
    Vector3 m_position, m_right, m_up, m_look;
    float m_fov = 0.f;
    incr_view ToIncr() const;
};

// GPU resources: RenderBackend12::CreateNewShaderBuffer / CreateNewShaderSurface / CreateNewSystemBuffer stand-ins
struct FShaderBuffer  { void* m_devicePtr = nullptr; size_t m_size = 0; };
struct FShaderSurface { void* m_devicePtr = nullptr; uint32_t m_width = 0, m_height = 0, m_texelBytes = 0; };
struct FSystemBuffer  { incr_view_constants m_viewConstants{}; };       // FViewConstants (the scene constants live in incr_scene)
struct FTexture       { void* m_devicePtr = nullptr; };

// inc/scene.h FScene: the device-resident scene + the light queries Renderer::Render makes
struct FScene {
    incr_scene* m_incrScene = nullptr;
    int    m_directionalLightIndex = -1;        // GetDirectionalLight(): global index of the first directional scene light
    size_t m_punctualLightCount = 0;            // GetPunctualLightCount()
    size_t m_meshletCount = 0;
    size_t m_primitiveCount = 0;
    const float* m_skylightSh9 = nullptr;       // device, 9 x float4 (m_skylight.m_shTextureIndex), may be NULL
    const uint32_t* m_skylightEnvmap = nullptr; // device, prefiltered cube with all levels (m_skylight.m_envmapTextureIndex), may be NULL
    uint32_t m_envmapRes = 0, m_envmapMips = 0;
    FTexture m_envBRDF; uint32_t m_envBrdfWidth = 0, m_envBrdfHeight = 0;     // Renderer::s_envBRDF lives in the renderer there
    int    GetDirectionalLight() const { return m_directionalLightIndex; }
    size_t GetPunctualLightCount() const { return m_punctualLightCount; }
};

namespace RenderJob {

// job-sync.h:8-70: token-ordered submission + the task a pass returns.  There, every pass takes a token on the calling
// thread (= the order of the Execute calls in Renderer::Render), records its command list on a PPL task and submits it to
// the DIRECT queue when the CPU fence says its predecessor has been submitted; Render joins the tasks at the end of the
// frame (when_all, renderer.cpp:2039-2041).  Here: one HIP stream is the DIRECT queue, "recording" is the marshalling
// of the pass's C-ABI descriptor, a small worker pool stands in for the PPL scheduler (m_threads == 0: record inline on
// the caller's thread), and Execute blocks until every earlier token has launched.
struct Result {                                    // job-sync.h:66-70: the task that was kicked off
    std::shared_future<int> m_task;
    int m_status = INCR_OK;                        // valid once the task is done (inline recording: immediately)
    bool m_taskDone = true;
    int wait() { if (m_task.valid()) { m_status = m_task.get(); m_taskDone = true; } return m_status; }
};

struct Sync {
    incr_context* m_ctx = nullptr;
    incr_stream m_stream = nullptr;
    size_t m_cpuFenceValue = 0, m_completed = 0;   // last token handed out / last token submitted (the CPU fence)
    int m_firstError = 0;
    std::string m_firstErrorMessage;
    std::mutex m_mutex;
    std::condition_variable m_turn;
    // recording pool
    int m_threads = 0;
    std::vector<std::thread> m_workers;
    std::deque<std::packaged_task<int()>> m_queue;
    std::mutex m_queueMutex;
    std::condition_variable m_queueCv;
    bool m_stop = false;
    // optional per-pass GPU timing (incr_renderer_set_profiling): one event after every submission of the frame
    bool m_profile = false;
    std::vector<std::pair<std::string, void*>> m_marks;     // (pass, hipEvent_t) of the current frame; [0] = frame start
    void BeginFrameProfile();
    void MarkPass(const char* passName);

    ~Sync() { SetThreads(0); }
    void SetThreads(int n) {
        if (n == m_threads) return;
        { std::lock_guard<std::mutex> l{m_queueMutex}; m_stop = true; }
        m_queueCv.notify_all();
        for (std::thread& t : m_workers) t.join();
        m_workers.clear(); m_stop = false; m_threads = n;
        for (int i = 0; i < n; ++i) m_workers.emplace_back([this] {
            for (;;) {
                std::packaged_task<int()> task;
                { std::unique_lock<std::mutex> l{m_queueMutex};
                  m_queueCv.wait(l, [this] { return m_stop || !m_queue.empty(); });
                  if (m_queue.empty()) return;
                  task = std::move(m_queue.front()); m_queue.pop_front(); }
                task();
            }
        });
    }
    size_t GetToken() { std::lock_guard<std::mutex> l{m_mutex}; return ++m_cpuFenceValue; }
    // Submits under the token order: waits until token - 1 has been submitted (job-sync.h:38-50), launches, signals.
    template <class Fn> int Execute(size_t token, Fn&& launch, const char* passName = nullptr) {
        std::unique_lock<std::mutex> l{m_mutex};
        m_turn.wait(l, [&] { return m_completed + 1 >= token; });
        const int rc = launch();
        if (m_profile) MarkPass(passName);
        if (rc != INCR_OK && m_firstError == INCR_OK) { m_firstError = rc; m_firstErrorMessage = incr_last_error(); }
        m_completed = token;
        l.unlock();
        m_turn.notify_all();
        return rc;
    }
    // Kicks off a pass's record-and-submit task: on the pool, or inline when there is none.
    template <class Fn> Result Spawn(Fn&& record) {
        Result r;
        if (m_threads == 0) { r.m_status = record(); return r; }
        std::packaged_task<int()> task(std::forward<Fn>(record));
        r.m_task = task.get_future().share(); r.m_taskDone = false;
        { std::lock_guard<std::mutex> l{m_queueMutex}; m_queue.push_back(std::move(task)); }
        m_queueCv.notify_one();
        return r;
    }
};

Result UpdateTLAS_Execute(Sync* jobSync, const FScene* scene);       // UpdateTLASPass::Execute(Sync*, const FScene*)
namespace UpdateTLASPass { inline Result Execute(Sync* jobSync, const FScene* scene) { return UpdateTLAS_Execute(jobSync, scene); } }

namespace PathTracing {      // render-jobs/path-tracing.inl:3-15
struct Desc {
    FShaderSurface* targetBuffer; FShaderSurface* historyBuffer;
    FShaderBuffer* lightPropertiesBuffer; FShaderBuffer* lightTransformsBuffer;     // owned by incr_scene here; may be NULL
    uint32_t currentSampleIndex; uint32_t resX; uint32_t resY;
    const FScene* scene; const FView* view; FConfig renderConfig;
    FShaderBuffer* rayStatsBuffer = nullptr;        // not in the reference: incr_ray_stats accumulation
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace TonemapPass {      // render-jobs/tonemap.inl:3-9
struct Desc { FShaderSurface* source; FShaderSurface* target; FConfig renderConfig; };
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace BatchCullingPass { // render-jobs/batch-culling.inl:3-12
struct Desc {
    FShaderBuffer* batchArgsBuffer_Default; FShaderBuffer* batchArgsBuffer_DoubleSided; FShaderBuffer* batchCountsBuffer;
    const FScene* sceneConstantBuffer;              // FSceneConstants only carries descriptor indices of the scene arrays
    FSystemBuffer* viewConstantBuffer;
    size_t drawCount; FConfig renderConfig;
    FShaderBuffer* renderStatsBuffer = nullptr;     // SpecialDescriptors::RenderStatsBufferUavIndex
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace LightCullingPass { // render-jobs/light-culling.inl:3-14
struct Desc {
    FShaderBuffer* culledLightCountBuffer; FShaderBuffer* culledLightListsBuffer; FShaderBuffer* lightGridBuffer;
    FShaderBuffer* lightPropertiesBuffer; FShaderBuffer* lightTransformsBuffer;
    FConfig renderConfig; const FScene* scene; const FView* view; Vector2 jitter;
    FSystemBuffer* viewConstantBuffer = nullptr;    // carries the matrices light-culling.inl:111-112 derives from view + jitter
    FShaderBuffer* renderStatsBuffer = nullptr;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace VisibilityPass {   // render-jobs/visibility-pass.inl:3-17
struct Desc {
    FShaderSurface* visBufferTarget; FShaderSurface* depthStencilTarget;
    FShaderBuffer* indirectArgsBuffer_Default; FShaderBuffer* indirectArgsBuffer_DoubleSided; FShaderBuffer* indirectCountsBuffer;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    uint32_t resX; uint32_t resY; size_t drawCount; FConfig renderConfig;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace GBufferComputePass { // render-jobs/gbuffer-compute-pass.inl:3-15
struct Desc {
    FShaderSurface* sourceVisBuffer; FShaderSurface* colorTarget; FShaderSurface* gbufferTargets[3]; FShaderSurface* depthStencilTarget;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    uint32_t resX; uint32_t resY; const FScene* scene; FConfig renderConfig;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace GBufferRasterPass { // render-jobs/gbuffer-raster-pass.inl:3-15 (decals)
struct Desc {
    FShaderSurface* sourceVisBuffer; FShaderSurface* colorTarget; FShaderSurface* gbufferTargets[3]; FShaderSurface* depthStencilTarget;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    uint32_t resX; uint32_t resY; const FScene* scene; FConfig renderConfig;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace HBAO {             // render-jobs/hbao.inl:3-14
struct Desc {
    FShaderSurface* aoTarget; FShaderSurface* bentNormalTarget; FShaderSurface* depthStencil; FShaderSurface* gbufferNormals;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    FConfig renderConfig; uint32_t resX; uint32_t resY;
    FShaderBuffer* rayStatsBuffer = nullptr;        // not in the reference: incr_ray_stats accumulation
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace SkyLightingPass {  // render-jobs/sky-lighting.inl:3-19
struct Desc {
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTex;
    FShaderSurface* gbufferBaseColorTex; FShaderSurface* gbufferNormalsTex; FShaderSurface* gbufferMetallicRoughnessAoTex;
    FShaderSurface* ambientOcclusionTex; FShaderSurface* bentNormalsTex; FTexture* envBRDFTex;
    FConfig renderConfig; const FScene* scene; const FView* view; Vector2 jitter; uint32_t resX; uint32_t resY;
    FSystemBuffer* viewConstantBuffer = nullptr;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace ForwardLightingPass { // render-jobs/forward-pass.inl:3-15
struct Desc {
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTarget;
    uint32_t resX; uint32_t resY;
    const FScene* scene; const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer; FConfig renderConfig;
    FShaderBuffer* rayStatsBuffer = nullptr;        // not in the reference: incr_ray_stats accumulation
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace DirectLightingPass { // render-jobs/direct-lighting.inl:3-16
struct Desc {
    int directionalLightIndex;
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTex;
    FShaderSurface* gbufferBaseColorTex; FShaderSurface* gbufferNormalsTex; FShaderSurface* gbufferMetallicRoughnessAoTex;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    FConfig renderConfig; uint32_t resX; uint32_t resY;
    FShaderBuffer* rayStatsBuffer = nullptr;        // not in the reference: incr_ray_stats accumulation
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace ClusteredLightingPass { // render-jobs/clustered-lighting.inl:3-17
struct Desc {
    FShaderBuffer* lightListsBuffer; FShaderBuffer* lightGridBuffer;
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTex;
    FShaderSurface* gbufferBaseColorTex; FShaderSurface* gbufferNormalsTex; FShaderSurface* gbufferMetallicRoughnessAoTex;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
    FConfig renderConfig; uint32_t resX; uint32_t resY;
};
Result Execute(Sync* jobSync, const Desc& passDesc, const bool bRequiresClear);
}
// FConfig::FusePasses: SkyLightingPass + DirectLightingPass + ClusteredLightingPass as one submission (incr_deferred_lighting);
// any Desc may be NULL = that pass is skipped this frame, exactly as Render skips it
namespace DeferredLightingPass {
Result Execute(Sync* jobSync, const SkyLightingPass::Desc* sky, const DirectLightingPass::Desc* direct, const ClusteredLightingPass::Desc* clustered, const bool bRequiresClear);
}
namespace DynamicSkyPass {   // render-jobs/dynamic-sky.inl:3-14
struct Desc {
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTarget;
    uint32_t resX; uint32_t resY;
    const FScene* scene; const FView* view; Vector2 jitter; FConfig renderConfig;
    FSystemBuffer* viewConstantBuffer = nullptr;    // resolution + scissor rows
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace EnvironmentmapPass { // render-jobs/environmentmap.inl:3-14
struct Desc {
    FShaderSurface* colorTarget; FShaderSurface* depthStencilTarget;
    uint32_t resX; uint32_t resY;
    const FScene* scene; const FView* view; Vector2 jitter; FConfig renderConfig;
    FSystemBuffer* viewConstantBuffer = nullptr;    // invParallaxViewProjMatrix + resolution + scissor rows
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace DebugVizPass {     // render-jobs/debug-visualization.inl:3-18
struct Desc {
    FShaderSurface* visBuffer; FShaderSurface* gbuffers[3]; FShaderSurface* target; FShaderSurface* depthBuffer;
    FShaderSurface* aoBuffer; FShaderSurface* bentNormalsBuffer; FShaderBuffer* indirectArgsBuffer;
    Vector2 jitter; FConfig renderConfig; const FScene* scene; const FView* view;
    const FScene* sceneConstantBuffer; FSystemBuffer* viewConstantBuffer;
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}
namespace TAAResolvePass {   // render-jobs/taa-resolve.inl:3-14
struct Desc {
    FShaderSurface* source; FShaderSurface* target;
    uint32_t resX; uint32_t resY; uint32_t historyIndex;
    Matrix invViewProjectionTransform; Matrix prevViewProjectionTransform;
    FShaderSurface* depthTexture;                   // depthTextureIndex in the reference
    FConfig renderConfig;
    FShaderSurface* historySource = nullptr;        // the reference resolves in place; here last frame's accumulation buffer
    uint32_t rowBegin = 0, rowEnd = 0;              // not in the reference: rows this GPU resolves (0,0 = the whole frame)
    FShaderSurface* tonemapTarget = nullptr;        // not in the reference: FConfig::FusePasses -- TonemapPass of `target` done by the resolve itself
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}

} // namespace RenderJob
