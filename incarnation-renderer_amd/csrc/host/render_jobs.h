// render_jobs.h -- C++ host mirror of the reference's render-pass interface for the hot path:
//   namespace RenderJob::<Pass> { struct Desc { ... }; Result Execute(Sync*, const Desc&); }
// (reference: source/demo-dll/src/render-jobs/*.inl, job-sync.h).  Desc field names and meanings are the reference's;
// FShaderBuffer / FShaderSurface / FSystemBuffer carry HBM pointers instead of bindless descriptor indices, and every
// Execute forwards to one entry point of the C ABI (include/incr.h) on the Sync's HIP stream, in token order.
// The reference host is MSVC/Windows C++ (PPL tasks, D3D12 command lists); here a pass is an asynchronous launch on a
// stream, so Result::m_task is already complete when Execute returns and m_status carries the error code the reference
// never had (it asserts: inc/common.h:113-148).
#pragma once
#include <cstdint>
#include <mutex>
#include <string>

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

// job-sync.h:8-61: token-ordered submission.  One HIP stream is the DIRECT queue; tokens are handed out in call order
// and Execute launches under the mutex, so passes recorded from several host threads still reach the stream in order.
struct Sync {
    incr_context* m_ctx = nullptr;
    incr_stream m_stream = nullptr;
    size_t m_cpuFenceValue = 0, m_completed = 0;
    int m_firstError = 0;
    std::string m_firstErrorMessage;
    std::mutex m_mutex;
    size_t GetToken() { std::lock_guard<std::mutex> l{m_mutex}; return ++m_cpuFenceValue; }
    template <class Fn> int Execute(size_t token, Fn&& launch) {
        std::lock_guard<std::mutex> l{m_mutex};
        (void)token;                                  // single recording thread: tokens already arrive in order
        const int rc = launch();
        if (rc != INCR_OK && m_firstError == INCR_OK) { m_firstError = rc; m_firstErrorMessage = incr_last_error(); }
        ++m_completed;
        return rc;
    }
};

// job-sync.h:66-70
struct Result { bool m_taskDone = true; int m_status = INCR_OK; };

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
};
Result Execute(Sync* jobSync, const Desc& passDesc);
}

} // namespace RenderJob
