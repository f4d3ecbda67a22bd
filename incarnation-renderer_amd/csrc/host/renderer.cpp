// renderer.cpp -- C++ host mirror of the hot-path half of the reference's renderer: the RenderJob::<Pass>::Execute
// bodies (one C-ABI call each) and Renderer::Render's path-trace and raster branches
// (reference: source/demo-dll/src/renderer.cpp:1358-2078, src/render-jobs/*.inl).  No GPU code here: everything goes
// through include/incr.h, exactly what a maintainer of the reference would link against.
#include "render_jobs.h"

#include <hip/hip_runtime_api.h>
#include <chrono>

#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

incr_view FView::ToIncr() const {
    incr_view v;
    v.view_transform = m_viewTransform; v.projection_transform = m_projectionTransform;
    const Vector3* src[4] = {&m_position, &m_right, &m_up, &m_look};
    float* dst[4] = {v.position, v.right, v.up, v.look};
    for (int i = 0; i < 4; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    v.fov = m_fov;
    return v;
}

namespace RenderJob {

// per-pass GPU time: events recorded on the submission stream (HIP events see only that stream)
void Sync::BeginFrameProfile() {
    if (!m_profile) return;
    for (auto& m : m_marks) (void)hipEventDestroy((hipEvent_t)m.second);
    m_marks.clear();
    MarkPass("frame_begin");
}
void Sync::MarkPass(const char* passName) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, (hipStream_t)m_stream);
    m_marks.emplace_back(passName ? passName : "pass", (void*)e);
}

namespace {
incr_lighting_inputs lighting_inputs(const FScene* scene, const FSystemBuffer* view, FShaderSurface* color, FShaderSurface* depth, FShaderSurface* g0,
                                     FShaderSurface* g1, FShaderSurface* g2, const FConfig& c) {
    incr_lighting_inputs in = {};
    in.scene = scene->m_incrScene; in.view = &view->m_viewConstants;
    in.color_target = (uint32_t*)color->m_devicePtr; in.depth_buffer = (const float*)depth->m_devicePtr;
    in.gbuffer_basecolor = (const uint32_t*)g0->m_devicePtr; in.gbuffer_normals = (const uint32_t*)g1->m_devicePtr;
    in.gbuffer_metallic_roughness_ao = (const uint32_t*)g2->m_devicePtr;
    in.lighting_only = c.Viewmode == 1 ? 1 : 0;                // Viewmode::LightingOnly -> LIGHTING_ONLY (clustered-lighting.inl:63)
    in.storage_rounding = c.StorageRounding;
    return in;
}
} // namespace

Result UpdateTLAS_Execute(Sync* jobSync, const FScene* scene) {           // update-tlas.inl:3-101
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    // transforms / visibility / root transform are unchanged since scene creation unless the owner passed new ones
    return jobSync->Execute(token, [&] { return incr_update_bvh(jobSync->m_ctx, scene->m_incrScene, nullptr, nullptr, nullptr, jobSync->m_stream); }, "update_bvh");
    });
}

Result PathTracing::Execute(Sync* jobSync, const Desc& p) {               // path-tracing.inl:17-299
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_pathtrace_desc d = {};
    d.scene = p.scene->m_incrScene;
    d.target_rgba16f = p.targetBuffer->m_devicePtr; d.history_r11g11b10 = p.historyBuffer->m_devicePtr;
    d.res_x = p.resX; d.res_y = p.resY;
    d.shard = {0, 0, 1};
    d.current_sample_index = p.currentSampleIndex; d.max_sample_count = p.renderConfig.MaxSampleCount;
    d.view = p.view->ToIncr();
    d.camera_aperture = p.renderConfig.Pathtracing_CameraAperture; d.camera_focal_length = p.renderConfig.Pathtracing_CameraFocalLength;
    d.env_sky_mode = p.renderConfig.EnvSkyMode; d.direct_lighting = p.renderConfig.EnableDirectLighting ? 1 : 0;
    d.viewmode = p.renderConfig.Viewmode; d.turbidity = p.renderConfig.Turbidity; d.sky_brightness = p.renderConfig.SkyBrightness;
    d.storage_rounding = p.renderConfig.StorageRounding;
    d.envmap_r11g11b10 = p.scene->m_skylightEnvmap; d.envmap_res = p.scene->m_envmapRes; d.envmap_mips = p.scene->m_envmapMips;   // :203 envmapTextureIndex
    d.ray_stats = p.rayStatsBuffer ? (incr_ray_stats*)p.rayStatsBuffer->m_devicePtr : nullptr;
    return jobSync->Execute(token, [&] { return incr_pathtrace(jobSync->m_ctx, &d, jobSync->m_stream); }, "pathtrace");
    });
}

Result TonemapPass::Execute(Sync* jobSync, const Desc& p) {               // tonemap.inl:11-136
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_tonemap_desc d = {};
    d.hdr_r11g11b10 = p.source->m_devicePtr; d.ldr_rgba8 = p.target->m_devicePtr;
    d.res_x = p.source->m_width; d.rows = p.source->m_height;
    d.exposure = p.renderConfig.Exposure; d.viewmode = p.renderConfig.Viewmode;
    return jobSync->Execute(token, [&] { return incr_tonemap(jobSync->m_ctx, &d, jobSync->m_stream); }, "tonemap");
    });
}

Result BatchCullingPass::Execute(Sync* jobSync, const Desc& p) {          // batch-culling.inl:14-131
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_batch_cull_desc d = {};
    d.scene = p.sceneConstantBuffer->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.default_list = (incr_draw_cmd*)p.batchArgsBuffer_Default->m_devicePtr;
    d.double_sided_list = (incr_draw_cmd*)p.batchArgsBuffer_DoubleSided->m_devicePtr;
    d.counts = (uint32_t*)p.batchCountsBuffer->m_devicePtr;
    d.render_stats = p.renderStatsBuffer ? (incr_render_stats*)p.renderStatsBuffer->m_devicePtr : nullptr;
    d.frustum_culling = p.renderConfig.FrustumCulling ? 1 : 0;
    d.use_primitives = p.renderConfig.UseMeshlets ? 0 : 1;                // cs_meshlet_cull_main : cs_primitive_cull_main
    return jobSync->Execute(token, [&] { return incr_batch_cull(jobSync->m_ctx, &d, jobSync->m_stream); }, "batch_cull");
    });
}

Result LightCullingPass::Execute(Sync* jobSync, const Desc& p) {          // light-culling.inl:16-152
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_light_cull_desc d = {};
    d.scene = p.scene->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.cluster_dim[0] = (uint32_t)p.renderConfig.LightClusterDimX; d.cluster_dim[1] = (uint32_t)p.renderConfig.LightClusterDimY;
    d.cluster_dim[2] = (uint32_t)p.renderConfig.LightClusterDimZ;
    d.max_lights_per_cluster = (uint32_t)p.renderConfig.MaxLightsPerCluster; d.cluster_depth_extent = p.renderConfig.ClusterDepthExtent;
    d.culled_light_lists = (uint32_t*)p.culledLightListsBuffer->m_devicePtr; d.light_grid = (incr_light_grid_data*)p.lightGridBuffer->m_devicePtr;
    d.culled_light_count = (uint32_t*)p.culledLightCountBuffer->m_devicePtr;
    d.render_stats = p.renderStatsBuffer ? (incr_render_stats*)p.renderStatsBuffer->m_devicePtr : nullptr;
    return jobSync->Execute(token, [&] { return incr_light_cull(jobSync->m_ctx, &d, jobSync->m_stream); }, "light_cull");
    });
}

Result VisibilityPass::Execute(Sync* jobSync, const Desc& p) {            // visibility-pass.inl:19-197
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_visibility_desc d = {};
    d.scene = p.sceneConstantBuffer->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.default_list = (const incr_draw_cmd*)p.indirectArgsBuffer_Default->m_devicePtr;
    d.double_sided_list = (const incr_draw_cmd*)p.indirectArgsBuffer_DoubleSided->m_devicePtr;
    d.counts = (const uint32_t*)p.indirectCountsBuffer->m_devicePtr;
    d.vis_buffer = (uint32_t*)p.visBufferTarget->m_devicePtr; d.depth_buffer = (float*)p.depthStencilTarget->m_devicePtr;
    d.use_primitives = p.renderConfig.UseMeshlets ? 0 : 1;                // visibility-pass.inl:96-97
    return jobSync->Execute(token, [&] { return incr_visibility(jobSync->m_ctx, &d, jobSync->m_stream); }, "visibility");
    });
}

Result GBufferComputePass::Execute(Sync* jobSync, const Desc& p) {        // gbuffer-compute-pass.inl:17-123
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_gbuffer_desc d = {};
    d.scene = p.scene->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.vis_buffer = (const uint32_t*)p.sourceVisBuffer->m_devicePtr;
    d.gbuffer_basecolor = (uint32_t*)p.gbufferTargets[0]->m_devicePtr; d.gbuffer_normals = (uint32_t*)p.gbufferTargets[1]->m_devicePtr;
    d.gbuffer_metallic_roughness_ao = (uint32_t*)p.gbufferTargets[2]->m_devicePtr;
    d.color_target = (uint32_t*)p.colorTarget->m_devicePtr; d.storage_rounding = p.renderConfig.StorageRounding;
    d.use_primitives = p.renderConfig.UseMeshlets ? 0 : 1;                // USING_MESHLETS, gbuffer-compute-pass.inl:56
    return jobSync->Execute(token, [&] { return incr_gbuffer(jobSync->m_ctx, &d, jobSync->m_stream); }, "gbuffer");
    });
}

static incr_sky_lighting_desc sky_desc(const SkyLightingPass::Desc& p) {
    incr_sky_lighting_desc d = {};
    d.in = lighting_inputs(p.scene, p.viewConstantBuffer, p.colorTarget, p.depthStencilTex, p.gbufferBaseColorTex, p.gbufferNormalsTex,
                           p.gbufferMetallicRoughnessAoTex, p.renderConfig);
    d.sh9_rgba = p.scene->m_skylightSh9;
    d.ambient_occlusion = (p.renderConfig.EnableHBAO && p.ambientOcclusionTex) ? (const uint8_t*)p.ambientOcclusionTex->m_devicePtr : nullptr;
    d.sky_brightness = p.renderConfig.SkyBrightness; d.diffuse_ibl = p.renderConfig.EnableDiffuseIBL ? 1 : 0;
    d.prefiltered_envmap = p.scene->m_skylightEnvmap; d.envmap_res = p.scene->m_envmapRes; d.envmap_mips = p.scene->m_envmapMips;
    d.env_brdf = p.envBRDFTex ? (const uint32_t*)p.envBRDFTex->m_devicePtr : nullptr;
    d.env_brdf_width = p.scene->m_envBrdfWidth; d.env_brdf_height = p.scene->m_envBrdfHeight;
    d.specular_ibl = p.renderConfig.EnableSpecularIBL ? 1 : 0;
    d.bent_normals = (p.renderConfig.EnableHBAO && p.bentNormalsTex) ? (const uint32_t*)p.bentNormalsTex->m_devicePtr : nullptr;
    d.use_bent_normals = p.renderConfig.UseBentNormals ? 1 : 0;          // sky-lighting.inl:65
    return d;
}
Result SkyLightingPass::Execute(Sync* jobSync, const Desc& p) {           // sky-lighting.inl:21-150
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    const incr_sky_lighting_desc d = sky_desc(p);
    return jobSync->Execute(token, [&] { return incr_sky_lighting(jobSync->m_ctx, &d, jobSync->m_stream); }, "sky_lighting");
    });
}

Result GBufferRasterPass::Execute(Sync* jobSync, const Desc& p) {         // gbuffer-raster-pass.inl:17-233
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_gbuffer_decals_desc d = {};
    d.scene = p.scene->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.depth_buffer = (const float*)p.depthStencilTarget->m_devicePtr;
    d.gbuffer_basecolor = (uint32_t*)p.gbufferTargets[0]->m_devicePtr; d.gbuffer_normals = (uint32_t*)p.gbufferTargets[1]->m_devicePtr;
    d.gbuffer_metallic_roughness_ao = (uint32_t*)p.gbufferTargets[2]->m_devicePtr;
    d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_gbuffer_decals(jobSync->m_ctx, &d, jobSync->m_stream); }, "gbuffer_decals");
    });
}

Result HBAO::Execute(Sync* jobSync, const Desc& p) {                      // hbao.inl:16-105
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_hbao_desc d = {};
    d.scene = p.sceneConstantBuffer->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.depth_buffer = (const float*)p.depthStencil->m_devicePtr; d.gbuffer_normals = (const uint32_t*)p.gbufferNormals->m_devicePtr;
    d.ao_target = (uint8_t*)p.aoTarget->m_devicePtr; d.bent_normal_target = (uint32_t*)p.bentNormalTarget->m_devicePtr;
    d.ray_stats = p.rayStatsBuffer ? (incr_ray_stats*)p.rayStatsBuffer->m_devicePtr : nullptr;
    d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_hbao(jobSync->m_ctx, &d, jobSync->m_stream); }, "hbao");
    });
}

Result ForwardLightingPass::Execute(Sync* jobSync, const Desc& p) {       // forward-pass.inl:17-190
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_forward_lighting_desc d = {};
    d.scene = p.scene->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.color_target = (uint32_t*)p.colorTarget->m_devicePtr; d.depth_buffer = (float*)p.depthStencilTarget->m_devicePtr;
    d.sh9_rgba = p.scene->m_skylightSh9;                                  // g_sceneCb.m_skylightProbeIndex
    d.prefiltered_envmap = p.scene->m_skylightEnvmap; d.envmap_res = p.scene->m_envmapRes; d.envmap_mips = p.scene->m_envmapMips;
    d.env_brdf = (const uint32_t*)p.scene->m_envBRDF.m_devicePtr; d.env_brdf_width = p.scene->m_envBrdfWidth; d.env_brdf_height = p.scene->m_envBrdfHeight;
    d.viewmode = p.renderConfig.Viewmode; d.direct_lighting = p.renderConfig.EnableDirectLighting ? 1 : 0;      // forward-pass.inl:95-101 macros
    d.diffuse_ibl = p.renderConfig.EnableDiffuseIBL ? 1 : 0; d.specular_ibl = p.renderConfig.EnableSpecularIBL ? 1 : 0;
    d.ray_stats = p.rayStatsBuffer ? (incr_ray_stats*)p.rayStatsBuffer->m_devicePtr : nullptr;
    d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_forward_lighting(jobSync->m_ctx, &d, jobSync->m_stream); }, "forward_lighting");
    });
}

static incr_direct_lighting_desc direct_desc(const DirectLightingPass::Desc& p) {
    incr_direct_lighting_desc d = {};
    d.in = lighting_inputs(p.sceneConstantBuffer, p.viewConstantBuffer, p.colorTarget, p.depthStencilTex, p.gbufferBaseColorTex, p.gbufferNormalsTex,
                           p.gbufferMetallicRoughnessAoTex, p.renderConfig);
    d.directional_light_index = p.directionalLightIndex;
    d.ray_stats = p.rayStatsBuffer ? (incr_ray_stats*)p.rayStatsBuffer->m_devicePtr : nullptr;
    return d;
}
Result DirectLightingPass::Execute(Sync* jobSync, const Desc& p) {        // direct-lighting.inl:18-112
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    const incr_direct_lighting_desc d = direct_desc(p);
    return jobSync->Execute(token, [&] { return incr_direct_lighting(jobSync->m_ctx, &d, jobSync->m_stream); }, "direct_lighting");
    });
}

static incr_clustered_lighting_desc clustered_desc(const ClusteredLightingPass::Desc& p, const bool bRequiresClear) {
    incr_clustered_lighting_desc d = {};
    d.in = lighting_inputs(p.sceneConstantBuffer, p.viewConstantBuffer, p.colorTarget, p.depthStencilTex, p.gbufferBaseColorTex, p.gbufferNormalsTex,
                           p.gbufferMetallicRoughnessAoTex, p.renderConfig);
    d.culled_light_lists = (const uint32_t*)p.lightListsBuffer->m_devicePtr; d.light_grid = (const incr_light_grid_data*)p.lightGridBuffer->m_devicePtr;
    d.cluster_dim[0] = (uint32_t)p.renderConfig.LightClusterDimX; d.cluster_dim[1] = (uint32_t)p.renderConfig.LightClusterDimY;
    d.cluster_dim[2] = (uint32_t)p.renderConfig.LightClusterDimZ;
    d.cluster_depth_extent = p.renderConfig.ClusterDepthExtent; d.requires_clear = bRequiresClear ? 1 : 0;
    return d;
}
Result ClusteredLightingPass::Execute(Sync* jobSync, const Desc& p, const bool bRequiresClear) {      // clustered-lighting.inl:19-132
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    const incr_clustered_lighting_desc d = clustered_desc(p, bRequiresClear);
    return jobSync->Execute(token, [&] { return incr_clustered_lighting(jobSync->m_ctx, &d, jobSync->m_stream); }, "clustered_lighting");
    });
}

// The three lighting passes of the deferred frame as one submission (one token, one kernel): renderer.cpp:1843-1920
Result DeferredLightingPass::Execute(Sync* jobSync, const SkyLightingPass::Desc* sky, const DirectLightingPass::Desc* direct,
                                     const ClusteredLightingPass::Desc* clustered, const bool bRequiresClear) {
    const size_t token = jobSync->GetToken();
    struct Parts { bool s, d, c; SkyLightingPass::Desc sky; DirectLightingPass::Desc direct; ClusteredLightingPass::Desc clustered; } parts = {};
    parts.s = sky != nullptr; parts.d = direct != nullptr; parts.c = clustered != nullptr;
    if (sky) parts.sky = *sky;
    if (direct) parts.direct = *direct;
    if (clustered) parts.clustered = *clustered;
    return jobSync->Spawn([=]() -> int {
    incr_sky_lighting_desc ds = {}; incr_direct_lighting_desc dd = {}; incr_clustered_lighting_desc dc = {};
    if (parts.s) ds = sky_desc(parts.sky);
    if (parts.d) dd = direct_desc(parts.direct);
    if (parts.c) dc = clustered_desc(parts.clustered, bRequiresClear);
    incr_deferred_lighting_desc d = {parts.s ? &ds : nullptr, parts.d ? &dd : nullptr, parts.c ? &dc : nullptr};
    return jobSync->Execute(token, [&] { return incr_deferred_lighting(jobSync->m_ctx, &d, jobSync->m_stream); }, "deferred_lighting");
    });
}

Result DynamicSkyPass::Execute(Sync* jobSync, const Desc& p) {            // dynamic-sky.inl:16-170
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_dynamic_sky_desc d = {};
    d.scene = p.scene->m_incrScene; d.view = &p.viewConstantBuffer->m_viewConstants;
    d.color_target = (uint32_t*)p.colorTarget->m_devicePtr; d.depth_buffer = (const float*)p.depthStencilTarget->m_devicePtr;
    d.turbidity = p.renderConfig.Turbidity; d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_dynamic_sky(jobSync->m_ctx, &d, jobSync->m_stream); }, "dynamic_sky");
    });
}

Result EnvironmentmapPass::Execute(Sync* jobSync, const Desc& p) {        // environmentmap.inl:16-150
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_environmentmap_desc d = {};
    d.view = &p.viewConstantBuffer->m_viewConstants;
    d.color_target = (uint32_t*)p.colorTarget->m_devicePtr; d.depth_buffer = (const float*)p.depthStencilTarget->m_devicePtr;
    d.envmap_r11g11b10 = p.scene->m_skylightEnvmap; d.envmap_res = p.scene->m_envmapRes; d.envmap_mips = p.scene->m_envmapMips;   // m_skylight.m_envmapTextureIndex
    d.sky_brightness = p.renderConfig.SkyBrightness; d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_environmentmap(jobSync->m_ctx, &d, jobSync->m_stream); }, "environmentmap");
    });
}

Result DebugVizPass::Execute(Sync* jobSync, const Desc& p) {              // debug-visualization.inl:20-170
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_debug_viz_desc d = {};
    d.view = &p.viewConstantBuffer->m_viewConstants; d.viewmode = p.renderConfig.Viewmode;
    d.vis_buffer = (const uint32_t*)p.visBuffer->m_devicePtr;
    d.gbuffer_basecolor = (const uint32_t*)p.gbuffers[0]->m_devicePtr; d.gbuffer_normals = (const uint32_t*)p.gbuffers[1]->m_devicePtr;
    d.gbuffer_metallic_roughness_ao = (const uint32_t*)p.gbuffers[2]->m_devicePtr;
    d.depth_buffer = (const float*)p.depthBuffer->m_devicePtr;
    // with HBAO off the frame clears both targets to 1 (renderer.cpp:1816-1841) = NULL here
    d.ambient_occlusion = (p.renderConfig.EnableHBAO && p.aoBuffer) ? (const uint8_t*)p.aoBuffer->m_devicePtr : nullptr;
    d.bent_normals = (p.renderConfig.EnableHBAO && p.bentNormalsBuffer) ? (const uint32_t*)p.bentNormalsBuffer->m_devicePtr : nullptr;
    d.color_target = (uint32_t*)p.target->m_devicePtr;
    d.light_cluster_slices = (uint32_t)p.renderConfig.LightClusterDimZ;    // :150
    d.use_primitives = p.renderConfig.UseMeshlets ? 0 : 1; d.storage_rounding = p.renderConfig.StorageRounding;
    return jobSync->Execute(token, [&] { return incr_debug_visualization(jobSync->m_ctx, &d, jobSync->m_stream); }, "debug_visualization");
    });
}

Result TAAResolvePass::Execute(Sync* jobSync, const Desc& p) {            // taa-resolve.inl:16-118
    const size_t token = jobSync->GetToken();
    return jobSync->Spawn([=]() -> int {
    incr_view_constants vc = {};
    vc.taa_inv_view_proj = p.invViewProjectionTransform; vc.prev_view_proj_transform = p.prevViewProjectionTransform;
    vc.res_x = p.resX; vc.res_y = p.resY;
    vc.scissor_y0 = p.rowBegin; vc.scissor_y1 = p.rowEnd;
    incr_taa_resolve_desc d = {};
    d.view = &vc;
    d.hdr_scene_color = (const uint32_t*)p.source->m_devicePtr; d.history_out = (uint32_t*)p.target->m_devicePtr;
    d.history_in = p.historySource ? (const uint32_t*)p.historySource->m_devicePtr : nullptr;
    d.depth_buffer = (const float*)p.depthTexture->m_devicePtr;
    d.history_index = p.historyIndex; d.storage_rounding = p.renderConfig.StorageRounding;
    if (p.tonemapTarget) { d.ldr_rgba8 = p.tonemapTarget->m_devicePtr; d.exposure = p.renderConfig.Exposure; d.viewmode = p.renderConfig.Viewmode; }
    return jobSync->Execute(token, [&] { return incr_taa_resolve(jobSync->m_ctx, &d, jobSync->m_stream); }, "taa_resolve");
    });
}

} // namespace RenderJob

// =====================================================================================================================
// Renderer: the statics of renderer.cpp:13-30 + Render (path-trace branch :1676-1709, raster branch :1711-2036)
// =====================================================================================================================
namespace Renderer {

struct FRenderState {          // inc/renderer.h FRenderState (fields Render reads)
    FConfig m_config;
    FScene* m_scene = nullptr;
    FView m_view, m_cullingView;
    uint32_t m_resX = 0, m_resY = 0;
};

struct State {
    incr_context* ctx = nullptr;
    RenderJob::Sync jobSync;                                   // s_jobSync
    uint32_t resX = 0, resY = 0;
    size_t meshletCount = 0;
    uint32_t pathtraceCurrentSampleIndex = 1;                  // s_pathtraceCurrentSampleIndex (renderer.cpp:23)
    Matrix prevViewProjectionTransform{};                      // s_prevViewProjectionTransform
    uint64_t frameIndex = 0;
    std::vector<Vector2> pixelJitterValues;                    // s_pixelJitterValues (:126-131)
    // surfaces (renderer.cpp:1364-1520; transient per frame there, persistent here)
    FShaderSurface hdrRasterSceneColor, hdrRaytraceSceneColor, depthBuffer, visBuffer, gbuffer_basecolor, gbuffer_normals, gbuffer_metallicRoughnessAo;
    FShaderSurface pathtraceHistoryBuffer, taaAccumulationBuffer[2], backBuffer;
    FShaderSurface aoBuffer, bentNormalsBuffer;                // R8_UNORM, R16G16_FLOAT (renderer.cpp:1443-1459)
    FShaderBuffer batchArgsBuffer_Default, batchArgsBuffer_DoubleSided, batchCountsBuffer, culledLightCountBuffer, culledLightListsBuffer, lightGridBuffer;
    FShaderBuffer renderStatsBuffer, rayStatsBuffer;
    std::vector<void*> allocations;
    // Per-frame transients (renderer.cpp:1385-1515 creates them inside Render; RenderBackend12 recycles them through a pool keyed by
    // the frame fence, backend-d3d12.cpp:616-789).  Here: stream-ordered allocations (hipMallocAsync from the device's default memory
    // pool, release threshold raised so that freed blocks stay cached) handed out at the top of Render and returned at its end with
    // the frame's fence, a HIP event.  A block is handed out again to the SAME stream at once (stream order is the fence) and to
    // another stream only after its fence has passed; blocks unused for kTransientTrimFrames frames go back to the driver.
    struct TransientBlock { void* ptr; uint64_t bytes; hipStream_t stream; hipEvent_t fence; uint64_t lastFrame; bool inUse; };
    std::vector<TransientBlock> transients;
    uint64_t transientBytes = 0, transientAllocs = 0, transientReuses = 0;
    int lastStatus = INCR_OK;
    // Sort-first multi-GPU: this GPU draws and shades rows [bandY0, bandY1) plus a one-row halo for the TAA
    // neighbourhood, resolves TAA for the band only and leaves gather + tonemap to the caller.  0,0 = the whole frame.
    uint32_t bandY0 = 0, bandY1 = 0;
    FShaderSurface* lastHdr = nullptr;                         // what the tonemap of this frame reads
    // FConfig::UseGraph: the frame's launches are captured, the instantiated graph is updated in place and replayed
    hipGraphExec_t graphExec = nullptr;                        // the executable graph of the frame being issued (one of graphExecs)
    hipGraphExec_t graphExecs[2] = {nullptr, nullptr};         // alternated: an exec is updated in place while the other one may still be running
    hipEvent_t graphDone[2] = {nullptr, nullptr};              // recorded behind each launch; waited for before that exec is updated again
    hipStream_t captureStream = nullptr;                       // launches are captured here and replayed on the renderer's stream (which may be the null stream)
    uint64_t graphFrames = 0, graphUpdates = 0, graphRebuilds = 0;
    int graphStatus = 0;                                       // 0 = ok / unused, else the HIP error that switched graph replay off
    double clockSeconds = 0.0;                                  // Tick(dt) total (inc/demo-interface.h:12)
    double hostSecondsLastFrame = 0.0;                          // wall time of Render on the calling thread (record + submit + join)
};
static constexpr uint32_t kBandSlackRows = 64;                  // TAA surfaces hold ceil(H/N)*N rows for an equal-sized all_gather

static int alloc(State& s, void** out, uint64_t bytes, bool clear) {
    bytes = (bytes + 3) & ~3ull;                 // whole words: the R8 surface of an odd-sized frame is cleared 32 bits at a time
    int rc = incr_buffer_create(s.ctx, bytes, out);
    if (rc) return rc;
    s.allocations.push_back(*out);
    return clear ? incr_buffer_clear(s.ctx, *out, 0, bytes, nullptr) : INCR_OK;
}
static int surface(State& s, FShaderSurface& f, uint32_t texelBytes, bool clear = false) {
    f.m_width = s.resX; f.m_height = s.resY; f.m_texelBytes = texelBytes;
    return alloc(s, &f.m_devicePtr, (uint64_t)s.resX * s.resY * texelBytes, clear);
}
static int buffer(State& s, FShaderBuffer& b, uint64_t bytes) { b.m_size = bytes; return alloc(s, &b.m_devicePtr, bytes ? bytes : 4, true); }

static Matrix identity() { Matrix m{}; m.m[0] = m.m[5] = m.m[10] = m.m[15] = 1.f; return m; }

static constexpr uint64_t kTransientTrimFrames = 64;
static int transient_acquire(State& s, void** out, uint64_t bytes, hipStream_t stream) {
    bytes = std::max<uint64_t>((bytes + 3) & ~3ull, 4);
    for (State::TransientBlock& b : s.transients) {
        if (b.inUse || b.bytes != bytes) continue;
        if (b.stream != stream && hipEventQuery(b.fence) != hipSuccess) { (void)hipGetLastError(); continue; }      // another queue: only behind the fence
        b.inUse = true; b.stream = stream; b.lastFrame = s.frameIndex; *out = b.ptr; s.transientReuses++;
        return INCR_OK;
    }
    State::TransientBlock b{nullptr, bytes, stream, nullptr, s.frameIndex, true};
    if (hipMallocAsync(&b.ptr, bytes, stream) != hipSuccess) { (void)hipGetLastError(); return INCR_ERR_OUT_OF_MEMORY; }
    if (hipEventCreateWithFlags(&b.fence, hipEventDisableTiming) != hipSuccess || hipMemsetAsync(b.ptr, 0, bytes, stream) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFreeAsync(b.ptr, stream); if (b.fence) (void)hipEventDestroy(b.fence); return INCR_ERR_HIP;
    }
    s.transients.push_back(b); s.transientBytes += bytes; s.transientAllocs++;
    *out = b.ptr;
    return INCR_OK;
}
// End of the frame: every block handed out this frame goes back with the frame's fence; long-unused blocks are freed (stream-ordered).
static void transient_release_frame(State& s, hipStream_t stream, bool capturing) {
    for (size_t i = 0; i < s.transients.size();) {
        State::TransientBlock& b = s.transients[i];
        if (b.inUse) { b.inUse = false; if (!capturing) (void)hipEventRecord(b.fence, stream); ++i; continue; }
        if (s.frameIndex - b.lastFrame > kTransientTrimFrames && !capturing) {
            (void)hipFreeAsync(b.ptr, b.stream); (void)hipEventDestroy(b.fence); s.transientBytes -= b.bytes;
            s.transients[i] = s.transients.back(); s.transients.pop_back();
            continue;
        }
        ++i;
    }
}
static int transient_surface(State& s, FShaderSurface& f, uint32_t texelBytes, hipStream_t stream) {
    f.m_width = s.resX; f.m_height = s.resY; f.m_texelBytes = texelBytes;
    return transient_acquire(s, &f.m_devicePtr, (uint64_t)s.resX * s.resY * texelBytes, stream);
}
static int transient_buffer(State& s, FShaderBuffer& b, uint64_t bytes, hipStream_t stream) { b.m_size = bytes; return transient_acquire(s, &b.m_devicePtr, bytes, stream); }

int Initialize(State& s, incr_context* ctx, incr_stream stream, uint32_t resX, uint32_t resY, size_t meshletCount, const FConfig& c) {
    s.ctx = ctx; s.jobSync.m_ctx = ctx; s.jobSync.m_stream = stream; s.resX = resX; s.resY = resY;
    s.prevViewProjectionTransform = identity();
    for (uint32_t i = 0; i < 16; ++i) { float j[2]; incr_pixel_jitter(i, resX, resY, j); s.pixelJitterValues.push_back({j[0], j[1]}); }   // renderer.cpp:126-131
    int rc = 0;
    // persistent: the accumulation surfaces (renderer.cpp:96-124 creates them in Initialize), the back buffer, the read-back counters
    FShaderSurface* kept[] = {&s.pathtraceHistoryBuffer, &s.backBuffer};
    for (FShaderSurface* f : kept) if (!rc) rc = surface(s, *f, 4, true);
    for (FShaderSurface& f : s.taaAccumulationBuffer) {
        f.m_width = resX; f.m_height = resY; f.m_texelBytes = 4;
        if (!rc) rc = alloc(s, &f.m_devicePtr, (uint64_t)resX * (resY + kBandSlackRows) * 4, true);
    }
    if (!rc) rc = buffer(s, s.renderStatsBuffer, sizeof(incr_render_stats));
    if (!rc) rc = buffer(s, s.rayStatsBuffer, sizeof(incr_ray_stats));
    s.meshletCount = meshletCount;
    // everything else is a per-frame transient (acquire_frame_transients): keep what the stream-ordered allocator frees cached
    hipMemPool_t pool = nullptr;
    int device = 0; (void)hipGetDevice(&device);          // incr_initialize made the context's device current
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) { uint64_t keep = ~0ull; (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep); }
    else (void)hipGetLastError();
    return rc;
}

// renderer.cpp:1385-1515: the frame's transient surfaces and buffers, in the reference's order
static int acquire_frame_transients(State& s, const FConfig& c, hipStream_t stream) {
    int rc = 0;
    FShaderSurface* four[] = {&s.hdrRasterSceneColor, &s.depthBuffer, &s.visBuffer, &s.gbuffer_basecolor, &s.gbuffer_normals, &s.gbuffer_metallicRoughnessAo, &s.bentNormalsBuffer};
    for (FShaderSurface* f : four) if (!rc) rc = transient_surface(s, *f, 4, stream);
    if (!rc) rc = transient_surface(s, s.hdrRaytraceSceneColor, 8, stream);
    if (!rc) rc = transient_surface(s, s.aoBuffer, 1, stream);
    const uint64_t clusters = (uint64_t)c.LightClusterDimX * c.LightClusterDimY * c.LightClusterDimZ;
    if (!rc) rc = transient_buffer(s, s.batchArgsBuffer_Default, (s.meshletCount + 1) * sizeof(incr_draw_cmd), stream);
    if (!rc) rc = transient_buffer(s, s.batchArgsBuffer_DoubleSided, (s.meshletCount + 1) * sizeof(incr_draw_cmd), stream);
    if (!rc) rc = transient_buffer(s, s.batchCountsBuffer, 8, stream);
    if (!rc) rc = transient_buffer(s, s.culledLightCountBuffer, 4, stream);
    if (!rc) rc = transient_buffer(s, s.culledLightListsBuffer, clusters * c.MaxLightsPerCluster * 4, stream);      // renderer.cpp:1506
    if (!rc) rc = transient_buffer(s, s.lightGridBuffer, clusters * 8, stream);                                    // renderer.cpp:1515
    return rc;
}

void Teardown(State& s) {
    s.jobSync.SetThreads(0);
    for (int i = 0; i < 2; ++i) {
        if (s.graphExecs[i]) { (void)hipGraphExecDestroy(s.graphExecs[i]); s.graphExecs[i] = nullptr; }
        if (s.graphDone[i]) { (void)hipEventDestroy(s.graphDone[i]); s.graphDone[i] = nullptr; }
    }
    s.graphExec = nullptr;
    for (State::TransientBlock& b : s.transients) { (void)hipFreeAsync(b.ptr, b.stream); (void)hipEventDestroy(b.fence); }
    s.transients.clear(); s.transientBytes = 0;
    if (s.captureStream) { (void)hipStreamDestroy(s.captureStream); s.captureStream = nullptr; }
    for (void* p : s.allocations) incr_buffer_destroy(s.ctx, p);
    s.allocations.clear();
}

// Renderer::Render (renderer.cpp:1358-2078), hot-path half.  Returns the first error of the frame.
int Render(State& s, const FRenderState& renderState) {
    using namespace RenderJob;
    const FConfig& c = renderState.m_config;
    const uint32_t resX = s.resX, resY = s.resY;
    const FScene* scene = renderState.m_scene;
    Sync* jobSync = &s.jobSync;
    jobSync->m_firstError = INCR_OK;
    const uint64_t frameIndex = s.frameIndex;

    // renderer.cpp:1631: jitter only with TAA in the normal view mode
    const Vector2 pixelJitter = (c.EnableTAA && c.Viewmode == 0) ? s.pixelJitterValues[frameIndex % 16] : Vector2{0.f, 0.f};
    // view constants (renderer.cpp:1634-1665)
    FSystemBuffer cbViewConstants;
    const incr_view view = renderState.m_view.ToIncr(), cull = (c.FreezeCulling ? renderState.m_cullingView : renderState.m_view).ToIncr();
    const float jitter[2] = {pixelJitter.x, pixelJitter.y};
    int rc = incr_make_view_constants(&view, &cull, jitter, &s.prevViewProjectionTransform, c.CameraNearPlane, resX, resY, &cbViewConstants.m_viewConstants);
    if (rc) return rc;
    const bool banded = !(s.bandY0 == 0 && s.bandY1 == 0);
    if (banded) {            // draw + shade the band and one halo row each side (TAA reads a 3x3 neighbourhood of the colour target)
        cbViewConstants.m_viewConstants.scissor_y0 = s.bandY0 ? s.bandY0 - 1 : 0;
        cbViewConstants.m_viewConstants.scissor_y1 = s.bandY1 + 1 < resY ? s.bandY1 + 1 : resY;
    }

    const auto hostT0 = std::chrono::steady_clock::now();
    jobSync->BeginFrameProfile();
    std::vector<Result> sceneRenderJobs;                                       // :1672 (joined at the end of the frame, :2039-2041)
    auto job = [&](const Result& r) { sceneRenderJobs.push_back(r); };
    // HIP-graph replay of the raster frame (not in the reference): launches of this frame are captured instead of executed, the
    // instantiated graph of the previous frame takes the new kernel arguments in place (hipGraphExecUpdate) and is launched
    // once.  The first two frames run directly (workspaces grow there); capture needs the launches on one host thread.
    // (a profiled frame is issued directly: its per-pass events would otherwise be recorded into the capture)
    const bool useGraph = c.UseGraph && !c.PathTrace && s.graphStatus == 0 && s.frameIndex >= 2 && !jobSync->m_profile;
    jobSync->SetThreads(useGraph ? 0 : c.RecordThreads);
    const incr_stream frameStream = jobSync->m_stream;
    if (useGraph) {
        hipError_t e = hipSuccess;
        if (!s.captureStream) e = hipStreamCreateWithFlags(&s.captureStream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamBeginCapture(s.captureStream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { s.graphStatus = (int)e; (void)hipGetLastError(); }
        else jobSync->m_stream = (incr_stream)s.captureStream;
    }
    const bool capturing = useGraph && s.graphStatus == 0;
    // the frame's transients come from the REAL stream (an allocation is not a capturable launch); in steady state every request is
    // served by the block it had last frame, so a replayed graph keeps its addresses and hipGraphExecUpdate sees unchanged arguments
    rc = acquire_frame_transients(s, c, (hipStream_t)frameStream);
    if (rc) { if (capturing) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(s.captureStream, &g); if (g) (void)hipGraphDestroy(g); jobSync->m_stream = frameStream; } transient_release_frame(s, (hipStream_t)frameStream, false); return rc; }

    job(UpdateTLASPass::Execute(jobSync, scene));                              // :1676

    if (c.PathTrace) {                                                         // :1679-1709
        if (s.pathtraceCurrentSampleIndex < c.MaxSampleCount) {
            PathTracing::Desc pathtraceDesc = {};
            pathtraceDesc.targetBuffer = &s.hdrRaytraceSceneColor; pathtraceDesc.historyBuffer = &s.pathtraceHistoryBuffer;
            pathtraceDesc.currentSampleIndex = s.pathtraceCurrentSampleIndex;
            pathtraceDesc.resX = resX; pathtraceDesc.resY = resY;
            pathtraceDesc.scene = scene; pathtraceDesc.view = &renderState.m_view; pathtraceDesc.renderConfig = c;
            pathtraceDesc.rayStatsBuffer = &s.rayStatsBuffer;
            job(PathTracing::Execute(jobSync, pathtraceDesc));
            s.pathtraceCurrentSampleIndex++;                                   // accumulate samples
        }
        TonemapPass::Desc tonemapDesc = {};
        tonemapDesc.source = &s.pathtraceHistoryBuffer; tonemapDesc.target = &s.backBuffer; tonemapDesc.renderConfig = c;
        job(TonemapPass::Execute(jobSync, tonemapDesc));
    } else {                                                                   // :1711-2036
        BatchCullingPass::Desc batchCullDesc = {};
        batchCullDesc.batchArgsBuffer_Default = &s.batchArgsBuffer_Default; batchCullDesc.batchArgsBuffer_DoubleSided = &s.batchArgsBuffer_DoubleSided;
        batchCullDesc.batchCountsBuffer = &s.batchCountsBuffer;
        batchCullDesc.sceneConstantBuffer = scene; batchCullDesc.viewConstantBuffer = &cbViewConstants;
        const size_t numDraws = c.UseMeshlets ? scene->m_meshletCount : scene->m_primitiveCount;   // :1375-1377
        batchCullDesc.drawCount = numDraws; batchCullDesc.renderConfig = c; batchCullDesc.renderStatsBuffer = &s.renderStatsBuffer;
        job(BatchCullingPass::Execute(jobSync, batchCullDesc));

        const size_t punctualLightCount = scene->GetPunctualLightCount();
        if (punctualLightCount > 0 && c.EnableDirectLighting) {                // :1727-1744
            LightCullingPass::Desc lightCullDesc = {};
            lightCullDesc.culledLightCountBuffer = &s.culledLightCountBuffer; lightCullDesc.culledLightListsBuffer = &s.culledLightListsBuffer;
            lightCullDesc.lightGridBuffer = &s.lightGridBuffer;
            lightCullDesc.renderConfig = c; lightCullDesc.scene = scene; lightCullDesc.view = &renderState.m_cullingView; lightCullDesc.jitter = pixelJitter;
            lightCullDesc.viewConstantBuffer = &cbViewConstants; lightCullDesc.renderStatsBuffer = &s.renderStatsBuffer;
            job(LightCullingPass::Execute(jobSync, lightCullDesc));
        }

        VisibilityPass::Desc visDesc = {};                                     // :1747-1762
        visDesc.visBufferTarget = &s.visBuffer; visDesc.depthStencilTarget = &s.depthBuffer;
        visDesc.indirectArgsBuffer_Default = &s.batchArgsBuffer_Default; visDesc.indirectArgsBuffer_DoubleSided = &s.batchArgsBuffer_DoubleSided;
        visDesc.indirectCountsBuffer = &s.batchCountsBuffer;
        visDesc.sceneConstantBuffer = scene; visDesc.viewConstantBuffer = &cbViewConstants;
        visDesc.resX = resX; visDesc.resY = resY; visDesc.drawCount = numDraws; visDesc.renderConfig = c;
        job(VisibilityPass::Execute(jobSync, visDesc));

        GBufferComputePass::Desc gbufferComputeDesc = {};                      // :1765-1781
        gbufferComputeDesc.sourceVisBuffer = &s.visBuffer; gbufferComputeDesc.colorTarget = &s.hdrRasterSceneColor;
        gbufferComputeDesc.gbufferTargets[0] = &s.gbuffer_basecolor; gbufferComputeDesc.gbufferTargets[1] = &s.gbuffer_normals;
        gbufferComputeDesc.gbufferTargets[2] = &s.gbuffer_metallicRoughnessAo; gbufferComputeDesc.depthStencilTarget = &s.depthBuffer;
        gbufferComputeDesc.sceneConstantBuffer = scene; gbufferComputeDesc.viewConstantBuffer = &cbViewConstants;
        gbufferComputeDesc.resX = resX; gbufferComputeDesc.resY = resY; gbufferComputeDesc.scene = scene; gbufferComputeDesc.renderConfig = c;
        job(GBufferComputePass::Execute(jobSync, gbufferComputeDesc));

        GBufferRasterPass::Desc gbufferRasterDesc = {};                        // decals :1783-1798 (a no-op for scenes without any)
        gbufferRasterDesc.sourceVisBuffer = &s.visBuffer; gbufferRasterDesc.colorTarget = &s.hdrRasterSceneColor;
        gbufferRasterDesc.gbufferTargets[0] = &s.gbuffer_basecolor; gbufferRasterDesc.gbufferTargets[1] = &s.gbuffer_normals;
        gbufferRasterDesc.gbufferTargets[2] = &s.gbuffer_metallicRoughnessAo; gbufferRasterDesc.depthStencilTarget = &s.depthBuffer;
        gbufferRasterDesc.sceneConstantBuffer = scene; gbufferRasterDesc.viewConstantBuffer = &cbViewConstants;
        gbufferRasterDesc.resX = resX; gbufferRasterDesc.resY = resY; gbufferRasterDesc.scene = scene; gbufferRasterDesc.renderConfig = c;
        job(GBufferRasterPass::Execute(jobSync, gbufferRasterDesc));

        if (c.EnableHBAO) {                                                    // :1801-1815; else both targets read as cleared to 1 (:1816-1841)
            HBAO::Desc hbaoDesc = {};
            hbaoDesc.aoTarget = &s.aoBuffer; hbaoDesc.bentNormalTarget = &s.bentNormalsBuffer;
            hbaoDesc.depthStencil = &s.depthBuffer; hbaoDesc.gbufferNormals = &s.gbuffer_normals;
            hbaoDesc.sceneConstantBuffer = scene; hbaoDesc.viewConstantBuffer = &cbViewConstants;
            hbaoDesc.renderConfig = c; hbaoDesc.resX = resX; hbaoDesc.resY = resY; hbaoDesc.rayStatsBuffer = &s.rayStatsBuffer;
            job(HBAO::Execute(jobSync, hbaoDesc));
        }

        // FConfig::FusePasses: the three lighting passes of the deferred frame are described as the reference describes them and
        // submitted as one (DeferredLightingPass); the forward frame keeps its sky pass
        const bool fuseLighting = c.FusePasses && !c.ForwardLighting;
        SkyLightingPass::Desc skyLightingDesc = {};
        if (c.EnableSkyLighting) {                                             // :1843-1863
            skyLightingDesc.colorTarget = &s.hdrRasterSceneColor; skyLightingDesc.depthStencilTex = &s.depthBuffer;
            skyLightingDesc.gbufferBaseColorTex = &s.gbuffer_basecolor; skyLightingDesc.gbufferNormalsTex = &s.gbuffer_normals;
            skyLightingDesc.gbufferMetallicRoughnessAoTex = &s.gbuffer_metallicRoughnessAo;
            skyLightingDesc.ambientOcclusionTex = &s.aoBuffer; skyLightingDesc.bentNormalsTex = &s.bentNormalsBuffer;
            skyLightingDesc.renderConfig = c; skyLightingDesc.scene = scene; skyLightingDesc.view = &renderState.m_view; skyLightingDesc.jitter = pixelJitter;
            skyLightingDesc.resX = resX; skyLightingDesc.resY = resY; skyLightingDesc.viewConstantBuffer = &cbViewConstants;
            skyLightingDesc.envBRDFTex = scene->m_envBRDF.m_devicePtr ? const_cast<FTexture*>(&scene->m_envBRDF) : nullptr;   // s_envBRDF.get()
            if (!fuseLighting) job(SkyLightingPass::Execute(jobSync, skyLightingDesc));
        }

        if (c.ForwardLighting) {                                               // :1865-1881
            ForwardLightingPass::Desc forwardDesc = {};
            forwardDesc.colorTarget = &s.hdrRasterSceneColor; forwardDesc.depthStencilTarget = &s.depthBuffer;
            forwardDesc.resX = resX; forwardDesc.resY = resY;
            forwardDesc.scene = scene; forwardDesc.sceneConstantBuffer = scene; forwardDesc.viewConstantBuffer = &cbViewConstants;
            forwardDesc.renderConfig = c; forwardDesc.rayStatsBuffer = &s.rayStatsBuffer;
            job(ForwardLightingPass::Execute(jobSync, forwardDesc));
        } else {                                                               // :1882-1926
            const int directionalLightIndex = scene->GetDirectionalLight();
            const bool haveDirect = directionalLightIndex != -1 && c.EnableDirectLighting, haveClustered = punctualLightCount > 0 && c.EnableDirectLighting;
            DirectLightingPass::Desc directLightingDesc = {};
            ClusteredLightingPass::Desc clusteredLightingDesc = {};
            if (haveDirect) {
                directLightingDesc.directionalLightIndex = directionalLightIndex;
                directLightingDesc.colorTarget = &s.hdrRasterSceneColor; directLightingDesc.depthStencilTex = &s.depthBuffer;
                directLightingDesc.gbufferBaseColorTex = &s.gbuffer_basecolor; directLightingDesc.gbufferNormalsTex = &s.gbuffer_normals;
                directLightingDesc.gbufferMetallicRoughnessAoTex = &s.gbuffer_metallicRoughnessAo;
                directLightingDesc.sceneConstantBuffer = scene; directLightingDesc.viewConstantBuffer = &cbViewConstants;
                directLightingDesc.renderConfig = c; directLightingDesc.resX = resX; directLightingDesc.resY = resY;
                directLightingDesc.rayStatsBuffer = &s.rayStatsBuffer;
                if (!fuseLighting) job(DirectLightingPass::Execute(jobSync, directLightingDesc));
            }
            const bool bRequiresClear = directionalLightIndex == -1;           // :1906
            if (haveClustered) {
                clusteredLightingDesc.lightListsBuffer = &s.culledLightListsBuffer; clusteredLightingDesc.lightGridBuffer = &s.lightGridBuffer;
                clusteredLightingDesc.colorTarget = &s.hdrRasterSceneColor; clusteredLightingDesc.depthStencilTex = &s.depthBuffer;
                clusteredLightingDesc.gbufferBaseColorTex = &s.gbuffer_basecolor; clusteredLightingDesc.gbufferNormalsTex = &s.gbuffer_normals;
                clusteredLightingDesc.gbufferMetallicRoughnessAoTex = &s.gbuffer_metallicRoughnessAo;
                clusteredLightingDesc.sceneConstantBuffer = scene; clusteredLightingDesc.viewConstantBuffer = &cbViewConstants;
                clusteredLightingDesc.renderConfig = c; clusteredLightingDesc.resX = resX; clusteredLightingDesc.resY = resY;
                if (!fuseLighting) job(ClusteredLightingPass::Execute(jobSync, clusteredLightingDesc, bRequiresClear));
            }
            if (fuseLighting && (c.EnableSkyLighting || haveDirect || haveClustered))
                job(DeferredLightingPass::Execute(jobSync, c.EnableSkyLighting ? &skyLightingDesc : nullptr, haveDirect ? &directLightingDesc : nullptr,
                                                  haveClustered ? &clusteredLightingDesc : nullptr, bRequiresClear));
        }

        if (c.EnvSkyMode == 0) {                                               // EnvSkyMode::HDRI :1927-1943
            EnvironmentmapPass::Desc envmapDesc = {};
            envmapDesc.colorTarget = &s.hdrRasterSceneColor; envmapDesc.depthStencilTarget = &s.depthBuffer;
            envmapDesc.resX = resX; envmapDesc.resY = resY;
            envmapDesc.scene = scene; envmapDesc.view = &renderState.m_view; envmapDesc.jitter = pixelJitter; envmapDesc.renderConfig = c;
            envmapDesc.viewConstantBuffer = &cbViewConstants;
            job(EnvironmentmapPass::Execute(jobSync, envmapDesc));
        } else {                                                               // :1944-1959
            DynamicSkyPass::Desc skyDesc = {};
            skyDesc.colorTarget = &s.hdrRasterSceneColor; skyDesc.depthStencilTarget = &s.depthBuffer;
            skyDesc.resX = resX; skyDesc.resY = resX;                          // sic: the reference passes resX twice (:1951-1952); unused by the pass
            skyDesc.scene = scene; skyDesc.view = &renderState.m_view; skyDesc.jitter = pixelJitter; skyDesc.renderConfig = c;
            skyDesc.viewConstantBuffer = &cbViewConstants;
            job(DynamicSkyPass::Execute(jobSync, skyDesc));
        }
        const bool bDebugView = c.Viewmode != 0 && c.Viewmode != 1;            // :1961 (Normal, LightingOnly)
        if (bDebugView) {                                                      // :1962-1984; the editor's highlight pass (:1986-2000) is UI
            DebugVizPass::Desc desc = {};
            desc.visBuffer = &s.visBuffer; desc.gbuffers[0] = &s.gbuffer_basecolor; desc.gbuffers[1] = &s.gbuffer_normals;
            desc.gbuffers[2] = &s.gbuffer_metallicRoughnessAo; desc.target = &s.hdrRasterSceneColor; desc.depthBuffer = &s.depthBuffer;
            desc.aoBuffer = &s.aoBuffer; desc.bentNormalsBuffer = &s.bentNormalsBuffer; desc.indirectArgsBuffer = nullptr;
            desc.jitter = pixelJitter; desc.renderConfig = c; desc.scene = scene; desc.view = &renderState.m_view;
            desc.sceneConstantBuffer = scene; desc.viewConstantBuffer = &cbViewConstants;
            job(DebugVizPass::Execute(jobSync, desc));
        }

        FShaderSurface* hdrOut = &s.hdrRasterSceneColor;
        bool tonemapFused = false;
        if (c.EnableTAA) {                                                     // :2003-2026
            TAAResolvePass::Desc resolveDesc = {};
            resolveDesc.source = &s.hdrRasterSceneColor;
            resolveDesc.target = &s.taaAccumulationBuffer[frameIndex & 1]; resolveDesc.historySource = &s.taaAccumulationBuffer[(frameIndex + 1) & 1];
            resolveDesc.resX = resX; resolveDesc.resY = resY; resolveDesc.historyIndex = (uint32_t)frameIndex;
            resolveDesc.prevViewProjectionTransform = s.prevViewProjectionTransform;
            resolveDesc.invViewProjectionTransform = cbViewConstants.m_viewConstants.taa_inv_view_proj;      // viewProjectionTransform.Invert()
            resolveDesc.depthTexture = &s.depthBuffer; resolveDesc.renderConfig = c;
            if (banded) { resolveDesc.rowBegin = s.bandY0; resolveDesc.rowEnd = s.bandY1 < resY ? s.bandY1 : resY; }
            if (c.FusePasses && !banded) { resolveDesc.tonemapTarget = &s.backBuffer; tonemapFused = true; }      // the resolve presents
            job(TAAResolvePass::Execute(jobSync, resolveDesc));
            s.prevViewProjectionTransform = cbViewConstants.m_viewConstants.unjittered_view_proj;            // :2025
            hdrOut = resolveDesc.target;
        }

        s.lastHdr = hdrOut;
        if (!banded && !tonemapFused) {       // with row bands the caller gathers the bands first, then calls Tonemap()
            TonemapPass::Desc tonemapDesc = {};                                // :2029-2036
            tonemapDesc.source = hdrOut; tonemapDesc.target = &s.backBuffer; tonemapDesc.renderConfig = c;
            job(TonemapPass::Execute(jobSync, tonemapDesc));
        }
    }
    for (Result& r : sceneRenderJobs) r.wait();                                // sceneRenderJoinTask.wait() :2039-2041
    if (capturing) {
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(s.captureStream, &graph);
        jobSync->m_stream = frameStream;
        if (e == hipSuccess && jobSync->m_firstError == INCR_OK) {
            // two executable graphs, alternated: the one updated in place here last ran two frames ago, and its launch has finished
            // (the event behind that launch is waited for) -- an update never touches kernel arguments a running replay still reads
            const int gi = (int)(s.frameIndex & 1u);
            hipGraphExec_t& exec = s.graphExecs[gi];
            if (!s.graphDone[gi]) e = hipEventCreateWithFlags(&s.graphDone[gi], hipEventDisableTiming);
            else e = hipEventSynchronize(s.graphDone[gi]);
            if (e == hipSuccess && exec) {
                hipGraphExecUpdateResult result; hipGraphNode_t errNode = nullptr;
                e = hipGraphExecUpdate(exec, graph, &errNode, &result);
                if (e == hipSuccess) s.graphUpdates++;
                else { (void)hipGetLastError(); (void)hipGraphExecDestroy(exec); exec = nullptr; e = hipSuccess; }    // topology changed (a pass toggled): rebuild
            }
            if (e == hipSuccess && !exec) { e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0); if (e == hipSuccess) s.graphRebuilds++; }
            if (e == hipSuccess) e = hipGraphLaunch(exec, (hipStream_t)jobSync->m_stream);
            if (e == hipSuccess) e = hipEventRecord(s.graphDone[gi], (hipStream_t)jobSync->m_stream);
            if (e == hipSuccess) s.graphFrames++;
            s.graphExec = exec;
        }
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {      // the frame is lost for replay: report, and issue every later frame directly
            s.graphStatus = (int)e; (void)hipGetLastError();
            if (jobSync->m_firstError == INCR_OK) { jobSync->m_firstError = INCR_ERR_HIP; jobSync->m_firstErrorMessage = std::string("HIP graph replay failed: ") + hipGetErrorString(e); }
        }
    }
    transient_release_frame(s, (hipStream_t)frameStream, false);      // back to the pool with this frame's fence (recorded on the real stream, behind the frame)
    s.hostSecondsLastFrame = std::chrono::duration<double>(std::chrono::steady_clock::now() - hostT0).count();
    s.frameIndex++;
    s.lastStatus = jobSync->m_firstError;
    if (jobSync->m_firstError != INCR_OK) (void)0;       // the message is in jobSync->m_firstErrorMessage
    return jobSync->m_firstError;
}

void ResetPathtraceAccumulation(State& s) { s.pathtraceCurrentSampleIndex = 1; }   // renderer.cpp:2080-2083

// Tonemap of the whole frame after the row bands were gathered (multi-GPU): renderer.cpp:2029-2036 on the presenting rank
int Tonemap(State& s, const FConfig& c) {
    if (!s.lastHdr) return INCR_ERR_INVALID_ARGUMENT;
    RenderJob::TonemapPass::Desc tonemapDesc = {};
    tonemapDesc.source = s.lastHdr; tonemapDesc.target = &s.backBuffer; tonemapDesc.renderConfig = c;
    RenderJob::Result r = RenderJob::TonemapPass::Execute(&s.jobSync, tonemapDesc);
    return r.wait();
}

} // namespace Renderer

// =====================================================================================================================
// C exports: the shape of the DLL interface (inc/demo-interface.h:10-16 Initialize / Render / Teardown), headless
// =====================================================================================================================
struct incr_renderer {
    Renderer::State state;
    FScene scene;
    FConfig config;
    std::string error;
};

extern "C" {

int incr_renderer_create(incr_context* ctx, incr_stream stream, uint32_t res_x, uint32_t res_y, const incr_renderer_scene* scene,
                         const incr_renderer_config* config, incr_renderer** out) {
    if (!ctx || !scene || !scene->scene || !out || !res_x || !res_y) return INCR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    incr_renderer* r = new (std::nothrow) incr_renderer();
    if (!r) return INCR_ERR_OUT_OF_MEMORY;
    r->scene.m_incrScene = const_cast<incr_scene*>(scene->scene);
    r->scene.m_directionalLightIndex = scene->directional_light_index;
    r->scene.m_punctualLightCount = scene->punctual_light_count;
    r->scene.m_meshletCount = scene->meshlet_count;
    r->scene.m_skylightSh9 = scene->skylight_sh9_rgba;
    r->scene.m_skylightEnvmap = scene->prefiltered_envmap; r->scene.m_envmapRes = scene->envmap_res; r->scene.m_envmapMips = scene->envmap_mips;
    r->scene.m_envBRDF.m_devicePtr = const_cast<uint32_t*>(scene->env_brdf);
    r->scene.m_envBrdfWidth = scene->env_brdf_width; r->scene.m_envBrdfHeight = scene->env_brdf_height;
    incr_renderer_set_config(r, config);
    r->scene.m_primitiveCount = scene->primitive_count;
    // draw lists sized for either path, so FConfig::UseMeshlets can change between frames (renderer.cpp:1375-1377 numDraws)
    const int rc = Renderer::Initialize(r->state, ctx, stream, res_x, res_y, std::max(scene->meshlet_count, scene->primitive_count), r->config);
    if (rc) { Renderer::Teardown(r->state); delete r; return rc; }
    *out = r;
    return INCR_OK;
}
void incr_renderer_destroy(incr_renderer* r) { if (!r) return; Renderer::Teardown(r->state); delete r; }

int incr_renderer_set_config(incr_renderer* r, const incr_renderer_config* c) {
    if (!r) return INCR_ERR_INVALID_ARGUMENT;
    if (!c) { r->config = FConfig(); return INCR_OK; }
    FConfig& f = r->config;
    f.PathTrace = c->path_trace != 0; f.Viewmode = c->viewmode; f.EnableTAA = c->enable_taa != 0; f.FrustumCulling = c->frustum_culling != 0;
    f.EnableDirectLighting = c->enable_direct_lighting != 0; f.EnableSkyLighting = c->enable_sky_lighting != 0; f.EnableDiffuseIBL = c->enable_diffuse_ibl != 0;
    f.Exposure = c->exposure; f.SkyBrightness = c->sky_brightness; f.CameraNearPlane = c->camera_near_plane;
    f.LightClusterDimX = (int)c->light_cluster_dim[0]; f.LightClusterDimY = (int)c->light_cluster_dim[1]; f.LightClusterDimZ = (int)c->light_cluster_dim[2];
    f.MaxLightsPerCluster = (int)c->max_lights_per_cluster; f.ClusterDepthExtent = c->cluster_depth_extent;
    f.MaxSampleCount = c->max_sample_count; f.Pathtracing_CameraAperture = c->camera_aperture; f.Pathtracing_CameraFocalLength = c->camera_focal_length;
    f.Turbidity = c->turbidity; f.EnvSkyMode = c->env_sky_mode; f.StorageRounding = c->storage_rounding;
    f.EnableSpecularIBL = c->enable_specular_ibl != 0;
    f.UseMeshlets = c->use_primitives == 0;
    f.EnableHBAO = c->enable_hbao != 0; f.UseBentNormals = c->use_bent_normals != 0;
    f.ForwardLighting = c->forward_lighting != 0;
    f.FusePasses = c->unfused_passes == 0; f.RecordThreads = c->record_threads > 0 ? (int)std::min(c->record_threads, 16) : 0; f.UseGraph = c->use_graph != 0;
    return INCR_OK;
}

int incr_renderer_render(incr_renderer* r, const incr_view* view, const incr_view* culling_view) {
    if (!r || !view) return INCR_ERR_INVALID_ARGUMENT;
    Renderer::FRenderState rs;
    rs.m_config = r->config; rs.m_scene = &r->scene; rs.m_resX = r->state.resX; rs.m_resY = r->state.resY;
    auto fill = [](FView& f, const incr_view& v) {
        f.m_viewTransform = v.view_transform; f.m_projectionTransform = v.projection_transform;
        f.m_position = {v.position[0], v.position[1], v.position[2]}; f.m_right = {v.right[0], v.right[1], v.right[2]};
        f.m_up = {v.up[0], v.up[1], v.up[2]}; f.m_look = {v.look[0], v.look[1], v.look[2]}; f.m_fov = v.fov;
    };
    fill(rs.m_view, *view); fill(rs.m_cullingView, culling_view ? *culling_view : *view);
    if (culling_view) rs.m_config.FreezeCulling = true;
    const int rc = Renderer::Render(r->state, rs);
    if (rc) r->error = r->state.jobSync.m_firstErrorMessage;
    return rc;
}
double incr_renderer_tick(incr_renderer* r, float dt) { if (!r) return 0.0; if (dt == dt && dt > 0.f) r->state.clockSeconds += (double)dt; return r->state.clockSeconds; }
void incr_renderer_reset_accumulation(incr_renderer* r) { if (r) Renderer::ResetPathtraceAccumulation(r->state); }
int incr_renderer_set_band(incr_renderer* r, uint32_t row_begin, uint32_t row_end) {
    if (!r || row_end < row_begin) return INCR_ERR_INVALID_ARGUMENT;
    if (row_end > r->state.resY + Renderer::kBandSlackRows) return INCR_ERR_INVALID_ARGUMENT;
    r->state.bandY0 = row_begin; r->state.bandY1 = row_end;
    return INCR_OK;
}
int incr_renderer_tonemap(incr_renderer* r) { return r ? Renderer::Tonemap(r->state, r->config) : INCR_ERR_INVALID_ARGUMENT; }
const char* incr_renderer_last_error(const incr_renderer* r) { return r ? r->error.c_str() : ""; }

// renderer.cpp:2064-2076 + Renderer::GetRenderStats: the statistics of the last frame, read back after the stream has drained
int incr_renderer_get_stats(incr_renderer* r, incr_render_stats* out_render_stats, incr_ray_stats* out_ray_stats) {
    if (!r) return INCR_ERR_INVALID_ARGUMENT;
    Renderer::State& s = r->state;
    int rc = incr_stream_synchronize(s.ctx, s.jobSync.m_stream);
    if (!rc && out_render_stats) rc = incr_buffer_download(s.ctx, out_render_stats, s.renderStatsBuffer.m_devicePtr, sizeof *out_render_stats, s.jobSync.m_stream);
    if (!rc && out_ray_stats) rc = incr_buffer_download(s.ctx, out_ray_stats, s.rayStatsBuffer.m_devicePtr, sizeof *out_ray_stats, s.jobSync.m_stream);
    if (rc) r->error = incr_last_error();
    return rc;
}

int incr_renderer_get_frame_info(incr_renderer* r, double* host_seconds, uint64_t* graph_frames, uint64_t* graph_updates, uint64_t* graph_rebuilds) {
    if (!r) return INCR_ERR_INVALID_ARGUMENT;
    if (host_seconds) *host_seconds = r->state.hostSecondsLastFrame;
    if (graph_frames) *graph_frames = r->state.graphFrames;
    if (graph_updates) *graph_updates = r->state.graphUpdates;
    if (graph_rebuilds) *graph_rebuilds = r->state.graphRebuilds;
    return INCR_OK;
}

int incr_renderer_set_profiling(incr_renderer* r, int enable) {
    if (!r) return INCR_ERR_INVALID_ARGUMENT;
    r->state.jobSync.m_profile = enable != 0;
    return INCR_OK;
}
// GPU milliseconds of every pass of the last profiled frame (direct issue only: a captured frame has no events)
int incr_renderer_get_pass_times(incr_renderer* r, char (*names)[32], float* ms, uint32_t capacity, uint32_t* count) {
    if (!r || !count) return INCR_ERR_INVALID_ARGUMENT;
    RenderJob::Sync& js = r->state.jobSync;
    *count = 0;
    if (js.m_marks.size() < 2) return INCR_OK;
    if (hipEventSynchronize((hipEvent_t)js.m_marks.back().second) != hipSuccess) return INCR_ERR_HIP;
    for (size_t i = 1; i < js.m_marks.size() && *count < capacity; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, (hipEvent_t)js.m_marks[i - 1].second, (hipEvent_t)js.m_marks[i].second) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (names) { strncpy(names[*count], js.m_marks[i].first.c_str(), 31); names[*count][31] = 0; }
        if (ms) ms[*count] = t;
        ++*count;
    }
    return INCR_OK;
}

void* incr_renderer_surface(incr_renderer* r, int which) {
    if (!r) return nullptr;
    Renderer::State& s = r->state;
    switch (which) {
    case INCR_SURFACE_BACK_BUFFER: return s.backBuffer.m_devicePtr;
    case INCR_SURFACE_HDR_RASTER: return s.hdrRasterSceneColor.m_devicePtr;
    case INCR_SURFACE_HDR_RAYTRACE: return s.hdrRaytraceSceneColor.m_devicePtr;
    case INCR_SURFACE_PATHTRACE_HISTORY: return s.pathtraceHistoryBuffer.m_devicePtr;
    case INCR_SURFACE_TAA_ACCUMULATION: return s.taaAccumulationBuffer[(s.frameIndex + 1) & 1].m_devicePtr;   // written by the last frame
    case INCR_SURFACE_DEPTH: return s.depthBuffer.m_devicePtr;
    case INCR_SURFACE_VISIBILITY: return s.visBuffer.m_devicePtr;
    case INCR_SURFACE_GBUFFER_BASECOLOR: return s.gbuffer_basecolor.m_devicePtr;
    case INCR_SURFACE_GBUFFER_NORMALS: return s.gbuffer_normals.m_devicePtr;
    case INCR_SURFACE_GBUFFER_METALLIC_ROUGHNESS_AO: return s.gbuffer_metallicRoughnessAo.m_devicePtr;
    case INCR_SURFACE_RENDER_STATS: return s.renderStatsBuffer.m_devicePtr;
    case INCR_SURFACE_RAY_STATS: return s.rayStatsBuffer.m_devicePtr;
    case INCR_SURFACE_BATCH_COUNTS: return s.batchCountsBuffer.m_devicePtr;
    case INCR_SURFACE_AMBIENT_OCCLUSION: return s.aoBuffer.m_devicePtr;
    case INCR_SURFACE_BENT_NORMALS: return s.bentNormalsBuffer.m_devicePtr;
    default: return nullptr;
    }
}

} // extern "C"
