/*
 * incr.h -- C ABI of libincr_hip.so: the MI355X-native replacement for the D3D12 dispatches of
 * Incarnation-Renderer's render loop.  One entry point per reference render pass; each takes a POD
 * `incr_<pass>_desc` that mirrors the pass's `RenderJob::<Pass>::Desc` (raw device pointers instead of
 * bindless descriptor indices, scalars copied from FConfig) and a HIP stream.  All GPU work of a
 * frame is enqueued on that one stream in call order -- the reference's execution order
 * (render-jobs/job-sync.h:27-61: token-ordered submission on one DIRECT queue).
 *
 * ONE FRAME IN FLIGHT PER CONTEXT: the path-state workspace of incr_pathtrace and the queues / depth surface of
 * incr_visibility are owned by the context, not by the stream.  All calls of a context must be ordered on one stream (or
 * ordered against each other by the caller); two frames of one context on two streams at the same time overwrite each
 * other's state.  Use one context per concurrent frame.  (The reference has one DIRECT queue and the same property.)
 *
 * Error convention: every function returns 0 on success, a negative incr_status otherwise, and never
 * aborts or throws (the reference has no error returns: inc/common.h:113-148).  incr_last_error() gives
 * the message for the calling thread.  There is NO CPU fallback: without a HIP device every launch
 * entry point returns INCR_ERR_NO_DEVICE.
 *
 * Citations are relative to /root/reference/source/demo-dll/.
 */
#ifndef INCR_H
#define INCR_H

#include "incr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum incr_status {
    INCR_OK = 0,
    INCR_ERR_INVALID_ARGUMENT = -1,
    INCR_ERR_NO_DEVICE = -2,
    INCR_ERR_HIP = -3,
    INCR_ERR_OUT_OF_MEMORY = -4,
    INCR_ERR_NOT_INITIALIZED = -5
} incr_status;

typedef struct incr_context incr_context;   /* one per process/GPU; stands in for RenderBackend12 (inc/backend-d3d12.h:478-536) */
typedef struct incr_scene incr_scene;       /* device-resident FScene (inc/scene.h:72-159) */
typedef void* incr_stream;                  /* hipStream_t; NULL = the default stream */

/* ---- lifetime: shape of the DLL exports Initialize/Teardown (inc/demo-interface.h:10-16), headless ---- */
int  incr_initialize(int device_index, incr_context** out_ctx);
int  incr_teardown(incr_context* ctx);
const char* incr_last_error(void);
const char* incr_version(void);
int  incr_device_count(void);               /* number of HIP devices visible (0 on a CPU-only box) */

/* ---- resource layer: RenderBackend12::CreateNewShaderBuffer / CreateNewShaderSurface stand-ins ---- */
int  incr_buffer_create(incr_context* ctx, uint64_t bytes, void** out_device_ptr);
int  incr_buffer_destroy(incr_context* ctx, void* device_ptr);
int  incr_buffer_upload(incr_context* ctx, void* device_dst, const void* host_src, uint64_t bytes, incr_stream stream);
int  incr_buffer_download(incr_context* ctx, void* host_dst, const void* device_src, uint64_t bytes, incr_stream stream);
int  incr_buffer_clear(incr_context* ctx, void* device_dst, uint32_t value32, uint64_t bytes, incr_stream stream);
int  incr_stream_synchronize(incr_context* ctx, incr_stream stream);

/* ---- scene: FScene::CreateGpuGeometryBuffers + CreateAccelerationStructures (src/scene.cpp:480-848) ---- */
int  incr_scene_create(incr_context* ctx, const incr_scene_desc* desc, incr_scene** out_scene);
int  incr_scene_destroy(incr_context* ctx, incr_scene* scene);
/* RenderJob::UpdateTLASPass::Execute (render-jobs/update-tlas.inl:3-101): re-derive the world-space
 * acceleration structure when mesh transforms / visibility / root transform changed.  Pass NULL to keep a field. */
int  incr_update_bvh(incr_context* ctx, incr_scene* scene, const incr_matrix* mesh_transforms,
                     const uint32_t* mesh_visibility, const incr_matrix* root_transform, incr_stream stream);
/* FScene::m_sunDir (scene.cpp:1594-1655 UpdateSunDirection) -- feeds the dynamic-sky environment. */
int  incr_scene_set_sun_dir(incr_context* ctx, incr_scene* scene, const float sun_dir[3]);
/* MaxAnisotropy of the raster passes' material sampler (incr_scene_desc::max_anisotropy) for the frames issued from now on: 1 .. 16, 0 = 16. */
int  incr_scene_set_max_anisotropy(incr_context* ctx, incr_scene* scene, uint32_t max_anisotropy);
/* The device-resident mip chain of scene texture `texture_index` (all levels, level 0 first; level l is max(w >> l, 1) x
 * max(h >> l, 1) RGBA8 texels): sizes, and a copy to host memory (tests compare it with the oracle's chain). */
int  incr_scene_texture_info(incr_context* ctx, const incr_scene* scene, uint32_t texture_index, uint32_t* width, uint32_t* height,
                             uint32_t* num_mips, uint64_t* chain_texels);
int  incr_scene_texture_download(incr_context* ctx, const incr_scene* scene, uint32_t texture_index, void* host_dst, uint64_t bytes);
/* Introspection used by tests and the bench (sizes of the flattened acceleration structure). */
typedef struct incr_scene_info {
    uint64_t num_triangles, num_bvh_nodes, bvh_bytes, triangle_bytes, shading_bytes, texture_bytes;
    uint32_t num_instances, bvh_max_depth;
} incr_scene_info;
int  incr_scene_get_info(incr_context* ctx, const incr_scene* scene, incr_scene_info* out_info);

/*
 * Framebuffer sharding: the frame is cut into row blocks of `row_block_height` rows;
 * this call renders blocks row_block_first, row_block_first + row_block_stride, ... and stores them
 * COMPACTLY (local row = k*row_block_height + r) in buffers of local_rows = incr_shard_local_rows() rows.
 * row_block_height == 0 means "the whole frame" (one rank).  Sampler keys always use the FULL frame
 * (pixelIdx = y*res_x + x, numPixels = res_x*res_y; shaders/raytracing/common.hlsli:10-11), so a shard is
 * bit-identical to the same pixels of a single-GPU frame.
 */
typedef struct incr_shard {
    uint32_t row_block_height;
    uint32_t row_block_first;
    uint32_t row_block_stride;
} incr_shard;
uint32_t incr_shard_local_rows(uint32_t res_y, const incr_shard* shard);

/* Ray statistics (SURVEY.md 8d "Ray count"): every traversal query, by category.  The kernels count into striped counters owned
 * by the context and fold them into the caller's struct at the end of the call (one frame in flight per context, see the top of
 * this file). */
typedef struct incr_ray_stats {
    uint64_t camera_rays;       /* TraceRay from rgsMain                      (pathtracing.hlsl:117) */
    uint64_t bounce_rays;       /* TraceRay from chsMain                      (pathtracing.hlsl:265) */
    uint64_t sky_shadow_rays;   /* RayQuery in GetSkyRadiance                 (lighting/common.hlsli:161-174) */
    uint64_t light_shadow_rays; /* RayQuery in GetDirectRadiance              (lighting/common.hlsli:102-115) */
    uint64_t shaded_hits;       /* chsMain invocations */
    uint64_t reserved[3];
} incr_ray_stats;

/*
 * RenderJob::PathTracing::Execute (render-jobs/path-tracing.inl:17-299) == clear target, DispatchRays(W,H,1)
 * (rgsMain/chsMain/ahsMain/ms* of shaders/raytracing/pathtracing.hlsl), then the 16x16 integrate dispatch
 * (shaders/raytracing/pathtrace-integrate.hlsl).  Desc mirrors path-tracing.inl:3-15 + the FConfig fields read.
 */
typedef struct incr_pathtrace_desc {
    const incr_scene* scene;
    void*    target_rgba16f;         /* device, local_rows*res_x*8 B   (Desc::targetBuffer, RGBA16F)           */
    void*    history_r11g11b10;      /* device, local_rows*res_x*4 B   (Desc::historyBuffer, R11G11B10_FLOAT)  */
    uint32_t res_x, res_y;           /* FULL frame resolution (Desc::resX/resY)                                */
    incr_shard shard;
    uint32_t current_sample_index;   /* Desc::currentSampleIndex: 1 .. MaxSampleCount-1 (renderer.cpp:23,1681-1699) */
    uint32_t max_sample_count;       /* FConfig::MaxSampleCount (256) -> sqrtSampleCount = (uint)sqrt            */
    incr_view view;                  /* Desc::view: viewTransform + projectionTransform                          */
    float    camera_aperture;        /* FConfig::Pathtracing_CameraAperture (0.01)                               */
    float    camera_focal_length;    /* FConfig::Pathtracing_CameraFocalLength (7)                               */
    int32_t  env_sky_mode;           /* FConfig::EnvSkyMode: 1 = dynamic sky (msDynamicSky); 0 = envmap (unsupported first cut) */
    int32_t  direct_lighting;        /* FConfig::EnableDirectLighting -> DIRECT_LIGHTING macro                   */
    int32_t  viewmode;               /* FConfig::Viewmode -> VIEWMODE macro (0..13; 1 lighting only, 2-5 first-hit material views, 7 mirror bounce, others = the normal path) */
    float    turbidity;              /* FConfig::Turbidity (2)                                                   */
    float    sky_brightness;         /* FConfig::SkyBrightness (25000)                                           */
    int32_t  storage_rounding;       /* 0 = round-toward-zero typed-UAV stores (D3D11.3 spec 3.2.3.3), 1 = nearest-even */
    incr_ray_stats* ray_stats;       /* device, optional (NULL = do not count); ACCUMULATED, caller clears        */
    /* GlobalCbLayout::envmapTextureIndex = FScene::m_skylight.m_envmapTextureIndex (path-tracing.inl:203): the prefiltered
     * sky cube that chsMain samples at level 0 for the sky next-event estimate (pathtracing.hlsl:240-243).  Device pointer to
     * all levels packed (incr_sky_ibl_info::envmap_r11g11b10).  NULL = evaluate 0.04 * Preetham(sun) analytically instead
     * (what level 0 of that cube approximates; DESIGN.md deviation Q15). */
    const uint32_t* envmap_r11g11b10;
    uint32_t envmap_res, envmap_mips;
    /* Not in the reference (which renders one sample per displayed frame, renderer.cpp:1681-1699): samples
     * current_sample_index .. current_sample_index + num_samples - 1 of every pixel traced as ONE wavefront (0 and 1 = one sample).
     * The history buffer ends up exactly as after num_samples successive calls -- the samples of a pixel are integrated one
     * after the other in sample order, each step re-quantised to R11G11B10 (pathtrace-integrate.hlsl:16-31) -- and the target
     * holds the last sample.  What it buys: a 1/8 framebuffer shard with 8 samples in flight launches a full frame's worth of
     * rays per traversal kernel (8-GPU runs of configs 4 / 5), and a 64-spp frame pays the launch tails 8 times less often. */
    uint32_t num_samples;
    uint32_t reserved0;
    /* Test instrumentation, optional (NULL = off), device, 12 x uint32 per LOCAL pixel: for path vertex k = 0..3 of the pixel's path
     * { InstanceIndex() << 16 | GeometryIndex(), PrimitiveIndex(), shadow word } of its closest hit -- 0xffffffff, 0xffffffff, 0 for a
     * miss; entries of vertices the path never traced keep what the caller cleared them to.  Shadow word: bit 31 = the sky sample's
     * shadow query was issued and found nothing; bits 0..30 = sum of (light loop index + 1) * 0x9E3779B1 over the lights whose
     * shadow query was issued and found nothing.  Lets the parity tests show that a pixel outside the colour tolerance is one whose
     * path took another triangle or whose shadow query flipped (a ray that grazes an edge), not a shading error.  num_samples <= 1. */
    uint32_t* debug_hit_chain;
} incr_pathtrace_desc;
int  incr_pathtrace(incr_context* ctx, const incr_pathtrace_desc* desc, incr_stream stream);
/*
 * Not in the reference: how a path-traced frame is issued.  The frame is cut into `pipelines` bands of 8-pixel tile rows, each
 * with its own path-state buffers and its own (library-owned) stream; the bands fork from `stream` and join it again inside
 * incr_pathtrace, so callers see one stream-ordered call.  Results are bit-identical for every count (1..8; 0 = the
 * default: 1 when the any-hit launches run on a side stream (incr_set_pathtrace_overlap, the default), else chosen per call from the number of
 * paths -- 2 for a frame, 4 for a framebuffer shard of fewer than 1.5 M pixels -- or the INCR_WF_PIPELINES environment variable read at incr_initialize).  One pipeline leaves the GPU waiting
 * for the longest rays of each of the eight traversal kernels of a frame; with two the other band's kernels run meanwhile.
 */
int  incr_set_pathtrace_pipelines(incr_context* ctx, uint32_t pipelines);
/*
 * Not in the reference: how the kernels of a pipeline are issued.  chsMain is a recursion -- shadow rays of a vertex, their light, then the continuation
 * ray -- but nothing a later vertex computes depends on the colour gathered so far (payload.color is only added to).  The library therefore runs a
 * pipeline's main chain camera rays -> shade -> closest hit -> shade -> ... without the shadow rays in it: they go into one queue, are traced by any-hit
 * launches on a SIDE stream beside the main chain, and the summation of every vertex (chsMain's additions in chsMain's order, bounce after bounce) is
 * deferred until they are done.  The image is bit-identical to the in-order issue.  Default on; INCR_WF_OVERLAP=0 at incr_initialize or enable = 0 issue
 * everything on the pipeline's one stream, any-hit launch and summation right behind the shading of their bounce: every launch then runs alone, which
 * is what per-kernel timing wants.
 */
int  incr_set_pathtrace_overlap(incr_context* ctx, int32_t enable);
/* Where the four summation launches (wf_accumulate_kernel) of a call run when the any-hit launches are on the side stream: 0 = at the end of the main stream, 1 = each right
 * behind the any-hit launch of its bounce on the side stream, -1 = by the number of paths (0 from 3 M paths up: a 4K frame; 1 below).  An issue order, not a result: every
 * placement leaves the same bits.  Not in the reference. */
int  incr_set_pathtrace_accumulate_placement(incr_context* ctx, int32_t placement);

/*
 * Per-kernel timing of the path tracer (stands in for the reference's PIX/Tracy GPU zones, inc/profiling.h:10-12):
 * between incr_profile_begin and incr_profile_end every incr_pathtrace call records HIP events around each of its
 * kernels on the stream it is launched on; incr_profile_end synchronises the device and sums the intervals.  With more
 * than one pipeline (above) intervals of different pipelines overlap in time: the sums price the launches, not the frame.
 */
typedef struct incr_pathtrace_profile {
    double generate_ms, extend_ms, shade_ms, shadow_gen_ms, shadow_ms, accumulate_ms;   /* summed over all recorded calls */
    uint32_t launches[5];            /* extend, shade, shadow_gen, shadow, accumulate launches that did work */
    uint32_t frames;                 /* recorded incr_pathtrace calls */
    uint32_t pipelines;              /* pipelines of the last recorded call */
    uint32_t overlap;                /* 1: the last recorded call ran its any-hit launches on the pipelines' side streams (incr_set_pathtrace_overlap, below): the
                                        launches of the two streams share the GPU while they are timed */
    uint32_t shadow_queue_overflows; /* (pipeline, call) pairs whose shadow-ray queue was too small: their excess rays were traced inline by the summation kernel --
                                        same image, slower.  The queue is sized from the scene's light grid (4 x (mean lights per occupied cell + 1) + 4 rays per path,
                                        at least 12); a count above 0 says a view is far denser than the scene's mean.  (round 4: reserved0) */
    uint64_t shadow_rays_inline;     /* the rays beyond the queue, over all recorded calls (round 4: the reserved extend_wait_ms; same size and offset) */
    uint64_t shadow_rays_asked;      /* shadow rays the recorded calls asked for (appended in round 5: the structure grew by 8 bytes) */
    uint64_t shadow_rays_answered;   /* ... of which the scene's sun cover map answered at their set-up: rays toward the directional light from origins with nothing
                                        above them are queued as null rays, their records stay "unoccluded" (appended in round 5: another 8 bytes) */
} incr_pathtrace_profile;
int  incr_profile_begin(incr_context* ctx);
int  incr_profile_end(incr_context* ctx, incr_pathtrace_profile* out_profile);

/*
 * RenderJob::TonemapPass::Execute (render-jobs/tonemap.inl:11-136; shaders/postprocess/tonemap.hlsl:30-59):
 * ldr = ACESFilm(hdr * Exposure(ev100)), written to an R8G8B8A8_UNORM_SRGB back buffer (sRGB encode on store).
 */
typedef struct incr_tonemap_desc {
    const void* hdr_r11g11b10;       /* device, rows*res_x*4 B  (Desc::source: path-trace history or TAA output) */
    void*    ldr_rgba8;              /* device, rows*res_x*4 B  (back buffer)                                     */
    uint32_t res_x, rows;
    float    exposure;               /* FConfig::Exposure (13.0); truncated to int ev100 (lighting/pbr.hlsli:78-81) */
    int32_t  viewmode;
} incr_tonemap_desc;
int  incr_tonemap(incr_context* ctx, const incr_tonemap_desc* desc, incr_stream stream);

/*
 * Self-test (not in the reference): the typed-UAV stores and loads of the render targets (R16G16B16A16_FLOAT, R11G11B10_FLOAT; D3D11.3
 * functional spec 3.2.3.3: round toward zero) run on the hardware's converters; this compares them with the bit-level reference
 * functions over every float32 pattern and every storage pattern.  out_mismatches[0..3] = f32 -> f16, f32 -> R11 / R10, f16 -> f32,
 * R11G11B10 -> f32; all zero on a correct build.  Blocks for about a second.
 */
int  incr_selftest_small_float(incr_context* ctx, uint64_t out_mismatches[4]);
/*
 * Self-test (not in the reference): shader divisions are evaluated as a * rcp(b) (HLSL a / b = DXIL FDiv, which GPUs lower to a reciprocal and a
 * multiplication) with rcp the CORRECTLY ROUNDED reciprocal -- on the device v_rcp_f32 + two fused Newton steps + v_div_fixup (csrc/pt_device.h rcp_cr),
 * in the CPU oracle 1.f / b.  This compares the device reciprocal with the IEEE quotient 1.f / x over every float32 pattern: out_mismatches[0] =
 * patterns with x and 1 / x both normal numbers (2^-126 <= |x| <= 2^126) where they differ, [1] = of those, by more than one ulp, [2] = differing
 * patterns outside that domain (the hardware reciprocal flushes denormals; no shader denominator lives there).  Blocks for about a second.
 */
int  incr_selftest_reciprocal(incr_context* ctx, uint64_t out_mismatches[3]);


/*
 * Inline ray queries against the scene's acceleration structure: the stand-in for DXR TraceRay / RayQuery
 * (call sites: shaders/raytracing/pathtracing.hlsl:117,265; shaders/lighting/common.hlsli:102-115).
 * All rays use RAY_FLAG_CULL_BACK_FACING_TRIANGLES like every call site of the reference.
 *   any_hit == 0: closest hit (non-opaque geometry runs the ahsMain alpha test); out_ids[i] = {hit, InstanceIndex,
 *                 GeometryIndex, PrimitiveIndex}, out_tuv[i] = {t, barycentric u, v}
 *   any_hit != 0: RayQuery + single Proceed(): out_ids[i].x = 1 iff any front-facing triangle intersects (tmin, tmax)
 * All pointers are DEVICE pointers: origins/dirs 3 floats per ray, out_ids 4 uint32 per ray, out_tuv 3 floats per ray.
 */
int  incr_trace_rays(incr_context* ctx, const incr_scene* scene, uint32_t num_rays, const float* origins, const float* dirs,
                     float tmin, float tmax, int any_hit, uint32_t* out_ids, float* out_tuv, incr_stream stream);

/* Host-only check of the scene packing + acceleration-structure build (no GPU needed): builds the flattened
 * BVH exactly as incr_scene_create does and verifies its invariants.  Returns INCR_OK and fills `info`
 * when every triangle is referenced by exactly one leaf and every node box encloses its subtree. */
int  incr_scene_validate_host(const incr_scene_desc* desc, incr_scene_info* out_info);

/* Un-interleave gathered shards: src is [stride][local_rows][res_x] texels of `texel_bytes`, dst the full frame. */
int  incr_unshard(incr_context* ctx, void* dst_full, const void* src_gathered, uint32_t res_x, uint32_t res_y,
                  uint32_t row_block_height, uint32_t num_shards, uint32_t texel_bytes, incr_stream stream);

/* =====================================================================================================
 * Raster path: visibility buffer + clustered deferred shading.
 * Frame order (src/renderer.cpp:1711-2036): batch cull -> light cull -> visibility -> g-buffer -> sky lighting ->
 * direct lighting -> clustered lighting -> TAA resolve -> tonemap.  All surfaces are linear res_x*res_y arrays in HBM.
 * ===================================================================================================== */

/*
 * MeshUtils::FixupMeshes (src/mesh-utils.cpp:560-638): tangents for a primitive whose material has a normal map and whose attributes
 * lack TANGENT -- MikkTSpace's genTangSpaceDefault (triangles, angular threshold 180 degrees) through callbacks that read POSITION /
 * NORMAL / TEXCOORD_0 by index and store float4(tangent, sign) at the indexed vertex (:28-112: of the faces around a vertex the last
 * in face order decides).  Host function.  positions / normals: vertex_count x 3 floats, uvs: vertex_count x 2, indices:
 * triangle_count x 3 (widened to 32 bits); out_tangents: vertex_count x 4 floats (0 where no face references the vertex).
 */
int      incr_generate_tangents(const float* positions, const float* normals, const float* uvs, const uint32_t* indices,
                                uint32_t vertex_count, uint32_t triangle_count, float* out_tangents);
/*
 * The image decode of the glTF load (src/scene.cpp:40, 98-117: tinygltf::LoadImageData = stb_image, four components requested): JPEG files through the
 * integer pipeline stb_image publishes -- baseline and progressive Huffman, the 12-bit "islow" IDCT, the 3:1 chroma up-sampling filters, 20-bit fixed-point
 * YCbCr -> RGB -- so that texel BYTES are the reference's (libjpeg-turbo rounds 1-2 codes differently).  Host function.  Call with rgba = NULL to get the
 * size, then with width x height x 4 bytes.  One- and three-component 8-bit files; anything else (CMYK, arithmetic coding, 12 bit) is INCR_ERR_INVALID_ARGUMENT.
 * PNG needs no counterpart: it is lossless.
 */
int      incr_decode_jpeg(const uint8_t* bytes, size_t size, int32_t* width, int32_t* height, uint8_t* rgba, size_t capacity);
/* MeshUtils::Meshletize (src/mesh-utils.cpp:640-835) as called by FScene::GenerateMeshlets (src/scene.cpp:503-560):
 * HOST-side load-time utility.  positions: vertex_count records of `position_stride_bytes` (>= 12) starting with xyz.
 * The returned set holds FGpuMeshlet ranges + bounding spheres (accessor/mesh/material fields are -1 for the caller to
 * fill), the unique-vertex index list and the 10:10:10 packed triangles. */
typedef struct incr_meshlet_set incr_meshlet_set;
int      incr_meshletize(const uint32_t* indices, uint32_t index_count, const float* positions, uint32_t vertex_count,
                         uint32_t position_stride_bytes, uint32_t max_verts, uint32_t max_prims, incr_meshlet_set** out_set);
uint32_t incr_meshlet_set_counts(const incr_meshlet_set* set, uint32_t* num_vertex_indices, uint32_t* num_triangles);
int      incr_meshlet_set_copy(const incr_meshlet_set* set, incr_gpu_meshlet* meshlets, uint32_t* vertex_indices, uint32_t* packed_triangles);
void     incr_meshlet_set_free(incr_meshlet_set* set);

/* The FViewConstants fields the raster passes read (inc/gpu-shared-types.h:167-192) plus the per-pass matrices that
 * light-culling.inl:111-112, sky-lighting.inl:124 and renderer.cpp:2007-2017 derive on their own (they associate the
 * jitter translation differently, so they are NOT bitwise the same matrices). */
typedef struct incr_view_constants {
    incr_matrix view_transform;
    incr_matrix proj_transform;              /* P * T(jitter)                                  renderer.cpp:1643 */
    incr_matrix view_proj_transform;         /* V * (P * T)                                    renderer.cpp:1644 */
    incr_matrix inv_view_proj_transform;     /* (V * (P * T))^-1                               renderer.cpp:1653 */
    incr_matrix inv_proj_transform;          /* (P * T)^-1                                     renderer.cpp:1656 */
    incr_matrix cull_view_proj_transform;    /* Vcull * (P * T)                                renderer.cpp:1657 */
    incr_matrix prev_view_proj_transform;    /* last frame's UNJITTERED V * P                  renderer.cpp:1655,2025 */
    incr_matrix lightcull_proj_transform;    /* Pcull * T                                      light-culling.inl:111 */
    incr_matrix lightcull_inv_view_proj;     /* ((Vcull * Pcull) * T)^-1                       light-culling.inl:112 */
    incr_matrix sky_inv_view_proj;           /* ((V * P) * T)^-1                               sky-lighting.inl:124 */
    incr_matrix unjittered_view_proj;        /* V * P: becomes prev_view_proj_transform of the next frame */
    incr_matrix taa_inv_view_proj;           /* (V * P)^-1                                     renderer.cpp:2017 */
    float    eye_pos[3];
    float    near_plane;                     /* FConfig::CameraNearPlane (1.0) */
    uint32_t res_x, res_y;
    float    jitter[2];
    uint32_t scissor_y0, scissor_y1;         /* rows [y0, y1) a pass touches (sort-first multi-GPU row bands); 0,0 = the whole frame */
    float    camera_up[3];                   /* FViewConstants::m_cameraUpVector = view.m_up                  renderer.cpp:1660 */
    float    pad0;
    incr_matrix envmap_inv_parallax_view_proj; /* ((V with translation zeroed) * P)^-1               environmentmap.inl:127-133 */
} incr_view_constants;
/* renderer.cpp:126-131: jitter = 2 * (Halton(i,2), Halton(i,3)) - 0.5) / res for i = frame_index % 16. */
void incr_pixel_jitter(uint32_t frame_index, uint32_t res_x, uint32_t res_y, float out_jitter[2]);
/* Host helper (no GPU): fills every matrix above.  culling_view NULL = view (FreezeCulling off); prev_view_proj NULL = identity. */
int  incr_make_view_constants(const incr_view* view, const incr_view* culling_view, const float jitter[2],
                              const incr_matrix* prev_view_proj, float near_plane, uint32_t res_x, uint32_t res_y,
                              incr_view_constants* out);

/* One entry of the culled draw lists: the live fields of FIndirectDrawWithRootConstants (batch-culling.hlsl:150-155). */
typedef struct incr_draw_cmd { uint32_t id; uint32_t vertex_count; } incr_draw_cmd;

/*
 * RenderJob::BatchCullingPass::Execute (render-jobs/batch-culling.inl; shaders/culling/batch-culling.hlsl cs_meshlet_cull_main):
 * per-meshlet visibility + Gribb-Hartmann frustum test against cull_view_proj_transform, split by FMaterial::m_doubleSided.
 * The reference appends with InterlockedAdd (any order); this implementation emits ASCENDING meshlet ids, one of
 * the reference's possible outcomes and the one the visibility pass's tie rule is defined on.
 */
typedef struct incr_batch_cull_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    incr_draw_cmd* default_list;       /* device, capacity num_meshlets   (batchArgsBuffer_Default)      */
    incr_draw_cmd* double_sided_list;  /* device, capacity num_meshlets   (batchArgsBuffer_DoubleSided)  */
    uint32_t* counts;                  /* device, 2 x u32: {default, double-sided}  (batchCountsBuffer)  */
    incr_render_stats* render_stats;   /* device, optional: m_culledObjects is stored                    */
    int32_t  frustum_culling;          /* FConfig::FrustumCulling */
    int32_t  use_primitives;           /* !FConfig::UseMeshlets: cs_primitive_cull_main, ids are primitive ids, vertex_count = index count */
} incr_batch_cull_desc;
int incr_batch_cull(incr_context* ctx, const incr_batch_cull_desc* desc, incr_stream stream);

/*
 * RenderJob::LightCullingPass::Execute (render-jobs/light-culling.inl; shaders/culling/light-culling.hlsl): one thread per
 * cluster tests every punctual scene light's range sphere against the cluster frustum.  light_grid[c] = {offset, count}
 * with offset = c * max_lights_per_cluster (the reference allocates offsets with an atomic, i.e. in any order; consumers
 * only follow (offset, count)).  Lists hold GLOBAL light indices in ascending scene-light order, as in the reference.
 */
typedef struct incr_light_cull_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    uint32_t cluster_dim[3];           /* FConfig::LightClusterDimX/Y/Z (16, 9, 24) */
    uint32_t max_lights_per_cluster;   /* FConfig::MaxLightsPerCluster (64), <= 128 */
    float    cluster_depth_extent;     /* FConfig::ClusterDepthExtent (200) */
    uint32_t* culled_light_lists;      /* device, max_lights_per_cluster * clusters u32 */
    incr_light_grid_data* light_grid;  /* device, clusters entries */
    uint32_t* culled_light_count;      /* device, 1 u32: total list entries */
    incr_render_stats* render_stats;   /* device, optional: m_culledLights is stored */
} incr_light_cull_desc;
int incr_light_cull(incr_context* ctx, const incr_light_cull_desc* desc, incr_stream stream);

/*
 * RenderJob::VisibilityPass::Execute (render-jobs/visibility-pass.inl; shaders/geo-raster/visibility-pass.hlsl
 * vs_meshlet_main / ps_meshlet_main): draws every command of the default list with back-face culling (CCW front)
 * and every command of the double-sided list without, depth test GREATER_EQUAL on reverse Z, alpha clip at 0.5,
 * vis = meshletId << 7 | triangle.  Clears: vis 0xFFFFF000, depth 0.  A software rasteriser on MI355X
 * (DESIGN.md "Visibility pass"): D3D top-left rule on 24.8 fixed-point positions, pixel centres at +0.5.
 */
typedef struct incr_visibility_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    const incr_draw_cmd* default_list;
    const incr_draw_cmd* double_sided_list;
    const uint32_t* counts;            /* device, from incr_batch_cull */
    uint32_t* vis_buffer;              /* device, res_x*res_y R32_UINT */
    float*    depth_buffer;            /* device, res_x*res_y D32_FLOAT */
    int32_t   use_primitives;          /* !FConfig::UseMeshlets: vs_primitive_main / ps_primitive_main, vis = primitiveId << 20 | triangle
                                          (at most 4096 primitives: the 12 bits of EncodePrimitiveVisibility, encoding.hlsli:4-11) */
} incr_visibility_desc;
int incr_visibility(incr_context* ctx, const incr_visibility_desc* desc, incr_stream stream);
/* Introspection for the bench: work-queue counters of the last incr_visibility call (synchronises the device):
 * [0] screen triangles after clip + cull, [1] band items of large triangles, [2] small triangles, [3] whole-triangle items. */
int incr_visibility_get_stats(incr_context* ctx, uint32_t out_counters[4]);

/*
 * RenderJob::GBufferComputePass::Execute (render-jobs/gbuffer-compute-pass.inl; shaders/geo-raster/gbuffer-compute.hlsl):
 * clears the colour target, then per pixel decodes the visibility id, interpolates attributes with screen-space
 * barycentrics, evaluates the material and writes the three G-buffers + emissive * 20000.
 */
typedef struct incr_gbuffer_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    const uint32_t* vis_buffer;
    uint32_t* gbuffer_basecolor;       /* device, R8G8B8A8_UNORM */
    uint32_t* gbuffer_normals;         /* device, R16G16_FLOAT (octahedral) */
    uint32_t* gbuffer_metallic_roughness_ao; /* device, R8G8B8A8_UNORM */
    uint32_t* color_target;            /* device, R11G11B10_FLOAT */
    int32_t  storage_rounding;         /* as incr_pathtrace_desc::storage_rounding */
    int32_t  use_primitives;           /* !FConfig::UseMeshlets -> USING_MESHLETS=0 (DecodePrimitiveVisibility, GetPrimitiveTriangleData) */
} incr_gbuffer_desc;
int incr_gbuffer(incr_context* ctx, const incr_gbuffer_desc* desc, incr_stream stream);

/*
 * RenderJob::GBufferRasterPass::Execute (render-jobs/gbuffer-raster-pass.inl; shaders/geo-raster/gbuffer-raster.hlsl): the
 * decals of the scene (incr_scene_desc::decal_primitives) drawn in order over the G-buffers after incr_gbuffer.  Depth
 * test GREATER_EQUAL against the scene depth, no depth writes, no alpha clip; base colour blended SRC_ALPHA / INV_SRC_ALPHA
 * (colour and alpha channels alike, re-quantised to UNORM8 after every fragment), normals replaced only by decals with a
 * normal map, metallic / roughness / ao only by decals with a metallic-roughness map.  A scene without decals makes this a
 * no-op.  Here: one set-up kernel files every decal triangle in a fixed slot, one kernel per pixel tile walks the slots in
 * draw order -- the blend order is the reference's draw order, whatever the scheduling.
 */
typedef struct incr_gbuffer_decals_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    const float* depth_buffer;
    uint32_t* gbuffer_basecolor;
    uint32_t* gbuffer_normals;
    uint32_t* gbuffer_metallic_roughness_ao;
    int32_t  storage_rounding;
} incr_gbuffer_decals_desc;
int incr_gbuffer_decals(incr_context* ctx, const incr_gbuffer_decals_desc* desc, incr_stream stream);

/*
 * RenderJob::HBAO::Execute (render-jobs/hbao.inl:16-105; shaders/postprocess/hbao.hlsl): ray-traced horizon-based
 * ambient occlusion.  Per pixel 4 azimuthal slices x 2 horizons x a binary search of at most 4 any-hit ray queries
 * (TMin 0.1, TMax 1, back faces culled) against the scene BVH, integrated against the projected normal; writes the
 * occlusion factor sky lighting multiplies by and the octahedral bent normal.  FConfig::EnableHBAO is off by default:
 * the frame then clears both targets to 1 (renderer.cpp:1816-1841 = ambient_occlusion NULL in incr_sky_lighting_desc).
 * cos / sin / acos are evaluated by fixed polynomials so the result does not depend on a math library.
 */
typedef struct incr_hbao_desc {
    const incr_scene* scene;
    const incr_view_constants* view;   /* inv_view_proj_transform, eye_pos, camera_up, resolution, scissor rows */
    const float*    depth_buffer;
    const uint32_t* gbuffer_normals;   /* device, R16G16_FLOAT (octahedral) */
    uint8_t*  ao_target;               /* device, R8_UNORM */
    uint32_t* bent_normal_target;      /* device, R16G16_FLOAT (octahedral) */
    incr_ray_stats* ray_stats;         /* device, optional: light_shadow_rays accumulated (the pass's ray queries) */
    int32_t  storage_rounding;
} incr_hbao_desc;
int incr_hbao(incr_context* ctx, const incr_hbao_desc* desc, incr_stream stream);

/* Inputs shared by the three deferred lighting passes. */
typedef struct incr_lighting_inputs {
    const incr_scene* scene;
    const incr_view_constants* view;
    uint32_t* color_target;            /* device, R11G11B10_FLOAT, read-modify-write (`colorTarget[xy] += radiance`) */
    const float*    depth_buffer;
    const uint32_t* gbuffer_basecolor;
    const uint32_t* gbuffer_normals;
    const uint32_t* gbuffer_metallic_roughness_ao;
    int32_t  lighting_only;            /* Viewmode::LightingOnly -> LIGHTING_ONLY macro */
    int32_t  storage_rounding;
} incr_lighting_inputs;

/* RenderJob::SkyLightingPass::Execute (render-jobs/sky-lighting.inl; shaders/lighting/sky-lighting.hlsl): SH9 diffuse IBL
 * + split-sum specular IBL (prefiltered environment cube x env-BRDF LUT, both from the precompute entry points below). */
typedef struct incr_sky_lighting_desc {
    incr_lighting_inputs in;
    const float* sh9_rgba;             /* device, 9 x float4: the skylight probe texture (RGBA32F 9x1); NULL = g_skylightProbeIndex -1 */
    const uint8_t* ambient_occlusion;  /* device, R8_UNORM; NULL = cleared to 1 (HBAO disabled, renderer.cpp:1818-1838) */
    float    sky_brightness;           /* FConfig::SkyBrightness */
    int32_t  diffuse_ibl;              /* FConfig::EnableDiffuseIBL */
    const uint32_t* prefiltered_envmap;/* device, R11G11B10_FLOAT cube, all levels packed (incr_sky_ibl_info::envmap_r11g11b10); NULL = g_envmapIndex -1 */
    uint32_t envmap_res, envmap_mips;
    const uint32_t* env_brdf;          /* device, R16G16_FLOAT LUT (incr_generate_env_brdf); NULL = g_envBrdfTextureIndex -1 */
    uint32_t env_brdf_width, env_brdf_height;
    int32_t  specular_ibl;             /* FConfig::EnableSpecularIBL */
    const uint32_t* bent_normals;      /* device, R16G16_FLOAT (octahedral) from incr_hbao; read when use_bent_normals; NULL = cleared to (1, 1) */
    int32_t  use_bent_normals;         /* FConfig::UseBentNormals -> USE_BENT_NORMALS (sky-lighting.inl:65): the SH9 lookup normal */
} incr_sky_lighting_desc;
int incr_sky_lighting(incr_context* ctx, const incr_sky_lighting_desc* desc, incr_stream stream);

/* RenderJob::DirectLightingPass::Execute (render-jobs/direct-lighting.inl; shaders/lighting/direct-lighting.hlsl): the sun,
 * with one ray-traced shadow query per lit pixel (lighting/common.hlsli:102-115). */
typedef struct incr_direct_lighting_desc {
    incr_lighting_inputs in;
    int32_t  directional_light_index;  /* FScene::GetDirectionalLight(): GLOBAL index, also used to index the transform list */
    incr_ray_stats* ray_stats;         /* device, optional: light_shadow_rays accumulated */
} incr_direct_lighting_desc;
int incr_direct_lighting(incr_context* ctx, const incr_direct_lighting_desc* desc, incr_stream stream);

/* RenderJob::ClusteredLightingPass::Execute (render-jobs/clustered-lighting.inl; shaders/lighting/clustered-lighting.hlsl). */
typedef struct incr_clustered_lighting_desc {
    incr_lighting_inputs in;
    const uint32_t* culled_light_lists;
    const incr_light_grid_data* light_grid;
    uint32_t cluster_dim[3];
    float    cluster_depth_extent;
    int32_t  requires_clear;           /* bRequiresClear: the colour target is cleared first when the scene has no sun (renderer.cpp:1906) */
} incr_clustered_lighting_desc;
int incr_clustered_lighting(incr_context* ctx, const incr_clustered_lighting_desc* desc, incr_stream stream);

/* SkyLightingPass + DirectLightingPass + ClusteredLightingPass (renderer.cpp:1843-1920) as ONE kernel: the G-buffers and
 * depth are read once per pixel and the colour target is read and written once.  Any of the three may be NULL (the pass is
 * skipped, as the reference skips it); the parts must address the same surfaces.  The colour is re-quantised to
 * R11G11B10_FLOAT between the parts, so the result is bit-identical to the three calls in that order (the unfused calls
 * stay available for per-pass comparison). */
typedef struct incr_deferred_lighting_desc {
    const incr_sky_lighting_desc* sky;
    const incr_direct_lighting_desc* direct;
    const incr_clustered_lighting_desc* clustered;
} incr_deferred_lighting_desc;
int incr_deferred_lighting(incr_context* ctx, const incr_deferred_lighting_desc* desc, incr_stream stream);

/* RenderJob::ForwardLightingPass::Execute (render-jobs/forward-pass.inl; shaders/geo-raster/forward-pass.hlsl), the
 * FConfig::ForwardLighting alternative to the deferred direct + clustered passes: clears the colour target to (0.8, 0.8, 1)
 * and depth to 0, draws every primitive of every visible mesh (no frustum culling, cull mode from the material, alpha test
 * clip(opacity - 0.5), depth GREATER_EQUAL with writes) and shades in the pixel shader: emissive * 20000 + every scene light
 * (sun with a ray-traced shadow) + SH9 diffuse + split-sum specular IBL.  Here: primitive visibility in scene order into a
 * library-owned id surface, then one shading kernel.  Viewmode 2..5 and 7 return material channels / reflections as the
 * shader does (its LightingOnly branch is dead code: `#if VIWEMODE == 1`).  At most 4096 primitives. */
typedef struct incr_forward_lighting_desc {
    const incr_scene* scene;
    const incr_view_constants* view;
    uint32_t* color_target;            /* device, R11G11B10_FLOAT */
    float*    depth_buffer;            /* device, cleared and written */
    const float* sh9_rgba;             /* as incr_sky_lighting_desc */
    const uint32_t* prefiltered_envmap; uint32_t envmap_res, envmap_mips;
    const uint32_t* env_brdf;          uint32_t env_brdf_width, env_brdf_height;
    int32_t  viewmode, direct_lighting, diffuse_ibl, specular_ibl;
    incr_ray_stats* ray_stats;         /* device, optional: light_shadow_rays accumulated */
    int32_t  storage_rounding;
} incr_forward_lighting_desc;
int incr_forward_lighting(incr_context* ctx, const incr_forward_lighting_desc* desc, incr_stream stream);

/* RenderJob::DynamicSkyPass::Execute (render-jobs/dynamic-sky.inl; shaders/environment-sky/dynamic-sky.hlsl): fills every
 * pixel at the far plane (depth <= 0.0001) with 1000 * Preetham sky radiance.  The reference shader looks the sky up along
 * the lat-long direction of the pixel's SCREEN uv (not the view ray); that is what this pass does too. */
typedef struct incr_dynamic_sky_desc {
    const incr_scene* scene;           /* m_sunDir */
    const incr_view_constants* view;   /* resolution + scissor rows */
    uint32_t* color_target;            /* device, R11G11B10_FLOAT */
    const float* depth_buffer;
    float    turbidity;                /* FConfig::Turbidity */
    int32_t  storage_rounding;
} incr_dynamic_sky_desc;
int incr_dynamic_sky(incr_context* ctx, const incr_dynamic_sky_desc* desc, incr_stream stream);

/* RenderJob::EnvironmentmapPass::Execute (render-jobs/environmentmap.inl; shaders/environment-sky/envmap.hlsl), the
 * background of EnvSkyMode::HDRI: every pixel at the far plane (depth <= 0.0001) becomes sky_brightness * envmap(view ray),
 * the ray taken through the translation-free view (view->envmap_inv_parallax_view_proj).  The cube is the scene's
 * m_skylight.m_envmapTextureIndex in the packed level layout of incr_sky_ibl_info::envmap_r11g11b10; it is sampled at
 * level 0 (the anisotropic sampler of the reference magnifies a 256^2 face at any display resolution). */
typedef struct incr_environmentmap_desc {
    const incr_view_constants* view;
    uint32_t* color_target;            /* device, R11G11B10_FLOAT */
    const float* depth_buffer;
    const uint32_t* envmap_r11g11b10;  /* device, cube, all levels packed */
    uint32_t envmap_res, envmap_mips;
    float    sky_brightness;           /* FConfig::SkyBrightness */
    int32_t  storage_rounding;
} incr_environmentmap_desc;
int incr_environmentmap(incr_context* ctx, const incr_environmentmap_desc* desc, incr_stream stream);

/* RenderJob::DebugVizPass::Execute (render-jobs/debug-visualization.inl; shaders/postprocess/debug-visualization.hlsl): when
 * FConfig::Viewmode is neither Normal (0) nor LightingOnly (1) a full-screen pass overwrites the HDR colour target with
 * Roughness 2, Metallic 3, BaseColor 4, ObjectIds 8 / TriangleIds 9 (Halton hue of the visibility id), Normalmap 10,
 * LightClusterSlices 11, AmbientOcclusion 12, BentNormals 13; the other modes write 0.  The mouse-pick store that feeds
 * the editor's highlight pass is UI and not part of this library. */
typedef struct incr_debug_viz_desc {
    const incr_view_constants* view;
    int32_t  viewmode;
    const uint32_t* vis_buffer;
    const uint32_t* gbuffer_basecolor;
    const uint32_t* gbuffer_normals;
    const uint32_t* gbuffer_metallic_roughness_ao;
    const float*    depth_buffer;
    const uint8_t*  ambient_occlusion; /* NULL = cleared to 1 */
    const uint32_t* bent_normals;      /* NULL = cleared to (1, 1) */
    uint32_t* color_target;            /* device, R11G11B10_FLOAT */
    uint32_t light_cluster_slices;     /* FConfig::LightClusterDimZ */
    int32_t  use_primitives;           /* USING_MESHLETS = 0 */
    int32_t  storage_rounding;
} incr_debug_viz_desc;
int incr_debug_visualization(incr_context* ctx, const incr_debug_viz_desc* desc, incr_stream stream);

/* RenderJob::TAAResolvePass::Execute (render-jobs/taa-resolve.inl; shaders/postprocess/taa-resolve.hlsl).  The reference
 * reads and writes ONE accumulation surface in place (a data race across pixels); here history is read from
 * `history_in` and the result written to `history_out` (they may not alias): the race-free reading of the shader. */
typedef struct incr_taa_resolve_desc {
    const incr_view_constants* view;   /* taa_inv_view_proj, prev_view_proj_transform, res */
    const uint32_t* hdr_scene_color;   /* device, R11G11B10_FLOAT */
    const uint32_t* history_in;        /* device, R11G11B10_FLOAT (ignored when history_index == 0) */
    uint32_t* history_out;
    const float* depth_buffer;
    uint32_t history_index;            /* frame index */
    int32_t  storage_rounding;
    /* optional TonemapPass fused into the resolve (render-jobs/tonemap.inl on the surface just written): when ldr_rgba8 is
     * not NULL every resolved texel is also tonemapped into it, bit-identical to incr_tonemap(history_out) afterwards */
    void*    ldr_rgba8;                /* device, res_x*res_y*4 B, or NULL */
    float    exposure;                 /* FConfig::Exposure */
    int32_t  viewmode;
} incr_taa_resolve_desc;
int incr_taa_resolve(incr_context* ctx, const incr_taa_resolve_desc* desc, incr_stream stream);

/* =====================================================================================================
 * Sky and IBL precompute: the producers of the envmap / skylight-probe / env-BRDF inputs.
 * ===================================================================================================== */
/* FScene::UpdateDynamicSky (src/scene.cpp:1657-1790): Preetham lat-long texture (x 0.04) -> 2x2 box mip chain ->
 * cube map per level -> GGX prefilter (1024 Hammersley samples, roughness = level / levels) -> SH9 projection + reduction.
 * The object owns the products (R11G11B10_FLOAT mip chains packed level after level; cubes as [level][face][y][x]);
 * pass *inout_ibl == NULL to create it, the same object again to refresh it after the sun moved. */
typedef struct incr_sky_ibl incr_sky_ibl;
typedef struct incr_sky_ibl_info {
    uint32_t cubemap_res, num_mips;             /* FConfig::EnvmapResolution (256), CalcMipCount(res, res) */
    const uint32_t* latlong_r11g11b10;          /* device: 2*res x res lat-long sky, all levels */
    const uint32_t* source_cube_r11g11b10;      /* device: unfiltered cube, all levels */
    const uint32_t* envmap_r11g11b10;           /* device: prefiltered cube = FScene::m_dynamicSkyEnvmap, all levels */
    const float*    sh9_rgba;                   /* device: 9 x float4 = FScene::m_dynamicSkySH */
    uint64_t latlong_texels, cube_texels;       /* texels over all levels (of the lat-long, of the prefiltered cube) */
    /* round 5 (appended: the fields above keep their offsets).  incr_cache_hdri: the lat-long is the decoded RGBA32F picture (latlong_r11g11b10 is NULL), the source
     * cube is latlong_height^2 with CalcMipCount(w, h, true) levels and the prefiltered cube half its size with one level fewer; for the dynamic sky these repeat the above. */
    const float*    latlong_rgba32f;            /* device: latlong_width x latlong_height x RGBA32F, all levels; NULL for the dynamic sky */
    uint32_t latlong_width, latlong_height;
    uint32_t source_cube_res, source_cube_mips;
} incr_sky_ibl_info;
int incr_update_dynamic_sky(incr_context* ctx, const incr_scene* scene, uint32_t envmap_resolution, float turbidity, int32_t storage_rounding,
                            incr_stream stream, incr_sky_ibl** inout_ibl);
/*
 * FTextureCache::CacheHDRI (src/demo.cpp:292-434), the light probe of EnvSkyMode::HDRI (FScene::ReloadEnvironment, src/scene.cpp:219-226): a Radiance .hdr lat-long
 * picture (DirectX::LoadFromHDRFile -> RGBA32F) -> GenerateMipMaps(TEX_FILTER_LINEAR) down to 4 x 4 -> ConvertLatlong2Cubemap at size = picture height ->
 * PrefilterCubemap at half that size with one level fewer (roughness = level / levels) -> ShEncode from the source picture.  `hdr_file` = the file's bytes (host);
 * the products are those of incr_update_dynamic_sky: envmap_r11g11b10 = FLightProbe::m_envmapTextureIndex, sh9_rgba = m_shTextureIndex.  2:1 pictures with a
 * power-of-two height.  incr_decode_hdr: the reader alone (host; rgba = NULL returns the size).
 */
int incr_cache_hdri(incr_context* ctx, const uint8_t* hdr_file, size_t size, int32_t storage_rounding, incr_stream stream, incr_sky_ibl** out_ibl);
int incr_decode_hdr(const uint8_t* bytes, size_t size, int32_t* width, int32_t* height, float* rgba, size_t capacity_floats);
int incr_sky_ibl_get_info(const incr_sky_ibl* ibl, incr_sky_ibl_info* out_info);
int incr_sky_ibl_destroy(incr_context* ctx, incr_sky_ibl* ibl);
/* Renderer::GenerateEnvBrdfTexture (src/renderer.cpp:169-245; split-sum-approx/brdf-integration.hlsl): R16G16_FLOAT LUT,
 * x = NoV, y = 1 - roughness, 1024 samples per texel.  out_rg16f: device, width*height u32. */
int incr_generate_env_brdf(incr_context* ctx, uint32_t width, uint32_t height, int32_t storage_rounding, void* out_rg16f, incr_stream stream);

/* =====================================================================================================
 * Texture ingest, GPU step: FScene::PrefilterNormalRoughnessTextures (src/scene.cpp:1157-1320;
 * shaders/content-pipeline/prefilter-normal-roughness.hlsl) -- vMF filtering of a normal map together with its
 * metallic-roughness map.  Level 0: normal map copied, metallic-roughness stored .bgra (metal in R, roughness in G).
 * Level k >= 1: every texel averages the vMF mean vectors of its 2^k x 2^k base-level footprint and converts the mean back
 * to (normal * 0.5 + 0.5, 0) and (average metalness, widened roughness, 0, 0).  Both outputs are R8G8B8A8_UNORM, levels
 * packed one after another (level k is (width >> k) x (height >> k), row-major); the reference's BC5 compression of the
 * read-back is not part of the path.  Blocks until the chain is written (the reference waits on its read-back here).
 * ===================================================================================================== */
typedef struct incr_prefilter_normal_roughness_desc {
    const uint8_t* normalmap;           /* device, width*height RGBA8 as decoded from the image (tangent-space normal in rgb) */
    const uint8_t* metallic_roughness;  /* device, width*height RGBA8 as decoded (glTF: roughness in G, metalness in B); same size (scene.cpp:1282) */
    uint32_t width, height;
    uint32_t num_mips;                  /* = incr_compressed_mip_count(width, height) */
    uint32_t reserved;                  /* 0: sums through a float4 pyramid (every base texel read once; levels >= 2 may differ from the shader's single
                                           running sum by float re-association, <= 1 LSB); 1: the shader's exact summation order (O(texels x levels)) */
    uint8_t* out_normalmap;             /* device, incr_mip_chain_texels(width, height, num_mips) * 4 bytes */
    uint8_t* out_metallic_roughness;    /* device, same size */
} incr_prefilter_normal_roughness_desc;
/* RenderUtils12::CalcMipCount(w, h, bCompressed = true) (inc/backend-d3d12.h:456-471): levels down to 4x4. */
uint32_t incr_compressed_mip_count(uint32_t width, uint32_t height);
uint64_t incr_mip_chain_texels(uint32_t width, uint32_t height, uint32_t num_mips);
int incr_prefilter_normal_roughness(incr_context* ctx, const incr_prefilter_normal_roughness_desc* desc, incr_stream stream);

/* =====================================================================================================
 * Frame level: the DLL exports Initialize / Render / Teardown (inc/demo-interface.h:10-16) over the C++ host mirror
 * of Renderer::Render (csrc/host/renderer.cpp: RenderJob::<Pass>::Execute + the path-trace and raster branches of
 * src/renderer.cpp:1676-2036).  One incr_renderer owns the frame's surfaces and the state Renderer keeps in statics
 * (s_pathtraceCurrentSampleIndex, s_prevViewProjectionTransform, jitter table, TAA accumulation).
 * ===================================================================================================== */
typedef struct incr_renderer incr_renderer;
typedef struct incr_renderer_scene {       /* what Renderer::Render asks FScene */
    const incr_scene* scene;
    int32_t  directional_light_index;      /* FScene::GetDirectionalLight() (-1 = none) */
    uint32_t punctual_light_count;         /* FScene::GetPunctualLightCount() */
    uint32_t meshlet_count;                /* FScene::m_meshletCount */
    const float* skylight_sh9_rgba;        /* device, 9 x float4 skylight probe (m_skylight.m_shTextureIndex); NULL = none */
    const uint32_t* prefiltered_envmap;    /* device, m_skylight.m_envmapTextureIndex: prefiltered cube, all levels; NULL = none */
    uint32_t envmap_res, envmap_mips;
    const uint32_t* env_brdf;              /* device, Renderer::s_envBRDF (R16G16_FLOAT); NULL = none */
    uint32_t env_brdf_width, env_brdf_height;
    uint32_t primitive_count;              /* FScene::m_primitiveCount (draw count when !FConfig::UseMeshlets, renderer.cpp:1375-1377) */
} incr_renderer_scene;
typedef struct incr_renderer_config {      /* FConfig fields the frame reads (inc/common.h:6-48) */
    int32_t  path_trace, viewmode, enable_taa, frustum_culling, enable_direct_lighting, enable_sky_lighting, enable_diffuse_ibl;
    float    exposure, sky_brightness, camera_near_plane;
    uint32_t light_cluster_dim[3], max_lights_per_cluster;
    float    cluster_depth_extent;
    uint32_t max_sample_count;
    float    camera_aperture, camera_focal_length, turbidity;
    int32_t  env_sky_mode, storage_rounding;
    int32_t  enable_specular_ibl;
    int32_t  use_primitives;               /* !FConfig::UseMeshlets: primitive variants of culling / visibility / G-buffer */
    int32_t  enable_hbao, use_bent_normals;/* FConfig::EnableHBAO, FConfig::UseBentNormals */
    int32_t  forward_lighting;             /* FConfig::ForwardLighting */
    /* how the frame is issued (not in the reference; the surfaces are bit-identical for every combination) */
    int32_t  unfused_passes;               /* 0: sky + sun + clustered lighting as one kernel and the tonemap inside the TAA resolve; 1: pass by pass */
    int32_t  record_threads;               /* > 0: passes are recorded on that many host threads and submitted in token order (job-sync.h) */
    int32_t  use_graph;                    /* 1: the raster frame is captured into a HIP graph, updated in place and replayed */
} incr_renderer_config;
enum {
    INCR_SURFACE_BACK_BUFFER = 0, INCR_SURFACE_HDR_RASTER, INCR_SURFACE_HDR_RAYTRACE, INCR_SURFACE_PATHTRACE_HISTORY,
    INCR_SURFACE_TAA_ACCUMULATION, INCR_SURFACE_DEPTH, INCR_SURFACE_VISIBILITY, INCR_SURFACE_GBUFFER_BASECOLOR,
    INCR_SURFACE_GBUFFER_NORMALS, INCR_SURFACE_GBUFFER_METALLIC_ROUGHNESS_AO, INCR_SURFACE_RENDER_STATS, INCR_SURFACE_RAY_STATS,
    INCR_SURFACE_BATCH_COUNTS, INCR_SURFACE_AMBIENT_OCCLUSION, INCR_SURFACE_BENT_NORMALS
};
int   incr_renderer_create(incr_context* ctx, incr_stream stream, uint32_t res_x, uint32_t res_y, const incr_renderer_scene* scene,
                           const incr_renderer_config* config /* NULL = FConfig defaults */, incr_renderer** out_renderer);
void  incr_renderer_destroy(incr_renderer* renderer);
int   incr_renderer_set_config(incr_renderer* renderer, const incr_renderer_config* config);
/* DLL export Tick(float dt) (inc/demo-interface.h:12; Demo::Tick, src/demo.cpp): advances the application clock.  The hot
 * path has one consumer of it -- the frame counter that selects the TAA jitter and history buffers advances per rendered
 * frame, not per tick -- so this only accumulates the time for callers that drive animation from it.  Returns the total. */
double incr_renderer_tick(incr_renderer* renderer, float dt_seconds);
/* Renderer::Render: one frame (one path-traced sample when config.path_trace, else one raster frame), enqueued on the
 * renderer's stream.  culling_view non-NULL = FConfig::FreezeCulling with that view. */
int   incr_renderer_render(incr_renderer* renderer, const incr_view* view, const incr_view* culling_view);
void  incr_renderer_reset_accumulation(incr_renderer* renderer);      /* Renderer::ResetPathtraceAccumulation */
/* Sort-first multi-GPU raster: this GPU renders rows [row_begin, row_end) of the frame (plus a one-row halo
 * for the TAA neighbourhood) and resolves TAA for the band only; culling and light culling stay replicated.  After each
 * frame the caller gathers every rank's band of INCR_SURFACE_TAA_ACCUMULATION into that same surface on all ranks (one
 * RCCL all_gather of equal bands: the surface has 64 slack rows for ceil(res_y / N) * N) and then calls
 * incr_renderer_tonemap on the presenting rank.  0,0 = the whole frame (single GPU; tonemap inside render). */
int   incr_renderer_set_band(incr_renderer* renderer, uint32_t row_begin, uint32_t row_end);
int   incr_renderer_tonemap(incr_renderer* renderer);
const char* incr_renderer_last_error(const incr_renderer* renderer);
/* Renderer::GetRenderStats (the read-back at the end of Render, renderer.cpp:2064-2076): culled objects / lights of the last
 * frame and the ray statistics accumulated so far.  Waits for the renderer's stream.  Either pointer may be NULL. */
int   incr_renderer_get_stats(incr_renderer* renderer, incr_render_stats* out_render_stats, incr_ray_stats* out_ray_stats);
/* Host-side cost of the last incr_renderer_render call and the graph-replay counters (all may be NULL). */
int   incr_renderer_get_frame_info(incr_renderer* renderer, double* host_seconds, uint64_t* graph_frames, uint64_t* graph_updates, uint64_t* graph_rebuilds);
/* Per-pass GPU time of the last frame: enable, render, read (events on the renderer's stream; direct issue, not graph replay). */
int   incr_renderer_set_profiling(incr_renderer* renderer, int enable);
int   incr_renderer_get_pass_times(incr_renderer* renderer, char (*names)[32], float* ms, uint32_t capacity, uint32_t* count);
void* incr_renderer_surface(incr_renderer* renderer, int which);      /* device pointer of a frame surface (INCR_SURFACE_*) */

#ifdef __cplusplus
}
#endif
#endif /* INCR_H */
