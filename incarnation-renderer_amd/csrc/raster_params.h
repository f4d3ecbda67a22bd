// raster_params.h -- kernel argument blocks of the raster path (raster.hip; filled in raster_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/incr.h"
#include "device_types.h"
#include "pt_params.h"

namespace incr {

// Device view of the scene for the raster passes.  The accessor -> bufferView -> buffer indirection of
// shaders/common/mesh-material.hlsli is resolved at scene creation: mv* arrays are indexed by the position of a
// vertex in the packed meshlet vertex-index buffer (meshlet.vertexBegin + local index), so a visibility-buffer
// pixel costs three 16-byte position loads instead of three dependent index/accessor/view chains.
struct RasterScene {
    const incr_gpu_meshlet* __restrict__ meshlets;
    const uint32_t* __restrict__ packedTris;
    const uint32_t* __restrict__ meshVisibility;
    const float4* __restrict__ mvPos;         // GetFloat3(uniqueVertIndex, positionAccessor), w = 0
    const float2* __restrict__ mvUv;          // GetFloat2(uniqueVertIndex, uvAccessor)
    const float4* __restrict__ mvNormal;      // GetFloat3(uniqueVertIndex, normalAccessor), w = 0
    const float4* __restrict__ mvTangent;     // GetFloat4(uniqueVertIndex, tangentAccessor)
    const float* __restrict__ meshLocalToWorld;   // 16 floats per mesh: mul(meshTransform, sceneRotation)
    const incr_material* __restrict__ materials;
    const DeviceTexture* __restrict__ textures;
    const uint32_t* const* __restrict__ alphaQuads;          // per texture: the alpha-quad chain the scan's alpha test reads (NULL: sample the texture)
    const MaterialTexels* __restrict__ materialTexels;      // per material: the interleaved chain of its three main maps (device_types.h), chain == NULL where not built
    const float* __restrict__ srgbLut;
    uint32_t numMeshlets, numMeshes;
    uint32_t maxAnisotropy;                   // g_anisoSampler's MaxAnisotropy (incr_scene_desc::max_anisotropy; 16 as the reference's root signatures)
};

// The primitive (non-meshlet) path, FConfig::UseMeshlets == false: per-primitive index / vertex arrays resolved out of the
// accessors at scene creation, and a static table that cuts every primitive into 64-triangle chunks.
struct PrimDraw { uint32_t indexBegin, indexCount, vertexBase; int32_t meshIndex, materialIndex; float sphere[4]; uint32_t pad[3]; };
static_assert(sizeof(PrimDraw) == 48, "PrimDraw layout");
struct RasterPrims {
    const PrimDraw* __restrict__ draws;
    const uint32_t* __restrict__ indices;     // GetUint(i, indexAccessor) widened to 32 bit
    const float4* __restrict__ pvPos; const float2* __restrict__ pvUv; const float4* __restrict__ pvNormal; const float4* __restrict__ pvTangent;
    const uint32_t* __restrict__ chunkPrim; const uint32_t* __restrict__ chunkFirstTri;
    uint32_t* __restrict__ primFlags;         // per frame: 0 not drawn, 1 default list, 2 double-sided list
    uint32_t numPrims, numChunks;
};

// Per-mesh, per-frame constants (mesh_frame_kernel): 176 B.
struct MeshFrame {
    float localToClip[16];     // mul(localToWorld, viewProj)                   gbuffer-compute.hlsl:196-197
    float planes[5][4];        // near, left, right, bottom, top in object space  batch-culling.hlsl:28-35
    float planeLen[5];         // length(plane.xyz)
    float pad[3];
};
static_assert(sizeof(MeshFrame) == 176, "MeshFrame layout");

struct ViewFrame {             // the matrices a pass reads, copied out of incr_view_constants
    float viewProj[16];
    float cullViewProj[16];
    float invViewProj[16];
    float invProj[16];
    float eye[3];
    float nearPlane;
    uint32_t W, H;
};

static constexpr uint32_t kMaxClusterSlices = 64;
static constexpr uint32_t kMaxLightsPerClusterCap = 128;     // MAX_LIGHTS_PER_CLUSTER default of light-culling.hlsl:14-16

// ---- kernel argument blocks (raster.hip) ----
struct LightCullParams {
    const incr_light* __restrict__ globalLights; const int32_t* __restrict__ lightIndices; const float* __restrict__ lightTransforms;
    uint32_t numSceneLights, numGlobalLights;
    uint32_t dim[3], maxLights;
    float slice[kMaxClusterSlices + 1];     // zNear * pow(zFar / zNear, k / dimZ): powf evaluated once on the host
    float proj[16], invViewProj[16];
    uint32_t* __restrict__ lists; incr_light_grid_data* __restrict__ grid; uint32_t* __restrict__ totalCount; incr_render_stats* __restrict__ stats;
};

// One screen triangle between the set-up and the scan kernel of the visibility pass: clockwise vertices in 24.8
// fixed point + depth, 48 B.  Alpha-tested triangles also carry the clip-space vertices and uvs of the original triangle.
struct alignas(16) SubTriRec { int32_t X0, Y0, X1, Y1, X2, Y2; float z0, z1, z2; uint32_t key; int32_t alphaTex; uint32_t pad; };
struct alignas(16) SubTriAlpha { float p[12]; float uv[6]; float pad[2]; };
static_assert(sizeof(SubTriRec) == 48 && sizeof(SubTriAlpha) == 80, "visibility queue records");

// Band-queue cursors: the part of the queue behind the scan grid's static round is cut into INCR_VIS_STRIPES contiguous segments, each with a cursor on a
// cache line of its own (raster.hip band_next)
#ifndef INCR_VIS_STRIPES
#define INCR_VIS_STRIPES 16
#endif
static constexpr size_t kVisCounterBytes = 256 + (size_t)INCR_VIS_STRIPES * 128;
// Screen bins (raster.hip scan_bin, the first phase of vis_scan_kernel): a workgroup scans a bin's triangles into an LDS copy of the bin and merges it into the surface.  A bin is 2^binShift pixels square
// (64 for frames of 4 M pixels and more, 32 below: enough workgroups to fill the chip); per bin three lists of triangle record indices.
static constexpr uint32_t kVisBinCapDefault = 1024;
enum VisBinList : uint32_t { kBinOpaque = 0, kBinAlpha = 1, kBinSmall = 2, kVisBinLists = 3 };
struct VisParams {
    RasterScene S;
    float viewProj[16];
    uint32_t W, H;
    const incr_draw_cmd* __restrict__ lists[2];
    const uint32_t* __restrict__ counts;
    unsigned long long* __restrict__ packed;         // W*H: depth bits << 32 | (order key + 1); 0 = empty
    // triangle queues (raster.hip "Work distribution")
    SubTriRec* __restrict__ recs; SubTriAlpha* __restrict__ alphas;
    uint2* __restrict__ bigQueue; uint32_t* __restrict__ smallQueue; uint32_t* __restrict__ wholeQueue;
    uint32_t* __restrict__ vqCounters;               // [0] records [1] band items [2] small [3] whole [4..6] consumer cursors; [64 + 32 s] band cursor of stripe s (kVisCounterBytes)
    uint32_t recCap, bigCap;
    uint32_t y0, y1;                                 // scissor rows [y0, y1) (incr_view_constants::scissor_y0/y1)
    // screen bins (origin: column 0, row y0): binCount[bin * 3 + list] triangles filed, the first binCap of them in binList[(bin * 3 + list) * binCap ..]; binCap == 0: no bins
    uint32_t* __restrict__ binCount; uint32_t* __restrict__ binList;
    uint32_t binsX, binsY, binCap, binShift;
    RasterPrims prims;                               // primitive path only
    int32_t forward;                                 // forward pass: every primitive of every visible mesh in scene order, cull mode from the material
};

// Decal G-buffer pass: `prims` carries the baked decal draws (meshIndex = decal index into decalLocalToWorld); every decal
// triangle owns two fixed record slots (near-plane clipping yields at most two screen triangles), key 0xffffffff = empty.
struct DecalParams {
    RasterScene S; RasterPrims prims;
    const float* __restrict__ decalLocalToWorld;
    float viewProj[16];
    uint32_t W, H, y0, y1; int32_t rounding;
    const float* __restrict__ depth;
    uint32_t* __restrict__ g0; uint32_t* __restrict__ g1; uint32_t* __restrict__ g2;
    SubTriRec* __restrict__ recs; uint32_t numSlots;
    uint32_t listCap;                                // entries of the per-tile LDS list before it is flushed (256 .. 1024)
};

struct ForwardParams {
    RasterScene S; RasterPrims prims; PtParams Pt;   // Pt: lights + BVH for the sun's shadow query
    uint32_t W, H, y0, y1; int32_t rounding;
    const uint32_t* __restrict__ vis; uint32_t* __restrict__ color;
    float viewProj[16], eye[3];
    const float4* __restrict__ sh9;
    const uint32_t* __restrict__ envmap; uint32_t envmapRes, envmapMips;
    const uint32_t* __restrict__ envBrdf; uint32_t envBrdfW, envBrdfH;
    int32_t viewmode, directLighting, diffuseIbl, specularIbl;
};

struct GBufferParams {
    RasterScene S;
    const MeshFrame* __restrict__ frames;
    uint32_t W, H;
    const uint32_t* __restrict__ vis;
    uint32_t* __restrict__ g0; uint32_t* __restrict__ g1; uint32_t* __restrict__ g2; uint32_t* __restrict__ color;
    int32_t rounding;
    uint32_t y0, y1;
    int32_t usePrims; RasterPrims prims;             // !USING_MESHLETS
};

struct LightingParams {
    uint32_t W, H;
    uint32_t* __restrict__ color; const float* __restrict__ depth;
    const uint32_t* __restrict__ g0; const uint32_t* __restrict__ g1; const uint32_t* __restrict__ g2;
    int32_t lightingOnly, rounding;
    float invViewProj[16], invProj[16];
    float eye[3];
    uint32_t y0, y1;
};

struct SkyParams {
    LightingParams L; const float4* __restrict__ sh9; const uint8_t* __restrict__ ao; float skyBrightness; int32_t diffuseIbl;
    const uint32_t* __restrict__ envmap; uint32_t envmapRes, envmapMips;       // prefiltered cube, all levels (NULL = no specular IBL)
    const uint32_t* __restrict__ envBrdf; uint32_t envBrdfW, envBrdfH;
    float skyInvViewProj[16];                                                  // sky-lighting.inl:124
    const uint32_t* __restrict__ bent; int32_t useBent;                        // USE_BENT_NORMALS
};

struct HbaoParams {
    uint32_t W, H, y0, y1; int32_t rounding;
    const float* __restrict__ depth; const uint32_t* __restrict__ normals;
    uint8_t* __restrict__ ao; uint32_t* __restrict__ bent;
    float invViewProj[16], eye[3], camUp[3];
    PtParams S;
};

struct DirectParams { LightingParams L; PtParams S; uint32_t sunIndex; };      // S.cover: the sun cover map of the scene, if it has one

struct ClusteredParams {
    LightingParams L; PtParams S;
    const uint32_t* __restrict__ lists; const incr_light_grid_data* __restrict__ grid;
    uint32_t dim[3];
    float scale, bias;
    int32_t requiresClear;
};

// sky + sun + clustered in one kernel (raster.hip deferred_lighting_kernel); the three parts address the same surfaces
struct FusedLightingParams { SkyParams sky; DirectParams direct; ClusteredParams clustered; int32_t haveSky, haveDirect, haveClustered; };

struct DynamicSkyParams {
    uint32_t W, H, y0, y1; int32_t rounding;
    uint32_t* __restrict__ color; const float* __restrict__ depth;
    incr_perez_distribution perez; SkySun sun;
};

struct EnvmapParams {
    uint32_t W, H, y0, y1; int32_t rounding;
    uint32_t* __restrict__ color; const float* __restrict__ depth;
    const uint32_t* __restrict__ envmap; uint32_t envmapRes, envmapMips;
    float skyBrightness, invParallaxViewProj[16];
};

struct DebugVizParams {
    uint32_t W, H, y0, y1; int32_t rounding, viewmode, usePrims;
    const uint32_t* __restrict__ vis; const uint32_t* __restrict__ g0; const uint32_t* __restrict__ g1; const uint32_t* __restrict__ g2;
    const float* __restrict__ depth; const uint8_t* __restrict__ ao; const uint32_t* __restrict__ bent;
    uint32_t* __restrict__ color;
    float numSlices, sliceScale, sliceBias;        // slices / log(1000 / 5), -log(5) * scale: logf once on the host
    float invProj[16];
};

struct TaaParams {
    uint32_t W, H, historyIndex; int32_t rounding;
    const uint32_t* __restrict__ hdr; const uint32_t* __restrict__ histIn; uint32_t* __restrict__ histOut; const float* __restrict__ depth;
    float invViewProj[16], prevViewProj[16];
    uint32_t y0, y1;
    uint32_t* __restrict__ ldr; int32_t ev100, viewmode;      // optional: tonemapped copy of the resolved texel (presenting path)
    const float* __restrict__ srgbSteps;                      // pt_device.h linear_to_srgb8
};

} // namespace incr
