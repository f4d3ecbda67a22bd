// pt_params.h -- kernel argument block of the path-tracing launch (host fills it in api.hip; render-jobs/path-tracing.inl:143-217 GlobalCbLayout).
#pragma once
#include <stdint.h>
#include "../../include/incr.h"
#include "device_types.h"

namespace incr {

// per-frame Preetham sun terms (host computed): sun direction, Yxy at the zenith, Perez F(0, thetaS)
struct SkySun { float s[3]; float YxyZenith[3]; float F0[3]; float pad[3]; };

// Exact unsigned division by a launch constant in 4 instructions (Granlund & Montgomery 1994, the add-shift form that covers every
// 32-bit numerator; generated on the host by make_fastdiv in api.hip): q = mulhi(magic, n); ((n - q) >> 1) + q >> more.
// The compiler does not specialise `n / P.W` or the sampler's `% N` for a uniform divisor: it emits the ~35-instruction
// per-lane sequence, a dozen times per shaded hit.
struct FastDiv { uint32_t d, magic, more; };

// Sun cover map (raster.hip sun_cover_*; consulted by the raster lighting passes and by the path tracer's shadow-ray set-up): the scene seen from the sun, an orthographic G x G grid over the BVH's bounds whose cell holds the greatest height (distance
// along the direction TO the sun) of any triangle over that cell.  A shadow ray toward the sun that starts above its cell's entry cannot hit anything -- the
// walk that would only prove that (94 % of the 4K Sponza-stat frame's sun rays are unoccluded, and an unoccluded ray visits every node along its whole length) is
// skipped.  Everything is derived on the device from what the shadow query itself reads (light table, triangle records, root node).
#ifndef INCR_SUN_COVER_G
#define INCR_SUN_COVER_G 2048
#endif
static constexpr uint32_t kSunCoverG = INCR_SUN_COVER_G;
struct SunCoverHeader { float U[3], V[3], L[3]; float u0, v0, invCell, eps; uint32_t valid, pad[3]; };
struct SunCover { const SunCoverHeader* __restrict__ header; const uint32_t* __restrict__ grid; };      // header == NULL: no map
struct PtParams {
    const BvhNode8* __restrict__ nodes;
    const TriIsect* __restrict__ tris;
    const TriShade* __restrict__ shade;
    const incr_material* __restrict__ materials;
    const DeviceTexture* __restrict__ textures;
    const MaterialTexels* __restrict__ materialTexels;   // per material: interleaved level-0 texels (device_types.h), texels == NULL where not built
    const float* __restrict__ srgbLut;
    const float* __restrict__ meshLocalToWorld;    // 16 floats per mesh: mul(meshTransform, sceneRotation)
    const incr_light* __restrict__ globalLights;
    const int32_t* __restrict__ lightIndices;
    const float* __restrict__ lightTransforms;     // 16 floats per scene light
    const float4* __restrict__ resolvedLights;     // 5 x float4 per light loop index (pt_shade.h load_light_resolved)
    uint32_t numGlobalLights, numSceneLights;
    int32_t lightCount;
    uint32_t numTris, numNodes;
    float projectionToWorld[16];
    float cameraMatrix[16];
    uint32_t W, H, numPixels, sampleIdx, sqrtSampleCount;
    FastDiv divW, divSqrtN, divN;                  // by W, sqrtSampleCount and sqrtSampleCount^2
    // samples in flight (incr_pathtrace_desc::num_samples): path id = sample * localPixels + local pixel; sample s of the call is
    // currentSampleIndex + s.  numSamples > 1: finished paths park their colour in sampleStage[path id] (RGBA16F bits) and
    // wf_integrate_samples_kernel folds a pixel's samples into the history in sample order
    uint32_t numSamples, localPixels;
    FastDiv divLocalPixels;
    uint2* __restrict__ sampleStage;
    uint32_t* __restrict__ hitChain;           // incr_pathtrace_desc::debug_hit_chain
    float aperture, focalLength, skyBrightness;
    int32_t directLighting, viewmode;
    incr_perez_distribution perez;
    SkySun sunMiss, sunEnv;
    uint32_t rowBlockHeight, rowBlockFirst, rowBlockStride, localRows;
    uint16_t* __restrict__ target;       // RGBA16F, localRows x W
    uint32_t* __restrict__ history;      // R11G11B10F, localRows x W
    int32_t storageRounding;
    incr_ray_stats* stats;
    // world-space light grid (scene_build.h LightGrid)
    float gridLo[3], gridInvCell[3];
    uint32_t gridDim[3];
    const uint32_t* __restrict__ gridCellOffsets;
    const uint32_t* __restrict__ gridCellLights;
    // prefiltered sky cube for the sky next-event estimate (NULL = analytic 0.04 * Preetham)
    const uint32_t* __restrict__ envmap;
    uint32_t envmapRes, envmapMips;
    int32_t envSkyMode;                       // 0 = HDRI cube in the miss shader (msEnvmap), 1 = dynamic Preetham sky (msDynamicSky)
    unsigned long long* statStripes;          // raster passes: 64 striped ray counters (16 words apart), folded into `stats` afterwards
    SunCover cover;                           // header == NULL: no map
};

// HIP events per pipeline of a recorded incr_pathtrace call (incr_profile_begin): [0] start, then a (start, end) pair per bounce and kernel kind --
// [1 + (bounce * 4 + kind) * 2 + end], kind 0 extend (bounce 0: wf_primary_kernel), 1 shade, 2 shadow (the any-hit launch issued after that bounce:
// it may cover several bounces' rays), 3 accumulate -- each recorded on the stream the kernel is launched on (pathtrace_wf.hip launcher)
static constexpr int kWfProfEvents = 33;

// Path state of the wavefront (multi-kernel) path tracer, SoA in HBM, one slot per local pixel (pathtrace_wf.hip).
struct WfBuffers {
    float4* extA;        // rays of the even bounces, list order, 2 x float4 per ray: {origin.xyz, path id} {direction.xyz, 0}
    float4* extB;        // rays of the odd bounces
    float4* stA;         // state travelling with the rays of the even bounces, 2 x float4: {payload.color} {payload.attenuation.xyz, 0}
    float4* stB;
    uint4*  hit;         // per list slot of the current bounce: {leaf-order triangle, t, u, v}
    float4* vertex;      // 3 x float4 per bounce and list slot ([bounce][slot]): what accumulate reads of a vertex (pathtrace_wf.hip "vertex record")
    float4* vertexFull;  // 5 x float4 per bounce and list slot: the rest of the record, written only for the vertices that are re-evaluated
    uint32_t* counters;  // 64 words: list lengths, traversal cursors, shadow-ray counts (pathtrace_wf.hip kCursorBase / kRayCountBase)
    float4* rayQ;        // shadow-ray queue (ONE for the four bounces), 2 x float4 per ray: {origin.xyz, vertex slot} {direction.xyz, index of the ray's record}
    float4* rayC;        // per queued ray: {attenuation * radiance of its light, 1 = unoccluded / 0 = occluded}
    uint32_t rayCap;     // rays the queue holds (soft limit: overflowing vertices are traced inside accumulate)
    uint32_t numPaths;
};



} // namespace incr
