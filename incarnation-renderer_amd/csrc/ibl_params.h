// ibl_params.h -- argument block of the dynamic-sky / IBL precompute chain (ibl.hip; filled in ibl_api.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/incr.h"
#include "pt_params.h"

namespace incr {

static constexpr uint32_t kMaxEnvMips = 16;

struct IblParams {
    uint32_t cubemapRes, mips, shMips, sampleCount;
    int32_t rounding;
    incr_perez_distribution perez; SkySun sun;
    uint32_t* latlong;            // 2N x N lat-long radiance, all levels packed
    uint32_t* srcCube;            // N x N x 6 cube converted from the lat-long levels, all levels packed
    uint32_t* envmap;             // GGX-prefiltered cube (FScene::m_dynamicSkyEnvmap), all levels packed
    float4* shScratch;            // SH projection + reduction levels (RGBA32F, 9 slices per level)
    float4* sh9;                  // 9 x RGBA32F (FScene::m_dynamicSkySH)
    size_t latlongOffset[kMaxEnvMips], cubeOffset[kMaxEnvMips];
};

// FTextureCache::CacheHDRI (ibl.hip incr_launch_cache_hdri)
struct HdriParams {
    uint32_t width, height, cubemapRes, mips, shMips, sampleCount;      // lat-long W x H (RGBA32F), source cube N = H, CalcMipCount(W, H, true) levels
    int32_t rounding;
    float4* latlong;              // all levels packed; level 0 uploaded by the caller
    uint32_t* srcCube;            // N^2 x 6, `mips` levels
    uint32_t* envmap;             // (N / 2)^2 x 6, `mips - 1` levels: FLightProbe::m_envmapTextureIndex
    float4* shScratch; float4* sh9;
    size_t latlongOffset[kMaxEnvMips], cubeOffset[kMaxEnvMips], envOffset[kMaxEnvMips];
};

} // namespace incr
