// ibl_api.hip -- C ABI of the sky / image-based-lighting precompute (include/incr.h "Sky and IBL precompute").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>

#include "api_internal.h"
#include "ibl_params.h"

using namespace incr;

extern "C" hipError_t incr_launch_update_dynamic_sky(const IblParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_env_brdf(uint32_t* out, uint32_t W, uint32_t H, uint32_t numSamples, int rounding, hipStream_t stream);

struct incr_sky_ibl {
    uint32_t cubemapRes = 0, mips = 0;
    void* block = nullptr;                 // one allocation behind every product
    uint32_t* latlong = nullptr; uint32_t* srcCube = nullptr; uint32_t* envmap = nullptr; float4* shScratch = nullptr; float4* sh9 = nullptr;
    uint64_t latlongTexels = 0, cubeTexels = 0;
};

namespace {
uint32_t mip_count(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 1 && h >= 1) { ++n; w >>= 1; h >>= 1; } return n; }      // RenderUtils12::CalcMipCount
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
}

extern "C" {

int incr_update_dynamic_sky(incr_context* ctx, const incr_scene* scene, uint32_t envmap_resolution, float turbidity, int32_t storage_rounding,
                            incr_stream stream, incr_sky_ibl** inout_ibl) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!scene || !inout_ibl) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint32_t N = envmap_resolution;
    if (N < 1 || N > 4096 || (N & (N - 1))) return fail(INCR_ERR_INVALID_ARGUMENT, "envmap_resolution %u must be a power of two in 1..4096", N);
    const uint32_t W = 2 * N, H = N, mips = mip_count(N, N), shMips = mip_count(W, H);
    incr_sky_ibl* ibl = *inout_ibl;
    if (ibl && ibl->cubemapRes != N) return fail(INCR_ERR_INVALID_ARGUMENT, "the IBL object was created for resolution %u", ibl->cubemapRes);
    IblParams P; memset(&P, 0, sizeof P);
    size_t llTexels = 0, cubeTexels = 0, shTexels = 0;
    for (uint32_t m = 0; m < mips; ++m) {
        P.latlongOffset[m] = llTexels; P.cubeOffset[m] = cubeTexels;
        llTexels += (size_t)std::max(W >> m, 1u) * std::max(H >> m, 1u);
        cubeTexels += (size_t)6 * std::max(N >> m, 1u) * std::max(N >> m, 1u);
    }
    for (uint32_t m = 0; m < shMips; ++m) shTexels += (size_t)9 * std::max(W >> m, 1u) * std::max(H >> m, 1u);
    if (!ibl) {
        ibl = new (std::nothrow) incr_sky_ibl();
        if (!ibl) return fail(INCR_ERR_OUT_OF_MEMORY, "host allocation failed");
        const size_t szLl = align256(llTexels * 4), szCube = align256(cubeTexels * 4), szSh = align256(shTexels * 16), szSh9 = 256;
        hipError_t e = hipMalloc(&ibl->block, szLl + 2 * szCube + szSh + szSh9);
        if (e != hipSuccess) { delete ibl; return fail(e == hipErrorOutOfMemory ? INCR_ERR_OUT_OF_MEMORY : INCR_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
        uint8_t* p = (uint8_t*)ibl->block;
        ibl->latlong = (uint32_t*)p; p += szLl; ibl->srcCube = (uint32_t*)p; p += szCube; ibl->envmap = (uint32_t*)p; p += szCube;
        ibl->shScratch = (float4*)p; p += szSh; ibl->sh9 = (float4*)p;
        ibl->cubemapRes = N; ibl->mips = mips; ibl->latlongTexels = llTexels; ibl->cubeTexels = cubeTexels;
        *inout_ibl = ibl;
    }
    P.cubemapRes = N; P.mips = mips; P.shMips = shMips; P.sampleCount = 1024;          // renderer.cpp:520
    P.rounding = storage_rounding;
    incr_perez_from_turbidity(turbidity, &P.perez);                                     // renderer.cpp:327-333
    const float* s = scene->host.sunDir;                                               // Vector3 L = sunDir; L.Normalize() (:335-336)
    const V3 L = normalize3(v3(s[0], s[1], s[2]));
    const float Ln[3] = {L.x, L.y, L.z};
    incr_sky_sun_terms(Ln, turbidity, P.perez, &P.sun);
    P.latlong = ibl->latlong; P.srcCube = ibl->srcCube; P.envmap = ibl->envmap; P.shScratch = ibl->shScratch; P.sh9 = ibl->sh9;
    HIP_TRY(incr_launch_update_dynamic_sky(&P, (hipStream_t)stream));
    return INCR_OK;
}

int incr_sky_ibl_get_info(const incr_sky_ibl* ibl, incr_sky_ibl_info* out) {
    if (!ibl || !out) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    out->cubemap_res = ibl->cubemapRes; out->num_mips = ibl->mips;
    out->latlong_r11g11b10 = ibl->latlong; out->source_cube_r11g11b10 = ibl->srcCube; out->envmap_r11g11b10 = ibl->envmap;
    out->sh9_rgba = (const float*)ibl->sh9; out->latlong_texels = ibl->latlongTexels; out->cube_texels = ibl->cubeTexels;
    return INCR_OK;
}

int incr_sky_ibl_destroy(incr_context* ctx, incr_sky_ibl* ibl) {
    if (!ibl) return INCR_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (ibl->block) (void)hipFree(ibl->block);
    delete ibl;
    return INCR_OK;
}

int incr_generate_env_brdf(incr_context* ctx, uint32_t width, uint32_t height, int32_t storage_rounding, void* out_rg16f, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!out_rg16f || !width || !height) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL output or zero size");
    HIP_TRY(incr_launch_env_brdf((uint32_t*)out_rg16f, width, height, 1024, storage_rounding, (hipStream_t)stream));   // renderer.cpp:218
    return INCR_OK;
}

} // extern "C"
