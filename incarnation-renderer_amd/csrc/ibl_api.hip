// ibl_api.hip -- C ABI of the sky / image-based-lighting precompute (include/incr.h "Sky and IBL precompute").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "api_internal.h"
#include "ibl_params.h"

using namespace incr;

extern "C" hipError_t incr_launch_update_dynamic_sky(const IblParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_env_brdf(uint32_t* out, uint32_t W, uint32_t H, uint32_t numSamples, int rounding, hipStream_t stream);
extern "C" hipError_t incr_launch_cache_hdri(const HdriParams* p, hipStream_t stream);

struct incr_sky_ibl {
    uint32_t cubemapRes = 0, mips = 0;
    void* block = nullptr;                 // one allocation behind every product
    uint32_t* latlong = nullptr; uint32_t* srcCube = nullptr; uint32_t* envmap = nullptr; float4* shScratch = nullptr; float4* sh9 = nullptr;
    uint64_t latlongTexels = 0, cubeTexels = 0;
    // an HDRI light probe (incr_cache_hdri): the source is an RGBA32F lat-long of its own size, the source cube has its own size and level count
    float4* latlongF = nullptr; uint32_t llW = 0, llH = 0, srcRes = 0, srcMips = 0;
};

namespace {
uint32_t mip_count(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 1 && h >= 1) { ++n; w >>= 1; h >>= 1; } return n; }      // RenderUtils12::CalcMipCount
uint32_t mip_count_compressed(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 4 && h >= 4) { ++n; w >>= 1; h >>= 1; } return n; }      // CalcMipCount(w, h, true): down to 4 x 4

// Radiance RGBE picture (.hdr), the subset DirectX::LoadFromHDRFile reads (demo.cpp:313): "#?RADIANCE" / "#?RGBE" signature, FORMAT=32-bit_rle_rgbe, the
// "-Y <height> +X <width>" orientation; scanlines flat, run-length coded per channel (2 2 hi lo), or old-style runs (1 1 1 n).  A texel is
// mantissa x 2^(exponent - 136), exponent 0 = black (Ward's real-pixel format as Walter's rgbe.c reads it), alpha 1; EXPOSURE lines multiply up and the
// pixels are divided by the product (the format's definition of radiance).  ext/directXTex ships as headers + .lib: its reader cannot be read -- pinned here, unpinned there.
bool hdr_header(const uint8_t* b, size_t n, size_t* dataOffset, uint32_t* W, uint32_t* H, float* exposure, const char** why) {
    size_t i = 0; bool first = true, format = false; *exposure = 1.f;
    auto line = [&](std::string& out) -> bool { out.clear(); while (i < n && b[i] != '\n') { out.push_back((char)b[i]); ++i; } if (i >= n) return false; ++i; if (!out.empty() && out.back() == '\r') out.pop_back(); return true; };
    std::string l;
    for (;;) {
        if (!line(l)) { *why = "truncated header"; return false; }
        if (first) { if (l.rfind("#?RADIANCE", 0) != 0 && l.rfind("#?RGBE", 0) != 0) { *why = "not a Radiance picture (#?RADIANCE)"; return false; } first = false; continue; }
        if (l.empty()) break;
        if (l.rfind("FORMAT=", 0) == 0) { if (l != "FORMAT=32-bit_rle_rgbe") { *why = "only FORMAT=32-bit_rle_rgbe"; return false; } format = true; }
        else if (l.rfind("EXPOSURE=", 0) == 0) { const float e = strtof(l.c_str() + 9, nullptr); if (e > 0.f) *exposure *= e; }
    }
    if (!format) { *why = "no FORMAT line"; return false; }
    if (!line(l)) { *why = "no resolution line"; return false; }
    unsigned h = 0, w = 0;
    if (sscanf(l.c_str(), "-Y %u +X %u", &h, &w) != 2 || !w || !h || w > 32768 || h > 32768) { *why = "resolution line must be -Y <height> +X <width>"; return false; }
    *W = w; *H = h; *dataOffset = i;
    return true;
}
bool hdr_pixels(const uint8_t* b, size_t n, size_t off, uint32_t W, uint32_t H, float exposure, float* rgba, const char** why) {
    std::vector<uint8_t> scan((size_t)W * 4);
    const float inv = exposure != 1.f ? 1.f / exposure : 1.f;
    size_t i = off;
    for (uint32_t y = 0; y < H; ++y) {
        if (W >= 8 && W < 32768 && i + 4 <= n && b[i] == 2 && b[i + 1] == 2 && !(b[i + 2] & 0x80)) {              // run-length coded, channel after channel
            if ((((uint32_t)b[i + 2] << 8) | b[i + 3]) != W) { *why = "scanline length differs from the width"; return false; }
            i += 4;
            for (int c = 0; c < 4; ++c) {
                uint32_t x = 0;
                while (x < W) {
                    if (i >= n) { *why = "truncated scanline"; return false; }
                    uint32_t count = b[i++];
                    if (count > 128) {
                        count &= 127u;
                        if (!count || x + count > W || i >= n) { *why = "bad run"; return false; }
                        const uint8_t v = b[i++];
                        for (uint32_t k = 0; k < count; ++k) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (!count || x + count > W || i + count > n) { *why = "bad literal run"; return false; }
                        for (uint32_t k = 0; k < count; ++k) scan[(size_t)(x++) * 4 + c] = b[i++];
                    }
                }
            }
        } else {                                                                                                   // flat, with old-style repeats (1 1 1 n)
            uint32_t x = 0; int shift = 0;
            while (x < W) {
                if (i + 4 > n) { *why = "truncated scanline"; return false; }
                const uint8_t* p = b + i; i += 4;
                if (p[0] == 1 && p[1] == 1 && p[2] == 1 && x > 0) {
                    const uint32_t count = (uint32_t)p[3] << shift;
                    if (x + count > W) { *why = "bad repeat"; return false; }
                    for (uint32_t k = 0; k < count; ++k, ++x) memcpy(&scan[(size_t)x * 4], &scan[(size_t)(x - 1) * 4], 4);
                    shift += 8;
                } else { memcpy(&scan[(size_t)x * 4], p, 4); ++x; shift = 0; }
            }
        }
        float* row = rgba + (size_t)y * W * 4;
        for (uint32_t x = 0; x < W; ++x) {
            const uint8_t* p = &scan[(size_t)x * 4];
            if (p[3] == 0) { row[4 * x] = row[4 * x + 1] = row[4 * x + 2] = 0.f; }
            else {
                const float f = ldexpf(1.f, (int)p[3] - 136);
                row[4 * x] = (float)p[0] * f * inv; row[4 * x + 1] = (float)p[1] * f * inv; row[4 * x + 2] = (float)p[2] * f * inv;
            }
            row[4 * x + 3] = 1.f;
        }
    }
    return true;
}
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

int incr_decode_hdr(const uint8_t* bytes, size_t size, int32_t* width, int32_t* height, float* rgba, size_t capacity_floats) {
    if (!bytes || !width || !height) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    size_t off; uint32_t W, H; float exposure; const char* why = "";
    if (!hdr_header(bytes, size, &off, &W, &H, &exposure, &why)) return fail(INCR_ERR_INVALID_ARGUMENT, "HDR file: %s", why);
    *width = (int32_t)W; *height = (int32_t)H;
    if (!rgba) return INCR_OK;
    if (capacity_floats < (size_t)W * H * 4) return fail(INCR_ERR_INVALID_ARGUMENT, "output holds %zu floats, the picture needs %zu", capacity_floats, (size_t)W * H * 4);
    if (!hdr_pixels(bytes, size, off, W, H, exposure, rgba, &why)) return fail(INCR_ERR_INVALID_ARGUMENT, "HDR file: %s", why);
    return INCR_OK;
}

int incr_cache_hdri(incr_context* ctx, const uint8_t* hdr_file, size_t size, int32_t storage_rounding, incr_stream stream, incr_sky_ibl** out_ibl) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!hdr_file || !out_ibl) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    size_t off; uint32_t W, H; float exposure; const char* why = "";
    if (!hdr_header(hdr_file, size, &off, &W, &H, &exposure, &why)) return fail(INCR_ERR_INVALID_ARGUMENT, "HDR file: %s", why);
    // a lat-long probe: 2 : 1, and powers of two so that every level halves exactly (what GenerateMipMaps' linear filter is pinned for, ibl.hip)
    if (W != 2 * H || (H & (H - 1)) || H < 8 || H > 8192) return fail(INCR_ERR_INVALID_ARGUMENT, "HDRI %u x %u: a 2:1 lat-long picture with a power-of-two height in 8..8192 is expected", W, H);
    const uint32_t N = H, mips = mip_count_compressed(W, H), shMips = mip_count(W, H), envN = N >> 1, envMips = mips - 1;
    if (mips < 2 || mips > kMaxEnvMips) return fail(INCR_ERR_INVALID_ARGUMENT, "HDRI %u x %u: %u levels", W, H, mips);
    std::vector<float> pixels;
    try { pixels.resize((size_t)W * H * 4); } catch (...) { return fail(INCR_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    if (!hdr_pixels(hdr_file, size, off, W, H, exposure, pixels.data(), &why)) return fail(INCR_ERR_INVALID_ARGUMENT, "HDR file: %s", why);
    HdriParams P; memset(&P, 0, sizeof P);
    size_t llTexels = 0, cubeTexels = 0, envTexels = 0, shTexels = 0;
    for (uint32_t m = 0; m < mips; ++m) {
        P.latlongOffset[m] = llTexels; P.cubeOffset[m] = cubeTexels;
        llTexels += (size_t)(W >> m) * (H >> m); cubeTexels += (size_t)6 * (N >> m) * (N >> m);
    }
    for (uint32_t m = 0; m < envMips; ++m) { P.envOffset[m] = envTexels; envTexels += (size_t)6 * (envN >> m) * (envN >> m); }
    for (uint32_t m = 0; m < shMips; ++m) shTexels += (size_t)9 * std::max(W >> m, 1u) * std::max(H >> m, 1u);
    incr_sky_ibl* ibl = new (std::nothrow) incr_sky_ibl();
    if (!ibl) return fail(INCR_ERR_OUT_OF_MEMORY, "host allocation failed");
    const size_t szLl = align256(llTexels * 16), szCube = align256(cubeTexels * 4), szEnv = align256(envTexels * 4), szSh = align256(shTexels * 16), szSh9 = 256;
    hipError_t e = hipMalloc(&ibl->block, szLl + szCube + szEnv + szSh + szSh9);
    if (e != hipSuccess) { delete ibl; return fail(e == hipErrorOutOfMemory ? INCR_ERR_OUT_OF_MEMORY : INCR_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e)); }
    uint8_t* p = (uint8_t*)ibl->block;
    ibl->latlongF = (float4*)p; p += szLl; ibl->srcCube = (uint32_t*)p; p += szCube; ibl->envmap = (uint32_t*)p; p += szEnv; ibl->shScratch = (float4*)p; p += szSh; ibl->sh9 = (float4*)p;
    ibl->cubemapRes = envN; ibl->mips = envMips; ibl->srcRes = N; ibl->srcMips = mips; ibl->llW = W; ibl->llH = H; ibl->latlongTexels = llTexels; ibl->cubeTexels = envTexels;
    // the upload is synchronous on purpose: `pixels` lives on this stack frame (load time, once per probe)
    e = hipMemcpy(ibl->latlongF, pixels.data(), (size_t)W * H * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(ibl->block); delete ibl; return fail(INCR_ERR_HIP, "upload failed: %s", hipGetErrorString(e)); }
    P.width = W; P.height = H; P.cubemapRes = N; P.mips = mips; P.shMips = shMips; P.sampleCount = 1024; P.rounding = storage_rounding;
    P.latlong = ibl->latlongF; P.srcCube = ibl->srcCube; P.envmap = ibl->envmap; P.shScratch = ibl->shScratch; P.sh9 = ibl->sh9;
    e = incr_launch_cache_hdri(&P, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipFree(ibl->block); delete ibl; return fail(INCR_ERR_HIP, "launch failed: %s", hipGetErrorString(e)); }
    *out_ibl = ibl;
    return INCR_OK;
}

int incr_sky_ibl_get_info(const incr_sky_ibl* ibl, incr_sky_ibl_info* out) {
    if (!ibl || !out) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    out->latlong_rgba32f = (const float*)ibl->latlongF; out->latlong_width = ibl->latlongF ? ibl->llW : 2 * ibl->cubemapRes; out->latlong_height = ibl->latlongF ? ibl->llH : ibl->cubemapRes;
    out->source_cube_res = ibl->latlongF ? ibl->srcRes : ibl->cubemapRes; out->source_cube_mips = ibl->latlongF ? ibl->srcMips : ibl->mips;
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
