// ibl.hip -- sky / image-based-lighting precompute on MI355X: the compute passes of FScene::UpdateDynamicSky
// (reference: source/demo-dll/src/scene.cpp:1657-1757 -> src/renderer.cpp:294-800) and Renderer::GenerateEnvBrdfTexture
// (src/renderer.cpp:169-245).  One lane per output texel; surfaces are packed mip chains in HBM (ibl_device.h).  These
// run when the sun moves, not per frame; the only heavy one is the GGX prefilter (1024 trilinear cube samples per texel).
#include <hip/hip_runtime.h>
#include <algorithm>

#include "ibl_device.h"
#include "ibl_params.h"

namespace incr {

// environment-sky/dynamic-sky-spherical-projection.hlsl:24-44
__global__ __launch_bounds__(256) void sky_latlong_kernel(uint32_t* __restrict__ out, uint32_t W, uint32_t H, const incr_perez_distribution perez, const SkySun sun, int rounding) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const f2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
    out[(size_t)y * W + x] = pack_r11g11b10(0.04f * CalculateSkyRadianceRGB(sun, normalize(latlong_dir(uv)), perez), rounding);
}
// postprocess/downsample.hlsl:14-32
__global__ __launch_bounds__(256) void downsample2d_kernel(const uint32_t* __restrict__ src, uint32_t srcW, uint32_t* __restrict__ dst, uint32_t dstW, uint32_t dstH, int rounding) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dstW || y >= dstH) return;
    const f3 a = unpack_r11g11b10(src[(size_t)(2 * y) * srcW + 2 * x]), b = unpack_r11g11b10(src[(size_t)(2 * y) * srcW + 2 * x + 1]);
    const f3 c = unpack_r11g11b10(src[(size_t)(2 * y + 1) * srcW + 2 * x]), d = unpack_r11g11b10(src[(size_t)(2 * y + 1) * srcW + 2 * x + 1]);
    dst[(size_t)y * dstW + x] = pack_r11g11b10(0.25f * (((a + b) + c) + d), rounding);
}
// DirectX::GenerateMipMaps(TEX_FILTER_LINEAR) on the RGBA32F lat-long of an HDRI (demo.cpp:318-320): a 2:1 level is the bilinear tap half way between four texels, weights
// 1/2 per axis, rows first -- (a/2 + b/2)/2 + (c/2 + d/2)/2 (every halving exact, two roundings); alpha rides along
__global__ __launch_bounds__(256) void downsample2d_f32_kernel(const float4* __restrict__ src, uint32_t srcW, float4* __restrict__ dst, uint32_t dstW, uint32_t dstH) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dstW || y >= dstH) return;
    const float4 a = src[(size_t)(2 * y) * srcW + 2 * x], b = src[(size_t)(2 * y) * srcW + 2 * x + 1], c = src[(size_t)(2 * y + 1) * srcW + 2 * x], d = src[(size_t)(2 * y + 1) * srcW + 2 * x + 1];
    dst[(size_t)y * dstW + x] = make_float4(0.5f * (0.5f * a.x + 0.5f * b.x) + 0.5f * (0.5f * c.x + 0.5f * d.x), 0.5f * (0.5f * a.y + 0.5f * b.y) + 0.5f * (0.5f * c.y + 0.5f * d.y),
                                            0.5f * (0.5f * a.z + 0.5f * b.z) + 0.5f * (0.5f * c.z + 0.5f * d.z), 0.5f * (0.5f * a.w + 0.5f * b.w) + 0.5f * (0.5f * c.w + 0.5f * d.w));
}
// content-pipeline/cubemapgen.hlsl:19-121, one level; TEXEL = uint32_t (R11G11B10: the dynamic sky) or float4 (RGBA32F: an HDRI)
template <class TEXEL>
__global__ __launch_bounds__(256) void cubemapgen_kernel(const TEXEL* __restrict__ latlong, uint32_t llW, uint32_t llH, uint32_t* __restrict__ cube, uint32_t n, int rounding) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= n || y >= n) return;
    const float ftx[6] = {K_PI_OVER_2, -K_PI_OVER_2, 0.f, 0.f, 0.f, K_PI}, fty[6] = {0.f, 0.f, K_PI_OVER_2, -K_PI_OVER_2, 0.f, 0.f};
    const float an = sinf(K_PI_OVER_4), ak = cosf(K_PI_OVER_4);
    f2 nn = {((float)x + 0.5f) / (float)n, ((float)y + 0.5f) / (float)n};
    nn = {(2.f * nn.x - 1.f) * an, (2.f * nn.y - 1.f) * an};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f2 uv;
        if (fty[i] == 0.f) { uv.x = atan2f(nn.x, ak); uv.y = atan2f(nn.y * cosf(uv.x), ak); uv.x += ftx[i]; }
        else if (fty[i] < 0.f) { const float d = sqrtf(dot(nn, nn)); uv.y = K_PI_OVER_2 - atan2f(d, ak); uv.x = K_PI_OVER_2 + atan2f(nn.y, nn.x); }
        else { const float d = sqrtf(dot(nn, nn)); uv.y = -K_PI_OVER_2 + atan2f(d, ak); uv.x = atan2f(nn.x, nn.y); }
        uv.x = uv.x * K_INV_PI; uv.y = uv.y / K_PI_OVER_2;
        while (uv.y < -1) { uv.y += 2; uv.x += 1; }
        while (uv.y > 1) { uv.y -= 2; uv.x += 1; }
        while (uv.x < -1) uv.x += 2;
        while (uv.x > 1) uv.x -= 2;
        uv = {0.5f * uv.x + 0.5f, 0.5f * uv.y + 0.5f};
        cube[((size_t)i * n + y) * n + x] = pack_r11g11b10(sample2d_bilinear_wrap(latlong, llW, llH, uv), rounding);
    }
}
DI f3 GetEnvDir(uint32_t face, f2 uv) {                              // split-sum-approx/prefilter.hlsl:26-41
    const float vx = 2.f * uv.x - 1.f, vy = -2.f * uv.y + 1.f;
    switch (face) {
    case 0: return normalize(mk3(1.f, vy, -vx));
    case 1: return normalize(mk3(-1.f, vy, vx));
    case 2: return normalize(mk3(vx, 1.f, -vy));
    case 3: return normalize(mk3(vx, -1.f, vy));
    case 4: return normalize(mk3(vx, vy, 1.f));
    default: return normalize(mk3(-vx, vy, -1.f));
    }
}
// split-sum-approx/prefilter.hlsl:43-95, one level, all six faces (blockIdx.z = face).  mipSize / cubemapSize are the shader's root constants: the OUTPUT level's size and
// the output cube's size (renderer.cpp:504-526) -- the dynamic sky filters into a cube of the source's size, CacheHDRI into one of half the size (demo.cpp:357-369)
__global__ __launch_bounds__(256) void prefilter_kernel(const CubeChain src, uint32_t mipSize, uint32_t cubemapSize, uint32_t* __restrict__ dst, uint32_t sampleCount, float roughness, int rounding) {
    const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15u), y = blockIdx.y * 16 + (threadIdx.x >> 4), face = blockIdx.z;
    if (x >= mipSize || y >= mipSize) return;
    const float Resolution = (float)cubemapSize;
    const f3 R = GetEnvDir(face, {(float)x / (float)mipSize, (float)y / (float)mipSize});
    const f3 N = R, V = R;
    const f3 up = fabsf(N.z) < 0.999f ? mk3(0.f, 0.f, 1.f) : mk3(1.f, 0.f, 0.f);       // TangentToWorld (common/math.hlsli:82-89)
    const f3 T = normalize(cross(up, N)), B = cross(N, T);
    f3 color = {0.f, 0.f, 0.f}; float totalWeight = 0.f;
    for (uint32_t i = 0; i < sampleCount; ++i) {
        const f2 Xi = Hammersley(i, (float)sampleCount);
        f3 H = SampleGGX(Xi, roughness);
        H = normalize(mk3((H.x * T.x + H.y * B.x) + H.z * N.x, (H.x * T.y + H.y * B.y) + H.z * N.y, (H.x * T.z + H.y * B.z) + H.z * N.z));
        const f3 L = normalize(reflect(-V, H));
        const float NoL = saturate(dot(N, L));
        if (NoL > 0.f) {
            const float NoH = hmax(dot(N, H), 0.0f), VoH = hmax(dot(V, H), 0.0f);
            const float D = GGX(NoH, roughness);
            const float pdf = (D * NoH / (4.0f * VoH)) + 0.0001f;
            const float saTexel = 4.f * K_PI / (6.0f * Resolution * Resolution);
            const float saSample = 1.f / ((float)sampleCount * pdf + 0.0001f);
            const float mipLevel = roughness == 0.0f ? 0.0f : 0.5f * log2f(saSample / saTexel);
            color = color + cube_sample_trilinear(src, L, mipLevel) * NoL;
            totalWeight += NoL;
        }
    }
    dst[((size_t)face * mipSize + y) * mipSize + x] = pack_r11g11b10(color / totalWeight, rounding);
}
// spherical-harmonics/projection.hlsl:17-49 (blockIdx.z unused: one lane writes the nine coefficient slices)
template <class TEXEL>
__global__ __launch_bounds__(256) void sh_project_kernel(const TEXEL* __restrict__ latlong, uint32_t W, uint32_t H, float4* __restrict__ out) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float K[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
    const f2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
    const f3 dir = latlong_dir(uv);
    float b[9];
    b[0] = K[0]; b[1] = K[1] * dir.y; b[2] = K[2] * dir.z; b[3] = K[3] * dir.x;
    b[4] = (K[4] * dir.x) * dir.y; b[5] = (K[5] * dir.y) * dir.z; b[6] = K[6] * ((3.0f * dir.z) * dir.z - 1.0f); b[7] = (K[7] * dir.x) * dir.z; b[8] = K[8] * (dir.x * dir.x - dir.y * dir.y);
    const f3 rad = load_radiance(latlong, (size_t)y * W + x);          // (an RGBA32F HDRI carries alpha 1, as a Load of R11G11B10 returns: the fourth channel is the same)
    const float sint = sinf(K_PI * uv.y), dTheta = K_PI / (float)H, dPhi = 2.f * K_PI / (float)W;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float w = b[i];
        out[((size_t)i * H + y) * W + x] = make_float4((((rad.x * w) * sint) * dTheta) * dPhi, (((rad.y * w) * sint) * dTheta) * dPhi,
                                                       (((rad.z * w) * sint) * dTheta) * dPhi, (((1.f * w) * sint) * dTheta) * dPhi);
    }
}
// spherical-harmonics/parallel-reduction.hlsl:14-28: one 2x2 sum per lane, blockIdx.z = coefficient slice
__global__ __launch_bounds__(64) void sh_reduce_kernel(const float4* __restrict__ src, uint32_t sw, uint32_t sh, float4* __restrict__ dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * 8 + (threadIdx.x & 7u), y = blockIdx.y * 8 + (threadIdx.x >> 3), i = blockIdx.z;
    if (x >= dw || y >= dh) return;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto at = [&](uint32_t xx, uint32_t yy) -> float4 { return (xx >= sw || yy >= sh) ? zero : src[((size_t)i * sh + yy) * sw + xx]; };
    const float4 a = at(2 * x, 2 * y), b = at(2 * x + 1, 2 * y), c = at(2 * x, 2 * y + 1), d = at(2 * x + 1, 2 * y + 1);
    dst[((size_t)i * dh + y) * dw + x] = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z, ((a.w + b.w) + c.w) + d.w);
}
// spherical-harmonics/export.hlsl:17-29
__global__ void sh_export_kernel(const float4* __restrict__ src, uint32_t w, uint32_t h, float4* __restrict__ out9) {
    const uint32_t i = threadIdx.x;
    if (i >= 9) return;
    const float4 a = src[((size_t)i * h) * w], b = w > 1 ? src[((size_t)i * h) * w + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    out9[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// split-sum-approx/brdf-integration.hlsl:23-67
__global__ __launch_bounds__(256) void env_brdf_kernel(uint32_t* __restrict__ out, uint32_t W, uint32_t H, uint32_t numSamples, int rounding) {
    const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15u), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const float NoV = (float)x / (float)W, Roughness = 1.f - (float)y / (float)H;
    const f3 V = {sqrtf(1.f - NoV * NoV), 0.f, NoV};
    float A = 0.f, Bv = 0.f;
    for (uint32_t i = 0; i < numSamples; ++i) {
        const f2 Xi = Hammersley(i, (float)numSamples);
        const f3 Hh = SampleGGX(Xi, Roughness);
        const f3 L = normalize(reflect(-V, Hh));
        const float NoL = saturate(L.z), NoH = saturate(Hh.z), VoH = saturate(dot(V, Hh));
        if (NoL > 0.f && NoV > 0.f) {
            const float a2 = Roughness * Roughness, k = 0.5f * a2 * a2;                 // G_Smith_IBL (lighting/pbr.hlsli:53-60)
            const float G = G_SchlickGGX(NoV, k) * G_SchlickGGX(NoL, k);
            const float G_Vis = G * VoH / (NoH * NoV);
            const float Fc = pow5(1.f - VoH);
            A += (1 - Fc) * G_Vis; Bv += Fc * G_Vis;
        }
    }
    out[(size_t)y * W + x] = f32_to_f16_bits(A / (float)numSamples, rounding) | (f32_to_f16_bits(Bv / (float)numSamples, rounding) << 16);
}

} // namespace incr

using namespace incr;
static inline dim3 grid64x4(uint32_t W, uint32_t H) { return dim3((W + 63) / 64, (H + 3) / 4); }

// ShEncode behind the projection (renderer.cpp:533-800): 2 x 2 sums level after level, then the export of the last two texels
static void sh_reduce_and_export(float4* shScratch, uint32_t W, uint32_t H, uint32_t shMips, float4* sh9, hipStream_t stream) {
    uint32_t sw = W, sh = H; size_t off = 0;
    for (uint32_t m = 0; m + 1 < shMips; ++m) {
        const uint32_t dw = std::max(W >> (m + 1), 1u), dh = std::max(H >> (m + 1), 1u);
        const size_t next = off + (size_t)sw * sh * 9;
        hipLaunchKernelGGL(sh_reduce_kernel, dim3((dw + 7) / 8, (dh + 7) / 8, 9), dim3(64), 0, stream, shScratch + off, sw, sh, shScratch + next, dw, dh);
        off = next; sw = dw; sh = dh;
    }
    hipLaunchKernelGGL(sh_export_kernel, dim3(1), dim3(64), 0, stream, shScratch + off, sw, sh, sh9);
}
// FScene::UpdateDynamicSky: the whole chain on one stream.  `shScratch` holds the SH projection mip chain.
extern "C" hipError_t incr_launch_update_dynamic_sky(const IblParams* p, hipStream_t stream) {
    const uint32_t N = p->cubemapRes, W = 2 * N, H = N, mips = p->mips;
    hipLaunchKernelGGL(sky_latlong_kernel, grid64x4(W, H), dim3(256), 0, stream, p->latlong, W, H, p->perez, p->sun, p->rounding);
    for (uint32_t m = 1; m < mips; ++m) {         // scene.cpp:1720-1730
        const uint32_t dw = std::max(W >> m, 1u), dh = std::max(H >> m, 1u);
        hipLaunchKernelGGL(downsample2d_kernel, grid64x4(dw, dh), dim3(256), 0, stream, p->latlong + p->latlongOffset[m - 1], std::max(W >> (m - 1), 1u),
                           p->latlong + p->latlongOffset[m], dw, dh, p->rounding);
    }
    for (uint32_t m = 0; m < mips; ++m) {         // renderer.cpp:450-466
        const uint32_t n = std::max(N >> m, 1u);
        hipLaunchKernelGGL(cubemapgen_kernel<uint32_t>, grid64x4(n, n), dim3(256), 0, stream, (const uint32_t*)(p->latlong + p->latlongOffset[m]), std::max(W >> m, 1u), std::max(H >> m, 1u),
                           p->srcCube + p->cubeOffset[m], n, p->rounding);
    }
    const CubeChain src = {N, mips, p->srcCube};
    for (uint32_t m = 0; m < mips; ++m) {         // renderer.cpp:504-526
        const uint32_t n = std::max(N >> m, 1u);
        hipLaunchKernelGGL(prefilter_kernel, dim3((n + 15) / 16, (n + 15) / 16, 6), dim3(256), 0, stream, src, n, N, p->envmap + p->cubeOffset[m], p->sampleCount,
                           (float)m / (float)mips, p->rounding);
    }
    // ShEncode (renderer.cpp:533-800)
    hipLaunchKernelGGL(sh_project_kernel<uint32_t>, grid64x4(W, H), dim3(256), 0, stream, (const uint32_t*)p->latlong, W, H, p->shScratch);
    sh_reduce_and_export(p->shScratch, W, H, p->shMips, p->sh9, stream);
    return hipGetLastError();
}
// FTextureCache::CacheHDRI (src/demo.cpp:292-434) behind the file read: RGBA32F lat-long (W x H = 2N x N) -> `mips` = CalcMipCount(W, H, true) levels -> an N^2 source cube of as many
// levels (level m from lat-long level m) -> the prefiltered cube of HALF the size with one level fewer, roughness m / (mips - 1) -> SH9 from level 0 of the lat-long.
extern "C" hipError_t incr_launch_cache_hdri(const HdriParams* p, hipStream_t stream) {
    const uint32_t N = p->cubemapRes, W = p->width, H = p->height, mips = p->mips;
    for (uint32_t m = 1; m < mips; ++m) {
        const uint32_t dw = std::max(W >> m, 1u), dh = std::max(H >> m, 1u);
        hipLaunchKernelGGL(downsample2d_f32_kernel, grid64x4(dw, dh), dim3(256), 0, stream, p->latlong + p->latlongOffset[m - 1], std::max(W >> (m - 1), 1u), p->latlong + p->latlongOffset[m], dw, dh);
    }
    for (uint32_t m = 0; m < mips; ++m) {         // renderer.cpp:450-466
        const uint32_t n = std::max(N >> m, 1u);
        hipLaunchKernelGGL(cubemapgen_kernel<float4>, grid64x4(n, n), dim3(256), 0, stream, (const float4*)(p->latlong + p->latlongOffset[m]), std::max(W >> m, 1u), std::max(H >> m, 1u),
                           p->srcCube + p->cubeOffset[m], n, p->rounding);
    }
    const CubeChain src = {N, mips, p->srcCube};
    const uint32_t envN = N >> 1, envMips = mips - 1;
    for (uint32_t m = 0; m < envMips; ++m) {      // demo.cpp:357-369 -> renderer.cpp:504-526: cubemapRes = filteredEnvmapSize, numMips = filteredEnvmapMips
        const uint32_t n = envN >> m;
        hipLaunchKernelGGL(prefilter_kernel, dim3((n + 15) / 16, (n + 15) / 16, 6), dim3(256), 0, stream, src, n, envN, p->envmap + p->envOffset[m], p->sampleCount,
                           (float)m / (float)envMips, p->rounding);
    }
    hipLaunchKernelGGL(sh_project_kernel<float4>, grid64x4(W, H), dim3(256), 0, stream, (const float4*)p->latlong, W, H, p->shScratch);      // demo.cpp:398: from the source image
    sh_reduce_and_export(p->shScratch, W, H, p->shMips, p->sh9, stream);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_env_brdf(uint32_t* out, uint32_t W, uint32_t H, uint32_t numSamples, int rounding, hipStream_t stream) {
    if (!W || !H) return hipSuccess;
    hipLaunchKernelGGL(env_brdf_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, stream, out, W, H, numSamples, rounding);
    return hipGetLastError();
}
