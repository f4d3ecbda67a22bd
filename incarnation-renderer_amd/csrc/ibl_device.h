// ibl_device.h -- device-side texture-unit stand-ins for the environment maps: packed mip chains of R11G11B10_FLOAT
// lat-long textures and cube maps, sampled with the pinned rules of DESIGN.md section 3 (2D: bilinear + WRAP; cube: D3D
// face table, bilinear inside the face, trilinear c0 (1 - f) + c1 f).  Same arithmetic as oracle/oracle_ibl.h.
#pragma once
#include "pt_device.h"

namespace incr {

struct CubeChain { uint32_t N, mips; const uint32_t* __restrict__ data; };     // levels packed [level][face][y][x]

DI uint32_t mip_dim(uint32_t n, uint32_t level) { const uint32_t d = n >> level; return d ? d : 1u; }
DI size_t mip_offset_2d(uint32_t W, uint32_t H, uint32_t level) { size_t o = 0; for (uint32_t k = 0; k < level; ++k) o += (size_t)mip_dim(W, k) * mip_dim(H, k); return o; }
DI size_t mip_offset_cube(uint32_t N, uint32_t level) { size_t o = 0; for (uint32_t k = 0; k < level; ++k) o += (size_t)6 * mip_dim(N, k) * mip_dim(N, k); return o; }

DI f3 sample2d_bilinear_wrap(const uint32_t* __restrict__ img, uint32_t W, uint32_t H, f2 uv) {
    float u = uv.x - floorf(uv.x), v = uv.y - floorf(uv.y);
    if (!(u >= 0.f && u <= 1.f)) u = 0.f;
    if (!(v >= 0.f && v <= 1.f)) v = 0.f;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;                      // u, v in [0, 1] => floor in [-1, W - 1]: wrap without integer division
    x0 = x0 < 0 ? x0 + (int)W : x0; y0 = y0 < 0 ? y0 + (int)H : y0;
    const int x1 = x0 + 1 == (int)W ? 0 : x0 + 1, y1 = y0 + 1 == (int)H ? 0 : y0 + 1;
    const f3 t00 = unpack_r11g11b10(img[(size_t)y0 * W + x0]), t10 = unpack_r11g11b10(img[(size_t)y0 * W + x1]);
    const f3 t01 = unpack_r11g11b10(img[(size_t)y1 * W + x0]), t11 = unpack_r11g11b10(img[(size_t)y1 * W + x1]);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}

DI void cube_face_uv(f3 d, uint32_t* face, float* u, float* v) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.f) { *face = 0; sc = -d.z; tc = -d.y; } else { *face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.f) { *face = 2; sc = d.x; tc = d.z; } else { *face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0.f) { *face = 4; sc = d.x; tc = -d.y; } else { *face = 5; sc = -d.x; tc = -d.y; } }
    *u = (sc / ma + 1.f) * 0.5f; *v = (tc / ma + 1.f) * 0.5f;
}
DI f3 cube_sample_level(const CubeChain& c, f3 dir, uint32_t level) {
    uint32_t face; float u, v;
    cube_face_uv(dir, &face, &u, &v);
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    const uint32_t n = mip_dim(c.N, level);
    const uint32_t* __restrict__ img = c.data + mip_offset_cube(c.N, level) + (size_t)face * n * n;
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    x = fminf(fmaxf(x, 0.f), (float)(n - 1)); y = fminf(fmaxf(y, 0.f), (float)(n - 1));
    const float x0f = floorf(x), y0f = floorf(y), fx = x - x0f, fy = y - y0f;
    const uint32_t x0 = (uint32_t)x0f, y0 = (uint32_t)y0f, x1 = min(x0 + 1, n - 1), y1 = min(y0 + 1, n - 1);
    const f3 t00 = unpack_r11g11b10(img[(size_t)y0 * n + x0]), t10 = unpack_r11g11b10(img[(size_t)y0 * n + x1]);
    const f3 t01 = unpack_r11g11b10(img[(size_t)y1 * n + x0]), t11 = unpack_r11g11b10(img[(size_t)y1 * n + x1]);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}
DI f3 cube_sample_trilinear(const CubeChain& c, f3 dir, float level) {
    if (!(level > 0.f)) level = 0.f;
    const float top = (float)(c.mips - 1);
    if (level > top) level = top;
    const float l0 = floorf(level), f = level - l0;
    const uint32_t a = (uint32_t)l0, b = min(a + 1, c.mips - 1);
    const f3 c0 = cube_sample_level(c, dir, a);
    if (f == 0.f) return c0;
    const f3 c1 = cube_sample_level(c, dir, b);
    return c0 * (1.f - f) + c1 * f;
}

// Radiance arriving along `dir` for the path tracer's sky next-event estimate: envmap.SampleLevel(g_envmapSampler, dir, 0)
// (pathtracing.hlsl:240-243; sampler s0: MIN_MAG_LINEAR_MIP_POINT) on the prefiltered sky cube when the caller provides
// it, else the analytic 0.04 * Preetham that level 0 of the cube is made from.
DI f3 sky_env_radiance(const PtParams& P, f3 dir) {
    if (P.envmap) return cube_sample_level(CubeChain{P.envmapRes, P.envmapMips, P.envmap}, dir, 0u);
    return 0.04f * CalculateSkyRadianceRGB(P.sunEnv, dir, P.perez);
}

// Miss shaders (pathtracing.hlsl:304-319): msEnvmap adds attenuation * skyBrightness * envmap.SampleLevel(dir, 0),
// msDynamicSky adds attenuation * 1000 * Preetham(dir).
DI f3 sky_miss_add(const PtParams& P, f3 attenuation, f3 dir) {
    if (P.envSkyMode == 0) return (attenuation * P.skyBrightness) * cube_sample_level(CubeChain{P.envmapRes, P.envmapMips, P.envmap}, dir, 0u);
    return attenuation * 1000 * CalculateSkyRadianceRGB(P.sunMiss, dir, P.perez);
}

DI f3 latlong_dir(f2 uv) {            // common/math.hlsli:46-79: LatlongUV2Polar + Polar2Cartesian(..., World)
    const float theta = K_PI * uv.y, phi = K_PI * (uv.x * 2.f - 1.f);
    const float sint = sinf(theta), cost = cosf(theta), sinp = sinf(phi), cosp = cosf(phi);
    return {sint * sinp, cost, sint * cosp};
}
DI f2 Hammersley(uint32_t i, float numSamples) {            // common/uniform-sampling.hlsli:141-150
    uint32_t bits = i;
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return {(float)i / numSamples, (float)bits / 4294967296.f};
}

} // namespace incr
