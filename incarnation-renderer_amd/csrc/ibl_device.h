// ibl_device.h -- device-side texture-unit stand-ins for the environment maps: packed mip chains of R11G11B10_FLOAT
// lat-long textures and cube maps, sampled with the pinned rules of DESIGN.md section 3 (2D: bilinear + WRAP; cube: D3D
// face table, bilinear with taps across face edges taken from the adjacent face (seamless), trilinear c0 (1 - f) + c1 f).  Same
// arithmetic as oracle/oracle_ibl.h.
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
    int x0, y0; float fx, fy;
    linear_address(u * (float)W, &x0, &fx); linear_address(v * (float)H, &y0, &fy);                      // u, v in [0, 1] => floor in [-1, W - 1]: wrap without integer division
    x0 = x0 < 0 ? x0 + (int)W : x0; y0 = y0 < 0 ? y0 + (int)H : y0;
    const int x1 = x0 + 1 == (int)W ? 0 : x0 + 1, y1 = y0 + 1 == (int)H ? 0 : y0 + 1;
    const f3 t00 = unpack_r11g11b10(img[(size_t)y0 * W + x0]), t10 = unpack_r11g11b10(img[(size_t)y0 * W + x1]);
    const f3 t01 = unpack_r11g11b10(img[(size_t)y1 * W + x0]), t11 = unpack_r11g11b10(img[(size_t)y1 * W + x1]);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}

// the same tap on an R32G32B32A32_FLOAT level (the Radiance HDRI of FTextureCache::CacheHDRI: DirectX::LoadFromHDRFile decodes to RGBA32F)
DI f3 sample2d_bilinear_wrap(const float4* __restrict__ img, uint32_t W, uint32_t H, f2 uv) {
    float u = uv.x - floorf(uv.x), v = uv.y - floorf(uv.y);
    if (!(u >= 0.f && u <= 1.f)) u = 0.f;
    if (!(v >= 0.f && v <= 1.f)) v = 0.f;
    int x0, y0; float fx, fy;
    linear_address(u * (float)W, &x0, &fx); linear_address(v * (float)H, &y0, &fy);
    x0 = x0 < 0 ? x0 + (int)W : x0; y0 = y0 < 0 ? y0 + (int)H : y0;
    const int x1 = x0 + 1 == (int)W ? 0 : x0 + 1, y1 = y0 + 1 == (int)H ? 0 : y0 + 1;
    const float4 a = img[(size_t)y0 * W + x0], b = img[(size_t)y0 * W + x1], c = img[(size_t)y1 * W + x0], d = img[(size_t)y1 * W + x1];
    const f3 t00 = {a.x, a.y, a.z}, t10 = {b.x, b.y, b.z}, t01 = {c.x, c.y, c.z}, t11 = {d.x, d.y, d.z};
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}
DI f3 load_radiance(const uint32_t* __restrict__ img, size_t i) { return unpack_r11g11b10(img[i]); }
DI f3 load_radiance(const float4* __restrict__ img, size_t i) { const float4 t = img[i]; return {t.x, t.y, t.z}; }

DI void cube_face_uv(f3 d, uint32_t* face, float* u, float* v) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.f) { *face = 0; sc = -d.z; tc = -d.y; } else { *face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.f) { *face = 2; sc = d.x; tc = d.z; } else { *face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0.f) { *face = 4; sc = d.x; tc = -d.y; } else { *face = 5; sc = -d.x; tc = -d.y; } }
    *u = (sc / ma + 1.f) * 0.5f; *v = (tc / ma + 1.f) * 0.5f;
}
// SEAMLESS cube filtering (D3D11.3 functional spec, TextureCube edge handling: a linear tap that falls off a face is taken from the
// adjacent face; every TextureCube.Sample / SampleLevel of the reference goes through it -- pathtracing.hlsl:243, 308, sky-lighting.hlsl:123,
// envmap.hlsl:47, prefilter.hlsl's taps).  cube_unfold: texel (x, y) of `face`, one step outside the face on exactly ONE side, -> the
// texel across that edge, in exact integer arithmetic: texel centres in doubled cube coordinates (a = 2x + 1 - n in the face plane, +-n
// along the major axis); the component that left the cube is pulled back onto its face (+-n) and the old major component moves half a
// texel in (+-(n - 1)).  At a cube CORNER three faces meet and the fourth tap of the 2 x 2 footprint does not exist: it is the
// average of the other three (the rule GL 4.x recommends and D3D hardware follows; pinned: sum in tap order, times float(1 / 3)).
DI void cube_unfold(uint32_t n, uint32_t* face, int* x, int* y) {
    const int N = (int)n, M = N - 1, a = 2 * *x + 1 - N, b = 2 * *y + 1 - N;
    int px, py, pz;
    switch (*face) {                                 // (sc, tc, ma) -> position on the cube, as cube_face_uv
        case 0: px = N; py = -b; pz = -a; break;
        case 1: px = -N; py = -b; pz = a; break;
        case 2: px = a; py = N; pz = b; break;
        case 3: px = a; py = -N; pz = -b; break;
        case 4: px = a; py = -b; pz = N; break;
        default: px = -a; py = -b; pz = -N; break;
    }
    const uint32_t major = *face >> 1;
    const bool ox = major != 0u && (px > N || px < -N), oy = major != 1u && (py > N || py < -N);      // else z left the cube
    if (major == 0u) px = px > 0 ? M : -M; else if (major == 1u) py = py > 0 ? M : -M; else pz = pz > 0 ? M : -M;
    uint32_t f;
    if (ox) { f = px < 0 ? 1u : 0u; px = px < 0 ? -N : N; }
    else if (oy) { f = py < 0 ? 3u : 2u; py = py < 0 ? -N : N; }
    else { f = pz < 0 ? 5u : 4u; pz = pz < 0 ? -N : N; }
    int a2, b2;
    switch (f) {
        case 0: a2 = -pz; b2 = -py; break;
        case 1: a2 = pz; b2 = -py; break;
        case 2: a2 = px; b2 = pz; break;
        case 3: a2 = px; b2 = -pz; break;
        case 4: a2 = px; b2 = -py; break;
        default: a2 = -px; b2 = -py; break;
    }
    *face = f; *x = (a2 + M) >> 1; *y = (b2 + M) >> 1;
}
DI f3 cube_texel(const uint32_t* __restrict__ level0, uint32_t n, uint32_t face, int x, int y) {      // (x, y) off the face on at most one side
    if (x < 0 || y < 0 || x >= (int)n || y >= (int)n) cube_unfold(n, &face, &x, &y);
    return unpack_r11g11b10(level0[((size_t)face * n + (uint32_t)y) * n + (uint32_t)x]);
}
DI f3 cube_sample_level(const CubeChain& c, f3 dir, uint32_t level) {
    uint32_t face; float u, v;
    cube_face_uv(dir, &face, &u, &v);
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    const uint32_t n = mip_dim(c.N, level);
    const uint32_t* __restrict__ lvl = c.data + mip_offset_cube(c.N, level);
    int x0, y0; float fx, fy;                                                      // u, v in [0, 1]: x0 in [-1, n - 1], x1 in [0, n]
    linear_address(u * (float)n, &x0, &fx); linear_address(v * (float)n, &y0, &fy);
    const int x1 = x0 + 1, y1 = y0 + 1;
    f3 t00, t10, t01, t11;
    if (x0 >= 0 && y0 >= 0 && x1 < (int)n && y1 < (int)n) {                         // the footprint lies inside the face
        const uint32_t* __restrict__ img = lvl + (size_t)face * n * n;
        t00 = unpack_r11g11b10(img[(size_t)y0 * n + x0]); t10 = unpack_r11g11b10(img[(size_t)y0 * n + x1]);
        t01 = unpack_r11g11b10(img[(size_t)y1 * n + x0]); t11 = unpack_r11g11b10(img[(size_t)y1 * n + x1]);
    } else {
        const bool xl = x0 < 0, xh = x1 >= (int)n, yl = y0 < 0, yh = y1 >= (int)n;
        const bool m00 = xl && yl, m10 = xh && yl, m01 = xl && yh, m11 = xh && yh;   // the tap beyond a cube corner (at most one)
        const f3 zero = {0.f, 0.f, 0.f};
        t00 = m00 ? zero : cube_texel(lvl, n, face, x0, y0); t10 = m10 ? zero : cube_texel(lvl, n, face, x1, y0);
        t01 = m01 ? zero : cube_texel(lvl, n, face, x0, y1); t11 = m11 ? zero : cube_texel(lvl, n, face, x1, y1);
        if (m00 || m10 || m01 || m11) {
            const f3 avg = (((t00 + t10) + t01) + t11) * 0.333333343f;
            if (m00) t00 = avg; else if (m10) t10 = avg; else if (m01) t01 = avg; else t11 = avg;
        }
    }
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
