// TEST INFRASTRUCTURE ONLY -- CPU oracle for the sky / image-based-lighting precompute that feeds the hot path
// (SURVEY.md 8f rank 2): FScene::UpdateDynamicSky (src/scene.cpp:1657-1790) = Preetham lat-long texture -> mip chain ->
// cube map -> GGX prefilter -> SH9 projection, and Renderer::GenerateEnvBrdfTexture (src/renderer.cpp:169-245).
// PARITY UNPINNED (see oracle_math.h).  Included at the end of oracle_pt.cpp.
//
// Pinned texture-unit rules: 2D bilinear with WRAP addressing on both axes, texel centre +0.5, float weights,
// (t00 wx0 + t10 fx) wy0 + (t01 wx0 + t11 fx) fy; "MIP_POINT" with an integer level reads exactly that level.  Cube maps:
// face = major axis (x before y before z on ties), D3D (sc, tc, ma) table, bilinear with SEAMLESS edges (taps beyond a face edge come
// from the adjacent face, the tap beyond a cube corner is the mean of the other three: D3D11.3 TextureCube edge handling), trilinear =
// c0 (1 - f) + c1 f between the two nearest levels, level clamped to [0, mips - 1].
// Surfaces are R11G11B10_FLOAT (radiance) / RGBA32F (SH) / R16G16_FLOAT (BRDF LUT), stores as `rounding` says.
#pragma once

struct MipChain2D { uint32_t W, H, mips; const uint32_t* data; };
static inline size_t mip_offset_2d(uint32_t W, uint32_t H, uint32_t level) { size_t o = 0; for (uint32_t k = 0; k < level; ++k) o += (size_t)std::max(W >> k, 1u) * std::max(H >> k, 1u); return o; }
static inline size_t mip_offset_cube(uint32_t N, uint32_t level) { size_t o = 0; for (uint32_t k = 0; k < level; ++k) o += (size_t)6 * std::max(N >> k, 1u) * std::max(N >> k, 1u); return o; }
static inline uint32_t mip_count(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 1 && h >= 1) { ++n; w >>= 1; h >>= 1; } return n; }   // RenderUtils12::CalcMipCount

struct float4f { float x, y, z, w; };          // an RGBA32F texel (the decoded Radiance picture of FTextureCache::CacheHDRI)
static inline float3 load_radiance(const uint32_t* img, size_t i) { return unpack_r11g11b10(img[i]); }
static inline float3 load_radiance(const float4f* img, size_t i) { return {img[i].x, img[i].y, img[i].z}; }
template <class TEXEL>
static float3 sample2d_bilinear_wrap(const TEXEL* img, uint32_t W, uint32_t H, float2 uv) {
    float u = uv.x - floorf(uv.x), v = uv.y - floorf(uv.y);
    if (!(u >= 0.f && u <= 1.f)) u = 0.f;
    if (!(v >= 0.f && v <= 1.f)) v = 0.f;
    int xl, yl; float fx, fy;
    linear_address(u * (float)W, &xl, &fx); linear_address(v * (float)H, &yl, &fy);
    const int x0 = ((xl % (int)W) + (int)W) % (int)W, y0 = ((yl % (int)H) + (int)H) % (int)H;
    const int x1 = (x0 + 1) % (int)W, y1 = (y0 + 1) % (int)H;
    const float3 t00 = load_radiance(img, (size_t)y0 * W + x0), t10 = load_radiance(img, (size_t)y0 * W + x1);
    const float3 t01 = load_radiance(img, (size_t)y1 * W + x0), t11 = load_radiance(img, (size_t)y1 * W + x1);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}

// cube map: all levels packed [level][face][y][x]
struct CubeChain { uint32_t N, mips; const uint32_t* data; };
static void cube_face_uv(float3 d, uint32_t* face, float* u, float* v) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.f) { *face = 0; sc = -d.z; tc = -d.y; } else { *face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.f) { *face = 2; sc = d.x; tc = d.z; } else { *face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0.f) { *face = 4; sc = d.x; tc = -d.y; } else { *face = 5; sc = -d.x; tc = -d.y; } }
    *u = (sc / ma + 1.f) * 0.5f; *v = (tc / ma + 1.f) * 0.5f;
}
// D3D11.3 TextureCube edge handling: a linear tap beyond a face edge comes from the adjacent face.  Restated geometrically: the texel
// centre as a point on the cube surface in doubled integer coordinates (face plane at +-n along the major axis, centres at odd
// offsets 2 i + 1 - n inside it); a centre one step beyond an edge has one in-plane coordinate of magnitude n + 1: fold it over the
// edge -- that coordinate becomes +-n (the new face's plane), the former major coordinate +-(n - 1) (the first texel row of the new
// face) -- and read the new face's (sc, tc) back.  The tap beyond a CORNER has no texel: average of the footprint's other three.
static void cube_fold_over_edge(uint32_t n, uint32_t* face, int* x, int* y) {
    const int N = (int)n;
    const int sc2 = 2 * *x + 1 - N, tc2 = 2 * *y + 1 - N;
    int p[3];
    switch (*face) {
        case 0: p[0] = N;  p[1] = -tc2; p[2] = -sc2; break;      // +X: sc = -z, tc = -y
        case 1: p[0] = -N; p[1] = -tc2; p[2] = sc2;  break;      // -X: sc = +z, tc = -y
        case 2: p[0] = sc2; p[1] = N;  p[2] = tc2;  break;       // +Y: sc = +x, tc = +z
        case 3: p[0] = sc2; p[1] = -N; p[2] = -tc2; break;       // -Y: sc = +x, tc = -z
        case 4: p[0] = sc2; p[1] = -tc2; p[2] = N;  break;       // +Z: sc = +x, tc = -y
        default: p[0] = -sc2; p[1] = -tc2; p[2] = -N; break;     // -Z: sc = -x, tc = -y
    }
    const int major = (int)(*face >> 1);
    int over = -1;
    for (int k = 0; k < 3; ++k) if (k != major && std::abs(p[k]) > N) over = k;
    p[major] = p[major] > 0 ? N - 1 : -(N - 1);
    p[over] = p[over] > 0 ? N : -N;
    const uint32_t nf = (uint32_t)over * 2u + (p[over] < 0 ? 1u : 0u);
    int s2, t2;
    switch (nf) {
        case 0: s2 = -p[2]; t2 = -p[1]; break;
        case 1: s2 = p[2];  t2 = -p[1]; break;
        case 2: s2 = p[0];  t2 = p[2];  break;
        case 3: s2 = p[0];  t2 = -p[2]; break;
        case 4: s2 = p[0];  t2 = -p[1]; break;
        default: s2 = -p[0]; t2 = -p[1]; break;
    }
    *face = nf; *x = (s2 + N - 1) / 2; *y = (t2 + N - 1) / 2;
}
static float3 cube_sample_level(const CubeChain& c, float3 dir, uint32_t level) {
    uint32_t face; float u, v;
    cube_face_uv(dir, &face, &u, &v);
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    const uint32_t n = std::max(c.N >> level, 1u);
    const uint32_t* lvl = c.data + mip_offset_cube(c.N, level);
    int xl, yl; float fx, fy;
    linear_address(u * (float)n, &xl, &fx); linear_address(v * (float)n, &yl, &fy);
    const int xs[2] = {xl, xl + 1}, ys[2] = {yl, yl + 1};
    float3 t[2][2]; bool missing[2][2]; bool anyMissing = false;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) {
        const bool offX = xs[i] < 0 || xs[i] >= (int)n, offY = ys[j] < 0 || ys[j] >= (int)n;
        missing[j][i] = offX && offY; anyMissing = anyMissing || missing[j][i];
        t[j][i] = float3{0.f, 0.f, 0.f};
        if (missing[j][i]) continue;
        uint32_t f = face; int tx = xs[i], ty = ys[j];
        if (offX || offY) cube_fold_over_edge(n, &f, &tx, &ty);
        t[j][i] = unpack_r11g11b10(lvl[((size_t)f * n + (uint32_t)ty) * n + (uint32_t)tx]);
    }
    if (anyMissing) {
        const float3 avg = (((t[0][0] + t[0][1]) + t[1][0]) + t[1][1]) * 0.333333343f;
        for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) if (missing[j][i]) t[j][i] = avg;
    }
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t[0][0] * wx0 + t[0][1] * fx) * wy0 + (t[1][0] * wx0 + t[1][1] * fx) * fy;
}
static float3 env_cube_level0(const uint32_t* data, uint32_t N, uint32_t mips, float3 dir) { return cube_sample_level({N, mips, data}, dir, 0u); }
static float3 cube_sample_trilinear(const CubeChain& c, float3 dir, float level) {
    if (!(level > 0.f)) level = 0.f;                       // also NaN
    const float top = (float)(c.mips - 1);
    if (level > top) level = top;
    const float l0 = floorf(level), f = level - l0;
    const uint32_t a = (uint32_t)l0, b = std::min(a + 1, c.mips - 1);
    const float3 c0 = cube_sample_level(c, dir, a);
    if (f == 0.f) return c0;
    const float3 c1 = cube_sample_level(c, dir, b);
    return c0 * (1.f - f) + c1 * f;
}

// common/math.hlsli:46-79
static inline float3 latlong_dir(float2 uv) {
    const float theta = k_Pi * uv.y, phi = k_Pi * (uv.x * 2.f - 1.f);
    const float sint = sinf(theta), cost = cosf(theta), sinp = sinf(phi), cosp = cosf(phi);
    return {sint * sinp, cost, sint * cosp};               // Polar2Cartesian(..., World) = p.xzy
}

// environment-sky/dynamic-sky-spherical-projection.hlsl:24-44 + renderer.cpp:294-360
static void sky_latlong(const float sunDir[3], float turbidity, uint32_t W, uint32_t H, uint32_t* out, int rounding) {
    incr_perez_distribution pc; perez_from_turbidity(turbidity, &pc);
    const SkySunTerms sun = sky_sun_terms(normalize(f3(sunDir[0], sunDir[1], sunDir[2])), turbidity, pc);
    for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) {
        const float2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
        out[(size_t)y * W + x] = pack_r11g11b10(0.04f * CalculateSkyRadianceRGB(sun, normalize(latlong_dir(uv)), pc), rounding);
    }
}
// postprocess/downsample.hlsl:14-32
static void downsample2d(const uint32_t* src, uint32_t srcW, uint32_t* dst, uint32_t dstW, uint32_t dstH, int rounding) {
    for (uint32_t y = 0; y < dstH; ++y) for (uint32_t x = 0; x < dstW; ++x) {
        const float3 a = unpack_r11g11b10(src[(size_t)(2 * y) * srcW + 2 * x]), b = unpack_r11g11b10(src[(size_t)(2 * y) * srcW + 2 * x + 1]);
        const float3 c = unpack_r11g11b10(src[(size_t)(2 * y + 1) * srcW + 2 * x]), d = unpack_r11g11b10(src[(size_t)(2 * y + 1) * srcW + 2 * x + 1]);
        dst[(size_t)y * dstW + x] = pack_r11g11b10(0.25f * (((a + b) + c) + d), rounding);
    }
}
// DirectX::GenerateMipMaps(TEX_FILTER_LINEAR) of the RGBA32F lat-long (demo.cpp:318-320), pinned for exact 2:1 levels: the bilinear tap half way between four
// texels, weights 1/2 per axis, rows first: (a/2 + b/2)/2 + (c/2 + d/2)/2
static void downsample2d_f32(const float4f* src, uint32_t srcW, float4f* dst, uint32_t dstW, uint32_t dstH) {
    for (uint32_t y = 0; y < dstH; ++y) for (uint32_t x = 0; x < dstW; ++x) {
        const float4f a = src[(size_t)(2 * y) * srcW + 2 * x], b = src[(size_t)(2 * y) * srcW + 2 * x + 1], c = src[(size_t)(2 * y + 1) * srcW + 2 * x], d = src[(size_t)(2 * y + 1) * srcW + 2 * x + 1];
        dst[(size_t)y * dstW + x] = {0.5f * (0.5f * a.x + 0.5f * b.x) + 0.5f * (0.5f * c.x + 0.5f * d.x), 0.5f * (0.5f * a.y + 0.5f * b.y) + 0.5f * (0.5f * c.y + 0.5f * d.y),
                                     0.5f * (0.5f * a.z + 0.5f * b.z) + 0.5f * (0.5f * c.z + 0.5f * d.z), 0.5f * (0.5f * a.w + 0.5f * b.w) + 0.5f * (0.5f * c.w + 0.5f * d.w)};
    }
}
// content-pipeline/cubemapgen.hlsl:19-121 for one level
template <class TEXEL>
static void cubemapgen_level(const TEXEL* latlong, uint32_t llW, uint32_t llH, uint32_t* cube, uint32_t n, int rounding) {
    const float faceTransform[6][2] = {{k_PiOver2, 0.f}, {-k_PiOver2, 0.f}, {0.f, k_PiOver2}, {0.f, -k_PiOver2}, {0.f, 0.f}, {k_Pi, 0.f}};
    const float an = sinf(k_PiOver4), ak = cosf(k_PiOver4);
    for (uint32_t y = 0; y < n; ++y) for (uint32_t x = 0; x < n; ++x) {
        float2 nn = {((float)x + 0.5f) / (float)n, ((float)y + 0.5f) / (float)n};
        nn = {(2.f * nn.x - 1.f) * an, (2.f * nn.y - 1.f) * an};
        for (int i = 0; i < 6; ++i) {
            const float ftx = faceTransform[i][0], fty = faceTransform[i][1];
            float2 uv;
            if (fty == 0.f) { uv.x = atan2f(nn.x, ak); uv.y = atan2f(nn.y * cosf(uv.x), ak); uv.x += ftx; }
            else if (fty < 0.f) { const float d = sqrtf(dot(nn, nn)); uv.y = k_PiOver2 - atan2f(d, ak); uv.x = k_PiOver2 + atan2f(nn.y, nn.x); }
            else { const float d = sqrtf(dot(nn, nn)); uv.y = -k_PiOver2 + atan2f(d, ak); uv.x = atan2f(nn.x, nn.y); }
            uv.x = uv.x * k_InvPi; uv.y = uv.y / k_PiOver2;
            while (uv.y < -1) { uv.y += 2; uv.x += 1; }
            while (uv.y > 1) { uv.y -= 2; uv.x += 1; }
            while (uv.x < -1) uv.x += 2;
            while (uv.x > 1) uv.x -= 2;
            uv = {0.5f * uv.x + 0.5f, 0.5f * uv.y + 0.5f};
            cube[((size_t)i * n + y) * n + x] = pack_r11g11b10(sample2d_bilinear_wrap(latlong, llW, llH, uv), rounding);
        }
    }
}
// common/uniform-sampling.hlsli:141-150
static inline float2 Hammersley(uint32_t i, float numSamples) {
    uint32_t bits = i;
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return {(float)i / numSamples, (float)bits / 4294967296.f};
}
static inline float3 GetEnvDir(uint32_t face, float2 uv) {                  // split-sum-approx/prefilter.hlsl:26-41
    const float vx = 2.f * uv.x - 1.f, vy = -2.f * uv.y + 1.f;
    switch (face) {
    case 0: return normalize(f3(1.f, vy, -vx));
    case 1: return normalize(f3(-1.f, vy, vx));
    case 2: return normalize(f3(vx, 1.f, -vy));
    case 3: return normalize(f3(vx, -1.f, vy));
    case 4: return normalize(f3(vx, vy, 1.f));
    default: return normalize(f3(-vx, vy, -1.f));
    }
}
// split-sum-approx/prefilter.hlsl:43-95 for one level (all six faces)
// mipSize / cubemapSize = the shader's root constants: the output level's size and the OUTPUT cube's size (renderer.cpp:504-526)
static void prefilter_level(const CubeChain& src, uint32_t mipSize, uint32_t cubemapSize, uint32_t* dst, uint32_t sampleCount, float roughness, int rounding) {
    const float Resolution = (float)cubemapSize;
    for (uint32_t face = 0; face < 6; ++face) for (uint32_t y = 0; y < mipSize; ++y) for (uint32_t x = 0; x < mipSize; ++x) {
        const float3 R = GetEnvDir(face, {(float)x / (float)mipSize, (float)y / (float)mipSize});
        const float3 N = R, V = R;
        const float3 up = fabsf(N.z) < 0.999f ? f3(0, 0, 1) : f3(1, 0, 0);   // TangentToWorld (common/math.hlsli:82-89)
        const float3 T = normalize(cross(up, N)), B = cross(N, T);
        float3 color = {0, 0, 0}; float totalWeight = 0.f;
        for (uint32_t i = 0; i < sampleCount; ++i) {
            const float2 Xi = Hammersley(i, (float)sampleCount);
            float3 H = SampleGGX(Xi, roughness);
            H = normalize(f3((H.x * T.x + H.y * B.x) + H.z * N.x, (H.x * T.y + H.y * B.y) + H.z * N.y, (H.x * T.z + H.y * B.z) + H.z * N.z));   // mul(H, float3x3(T,B,N))
            const float3 L = normalize(reflect(-V, H));
            const float NoL = saturate(dot(N, L));
            if (NoL > 0.f) {
                const float NoH = hmax(dot(N, H), 0.0f), VoH = hmax(dot(V, H), 0.0f);
                const float D = GGX(NoH, roughness);
                const float pdf = (D * NoH / (4.0f * VoH)) + 0.0001f;
                const float saTexel = 4.f * k_Pi / (6.0f * Resolution * Resolution);
                const float saSample = 1.f / ((float)sampleCount * pdf + 0.0001f);
                const float mipLevel = roughness == 0.0f ? 0.0f : 0.5f * log2f(saSample / saTexel);
                color += cube_sample_trilinear(src, L, mipLevel) * NoL;
                totalWeight += NoL;
            }
        }
        dst[((size_t)face * mipSize + y) * mipSize + x] = pack_r11g11b10(color / totalWeight, rounding);
    }
}
// spherical-harmonics/{projection,parallel-reduction,export}.hlsl + renderer.cpp:533-800 (srcMipOffset 0)
template <class TEXEL>
static void sh_encode(const TEXEL* latlong, uint32_t W, uint32_t H, float out9x4[36]) {
    static const float K[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
    std::vector<float4> cur((size_t)W * H * 9), next;
    const float dTheta = k_Pi / (float)H, dPhi = 2.f * k_Pi / (float)W;
    for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) {
        const float2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
        const float3 dir = latlong_dir(uv);
        float b[9];
        b[0] = K[0]; b[1] = K[1] * dir.y; b[2] = K[2] * dir.z; b[3] = K[3] * dir.x;
        b[4] = (K[4] * dir.x) * dir.y; b[5] = (K[5] * dir.y) * dir.z; b[6] = K[6] * ((3.0f * dir.z) * dir.z - 1.0f); b[7] = (K[7] * dir.x) * dir.z; b[8] = K[8] * (dir.x * dir.x - dir.y * dir.y);
        const float3 rad = load_radiance(latlong, (size_t)y * W + x);
        const float sint = sinf(k_Pi * uv.y);
        for (int i = 0; i < 9; ++i) {
            const float w = b[i];
            cur[((size_t)i * H + y) * W + x] = {(((rad.x * w) * sint) * dTheta) * dPhi, (((rad.y * w) * sint) * dTheta) * dPhi, (((rad.z * w) * sint) * dTheta) * dPhi,
                                                (((1.f * w) * sint) * dTheta) * dPhi};      // Texture2D.Load of an R11G11B10 texel has alpha 1
        }
    }
    uint32_t w = W, h = H;
    const uint32_t mips = mip_count(W, H);
    for (uint32_t m = 0; m + 1 < mips; ++m) {
        const uint32_t dw = std::max(W >> (m + 1), 1u), dh = std::max(H >> (m + 1), 1u);
        next.assign((size_t)dw * dh * 9, {0, 0, 0, 0});
        for (int i = 0; i < 9; ++i) for (uint32_t y = 0; y < dh; ++y) for (uint32_t x = 0; x < dw; ++x) {
            auto at = [&](uint32_t xx, uint32_t yy) -> float4 { if (xx >= w || yy >= h) return {0, 0, 0, 0}; return cur[((size_t)i * h + yy) * w + xx]; };
            next[((size_t)i * dh + y) * dw + x] = ((at(2 * x, 2 * y) + at(2 * x + 1, 2 * y)) + at(2 * x, 2 * y + 1)) + at(2 * x + 1, 2 * y + 1);
        }
        cur.swap(next); w = dw; h = dh;
    }
    for (int i = 0; i < 9; ++i) {
        const float4 a = cur[((size_t)i * h + 0) * w + 0], b = w > 1 ? cur[((size_t)i * h + 0) * w + 1] : float4{0, 0, 0, 0};
        const float4 s = a + b;
        out9x4[i * 4 + 0] = s.x; out9x4[i * 4 + 1] = s.y; out9x4[i * 4 + 2] = s.z; out9x4[i * 4 + 3] = s.w;
    }
}
// split-sum-approx/brdf-integration.hlsl:23-67 (G_Smith_IBL: lighting/pbr.hlsli:53-60)
static void env_brdf(uint32_t W, uint32_t H, uint32_t numSamples, uint32_t* outRG16F, int rounding) {
    for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W; ++x) {
        const float NoV = (float)x / (float)W, Roughness = 1.f - (float)y / (float)H;
        const float3 V = {sqrtf(1.f - NoV * NoV), 0.f, NoV};
        float A = 0.f, Bv = 0.f;
        for (uint32_t i = 0; i < numSamples; ++i) {
            const float2 Xi = Hammersley(i, (float)numSamples);
            const float3 Hh = SampleGGX(Xi, Roughness);
            const float3 L = normalize(reflect(-V, Hh));
            const float NoL = saturate(L.z), NoH = saturate(Hh.z), VoH = saturate(dot(V, Hh));
            if (NoL > 0.f && NoV > 0.f) {
                const float a2 = Roughness * Roughness, k = 0.5f * a2 * a2;
                const float G = G_SchlickGGX(NoV, k) * G_SchlickGGX(NoL, k);
                const float G_Vis = G * VoH / (NoH * NoV);
                const float Fc = pow5(1.f - VoH);
                A += (1 - Fc) * G_Vis; Bv += Fc * G_Vis;
            }
        }
        outRG16F[(size_t)y * W + x] = (uint32_t)f32_to_f16(A / (float)numSamples, rounding) | ((uint32_t)f32_to_f16(Bv / (float)numSamples, rounding) << 16);
    }
}

// FScene::UpdateDynamicSky (src/scene.cpp:1657-1757): every product of the chain, packed level after level
static void update_dynamic_sky(const float sunDir[3], float turbidity, uint32_t cubemapRes, uint32_t sampleCount, int rounding,
                               uint32_t* latlong, uint32_t* srcCube, uint32_t* envmap, float sh9x4[36]) {
    const uint32_t W = 2 * cubemapRes, H = cubemapRes, mips = mip_count(cubemapRes, cubemapRes);
    sky_latlong(sunDir, turbidity, W, H, latlong, rounding);
    for (uint32_t m = 1; m < mips; ++m)           // scene.cpp:1720-1730: while (mipResY >= 1)
        downsample2d(latlong + mip_offset_2d(W, H, m - 1), std::max(W >> (m - 1), 1u), latlong + mip_offset_2d(W, H, m), std::max(W >> m, 1u), std::max(H >> m, 1u), rounding);
    for (uint32_t m = 0; m < mips; ++m)           // renderer.cpp:450-466
        cubemapgen_level(latlong + mip_offset_2d(W, H, m), std::max(W >> m, 1u), std::max(H >> m, 1u), srcCube + mip_offset_cube(cubemapRes, m), std::max(cubemapRes >> m, 1u), rounding);
    const CubeChain src = {cubemapRes, mips, srcCube};
    for (uint32_t m = 0; m < mips; ++m)           // renderer.cpp:504-526: roughness = mipIndex / numMips
        prefilter_level(src, std::max(cubemapRes >> m, 1u), cubemapRes, envmap + mip_offset_cube(cubemapRes, m), sampleCount, (float)m / (float)mips, rounding);
    sh_encode(latlong, W, H, sh9x4);
}
// FTextureCache::CacheHDRI (src/demo.cpp:292-434) behind the file read: `latlong` holds level 0 (W x H RGBA32F, W = 2 H) on entry and the whole chain on return
static inline uint32_t mip_count_compressed(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 4 && h >= 4) { ++n; w >>= 1; h >>= 1; } return n; }   // CalcMipCount(w, h, true)
static void cache_hdri(float4f* latlong, uint32_t W, uint32_t H, uint32_t sampleCount, int rounding, uint32_t* srcCube, uint32_t* envmap, float sh9x4[36]) {
    const uint32_t N = H, mips = mip_count_compressed(W, H), envN = N >> 1, envMips = mips - 1;
    for (uint32_t m = 1; m < mips; ++m) downsample2d_f32(latlong + mip_offset_2d(W, H, m - 1), W >> (m - 1), latlong + mip_offset_2d(W, H, m), W >> m, H >> m);      // demo.cpp:318-320
    for (uint32_t m = 0; m < mips; ++m) cubemapgen_level(latlong + mip_offset_2d(W, H, m), W >> m, H >> m, srcCube + mip_offset_cube(N, m), N >> m, rounding);      // :352
    const CubeChain src = {N, mips, srcCube};
    for (uint32_t m = 0; m < envMips; ++m)        // :357-369: PrefilterCubemap(src, uavs, filteredEnvmapSize, 0, filteredEnvmapMips)
        prefilter_level(src, envN >> m, envN, envmap + mip_offset_cube(envN, m), sampleCount, (float)m / (float)envMips, rounding);
    sh_encode(latlong, W, H, sh9x4);              // :398: from the source picture
}

extern "C" {
uint64_t orc_ibl_sizes(uint32_t cubemapRes, uint64_t* latlongTexels, uint64_t* cubeTexels) {
    const uint32_t mips = mip_count(cubemapRes, cubemapRes);
    if (latlongTexels) *latlongTexels = mip_offset_2d(2 * cubemapRes, cubemapRes, mips);
    if (cubeTexels) *cubeTexels = mip_offset_cube(cubemapRes, mips);
    return mips;
}
void orc_update_dynamic_sky(const float sunDir[3], float turbidity, uint32_t cubemapRes, uint32_t sampleCount, int rounding,
                            uint32_t* latlong, uint32_t* srcCube, uint32_t* envmap, float* sh9x4) {
    update_dynamic_sky(sunDir, turbidity, cubemapRes, sampleCount, rounding, latlong, srcCube, envmap, sh9x4);
}
// level counts and texel totals of incr_cache_hdri's products for a W x H picture: returns the source levels
uint64_t orc_hdri_sizes(uint32_t W, uint32_t H, uint64_t* latlongTexels, uint64_t* srcCubeTexels, uint64_t* envTexels) {
    const uint32_t mips = mip_count_compressed(W, H);
    if (latlongTexels) *latlongTexels = mip_offset_2d(W, H, mips);
    if (srcCubeTexels) *srcCubeTexels = mip_offset_cube(H, mips);
    if (envTexels) *envTexels = mip_offset_cube(H >> 1, mips - 1);
    return mips;
}
void orc_cache_hdri(float* latlongRgba, uint32_t W, uint32_t H, uint32_t sampleCount, int rounding, uint32_t* srcCube, uint32_t* envmap, float* sh9x4) {
    cache_hdri((float4f*)latlongRgba, W, H, sampleCount, rounding, srcCube, envmap, sh9x4);
}
void orc_env_brdf(uint32_t W, uint32_t H, uint32_t numSamples, uint32_t* outRG16F, int rounding) { env_brdf(W, H, numSamples, outRG16F, rounding); }
void orc_cube_sample(const uint32_t* cube, uint32_t N, uint32_t mips, const float dir[3], float level, float out[3]) {
    const float3 c = cube_sample_trilinear({N, mips, cube}, {dir[0], dir[1], dir[2]}, level); out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
} // extern "C"
