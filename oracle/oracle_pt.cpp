// TEST INFRASTRUCTURE ONLY -- CPU oracle for the path-tracing half of the Incarnation render hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library;
// the shipped product (incarnation-renderer_amd/csrc) never links or loads it.
//
// PARITY UNPINNED: the reference is a Windows-only D3D12 / HLSL SM6.6 / DXR application with no tests
// and no golden vectors (SURVEY.md sections 4 and 8c).  This file restates, in scalar C++, the HLSL the
// reference compiles at run time; the closed fixed-function stages (DXR traversal + ray/triangle test,
// texture units, typed-UAV format conversion) are replaced by the explicit rules listed in DESIGN.md
// ("pinned fixed-function choices").  Every function cites the reference lines it follows, relative to
// /root/reference/source/demo-dll/.
#include "oracle_math.h"
#include "../include/incr.h"   // data contract + pass descs only (no product code)

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

using namespace orc;

// ------------------------------------------------------------------------------------------------
// Sampler -- shaders/common/uniform-sampling.hlsli:152-201 (Kensler 2013 CMJ).  uint32 wrap-around.
// ------------------------------------------------------------------------------------------------
static uint32_t CMJ_Permute(uint32_t i, uint32_t l, uint32_t p) {            // :152-176
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8; i *= 0x0929eb3f;
        i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1 | p >> 27;
        i *= 0x6935fa69;
        i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2; i *= 0x9e501cc3;
        i ^= (i & w) >> 2; i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
static float CMJ_RandFloat(uint32_t i, uint32_t p) {                          // :178-188
    i ^= p;
    i ^= i >> 17;
    i ^= i >> 10; i *= 0xb36534e5;
    i ^= i >> 12;
    i ^= i >> 21; i *= 0x93fc4795;
    i ^= 0xdf6e307f;
    i ^= i >> 17; i *= 1 | p >> 18;
    return (float)i * (1.0f / 4294967808.0f);
}
static float2 CorrelatedMultiJitteredSampling(uint32_t sampleIdx, uint32_t nx, uint32_t ny, uint32_t pattern) { // :192-201
    uint32_t N = nx * ny;
    sampleIdx = CMJ_Permute(sampleIdx, N, pattern * 0x51633e2d);
    uint32_t sx = CMJ_Permute(sampleIdx % nx, nx, pattern * 0x68bc21eb);
    uint32_t sy = CMJ_Permute(sampleIdx / nx, ny, pattern * 0x02e5be93);
    float jx = CMJ_RandFloat(sampleIdx, pattern * 0x967a889b);
    float jy = CMJ_RandFloat(sampleIdx, pattern * 0x368cc8b7);
    return {qdiv((float)sx + qdiv((float)sy + jx, (float)ny), (float)nx), qdiv((float)sampleIdx + jy, (float)N)};
}
// shaders/raytracing/common.hlsli:8-14
static float2 SamplePoint(uint32_t pixelIdx, uint32_t sampleIdx, uint32_t& setIdx, uint32_t sqrtSampleCount, uint32_t numPixels) {
    const uint32_t permutationIdx = setIdx * numPixels + pixelIdx;
    setIdx += 1;
    return CorrelatedMultiJitteredSampling(sampleIdx, sqrtSampleCount, sqrtSampleCount, permutationIdx);
}
// shaders/common/uniform-sampling.hlsli:125-139 (Halton, used for TAA jitter by renderer.cpp:127-132)
static float Halton(uint32_t sampleIndex, uint32_t base) {
    float result = 0.f, f = 1.f;
    while (sampleIndex > 0) {
        f = f / (float)base;
        result += f * (float)(sampleIndex % base);
        sampleIndex = sampleIndex / base;
    }
    return result;
}

static const float k_Pi = 3.14159265358979323846f;
static const float k_InvPi = 0.31830988618379067154f;
static const float k_PiOver2 = 1.57079632679489661923f;
static const float k_PiOver4 = 0.78539816339744830961f;

// shaders/common/uniform-sampling.hlsli:20-45
static float2 ConcentricSampleDisk(float2 u) {
    float2 uOffset = {2.f * u.x - 1.f, 2.f * u.y - 1.f};
    if (uOffset.x == 0.f && uOffset.y == 0.f) return {0.f, 0.f};
    float theta, r;
    if (fabsf(uOffset.x) > fabsf(uOffset.y)) { r = uOffset.x; theta = k_PiOver4 * qdiv(uOffset.y, uOffset.x); }
    else { r = uOffset.y; theta = k_PiOver2 - k_PiOver4 * qdiv(uOffset.x, uOffset.y); }
    return {r * cosf(theta), r * sinf(theta)};
}
// :61-66
static float3 CosineSampleHemisphere(float2 u) {
    float2 d = ConcentricSampleDisk(u);
    float z = sqrtf(hmax(0.f, 1.f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
static float CosineHemispherePdf(float cosTheta) { return cosTheta * k_InvPi; }  // :68-71

// shaders/common/bxdf-sampling.hlsli:14-48
static float GGX(float NoH, float Roughness) {
    float a = Roughness * Roughness;
    float a2 = a * a;
    NoH = hmax(NoH, 0.f);
    float NoH2 = NoH * NoH;
    float nom = a2;
    float denom = (NoH2 * (a2 - 1.f) + 1.f);
    denom = k_Pi * denom * denom;
    return qdiv(nom, hmax(denom, 0.0001f));
}
static float3 SampleGGX(float2 u, float roughness) {
    float a = roughness * roughness;
    float Phi = 2 * k_Pi * u.x;
    float CosTheta = sqrtf(qdiv(1.f - u.y, 1.f + (a * a - 1.f) * u.y));
    float SinTheta = sqrtf(1.f - CosTheta * CosTheta);
    // Polar2Cartesian(sint, cost, phi, Tangent)  shaders/common/math.hlsli:56-64
    return {SinTheta * sinf(Phi), SinTheta * cosf(Phi), CosTheta};
}
static float GGXPdf(float NoH, float roughness) { return GGX(NoH, roughness) * NoH; }

// shaders/lighting/pbr.hlsli
static float G_SmithGGXCorrelated(float NoV, float NoL, float roughness) {     // :8-14
    float a2 = roughness * roughness;
    float GGXV = NoL * sqrtf(NoV * NoV * (1.0f - a2) + a2);
    float GGXL = NoV * sqrtf(NoL * NoL * (1.0f - a2) + a2);
    return qdiv(0.5f, hmax(GGXV + GGXL, 0.00001f));
}
static float3 F_Schlick(float u, float3 f0) {                                  // :22-26
    float f = pow5(1.0f - u);
    return f + f0 * (1.0f - f);
}
static float G_SchlickGGX(float u, float k) { return qdiv(u, u * (1.f - k) + k); } // :37-40
static float G_Smith_Direct(float NoV, float NoL, float roughness) {           // :43-50
    float a2 = roughness * roughness;
    float k = (a2 + 1) * (a2 + 1) / 8.f;
    return G_SchlickGGX(NoV, k) * G_SchlickGGX(NoL, k);
}
static float Fd_Lambert() { return k_InvPi; }
static float Exposure(int ev100) { return 1.0f / (ldexpf(1.f, ev100) * 1.2f); } // :78-81  pow(2, ev100) exact
static float3 ACESFilm(float3 x) {                                              // :84-92
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return saturate((x * (a * x + b)) / (x * (c * x + d) + e));
}

// ------------------------------------------------------------------------------------------------
// Preetham sky -- shaders/environment-sky/preetham.hlsli:6-89, Perez constants render-jobs/path-tracing.inl:134-140
// Pinned: cos(acos(saturate(c))) is evaluated as saturate(c) (algebraic identity; removes a 0 * inf
// hazard at the horizon that depends on the ulp behaviour of the hardware cos).
// ------------------------------------------------------------------------------------------------
static void perez_from_turbidity(float t, incr_perez_distribution* p) {
    // path-tracing.inl:134-140: double arithmetic (double literals * float t), rounded to float by the Vector4 ctor
    const double A[3] = {0.1787 * t - 1.4630, -0.0193 * t - 0.2592, -0.0167 * t - 0.2608};
    const double B[3] = {-0.3554 * t + 0.4275, -0.0665 * t + 0.0008, -0.0950 * t + 0.0092};
    const double C[3] = {-0.0227 * t + 5.3251, -0.0004 * t + 0.2125, -0.0079 * t + 0.2102};
    const double D[3] = {0.1206 * t - 2.5771, -0.0641 * t - 0.8989, -0.0441 * t - 1.6537};
    const double E[3] = {-0.0670 * t + 0.3703, -0.0033 * t + 0.0452, -0.0109 * t + 0.0529};
    for (int i = 0; i < 3; ++i) { p->A[i] = (float)A[i]; p->B[i] = (float)B[i]; p->C[i] = (float)C[i]; p->D[i] = (float)D[i]; p->E[i] = (float)E[i]; }
    p->A[3] = p->B[3] = p->C[3] = p->D[3] = p->E[3] = 0.f;
}
static float3 Perez(float cosTheta, float gamma, float cosGamma, const incr_perez_distribution& pc) { // :36-51
    float3 A = {pc.A[0], pc.A[1], pc.A[2]}, B = {pc.B[0], pc.B[1], pc.B[2]}, C = {pc.C[0], pc.C[1], pc.C[2]},
           D = {pc.D[0], pc.D[1], pc.D[2]}, E = {pc.E[0], pc.E[1], pc.E[2]};
    float3 e1 = {expf(B.x / cosTheta), expf(B.y / cosTheta), expf(B.z / cosTheta)};
    float3 e2 = {expf(D.x * gamma), expf(D.y * gamma), expf(D.z * gamma)};
    return (1.0f + A * e1) * ((1.0f + C * e2) + (E * cosGamma) * cosGamma);
}
static float3 CalculateZenithLuminanceAndChromaticity(float t, float thetaS) {   // :53-74
    float chi = (4.f / 9.f - t / 120.f) * (k_Pi - 2.f * thetaS);
    float Y_zenith = (4.0453f * t - 4.9710f) * tanf(chi) - 0.2155f * t + 2.4192f;
    float theta2 = thetaS * thetaS, theta3 = theta2 * thetaS, T = t, T2 = t * t;
    float x_zenith = (0.00165f * theta3 - 0.00375f * theta2 + 0.00209f * thetaS + 0.0f) * T2 +
                     (-0.02903f * theta3 + 0.06377f * theta2 - 0.03202f * thetaS + 0.00394f) * T +
                     (0.11693f * theta3 - 0.21196f * theta2 + 0.06052f * thetaS + 0.25886f);
    float y_zenith = (0.00275f * theta3 - 0.00610f * theta2 + 0.00317f * thetaS + 0.0f) * T2 +
                     (-0.04214f * theta3 + 0.08970f * theta2 - 0.04153f * thetaS + 0.00516f) * T +
                     (0.15346f * theta3 - 0.26756f * theta2 + 0.06670f * thetaS + 0.26688f);
    return {Y_zenith, x_zenith, y_zenith};
}
static float3 Yxy2RGB(float3 Yxy) {                                               // :6-34
    float Y = Yxy.x, x = Yxy.y, y = Yxy.z;
    float X = x * (Y / y);
    float Z = (1.0f - x - y) * (Y / y);
    float3 RGB;
    RGB.x = 3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z;
    RGB.y = -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z;
    RGB.z = 0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z;
    return RGB;
}
// Per-frame sun terms (depend only on s and turbidity): Yxy_zenith and F(0, thetaS).
struct SkySunTerms { float3 s; float3 YxyZenith; float3 F0; };
static SkySunTerms sky_sun_terms(float3 s, float t, const incr_perez_distribution& pc) {
    float cosS = saturate(s.y);
    float thetaS = acosf(cosS);
    float3 Yz = CalculateZenithLuminanceAndChromaticity(t, thetaS);
    float3 F0 = Perez(1.f, thetaS, cosS, pc);                                      // Perez(0.f, thetaS)
    return {s, Yz, F0};
}
static float3 CalculateSkyRadianceRGB(const SkySunTerms& sun, float3 e, const incr_perez_distribution& pc) { // :76-89
    float cosE = saturate(e.y);
    float cosG = saturate(dot(sun.s, e));
    float gammaE = acosf(cosG);
    float3 F = Perez(cosE, gammaE, cosG, pc);
    float3 Yxy = sun.YxyZenith * (F / sun.F0);
    return Yxy2RGB(Yxy);
}

// ------------------------------------------------------------------------------------------------
// Scene access -- shaders/common/mesh-material.hlsli:5-191
// ------------------------------------------------------------------------------------------------
struct Scene {
    incr_scene_desc d;
    std::vector<uint32_t> primOffset;      // prefix sum of primitive_counts (GetPrimitive :178-191)
    float srgb_lut[256];
    // flattened world-space geometry + BVH
    struct Tri { float3 v0, e1, e2; uint32_t inst, geom, prim, flags; };   // flags bit0 = opaque, bit1 = winding flipped
    std::vector<Tri> tris;
    struct Node { float lo[3], hi[3]; uint32_t left, right, first, count; };
    std::vector<Node> nodes;
    std::vector<Mat4> instLocalToWorld;    // per instance (visible mesh)
    std::vector<uint32_t> instMesh;
    std::vector<uint32_t> leafIdx;
    // mip chains of the textures (levels packed, level 0 first), built at scene creation: see build_texture_mips()
    std::vector<std::vector<uint32_t>> texChain;
    std::vector<uint32_t> texMips;
    std::vector<uint32_t> texFormat;       // what the sampler decodes: the desc's format, or what a block-compressed chain was expanded to
    uint32_t maxAnisotropy = 16;           // incr_scene_desc::max_anisotropy (0 = 16)
};

static inline bool acc_valid(const Scene& s, int accessorIndex) { return accessorIndex >= 0 && (uint32_t)accessorIndex < s.d.num_accessors; }
static const uint8_t* acc_base(const Scene& s, int accessorIndex, uint32_t* stride) {
    const incr_mesh_accessor& a = s.d.accessors[accessorIndex];
    const incr_mesh_buffer_view& v = s.d.buffer_views[a.bufferViewIndex];
    *stride = a.byteStride;
    return (const uint8_t*)s.d.buffers[v.bufferSrvIndex] + a.byteOffset + v.byteOffset;
}
static void GetUint3(const Scene& s, int index, int accessorIndex, uint32_t out[3]) {      // :45-90
    out[0] = out[1] = out[2] = 0;
    if (index == -1 || !acc_valid(s, accessorIndex)) return;
    uint32_t stride; const uint8_t* b = acc_base(s, accessorIndex, &stride);
    if (stride == 4) { memcpy(out, b + (size_t)index * 4, 12); }
    else if (stride == 2) { uint16_t t[3]; memcpy(t, b + (size_t)index * 2, 6); out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; }
}
static float4 GetFloat4(const Scene& s, int index, int accessorIndex) {                     // :92-113
    float4 r = {0, 0, 0, 0};
    if (index == -1 || !acc_valid(s, accessorIndex)) return r;
    uint32_t stride; const uint8_t* b = acc_base(s, accessorIndex, &stride);
    if (stride == 16) memcpy(&r, b + (size_t)index * 16, 16);
    return r;
}
static float3 GetFloat3(const Scene& s, int index, int accessorIndex) {                     // :115-140
    float3 r = {0, 0, 0};
    if (index == -1 || !acc_valid(s, accessorIndex)) return r;
    uint32_t stride; const uint8_t* b = acc_base(s, accessorIndex, &stride);
    if (stride == 12 || stride == 16) memcpy(&r, b + (size_t)index * stride, 12);
    return r;
}
static float2 GetFloat2(const Scene& s, int index, int accessorIndex) {                     // :142-170
    float2 r = {0, 0};
    if (index == -1 || !acc_valid(s, accessorIndex)) return r;
    uint32_t stride; const uint8_t* b = acc_base(s, accessorIndex, &stride);
    if (stride == 8 || stride == 12 || stride == 16) memcpy(&r, b + (size_t)index * stride, 8);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Texture unit stand-in.  D3D bilinear, wrap addressing, mip 0 (SampleLevel(...,0) under PATH_TRACING,
// shaders/material/common.hlsli:4-8).  Pinned: texel centre at +0.5, float weights,
// out = (t00*(1-fx) + t10*fx)*(1-fy) + (t01*(1-fx) + t11*fx)*fy.
// ------------------------------------------------------------------------------------------------
static float4 decode_texel_word(const Scene& s, uint32_t format, uint32_t c) {
    const uint32_t p[4] = {c & 0xffu, (c >> 8) & 0xffu, (c >> 16) & 0xffu, c >> 24};
    const float inv255 = 1.f / 255.f;
    switch (format) {
    case INCR_TEX_RGBA8_SRGB:       return {s.srgb_lut[p[0]], s.srgb_lut[p[1]], s.srgb_lut[p[2]], (float)p[3] * inv255};
    case INCR_TEX_RG_SNORM:         return {2.f * ((float)p[0] * inv255) - 1.f, 2.f * ((float)p[1] * inv255) - 1.f, 0.f, 1.f};
    case INCR_TEX_BGRA8_SWIZZLE_RG: return {(float)p[2] * inv255, (float)p[1] * inv255, 0.f, 1.f};
    // expanded BC4 / BC5 texels (expand_bc_chain below): two 16-bit numerators over 35 x 255, or signed over 35 x 127
    //   (value = numerator x the float nearest 1 / 8925 resp. 1 / 4445: the end points come out as exactly 0, 1 and -1)
    case INCR_TEX_RG16_OVER_8925:   { const float r = 1.f / 8925.f; return {(float)(c & 0xffffu) * r, (float)(c >> 16) * r, 0.f, 1.f}; }
    case INCR_TEX_RG16S_OVER_4445:  { const float r = 1.f / 4445.f; return {(float)(int16_t)(c & 0xffffu) * r, (float)(int16_t)(c >> 16) * r, 0.f, 1.f}; }
    default:                        return {(float)p[0] * inv255, (float)p[1] * inv255, (float)p[2] * inv255, (float)p[3] * inv255};
    }
}
// bilinear tap on level `level` of the texture's chain (level l is max(W >> l, 1) x max(H >> l, 1))
static float4 SampleBilinearLevel(const Scene& s, int texIndex, uint32_t level, float2 uv) {
    const incr_texture_desc& t = s.d.textures[texIndex];
    int W = (int)t.width, H = (int)t.height;
    size_t off = 0;
    for (uint32_t l = 0; l < level; ++l) { off += (size_t)W * H; W = W > 1 ? W >> 1 : 1; H = H > 1 ? H >> 1 : 1; }
    const uint32_t* px = s.texChain[(size_t)texIndex].data() + off;
    float fu = uv.x - floorf(uv.x), fv = uv.y - floorf(uv.y);
    if (!(fu >= 0.f && fu <= 1.f)) fu = 0.f;      // NaN / inf coordinates sample texel (0,0)
    if (!(fv >= 0.f && fv <= 1.f)) fv = 0.f;
    int xl, yl; float fx, fy;
    linear_address(fu * (float)W, &xl, &fx); linear_address(fv * (float)H, &yl, &fy);
    int x0 = (xl + W) % W, y0 = (yl + H) % H;
    int x1 = (x0 + 1) % W, y1 = (y0 + 1) % H;
    const uint32_t fmt = s.texFormat[(size_t)texIndex];
    float4 t00 = decode_texel_word(s, fmt, px[(size_t)y0 * W + x0]), t10 = decode_texel_word(s, fmt, px[(size_t)y0 * W + x1]);
    float4 t01 = decode_texel_word(s, fmt, px[(size_t)y1 * W + x0]), t11 = decode_texel_word(s, fmt, px[(size_t)y1 * W + x1]);
    float wx0 = 1.f - fx, wy0 = 1.f - fy;
    float4 r;
    r.x = (t00.x * wx0 + t10.x * fx) * wy0 + (t01.x * wx0 + t11.x * fx) * fy;
    r.y = (t00.y * wx0 + t10.y * fx) * wy0 + (t01.y * wx0 + t11.y * fx) * fy;
    r.z = (t00.z * wx0 + t10.z * fx) * wy0 + (t01.z * wx0 + t11.z * fx) * fy;
    r.w = (t00.w * wx0 + t10.w * fx) * wy0 + (t01.w * wx0 + t11.w * fx) * fy;
    return r;
}
static float4 SampleLevel0(const Scene& s, int texIndex, float2 uv) { return SampleBilinearLevel(s, texIndex, 0u, uv); }

// ---- Sample(s, uv) of the raster passes (material/common.hlsli:6-8: TEX_SAMPLE = Sample outside PATH_TRACING) ----
// g_anisoSampler: FILTER_ANISOTROPIC, MaxAnisotropy = the root-signature default 16 (gbuffer-compute.hlsl:20, visibility-pass.hlsl:11,
// gbuffer-raster.hlsl:11; ext/d3d12 d3dx12_root_signature.h:277).  D3D leaves the anisotropic footprint to the GPU; pinned here
//  as the textbook line of taps (Vulkan 1.3 spec, "Texel Anisotropic Filtering"): with rho_max, rho_min the lengths
// of the longer / shorter screen-space UV derivative in texels,
//     N = min(ceil(rho_max / rho_min), MaxAnisotropy),  lod = log2(rho_max / min(rho_max / rho_min, MaxAnisotropy)),
//     colour = 1/N * sum_{i=1..N} trilinear(uv + longer derivative * (i / (N + 1) - 1/2), lod).
// Evaluated on SQUARED lengths so that no square root has to agree between machines; the logarithm is exponent + a degree-4
// polynomial of the mantissa; lod clamped to the chain, fraction truncated to 8 bits, trilinear = lerp of two bilinear taps.
// MaxAnisotropy = 1 degenerates to the isotropic rule of the D3D11.3 functional spec 7.18.11.
static float pinned_log2(float x) {
    uint32_t b; memcpy(&b, &x, 4);
    const float e = (float)((int)(b >> 23) - 127);
    const uint32_t mb = (b & 0x7fffffu) | 0x3f800000u; float m; memcpy(&m, &mb, 4);
    const float t = m - 1.f;
    const float p = 1.4425331354141235f + t * (-0.7161056399345398f + t * (0.4395408034324646f + t * (-0.2250901162624359f + t * 0.05923847481608391f)));
    return e + t * p;
}
struct AnisoLine { float lod; uint32_t taps; float2 along; };
static AnisoLine aniso_line(const incr_texture_desc& t, float2 ddx, float2 ddy, uint32_t maxAnisotropy) {
    const float W = (float)t.width, H = (float)t.height;
    const float ax = ddx.x * W, ay = ddx.y * H, bx = ddy.x * W, by = ddy.y * H;
    const float lenX2 = ax * ax + ay * ay, lenY2 = bx * bx + by * by;
    AnisoLine r;
    float rhoMax2, rhoMin2;
    if (lenX2 >= lenY2) { rhoMax2 = lenX2; rhoMin2 = lenY2; r.along = ddx; } else { rhoMax2 = lenY2; rhoMin2 = lenX2; r.along = ddy; }
    r.lod = 0.f; r.taps = 1;
    if (!(rhoMax2 >= 1.17549435e-38f) || !(rhoMax2 <= 3.0e38f)) return r;           // no usable footprint: level 0, one tap
    for (uint32_t n = 1; n < maxAnisotropy; ++n) {                                    // ceil(rho_max / rho_min) without the root
        if (!((float)(n * n) * rhoMin2 < rhoMax2)) break;
        r.taps = n + 1;
    }
    const float clampedMin2 = fmaxf(rhoMin2, rhoMax2 / (float)(maxAnisotropy * maxAnisotropy));
    if (clampedMin2 >= 1.17549435e-38f) r.lod = 0.5f * pinned_log2(clampedMin2);
    return r;
}
static float4 sample_trilinear(const Scene& s, int texIndex, float lod, float2 uv) {
    const uint32_t numMips = s.texMips[(size_t)texIndex];
    lod = fminf(fmaxf(lod, 0.f), (float)(numMips - 1u));
    const float l0f = floorf(lod);
    const float frac = floorf((lod - l0f) * 256.f) * (1.f / 256.f);
    const uint32_t l0 = (uint32_t)l0f;
    const float4 a = SampleBilinearLevel(s, texIndex, l0, uv);
    if (frac == 0.f || l0 + 1u >= numMips) return a;
    const float4 b = SampleBilinearLevel(s, texIndex, l0 + 1u, uv);
    return {a.x + (b.x - a.x) * frac, a.y + (b.y - a.y) * frac, a.z + (b.z - a.z) * frac, a.w + (b.w - a.w) * frac};
}
static float4 SampleGrad(const Scene& s, int texIndex, float2 uv, float2 ddx, float2 ddy) {
    const AnisoLine line = aniso_line(s.d.textures[texIndex], ddx, ddy, s.maxAnisotropy);
    if (line.taps == 1) return sample_trilinear(s, texIndex, line.lod, uv);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = 1; i <= line.taps; ++i) {
        const float where = (float)i / (float)(line.taps + 1u) - 0.5f;
        const float4 c = sample_trilinear(s, texIndex, line.lod, {uv.x + line.along.x * where, uv.y + line.along.y * where});
        acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    }
    const float n = (float)line.taps;
    return {acc.x / n, acc.y / n, acc.z / n, acc.w / n};
}

// shaders/material/common.hlsli:10-115 (fields the hot path reads)
struct MatProps {
    float3 emissive, basecolor; float opacity; bool bHasNormalmap; float3 normalmap;
    float metallic, roughness, ao, aoblend;
};
template <class Sampler> static MatProps EvaluateMaterialPropertiesT(const incr_material& mat, float2 uv, Sampler sample) {
    MatProps o;
    o.emissive = {mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]};
    if (mat.emissiveTextureIndex != -1) { float4 t = sample(mat.emissiveTextureIndex, uv); o.emissive *= f3(t.x, t.y, t.z); }
    o.basecolor = {mat.baseColorFactor[0], mat.baseColorFactor[1], mat.baseColorFactor[2]};
    o.opacity = 1.f;   // Q12: left uninitialised by the reference when there is no base-colour texture; pinned to 1
    if (mat.baseColorTextureIndex != -1) {
        float4 t = sample(mat.baseColorTextureIndex, uv);
        o.basecolor *= f3(t.x, t.y, t.z);
        o.opacity = t.w;
    }
    o.bHasNormalmap = mat.normalTextureIndex != -1;
    o.normalmap = {0, 0, 1};
    if (o.bHasNormalmap) {
        float4 t = sample(mat.normalTextureIndex, uv);
        float2 nxy = {t.x, t.y};
        float nz = sqrtf(1.f - dot(nxy, nxy));
        o.normalmap = {nxy.x, nxy.y, nz};
    }
    o.metallic = mat.metallicFactor;
    o.roughness = mat.roughnessFactor;
    if (mat.metallicRoughnessTextureIndex != -1) {
        float4 t = sample(mat.metallicRoughnessTextureIndex, uv);
        o.metallic = t.x; o.roughness = t.y;
    }
    o.aoblend = mat.aoStrength;
    o.ao = 1.f;
    if (mat.aoTextureIndex != -1) o.ao = sample(mat.aoTextureIndex, uv).x;
    return o;
}

static MatProps EvaluateMaterialProperties(const Scene& s, const incr_material& mat, float2 uv) {                     // PATH_TRACING: SampleLevel 0
    return EvaluateMaterialPropertiesT(mat, uv, [&](int tex, float2 st) { return SampleLevel0(s, tex, st); });
}
static MatProps EvaluateMaterialPropertiesGrad(const Scene& s, const incr_material& mat, float2 uv, float2 ddx, float2 ddy) {   // raster: Sample
    return EvaluateMaterialPropertiesT(mat, uv, [&](int tex, float2 st) { return SampleGrad(s, tex, st, ddx, ddy); });
}

// ------------------------------------------------------------------------------------------------
// Acceleration structure + traversal: stand-in for DXR.
//  * geometry flattened to world space: p_world = p * (meshTransform * rootTransform)   (update-tlas.inl:40)
//  * one instance per VISIBLE mesh, InstanceIndex = running index (update-tlas.inl:18-48)
//  * GeometryIndex = primitive index within the mesh, OPAQUE iff alphaMode == Opaque (scene.cpp:736)
//  * facing is an object-space property (DXR spec): front = clockwise in the left-handed object frame,
//    i.e. det = dot(e1, cross(d, e2)) > 0 with object-space vertex order; when the instance transform
//    mirrors (det3 < 0, e.g. the RH2LH root) v1 and v2 are swapped in the flattened triangle and the
//    barycentrics swapped back on hit.
//  * ray/triangle: Moeller-Trumbore on (v0, e1, e2), interval tmin det < T < tmax det (see tri_intersect), closest hit with
//    ties broken towards the smaller flattened triangle id (traversal-order independent).
// ------------------------------------------------------------------------------------------------
struct Hit { float t, u, v; uint32_t tri; };

// The hit predicate is pinned on the UNSCALED Moeller-Trumbore numerators (no division decides a hit: round 4; rounds 1-3 compared the
// quotients): front-facing, 0 <= U <= det, 0 <= V, U + V <= det, tmin det < T < tmax det.  t, u, v of an accepted hit are the
// numerators times the correctly rounded reciprocal of det.  DXR leaves the watertightness of the test to the implementation; this
// is one such choice, shared by the closest-hit and the any-hit queries.
static inline bool tri_intersect(const Scene::Tri& T, float3 o, float3 d, float tmin, float tmax, float* t, float* u, float* v) {
    float3 p = cross(d, T.e2);
    float det = dot(T.e1, p);
    if (!(det > 0.f)) return false;                 // RAY_FLAG_CULL_BACK_FACING_TRIANGLES (and degenerate)
    float3 sv = o - T.v0;
    float un = dot(sv, p);
    if (!(un >= 0.f && un <= det)) return false;
    float3 q = cross(sv, T.e1);
    float vn = dot(d, q);
    if (!(vn >= 0.f && un + vn <= det)) return false;
    float tn = dot(T.e2, q);
    if (!(tn > tmin * det && tn < tmax * det)) return false;
    float inv = 1.f / det;
    *t = tn * inv; *u = un * inv; *v = vn * inv;
    return true;
}
static inline bool box_intersect(const Scene::Node& n, float3 o, float3 invd, float tmin, float tmax) {
    float t0 = tmin, t1 = tmax;
    const float oo[3] = {o.x, o.y, o.z}, ii[3] = {invd.x, invd.y, invd.z};
    for (int a = 0; a < 3; ++a) {
        float ta = (n.lo[a] - oo[a]) * ii[a], tb = (n.hi[a] - oo[a]) * ii[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
        // widen by 2 ulp-ish so that the (exact) triangle test decides, not the box test
        t0 = fmaxf(t0, tn * 0.9999990f - 1e-30f); t1 = fminf(t1, tf * 1.0000010f + 1e-30f);
    }
    return t0 <= t1;
}

static void build_bvh(Scene& s) {
    const size_t n = s.tris.size();
    std::vector<float3> cen(n), lo(n), hi(n);
    for (size_t i = 0; i < n; ++i) {
        const Scene::Tri& T = s.tris[i];
        float3 a = T.v0, b = T.v0 + T.e1, c = T.v0 + T.e2;
        lo[i] = {fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z))};
        hi[i] = {fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z))};
        cen[i] = (lo[i] + hi[i]) * 0.5f;
    }
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)i;
    s.nodes.clear();
    s.nodes.reserve(n ? 2 * n : 1);
    struct Work { uint32_t node, first, count; };
    std::vector<Work> stack;
    s.nodes.push_back({});
    stack.push_back({0, 0, (uint32_t)n});
    while (!stack.empty()) {
        Work w = stack.back(); stack.pop_back();
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = w.first; i < w.first + w.count; ++i) {
            uint32_t t = idx[i];
            const float l[3] = {lo[t].x, lo[t].y, lo[t].z}, h[3] = {hi[t].x, hi[t].y, hi[t].z}, c[3] = {cen[t].x, cen[t].y, cen[t].z};
            for (int a = 0; a < 3; ++a) { blo[a] = fminf(blo[a], l[a]); bhi[a] = fmaxf(bhi[a], h[a]); clo[a] = fminf(clo[a], c[a]); chi[a] = fmaxf(chi[a], c[a]); }
        }
        Scene::Node nd{};
        for (int a = 0; a < 3; ++a) { nd.lo[a] = blo[a]; nd.hi[a] = bhi[a]; }
        int axis = 0; float ext = chi[0] - clo[0];
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > ext) { ext = chi[a] - clo[a]; axis = a; }
        if (w.count <= 4 || !(ext > 0.f)) { nd.first = w.first; nd.count = w.count; s.nodes[w.node] = nd; continue; }
        uint32_t mid = w.first + w.count / 2;
        std::nth_element(idx.begin() + w.first, idx.begin() + mid, idx.begin() + w.first + w.count, [&](uint32_t a, uint32_t b) {
            const float ca = axis == 0 ? cen[a].x : axis == 1 ? cen[a].y : cen[a].z;
            const float cb = axis == 0 ? cen[b].x : axis == 1 ? cen[b].y : cen[b].z;
            return ca < cb || (ca == cb && a < b);
        });
        nd.count = 0;
        nd.left = (uint32_t)s.nodes.size(); s.nodes.push_back({});
        nd.right = (uint32_t)s.nodes.size(); s.nodes.push_back({});
        s.nodes[w.node] = nd;
        stack.push_back({nd.left, w.first, mid - w.first});
        stack.push_back({nd.right, mid, w.first + w.count - mid});
    }
    s.leafIdx = std::move(idx);   // leaves hold ranges of leafIdx[]; triangle ids stay the flatten-order ids
}

// scene flattening -------------------------------------------------------------------------------
static Mat4 to_mat4(const incr_matrix& m) { Mat4 r; memcpy(r.m, m.m, 64); return r; }

static void flatten_scene(Scene& s) {
    const incr_scene_desc& d = s.d;
    s.primOffset.assign(d.num_meshes + 1, 0);
    for (uint32_t i = 0; i < d.num_meshes; ++i) s.primOffset[i + 1] = s.primOffset[i] + d.primitive_counts[i];
    s.tris.clear(); s.instLocalToWorld.clear(); s.instMesh.clear();
    const Mat4 root = to_mat4(d.root_transform);
    for (uint32_t mesh = 0; mesh < d.num_meshes; ++mesh) {
        if (d.mesh_visibility && !d.mesh_visibility[mesh]) continue;
        const uint32_t inst = (uint32_t)s.instMesh.size();
        const Mat4 l2w = mat_mul(to_mat4(d.mesh_transforms[mesh]), root);      // update-tlas.inl:40
        const bool mirrored = mat_det3(l2w) < 0.f;
        s.instMesh.push_back(mesh);
        s.instLocalToWorld.push_back(l2w);
        for (uint32_t g = 0; g < d.primitive_counts[mesh]; ++g) {
            const incr_gpu_primitive& prim = d.primitives[s.primOffset[mesh] + g];
            const bool opaque = d.materials[prim.materialIndex].alphaMode == INCR_ALPHA_OPAQUE;   // scene.cpp:736
            const uint32_t ntri = (uint32_t)prim.indexCount / 3;
            for (uint32_t p = 0; p < ntri; ++p) {
                uint32_t idx[3];
                GetUint3(s, (int)(p * 3), prim.indexAccessor, idx);
                float3 v[3];
                for (int k = 0; k < 3; ++k) v[k] = mul_point(GetFloat3(s, (int)idx[k], prim.positionAccessor), l2w);
                Scene::Tri T;
                T.v0 = v[0];
                T.e1 = (mirrored ? v[2] : v[1]) - v[0];
                T.e2 = (mirrored ? v[1] : v[2]) - v[0];
                T.inst = inst; T.geom = g; T.prim = p;
                T.flags = (opaque ? 1u : 0u) | (mirrored ? 2u : 0u);
                s.tris.push_back(T);
            }
        }
    }
    build_bvh(s);
}

// traversal ---------------------------------------------------------------------------------------
struct RayStats { uint64_t camera = 0, bounce = 0, skyShadow = 0, lightShadow = 0, hits = 0; };

// ahsMain alpha test -- shaders/raytracing/pathtracing.hlsl:270-302
static bool any_hit_accepts(const Scene& s, const Scene::Tri& T, float u, float v) {
    const uint32_t mesh = s.instMesh[T.inst];
    const incr_gpu_primitive& prim = s.d.primitives[s.primOffset[mesh] + T.geom];
    uint32_t idx[3];
    GetUint3(s, (int)(T.prim * (uint32_t)prim.indicesPerTriangle), prim.indexAccessor, idx);
    float2 uv0 = GetFloat2(s, (int)idx[0], prim.uvAccessor), uv1 = GetFloat2(s, (int)idx[1], prim.uvAccessor), uv2 = GetFloat2(s, (int)idx[2], prim.uvAccessor);
    float2 uv = uv0 + u * (uv1 - uv0) + v * (uv2 - uv0);
    const incr_material& mat = s.d.materials[prim.materialIndex];
    if (mat.baseColorTextureIndex != -1) {
        float alpha = SampleLevel0(s, mat.baseColorTextureIndex, uv).w;
        if (alpha < 0.5f) return false;    // IgnoreHit()
    }
    return true;
}

// TraceRay(..., RAY_FLAG_CULL_BACK_FACING_TRIANGLES, ...) closest hit
// A ray whose origin or direction is not finite hits nothing.  (Every ray/triangle test fails on NaN/inf operands anyway;
// stating it up front spares the traversal a walk over the whole tree, whose box tests cannot reject NaN.)
static inline bool finite_ray(float3 o, float3 d) {
    return std::isfinite(o.x) && std::isfinite(o.y) && std::isfinite(o.z) && std::isfinite(d.x) && std::isfinite(d.y) && std::isfinite(d.z);
}
static bool trace_closest(const Scene& s, float3 o, float3 d, float tmin, float tmax, Hit* hit) {
    if (s.nodes.empty() || s.tris.empty() || !finite_ray(o, d)) return false;
    float3 invd = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    bool found = false; float best = tmax; uint32_t bestTri = 0xffffffffu; float bu = 0, bv = 0;
    while (sp) {
        const Scene::Node& n = s.nodes[stack[--sp]];
        if (!box_intersect(n, o, invd, tmin, best)) continue;
        if (n.count) {
            for (uint32_t i = 0; i < n.count; ++i) {
                const uint32_t ti = s.leafIdx[n.first + i];
                const Scene::Tri& T = s.tris[ti];
                float t, u, v;
                // interval upper bound is inclusive of the current best so that equal-t ties can be resolved by id
                if (!tri_intersect(T, o, d, tmin, tmax, &t, &u, &v)) continue;
                if (found && !(t < best || (t == best && ti < bestTri))) continue;
                if (T.flags & 2u) { float tmp = u; u = v; v = tmp; }   // undo the mirrored-instance vertex swap
                if (!(T.flags & 1u) && !any_hit_accepts(s, T, u, v)) continue;
                found = true; best = t; bestTri = ti; bu = u; bv = v;
            }
        } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
    if (found) { hit->t = best; hit->u = bu; hit->v = bv; hit->tri = bestTri; }
    return found;
}
// RayQuery<CULL_BACK_FACING> + single Proceed(): occluded iff ANY front-facing triangle lies in the interval
// (shaders/lighting/common.hlsli:102-115).
static bool trace_any(const Scene& s, float3 o, float3 d, float tmin, float tmax) {
    if (s.nodes.empty() || s.tris.empty() || !finite_ray(o, d)) return false;
    float3 invd = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const Scene::Node& n = s.nodes[stack[--sp]];
        if (!box_intersect(n, o, invd, tmin, tmax)) continue;
        if (n.count) {
            for (uint32_t i = 0; i < n.count; ++i) {
                float t, u, v;
                if (tri_intersect(s.tris[s.leafIdx[n.first + i]], o, d, tmin, tmax, &t, &u, &v)) return true;
            }
        } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// Lighting -- shaders/lighting/common.hlsli
// ------------------------------------------------------------------------------------------------
// visible (optional): set when the light's shadow query was issued and found nothing (the hit-chain instrumentation)
static float3 GetDirectRadiance(const Scene& s, const incr_light& light, const Mat4& lightTransform, float3 worldPos, float3 basecolor,
                                float metallic, float roughness, float3 N, float3 V, bool pathTracing, RayStats* rs, bool* visible = nullptr) {  // :8-143
    float3 L = {0, 0, 0};
    float3 radianceIn = {0, 0, 0};
    const float3 color = {light.color[0], light.color[1], light.color[2]};
    const float* m = lightTransform.m;
    if (light.type == INCR_LIGHT_DIRECTIONAL) {
        // mul(float3(0,0,-1), float3x3(rows)) = -row2
        L = normalize(f3((0.f * m[0] + 0.f * m[4]) + -1.f * m[8], (0.f * m[1] + 0.f * m[5]) + -1.f * m[9], (0.f * m[2] + 0.f * m[6]) + -1.f * m[10]));
        radianceIn = 100 * light.intensity * color;
    } else if (light.type == INCR_LIGHT_POINT) {
        float3 lightPosition = {m[12], m[13], m[14]};
        float3 lightVec = lightPosition - worldPos;
        float distanceSquared = dot(lightVec, lightVec);
        float distance = sqrtf(distanceSquared);
        L = lightVec / distance;
        float radialAttenuation;
        if (light.range > 0.f) radialAttenuation = qdiv(hmax(hmin(1.f, 1.f - pow4(qdiv(distance, light.range))), 0.f), distanceSquared);
        else radialAttenuation = qdiv(1.f, hmax(distanceSquared, 0.0001f));
        radianceIn = 10000 * light.intensity * color * radialAttenuation;
    } else if (light.type == INCR_LIGHT_SPOT) {
        float3 spotPosition = {m[12], m[13], m[14]};
        float3 spotDirection = normalize(f3((0.f * m[0] + 0.f * m[4]) + -1.f * m[8], (0.f * m[1] + 0.f * m[5]) + -1.f * m[9], (0.f * m[2] + 0.f * m[6]) + -1.f * m[10]));
        float3 lightVec = spotPosition - worldPos;
        float distanceSquared = dot(lightVec, lightVec);
        float distance = sqrtf(distanceSquared);
        L = lightVec / distance;
        float angularAttenuation = 0.f;
        float cd = dot(spotDirection, L);
        float cosineOuterAngle = cosf(light.spotAngles[1]);
        if (cd > cosineOuterAngle) {
            float cosineInnerAngle = cosf(light.spotAngles[0]);
            float lightAngleScale = qdiv(1.f, hmax(0.001f, cosineInnerAngle - cosineOuterAngle));
            float lightAngleOffset = -cosineOuterAngle * lightAngleScale;
            angularAttenuation = saturate(cd * lightAngleScale + lightAngleOffset);
            angularAttenuation *= angularAttenuation;
        }
        float radialAttenuation = 0.f;
        if (angularAttenuation > 0.f) {
            if (light.range > 0.f) { /* Q9: inner-scope redeclaration at :73 leaves the outer value 0 */ }
            else radialAttenuation = qdiv(1.f, hmax(distanceSquared, 0.0001f));
        }
        radianceIn = 10000 * light.intensity * color * angularAttenuation * radialAttenuation;
    }
    float3 radianceOut = {0, 0, 0};
    float NoL = saturate(dot(N, L));
    if (NoL > 0.f && (radianceIn.x > 0.f || radianceIn.y > 0.f || radianceIn.z > 0.f)) {
        float lightVisibility = 1.f;
        if (pathTracing || light.type == INCR_LIGHT_DIRECTIONAL) {
            if (rs) rs->lightShadow++;
            if (trace_any(s, worldPos, L, 0.1f, 1000.f)) lightVisibility = 0.f;
            else if (visible) *visible = true;
        }
        if (lightVisibility > 0.f) {
            float NoV = saturate(dot(N, V));
            float3 F0 = metallic * basecolor + (1.f - metallic) * 0.04f;
            float3 albedo = (1.f - metallic) * basecolor;
            float3 H = normalize(L + V);
            float NoH = saturate(dot(N, H));
            float VoH = saturate(dot(V, H));
            float D = GGX(NoH, roughness);
            float3 F = F_Schlick(VoH, F0);
            float G = G_SmithGGXCorrelated(NoV, NoL, roughness);
            float3 Fd = albedo * Fd_Lambert();
            float3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
            float3 irradiance = radianceIn * NoL;
            radianceOut = (Fr + (1.f - F) * Fd) * irradiance * lightVisibility;
        }
    }
    return radianceOut;
}

static float3 GetSkyRadiance(const Scene& s, float3 radianceIn, float3 L, float3 worldPos, const MatProps& matInfo, float3 N, float3 V, RayStats* rs, bool* visible = nullptr) { // :145-202
    float3 radianceOut = {0, 0, 0};
    float NoL = saturate(dot(N, L));
    if (NoL > 0.f && (radianceIn.x > 0.f || radianceIn.y > 0.f || radianceIn.z > 0.f)) {
        float lightVisibility = 1.f;
        if (rs) rs->skyShadow++;
        if (trace_any(s, worldPos, L, 0.1f, 1000.f)) lightVisibility = 0.f;
        else if (visible) *visible = true;
        if (lightVisibility > 0.f) {
            float NoV = dot(N, V);
            float3 F0 = matInfo.metallic * matInfo.basecolor + (1.f - matInfo.metallic) * 0.04f;
            float3 albedo = (1.f - matInfo.metallic) * matInfo.basecolor;
            float3 H = normalize(L + V);
            float NoH = hmax(0.0001f, dot(N, H));
            float VoH = hmax(0.0001f, dot(V, H));
            float D = GGX(NoH, matInfo.roughness);
            float3 F = F_Schlick(VoH, F0);
            float G = G_Smith_Direct(NoV, NoL, matInfo.roughness);
            float3 Fd = albedo * Fd_Lambert();
            float3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
            float3 irradiance = radianceIn * NoL;
            radianceOut = (Fr + (1.f - F) * Fd) * irradiance * lightVisibility;
        }
    }
    return radianceOut / CosineHemispherePdf(L.z);
}

// ------------------------------------------------------------------------------------------------
// Path tracer -- shaders/raytracing/pathtracing.hlsl + shaders/raytracing/common.hlsli
// ------------------------------------------------------------------------------------------------
typedef incr_pathtrace_desc PtDesc;   // oracle side: target/history/ray_stats are HOST pointers, `scene` is ignored

struct Frame {            // per-call constants (path-tracing.inl:134-217 GlobalCbLayout)
    Mat4 projectionToWorld, cameraMatrix, sceneRotation;
    uint32_t W, H, numPixels, sampleIdx, sqrtSampleCount;
    float aperture, focalLength, skyBrightness, turbidity;
    int directLighting, viewmode, lightCount, envSkyMode;
    incr_perez_distribution perez;
    SkySunTerms sunMiss, sunEnv;
    const uint32_t* envmap; uint32_t envmapRes, envmapMips;     // GlobalCbLayout::envmapTextureIndex (NULL: analytic stand-in)
};
static float3 env_cube_level0(const uint32_t* data, uint32_t N, uint32_t mips, float3 dir);      // oracle_ibl.h

struct Ray { float3 o, d; float tmin, tmax; };

static Ray GenerateCameraRay(const Frame& f, float2 index, float2 randomSample) {     // common.hlsli:25-53
    float2 xy = {index.x + 0.5f, index.y + 0.5f};
    float2 screenPos = {xy.x / (float)f.W * 2.0f - 1.0f, xy.y / (float)f.H * 2.0f - 1.0f};
    screenPos.y = -screenPos.y;
    const float* cm = f.cameraMatrix.m;
    float3 cameraPos = {cm[12], cm[13], cm[14]};
    float4 world = mul_vec4({screenPos.x, screenPos.y, 0.0001f, 1.f}, f.projectionToWorld);
    float3 wxyz = {world.x / world.w, world.y / world.w, world.z / world.w};
    float3 primaryRayDir = normalize(wxyz - cameraPos);
    float3 focalPoint = cameraPos + f.focalLength * primaryRayDir;
    float3 cameraRight = {cm[0], cm[1], cm[2]};
    float3 cameraUp = {cm[4], cm[5], cm[6]};
    float2 offset = ConcentricSampleDisk(randomSample);
    float3 rayOrigin = cameraPos + (f.aperture * offset.x) * cameraRight + (f.aperture * offset.y) * cameraUp;
    Ray r;
    r.o = rayOrigin;
    r.d = normalize(focalPoint - rayOrigin);
    r.tmin = 0.001f; r.tmax = 10000.0f;
    return r;
}

// chain (optional, 12 words): incr_pathtrace_desc::debug_hit_chain of this pixel
static float4 trace_pixel(const Scene& s, const Frame& f, uint32_t px, uint32_t py, RayStats* rs, uint32_t* chain = nullptr) {
    // rgsMain -- pathtracing.hlsl:89-119
    uint32_t raygenSampleSetIdx = 0;
    const uint32_t pixelIdx = py * f.W + px;
    float2 subPixelJitter = SamplePoint(pixelIdx, f.sampleIdx, raygenSampleSetIdx, f.sqrtSampleCount, f.numPixels);
    float2 apertureSample = SamplePoint(pixelIdx, f.sampleIdx, raygenSampleSetIdx, f.sqrtSampleCount, f.numPixels);
    Ray ray = GenerateCameraRay(f, {(float)px + subPixelJitter.x, (float)py + subPixelJitter.y}, apertureSample);

    float4 color = {0, 0, 0, 0};
    int pathLength = 0;
    float3 attenuation = {1.f, 1.f, 1.f};
    uint32_t sampleSetIndex = 0;                // Q3: the payload restarts at set 0
    const float k_rayOffset = 0.001f;
    bool first = true;

    for (;;) {
        if (first) rs->camera++; else rs->bounce++;
        first = false;
        Hit hit;
        if (!trace_closest(s, ray.o, ray.d, ray.tmin, ray.tmax, &hit)) {
            if (chain) { chain[3 * pathLength] = 0xffffffffu; chain[3 * pathLength + 1] = 0xffffffffu; chain[3 * pathLength + 2] = 0u; }
            float3 add;
            if (f.envSkyMode == 0) {                   // msEnvmap -- pathtracing.hlsl:304-309 (EnvSkyMode::HDRI)
                add = (attenuation * f.skyBrightness) * env_cube_level0(f.envmap, f.envmapRes, f.envmapMips, ray.d);
            } else {                                   // msDynamicSky -- pathtracing.hlsl:311-319
                float3 sky = CalculateSkyRadianceRGB(f.sunMiss, ray.d, f.perez);
                add = attenuation * 1000 * sky;
            }
            color.x += add.x; color.y += add.y; color.z += add.z;
            break;
        }
        // chsMain -- pathtracing.hlsl:121-268
        rs->hits++;
        pathLength += 1;
        const Scene::Tri& T = s.tris[hit.tri];
        if (chain) { chain[3 * (pathLength - 1)] = T.inst << 16 | (T.geom & 0xffffu); chain[3 * (pathLength - 1) + 1] = T.prim; }
        uint32_t visWord = 0u;
        const uint32_t mesh = s.instMesh[T.inst];
        const incr_gpu_primitive& prim = s.d.primitives[s.primOffset[mesh] + T.geom];       // GetPrimitive
        const Mat4 localToWorld = mat_mul(to_mat4(s.d.mesh_transforms[prim.meshIndex]), f.sceneRotation);
        float3 hitPosition = ray.o + hit.t * ray.d;
        uint32_t indices[3];
        GetUint3(s, (int)(T.prim * (uint32_t)prim.indicesPerTriangle), prim.indexAccessor, indices);
        float3 vn[3]; float4 vt[3]; float2 vuv[3];
        for (int k = 0; k < 3; ++k) {
            vn[k] = GetFloat3(s, (int)indices[k], prim.normalAccessor);
            vt[k] = GetFloat4(s, (int)indices[k], prim.tangentAccessor);
            vuv[k] = GetFloat2(s, (int)indices[k], prim.uvAccessor);
        }
        const float bx = hit.u, by = hit.v;
        float4 packedT = vt[0] + bx * (vt[1] - vt[0]) + by * (vt[2] - vt[0]);        // HitAttribute common.hlsli:126-145
        float3 N = normalize(vn[0] + bx * (vn[1] - vn[0]) + by * (vn[2] - vn[0]));
        float3 Tn = normalize(f3(packedT.x, packedT.y, packedT.z));
        float3 B = cross(N, Tn) * packedT.w;
        N = mul_dir(N, localToWorld);            // Q6: direct transform, no inverse-transpose, not re-normalised
        Tn = mul_dir(Tn, localToWorld);
        B = mul_dir(B, localToWorld);
        const float3 Ntbn = N;                   // float3x3 tangentToWorld = float3x3(T, B, N) is built HERE (:166), before normal mapping
        float2 uv = vuv[0] + bx * (vuv[1] - vuv[0]) + by * (vuv[2] - vuv[0]);
        const incr_material& material = s.d.materials[prim.materialIndex];
        MatProps matInfo = EvaluateMaterialProperties(s, material, uv);

        if (f.viewmode == 1) matInfo.basecolor = f3(0.5f);
        else if (f.viewmode == 2) { color = {matInfo.roughness, matInfo.roughness, matInfo.roughness, matInfo.roughness}; break; }
        else if (f.viewmode == 3) { color = {matInfo.metallic, matInfo.metallic, matInfo.metallic, matInfo.metallic}; break; }
        else if (f.viewmode == 4) { color = {matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, 1.f}; break; }
        else if (f.viewmode == 5) { color = {matInfo.emissive.x, matInfo.emissive.y, matInfo.emissive.z, 1.f}; break; }

        {   // Q5 emissive
            float3 add = attenuation * matInfo.emissive * 20000;
            color.x += add.x; color.y += add.y; color.z += add.z;
            attenuation *= saturate(1.f - matInfo.emissive);
        }
        if (matInfo.bHasNormalmap) {
            // mul(normalmap, float3x3(T, B, N)) = n.x*T + n.y*B + n.z*N
            float3 nm = matInfo.normalmap;
            N = normalize(f3((nm.x * Tn.x + nm.y * B.x) + nm.z * N.x, (nm.x * Tn.y + nm.y * B.y) + nm.z * N.y, (nm.x * Tn.z + nm.y * B.z) + nm.z * N.z));
        }
        float3 V = -ray.d;

        if (f.viewmode == 7 && pathLength < 2) {         // VIEWMODE 7, Reflections -- pathtracing.hlsl:212-224: the first hit sends ONE mirror ray with
            // the same payload (colour, attenuation, path length; no SamplePoint is consumed: the sample set index stays) and returns --
            // no lights, no sky sample at this vertex, no test of the attenuation
            if (chain) chain[3 * (pathLength - 1) + 2] = 0u;
            ray.o = hitPosition; ray.d = normalize(reflect(-V, N)); ray.tmin = k_rayOffset; ray.tmax = 10000.0f;
            continue;
        }

        if (f.directLighting) {
            for (int lightIndex = 0; lightIndex < f.lightCount; ++lightIndex) {
                // D3D robust buffer access: out-of-range loads return 0
                int lightId = (uint32_t)lightIndex < s.d.num_scene_lights ? s.d.light_indices[lightIndex] : 0;
                incr_light light{};
                if ((uint32_t)lightId < s.d.num_global_lights) light = s.d.global_lights[lightId];
                Mat4 lightTransform{};
                if ((uint32_t)lightId < s.d.num_scene_lights) lightTransform = to_mat4(s.d.light_transforms[lightId]);
                bool vis = false;
                float3 r = attenuation * GetDirectRadiance(s, light, lightTransform, hitPosition, matInfo.basecolor, matInfo.metallic, matInfo.roughness, N, V, true, rs, &vis);
                if (vis) visWord += ((uint32_t)lightIndex + 1u) * 0x9E3779B1u;
                color.x += r.x; color.y += r.y; color.z += r.z;
            }
        }
        {   // Skylight contribution :239-244 (Q1: tangent-space direction used as a world direction)
            float2 u = SamplePoint(pixelIdx, f.sampleIdx, sampleSetIndex, f.sqrtSampleCount, f.numPixels);
            float3 sampleDir = CosineSampleHemisphere(u);
            // envmap.SampleLevel(g_envmapSampler, sampleDir, 0) on the prefiltered sky cube when the caller supplies it
            // (oracle_ibl.h); else Q15: level 0 of that cube ~= 0.04 * Preetham(scene sun dir), evaluated analytically
            float3 radianceIn = f.skyBrightness * (f.envmap ? env_cube_level0(f.envmap, f.envmapRes, f.envmapMips, sampleDir)
                                                            : 0.04f * CalculateSkyRadianceRGB(f.sunEnv, sampleDir, f.perez));
            bool skyVis = false;
            float3 r = attenuation * GetSkyRadiance(s, radianceIn, sampleDir, hitPosition, matInfo, N, V, rs, &skyVis);
            color.x += r.x; color.y += r.y; color.z += r.z;
            if (chain) chain[3 * (pathLength - 1) + 2] = (visWord & 0x7fffffffu) | (skyVis ? 0x80000000u : 0u);      // which shadow queries of this vertex came back unoccluded
        }
        if (pathLength < 4) {   // MAX_RECURSION_DEPTH
            // GenerateIndirectRadianceRay -- common.hlsli:55-123 (set index taken BY VALUE: Q3)
            uint32_t setCopy = sampleSetIndex;
            float2 u = SamplePoint(pixelIdx, f.sampleIdx, setCopy, f.sqrtSampleCount, f.numPixels);
            float3 L, H;
            const float t = 0.5f;
            auto mulTBNg = [&](float3 v) {   // mul(v, tangentToWorld)
                return f3((v.x * Tn.x + v.y * B.x) + v.z * Ntbn.x, (v.x * Tn.y + v.y * B.y) + v.z * Ntbn.y, (v.x * Tn.z + v.y * B.z) + v.z * Ntbn.z);
            };
            if (u.x < t) {
                u.x = 2.f * u.x;
                L = CosineSampleHemisphere(u);
                L = normalize(mulTBNg(L));
                H = normalize(L + V);
            } else {
                u.x = 2.f * (u.x - 0.5f);
                H = SampleGGX(u, matInfo.roughness);
                H = normalize(mulTBNg(H));
                L = normalize(reflect(-V, H));
            }
            float NoL = hmax(dot(N, L), 0.001f);
            float NoV = hmax(dot(N, V), 0.001f);
            float NoH = hmax(dot(N, H), 0.001f);
            float VoH = hmax(dot(V, H), 0.001f);
            float pdf = t * CosineHemispherePdf(NoL) + qdiv((1.f - t) * GGXPdf(NoH, matInfo.roughness), 4.f * VoH);
            float3 F0 = matInfo.metallic * matInfo.basecolor + (1.f - matInfo.metallic) * 0.04f;
            float3 albedo = (1.f - matInfo.metallic) * matInfo.basecolor;
            float D = GGX(NoH, matInfo.roughness);
            float3 F = F_Schlick(VoH, F0);
            float G = G_Smith_Direct(NoV, NoL, matInfo.roughness);
            float3 Fd = albedo * Fd_Lambert();
            float3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
            float3 brdf = (Fr + (1.f - F) * Fd);
            float3 outAttenuation = brdf * NoL / pdf;
            attenuation *= outAttenuation;
            // Q2: any(attenuation) > 0.001  ==  any component != 0  (NaN counts as non-zero)
            if (attenuation.x != 0.f || attenuation.y != 0.f || attenuation.z != 0.f) {
                ray.o = hitPosition; ray.d = L; ray.tmin = k_rayOffset; ray.tmax = 10000.0f;
                continue;
            }
        }
        break;
    }
    return color;
}

// frame constants -- render-jobs/path-tracing.inl:134-217 -------------------------------------------
static Frame make_frame(const Scene& s, const PtDesc& d) {
    Frame f{};
    const Mat4 V = to_mat4(d.view.view_transform), P = to_mat4(d.view.projection_transform);
    f.projectionToWorld = mat_inverse(mat_mul(V, P));
    f.cameraMatrix = mat_inverse(V);
    f.sceneRotation = to_mat4(s.d.root_transform);
    f.W = d.res_x; f.H = d.res_y; f.numPixels = d.res_x * d.res_y;
    f.sampleIdx = d.current_sample_index;
    f.sqrtSampleCount = (uint32_t)std::sqrt((double)d.max_sample_count);     // :207 std::sqrt(MaxSampleCount)
    f.aperture = d.camera_aperture; f.focalLength = d.camera_focal_length;
    f.skyBrightness = d.sky_brightness; f.turbidity = d.turbidity;
    f.directLighting = d.direct_lighting; f.viewmode = d.viewmode; f.envSkyMode = d.env_sky_mode;
    f.lightCount = (int)s.d.num_global_lights;                               // :198 m_globalLightList.size()
    perez_from_turbidity(d.turbidity, &f.perez);
    // Sun direction for the miss shader (:181-189)
    float3 L = {1.f, 0.1f, 1.f};
    int sun = -1;                                                            // FScene::GetDirectionalLight scene.cpp:1577-1587
    for (uint32_t i = 0; i < s.d.num_scene_lights; ++i) {
        int li = s.d.light_indices[i];
        if ((uint32_t)li < s.d.num_global_lights && s.d.global_lights[li].type == INCR_LIGHT_DIRECTIONAL) { sun = li; break; }
    }
    if (sun != -1 && (uint32_t)sun < s.d.num_scene_lights) {
        const float* m = s.d.light_transforms[sun].m;
        L = f3(-m[8], -m[9], -m[10]);                                        // (0,0,-1,0) * M with translation zeroed
    }
    L = normalize(L);
    f.sunMiss = sky_sun_terms(L, d.turbidity, f.perez);
    f.sunEnv = sky_sun_terms(f3(s.d.sun_dir[0], s.d.sun_dir[1], s.d.sun_dir[2]), d.turbidity, f.perez);
    f.envmap = d.envmap_r11g11b10; f.envmapRes = d.envmap_res; f.envmapMips = d.envmap_mips;
    return f;
}

static uint32_t shard_local_rows(uint32_t res_y, const incr_shard& sh) {
    if (sh.row_block_height == 0) return res_y;
    const uint32_t nblk = (res_y + sh.row_block_height - 1) / sh.row_block_height;
    const uint32_t stride = sh.row_block_stride ? sh.row_block_stride : 1;
    const uint32_t mine = (nblk + stride - 1) / stride;     // same for every rank (equal-size all_gather); surplus rows are padding
    return mine * sh.row_block_height;
}
// local row -> global row, or -1 when the local row is padding past the end of the frame
static int64_t shard_global_row(uint32_t res_y, const incr_shard& sh, uint32_t local_row) {
    if (sh.row_block_height == 0) return local_row < res_y ? local_row : -1;
    const uint32_t stride = sh.row_block_stride ? sh.row_block_stride : 1;
    const uint32_t k = local_row / sh.row_block_height, r = local_row % sh.row_block_height;
    const uint64_t y = (uint64_t)(sh.row_block_first + (uint64_t)k * stride) * sh.row_block_height + r;
    return y < res_y ? (int64_t)y : -1;
}

// Mip chains (FScene::LoadTexture, src/scene.cpp:1072-1090: GenerateMipMaps(TEX_FILTER_LINEAR) down to 4x4; PrefilterNormalRoughnessTextures
// :1157-1320 for a material's normal + metallic-roughness pair).  Pinned as in the product (csrc/texture_mips.hip): 2x2 box, odd sizes clamp
// the second column / row, sRGB colour averaged in linear light and re-encoded to the nearest entry of the decode table, other channels
// average bytes with round-to-nearest.
extern "C" uint32_t orc_prefilter_normal_roughness(const uint32_t* srcN, const uint32_t* srcMR, uint32_t W, uint32_t H, uint32_t* outN, uint32_t* outMR);
static uint32_t srgb_encode_nearest(const float* lut, float v) {
    uint32_t lo = 0u, hi = 255u;
    for (int it = 0; it < 8; ++it) { const uint32_t mid = (lo + hi + 1u) >> 1; if (lut[mid] <= v) lo = mid; else hi = mid - 1u; }
    if (lo < 255u && (lut[lo + 1u] - v) < (v - lut[lo])) ++lo;
    return lo;
}
// ---- block-compressed chains at the boundary (incr_types.h formats 4-7): what the reference uploads after DirectX::Compress
// (scene.cpp:1100, 1401).  The block layouts are those of the D3D11.3 functional spec 19.5 (BC1 colour block inside BC3; BC4 blocks for
// BC3's alpha and for each channel of BC5).  Pinned: a BC4 / BC5 texel keeps its palette entry EXACTLY, as the
// numerator over 7 x 5 x 255 (x 127 signed); a BC3 texel becomes 8 bits per channel, interpolants rounded to nearest.
struct Bc4Palette { int32_t times35[8]; };          // the eight values of one BC4 block, each multiplied by 35
static Bc4Palette bc4_palette(const uint8_t* block, bool snorm) {
    int32_t a = block[0], b = block[1];
    if (snorm) { a = (int8_t)block[0]; b = (int8_t)block[1]; a = std::max(a, -127); b = std::max(b, -127); }
    Bc4Palette p;
    p.times35[0] = 35 * a; p.times35[1] = 35 * b;
    if (a > b) {
        for (int k = 1; k <= 6; ++k) p.times35[1 + k] = 5 * ((7 - k) * a + k * b);          // (7 - k) / 7 a + k / 7 b
    } else {
        for (int k = 1; k <= 4; ++k) p.times35[1 + k] = 7 * ((5 - k) * a + k * b);          // (5 - k) / 5 a + k / 5 b
        p.times35[6] = snorm ? -35 * 127 : 0;
        p.times35[7] = snorm ? 35 * 127 : 35 * 255;
    }
    return p;
}
static uint32_t bc4_selector(const uint8_t* block, uint32_t texel) {                        // 3 bits per texel, little-endian, after the two endpoints
    uint64_t v = 0; for (int k = 5; k >= 0; --k) v = (v << 8) | block[2 + k];
    return (uint32_t)((v >> (3 * texel)) & 7u);
}
static void expand_bc_level(const uint8_t* blocks, uint32_t w, uint32_t h, uint32_t format, uint32_t* out) {
    const uint32_t bw = (w + 3) / 4, bytes = format == INCR_TEX_BC4_UNORM ? 8 : 16;
    for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
        const uint8_t* blk = blocks + ((size_t)(y / 4) * bw + x / 4) * bytes;
        const uint32_t texel = (y % 4) * 4 + x % 4;
        uint32_t word = 0;
        if (format == INCR_TEX_BC3_SRGB) {
            const uint32_t alpha = (uint32_t)(bc4_palette(blk, false).times35[bc4_selector(blk, texel)] + 17) / 35u;
            const uint32_t e[2] = {(uint32_t)blk[8] | ((uint32_t)blk[9] << 8), (uint32_t)blk[10] | ((uint32_t)blk[11] << 8)};
            uint32_t rgb[2][3];
            for (int k = 0; k < 2; ++k) {
                const uint32_t r5 = e[k] >> 11, g6 = (e[k] >> 5) & 63u, b5 = e[k] & 31u;
                rgb[k][0] = (r5 << 3) | (r5 >> 2); rgb[k][1] = (g6 << 2) | (g6 >> 4); rgb[k][2] = (b5 << 3) | (b5 >> 2);
            }
            const uint32_t selectors = (uint32_t)blk[12] | ((uint32_t)blk[13] << 8) | ((uint32_t)blk[14] << 16) | ((uint32_t)blk[15] << 24);
            const uint32_t sel = (selectors >> (2 * texel)) & 3u;
            word = alpha << 24;
            for (int ch = 0; ch < 3; ++ch) {
                uint32_t v;
                switch (sel) {
                case 0: v = rgb[0][ch]; break;
                case 1: v = rgb[1][ch]; break;
                case 2: v = (2 * rgb[0][ch] + rgb[1][ch] + 1) / 3; break;          // BC3's colour block is always in four-colour mode
                default: v = (rgb[0][ch] + 2 * rgb[1][ch] + 1) / 3; break;
                }
                word |= v << (8 * ch);
            }
        } else {
            const bool snorm = format == INCR_TEX_BC5_SNORM;
            const int32_t r = bc4_palette(blk, snorm).times35[bc4_selector(blk, texel)];
            const int32_t g = format == INCR_TEX_BC4_UNORM ? 0 : bc4_palette(blk + 8, snorm).times35[bc4_selector(blk + 8, texel)];
            word = ((uint32_t)r & 0xffffu) | ((uint32_t)g << 16);
        }
        out[(size_t)y * w + x] = word;
    }
}
static void build_texture_mips(Scene& s) {
    const uint32_t nt = s.d.num_textures;
    s.texChain.assign(nt, {}); s.texMips.assign(nt, 1u); s.texFormat.assign(nt, 0u);
    s.maxAnisotropy = s.d.max_anisotropy ? std::min(s.d.max_anisotropy, 16u) : 16u;
    auto dim = [](uint32_t d, uint32_t l) { return std::max(d >> l, 1u); };
    for (uint32_t i = 0; i < nt; ++i) {
        const incr_texture_desc& t = s.d.textures[i];
        s.texFormat[i] = t.format;
        if (t.format >= INCR_TEX_BC3_SRGB && t.format <= INCR_TEX_BC5_SNORM) {          // a compressed chain: expanded level by level, nothing generated
            s.texMips[i] = t.mip_levels ? t.mip_levels : 1u;
            s.texFormat[i] = t.format == INCR_TEX_BC3_SRGB ? (uint32_t)INCR_TEX_RGBA8_SRGB : (t.format == INCR_TEX_BC5_SNORM ? (uint32_t)INCR_TEX_RG16S_OVER_4445 : (uint32_t)INCR_TEX_RG16_OVER_8925);
            size_t texels = 0; for (uint32_t l = 0; l < s.texMips[i]; ++l) texels += (size_t)dim(t.width, l) * dim(t.height, l);
            s.texChain[i].assign(texels, 0u);
            const uint8_t* src = t.texels; uint32_t* dst = s.texChain[i].data();
            for (uint32_t l = 0; l < s.texMips[i]; ++l) {
                const uint32_t w = dim(t.width, l), h = dim(t.height, l);
                expand_bc_level(src, w, h, t.format, dst);
                src += (size_t)((w + 3) / 4) * ((h + 3) / 4) * (t.format == INCR_TEX_BC4_UNORM ? 8 : 16); dst += (size_t)w * h;
            }
            continue;
        }
        uint32_t n = 0; for (uint32_t w = t.width, h = t.height; w >= 4 && h >= 4; w >>= 1, h >>= 1) ++n;
        s.texMips[i] = n ? n : 1u;
        size_t texels = 0; for (uint32_t l = 0; l < s.texMips[i]; ++l) texels += (size_t)dim(t.width, l) * dim(t.height, l);
        s.texChain[i].assign(texels, 0u);
        memcpy(s.texChain[i].data(), t.texels, (size_t)t.width * t.height * 4);
    }
    std::vector<uint8_t> have(nt, 0);
    for (uint32_t i = 0; i < nt; ++i) have[i] = s.d.textures[i].format >= INCR_TEX_BC3_SRGB ? 1 : 0;
    for (uint32_t mi = 0; mi < s.d.num_materials; ++mi) {
        const int32_t a = s.d.materials[mi].normalTextureIndex, b = s.d.materials[mi].metallicRoughnessTextureIndex;
        if (a < 0 || b < 0 || a == b || (uint32_t)a >= nt || (uint32_t)b >= nt || have[(size_t)a] || have[(size_t)b]) continue;
        const incr_texture_desc& N = s.d.textures[a]; const incr_texture_desc& M = s.d.textures[b];
        const uint32_t mips = s.texMips[(size_t)a];
        if (N.width != M.width || N.height != M.height || mips < 2 || N.format != INCR_TEX_RG_SNORM || M.format != INCR_TEX_BGRA8_SWIZZLE_RG) continue;
        if ((N.width >> (mips - 1)) << (mips - 1) != N.width || (N.height >> (mips - 1)) << (mips - 1) != N.height) continue;
        std::vector<uint32_t> outN(s.texChain[(size_t)a].size()), outM(outN.size());
        orc_prefilter_normal_roughness(s.texChain[(size_t)a].data(), s.texChain[(size_t)b].data(), N.width, N.height, outN.data(), outM.data());
        for (size_t k = (size_t)N.width * N.height; k < outN.size(); ++k) {
            s.texChain[(size_t)a][k] = outN[k];
            s.texChain[(size_t)b][k] = (outM[k] & 0x0000ff00u) | ((outM[k] & 0xffu) << 16);      // back to the glTF channel order the sampler swizzles
        }
        have[(size_t)a] = have[(size_t)b] = 1;
    }
    for (uint32_t i = 0; i < nt; ++i) {
        if (have[i]) continue;
        const incr_texture_desc& t = s.d.textures[i];
        uint32_t* level = s.texChain[i].data();
        for (uint32_t l = 0; l + 1 < s.texMips[i]; ++l) {
            const uint32_t sw = dim(t.width, l), sh = dim(t.height, l), dw = dim(t.width, l + 1), dh = dim(t.height, l + 1);
            uint32_t* next = level + (size_t)sw * sh;
            for (uint32_t y = 0; y < dh; ++y) for (uint32_t x = 0; x < dw; ++x) {
                const uint32_t x0 = std::min(2u * x, sw - 1u), x1 = std::min(2u * x + 1u, sw - 1u), y0 = std::min(2u * y, sh - 1u), y1 = std::min(2u * y + 1u, sh - 1u);
                const uint32_t c[4] = {level[(size_t)y0 * sw + x0], level[(size_t)y0 * sw + x1], level[(size_t)y1 * sw + x0], level[(size_t)y1 * sw + x1]};
                uint32_t out = 0u;
                for (int ch = 0; ch < 4; ++ch) {
                    const uint32_t bb[4] = {(c[0] >> (8 * ch)) & 0xffu, (c[1] >> (8 * ch)) & 0xffu, (c[2] >> (8 * ch)) & 0xffu, (c[3] >> (8 * ch)) & 0xffu};
                    uint32_t r;
                    if (t.format == INCR_TEX_RGBA8_SRGB && ch < 3) r = srgb_encode_nearest(s.srgb_lut, ((s.srgb_lut[bb[0]] + s.srgb_lut[bb[1]]) + (s.srgb_lut[bb[2]] + s.srgb_lut[bb[3]])) * 0.25f);
                    else r = (uint32_t)((float)(bb[0] + bb[1] + bb[2] + bb[3]) * 0.25f + 0.5f);
                    out |= r << (8 * ch);
                }
                next[(size_t)y * dw + x] = out;
            }
            level = next;
        }
    }
}

extern "C" {

// --- KAT-level exports (golden vectors, unit tests) -------------------------------------------------
uint32_t orc_cmj_permute(uint32_t i, uint32_t l, uint32_t p) { return CMJ_Permute(i, l, p); }
float    orc_cmj_randfloat(uint32_t i, uint32_t p) { return CMJ_RandFloat(i, p); }
void     orc_cmj_sample(uint32_t sampleIdx, uint32_t nx, uint32_t ny, uint32_t pattern, float out[2]) {
    float2 r = CorrelatedMultiJitteredSampling(sampleIdx, nx, ny, pattern); out[0] = r.x; out[1] = r.y;
}
void     orc_sample_point(uint32_t pixelIdx, uint32_t sampleIdx, uint32_t setIdx, uint32_t sqrtN, uint32_t numPixels, float out[2]) {
    float2 r = SamplePoint(pixelIdx, sampleIdx, setIdx, sqrtN, numPixels); out[0] = r.x; out[1] = r.y;
}
float    orc_halton(uint32_t i, uint32_t base) { return Halton(i, base); }
void     orc_concentric_disk(const float u[2], float out[2]) { float2 r = ConcentricSampleDisk({u[0], u[1]}); out[0] = r.x; out[1] = r.y; }
void     orc_cosine_hemisphere(const float u[2], float out[3]) { float3 r = CosineSampleHemisphere({u[0], u[1]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
float    orc_ggx(float NoH, float r) { return GGX(NoH, r); }
float    orc_ggx_pdf(float NoH, float r) { return GGXPdf(NoH, r); }
void     orc_sample_ggx(const float u[2], float r, float out[3]) { float3 h = SampleGGX({u[0], u[1]}, r); out[0] = h.x; out[1] = h.y; out[2] = h.z; }
float    orc_g_smith_direct(float NoV, float NoL, float r) { return G_Smith_Direct(NoV, NoL, r); }
float    orc_g_smith_ggx_correlated(float NoV, float NoL, float r) { return G_SmithGGXCorrelated(NoV, NoL, r); }
void     orc_f_schlick(float u, const float f0[3], float out[3]) { float3 r = F_Schlick(u, {f0[0], f0[1], f0[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
float    orc_exposure(int ev100) { return Exposure(ev100); }
void     orc_aces_film(const float x[3], float out[3]) { float3 r = ACESFilm({x[0], x[1], x[2]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void     orc_perez_from_turbidity(float t, incr_perez_distribution* p) { perez_from_turbidity(t, p); }
void     orc_sky_radiance(const float s[3], const float e[3], float t, float out[3]) {
    incr_perez_distribution pc; perez_from_turbidity(t, &pc);
    SkySunTerms sun = sky_sun_terms({s[0], s[1], s[2]}, t, pc);
    float3 r = CalculateSkyRadianceRGB(sun, {e[0], e[1], e[2]}, pc); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint16_t orc_f32_to_f16(float f, int rounding) { return f32_to_f16(f, rounding); }
float    orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
uint32_t orc_pack_r11g11b10(const float c[3], int rounding) { return pack_r11g11b10({c[0], c[1], c[2]}, rounding); }
void     orc_unpack_r11g11b10(uint32_t p, float out[3]) { float3 r = unpack_r11g11b10(p); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void     orc_mat_inverse(const incr_matrix* m, incr_matrix* out) { Mat4 r = mat_inverse(to_mat4(*m)); memcpy(out->m, r.m, 64); }
void     orc_mat_mul(const incr_matrix* a, const incr_matrix* b, incr_matrix* out) { Mat4 r = mat_mul(to_mat4(*a), to_mat4(*b)); memcpy(out->m, r.m, 64); }
uint32_t orc_shard_local_rows(uint32_t res_y, const incr_shard* sh) { return shard_local_rows(res_y, *sh); }

// --- scene ---------------------------------------------------------------------------------------
void* orc_scene_create(const incr_scene_desc* desc) {
    Scene* s = new Scene();
    s->d = *desc;     // borrows the caller's host arrays for the lifetime of the oracle scene
    for (int i = 0; i < 256; ++i) {     // D3D sRGB -> linear (D3D11.3 spec 3.2.3.x): exact formula in double, rounded once
        double c = i / 255.0;
        s->srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    build_texture_mips(*s);
    flatten_scene(*s);
    return s;
}
void orc_scene_destroy(void* scene) { delete (Scene*)scene; }
uint64_t orc_scene_num_triangles(void* scene) { return ((Scene*)scene)->tris.size(); }

uint32_t orc_texture_chain(void* scene, uint32_t texIndex, uint32_t* outTexels, uint64_t capacity) {      // returns the level count; copies min(chain, capacity) texels
    const Scene& s = *(Scene*)scene;
    if (texIndex >= s.texChain.size()) return 0;
    const std::vector<uint32_t>& c = s.texChain[texIndex];
    if (outTexels) memcpy(outTexels, c.data(), std::min<uint64_t>(capacity, c.size()) * 4);
    return s.texMips[texIndex];
}
void orc_sample_texture_grad(void* scene, int texIndex, const float uv[2], const float ddx[2], const float ddy[2], float out[4]) {
    float4 r = SampleGrad(*(Scene*)scene, texIndex, {uv[0], uv[1]}, {ddx[0], ddx[1]}, {ddy[0], ddy[1]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_sample_texture(void* scene, int texIndex, const float uv[2], float out[4]) {
    float4 r = SampleLevel0(*(Scene*)scene, texIndex, {uv[0], uv[1]}); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// primary-visibility records for a set of pixels: out[i] = {instance, geometry, primitive, hit?} and t,u,v
void orc_camera_rays(void* scene, const PtDesc* d, uint32_t n, const uint32_t* px, const uint32_t* py, float* origins, float* dirs) {
    const Scene& s = *(Scene*)scene;
    Frame f = make_frame(s, *d);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t set = 0; const uint32_t pixelIdx = py[i] * f.W + px[i];
        float2 j = SamplePoint(pixelIdx, f.sampleIdx, set, f.sqrtSampleCount, f.numPixels);
        float2 a = SamplePoint(pixelIdx, f.sampleIdx, set, f.sqrtSampleCount, f.numPixels);
        Ray r = GenerateCameraRay(f, {(float)px[i] + j.x, (float)py[i] + j.y}, a);
        origins[i * 3 + 0] = r.o.x; origins[i * 3 + 1] = r.o.y; origins[i * 3 + 2] = r.o.z;
        dirs[i * 3 + 0] = r.d.x; dirs[i * 3 + 1] = r.d.y; dirs[i * 3 + 2] = r.d.z;
    }
}
// generic ray queries against the oracle BVH: ids[i] = {hit, instance, geometry, primitive}, tuv[i] = {t,u,v}
void orc_trace_rays(void* scene, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax, int any_hit, uint32_t* ids, float* tuv) {
    const Scene& s = *(Scene*)scene;
    for (uint32_t i = 0; i < n; ++i) {
        float3 o = {origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]}, d = {dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]};
        ids[i * 4 + 0] = ids[i * 4 + 1] = ids[i * 4 + 2] = ids[i * 4 + 3] = 0; tuv[i * 3] = tuv[i * 3 + 1] = tuv[i * 3 + 2] = 0;
        if (any_hit) { ids[i * 4] = trace_any(s, o, d, tmin, tmax) ? 1u : 0u; continue; }
        Hit h;
        if (trace_closest(s, o, d, tmin, tmax, &h)) {
            const Scene::Tri& T = s.tris[h.tri];
            ids[i * 4] = 1; ids[i * 4 + 1] = T.inst; ids[i * 4 + 2] = T.geom; ids[i * 4 + 3] = T.prim;
            tuv[i * 3] = h.t; tuv[i * 3 + 1] = h.u; tuv[i * 3 + 2] = h.v;
        }
    }
}

// --- RenderJob::PathTracing::Execute: DispatchRays + integrate ---------------------------------------
// `float_target` (optional, local_rows*res_x*4 floats) receives the un-quantised payload colour.
int orc_pathtrace(void* scene, const PtDesc* d, int num_threads, float* float_target) {
    const Scene& s = *(Scene*)scene;
    if (d->env_sky_mode != 1 && !(d->env_sky_mode == 0 && d->envmap_r11g11b10)) return -1;
    const Frame f = make_frame(s, *d);
    const uint32_t localRows = shard_local_rows(d->res_y, d->shard);
    uint16_t* target = (uint16_t*)d->target_rgba16f;
    uint32_t* history = (uint32_t*)d->history_r11g11b10;
    const uint32_t W = d->res_x;
    const uint32_t tilesX = (W + 31) / 32, tilesY = (localRows + 31) / 32;
    std::atomic<uint32_t> next{0};
    if (num_threads < 1) num_threads = 1;
    std::vector<RayStats> stats((size_t)num_threads);
    auto worker = [&](int tid) {
        RayStats& rs = stats[(size_t)tid];
        for (;;) {
            const uint32_t tile = next.fetch_add(1);
            if (tile >= tilesX * tilesY) break;
            const uint32_t tx = tile % tilesX, ty = tile / tilesX;
            for (uint32_t ly = ty * 32; ly < std::min(localRows, ty * 32 + 32); ++ly) {
                const int64_t gy = shard_global_row(d->res_y, d->shard, ly);
                for (uint32_t x = tx * 32; x < std::min(W, tx * 32 + 32); ++x) {
                    const size_t li = (size_t)ly * W + x;
                    float4 c = {0, 0, 0, 0};
                    if (gy >= 0) c = trace_pixel(s, f, x, (uint32_t)gy, &rs, d->debug_hit_chain ? d->debug_hit_chain + li * 12 : nullptr);
                    if (float_target) { float_target[li * 4] = c.x; float_target[li * 4 + 1] = c.y; float_target[li * 4 + 2] = c.z; float_target[li * 4 + 3] = c.w; }
                    // destUav[xy] = payload.color  -> RGBA16F typed store (:118)
                    uint16_t h[4] = {f32_to_f16(c.x, d->storage_rounding), f32_to_f16(c.y, d->storage_rounding),
                                     f32_to_f16(c.z, d->storage_rounding), f32_to_f16(c.w, d->storage_rounding)};
                    if (target) memcpy(target + li * 4, h, 8);
                    // pathtrace-integrate.hlsl:16-31
                    if (history) {
                        float3 cur = {f16_to_f32(h[0]), f16_to_f32(h[1]), f16_to_f32(h[2])};
                        float3 hist = unpack_r11g11b10(history[li]);
                        const float n = (float)d->current_sample_index;
                        float3 avg = ((n - 1.f) * hist + cur) / n;
                        if (d->current_sample_index <= 1) avg = cur / n;   // history is freshly cleared at the first sample
                        history[li] = pack_r11g11b10(avg, d->storage_rounding);
                    }
                }
            }
        }
    };
    if (num_threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < num_threads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    if (d->ray_stats) {
        for (const RayStats& r : stats) {
            d->ray_stats->camera_rays += r.camera; d->ray_stats->bounce_rays += r.bounce;
            d->ray_stats->sky_shadow_rays += r.skyShadow; d->ray_stats->light_shadow_rays += r.lightShadow;
            d->ray_stats->shaded_hits += r.hits;
        }
    }
    return 0;
}

// --- TonemapPass: shaders/postprocess/tonemap.hlsl:30-59 -> R8G8B8A8_UNORM_SRGB back buffer ----------
static uint8_t linear_to_srgb8(float c) {
    // D3D11.3 spec FLOAT -> UNORM_SRGB: clamp, encode, scale, round to nearest
    if (!(c > 0.f)) c = 0.f;          // NaN -> 0
    if (c > 1.f) c = 1.f;
    float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f;
    return (uint8_t)floorf(e * 255.f + 0.5f);
}
void orc_tonemap(const uint32_t* hdr_r11g11b10, uint8_t* ldr_rgba8, uint32_t res_x, uint32_t rows, float exposure, int viewmode) {
    const float e = Exposure((int)exposure);
    for (size_t i = 0; i < (size_t)res_x * rows; ++i) {
        float3 hdr = unpack_r11g11b10(hdr_r11g11b10[i]);
        float3 ldr;
        if (viewmode != 0 && viewmode != 1) ldr = hdr;
        else ldr = ACESFilm(hdr * e);
        ldr_rgba8[i * 4 + 0] = linear_to_srgb8(ldr.x); ldr_rgba8[i * 4 + 1] = linear_to_srgb8(ldr.y);
        ldr_rgba8[i * 4 + 2] = linear_to_srgb8(ldr.z); ldr_rgba8[i * 4 + 3] = 255;
    }
}

} // extern "C"

// sky / IBL precompute feeding the hot path; its cube-map sampler is also what the sky-lighting pass uses
#include "oracle_ibl.h"

// raster half of the hot path: shares the scene access, texture unit and BRDF code above
#include "oracle_raster.h"
