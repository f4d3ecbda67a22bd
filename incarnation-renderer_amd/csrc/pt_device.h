// pt_device.h -- device-side math, sampling, BRDF, sky, texture and storage-format helpers for the
// path-tracing kernels (gfx950).  Compiled with -ffp-contract=off: every +,-,*,/ and sqrt below is a
// single correctly-rounded IEEE operation, so results are reproducible against the CPU oracle.
// Reference files restated here (relative to /root/reference/source/demo-dll/shaders/):
//   common/uniform-sampling.hlsli, common/bxdf-sampling.hlsli, common/math.hlsli, lighting/pbr.hlsli,
//   environment-sky/preetham.hlsli, material/common.hlsli, raytracing/common.hlsli
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/incr.h"
#include "device_types.h"
#include "pt_params.h"

namespace incr {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

#define DI __device__ __forceinline__

// Division.  HLSL's a / b is DXIL FDiv, which GPUs lower to a reciprocal and a multiplication; INCR_DIV selects how it is evaluated here:
//   0  IEEE a / b (v_div_scale x 2, v_rcp, 6 fused ops, v_div_fmas, v_div_fixup: 11 instructions, 14 issue slots)
//   1  a * rcp(b) with rcp(b) the CORRECTLY ROUNDED reciprocal: v_rcp_f32 (1 ulp) + two Newton steps in fused arithmetic (Markstein) + v_div_fixup
//      for zero / infinity / NaN -- 6 instructions + the multiplication
//   2  a * rcp(b) with one Newton step (faithful, correctly rounded in all but ~2^-22 of the cases), 3 instructions + the multiplication
#ifndef INCR_DIV
#define INCR_DIV 1          // pinned in the oracle as a * (1.f / b) (oracle_math.h): 1 reproduces it bit for bit; 0 and 2 are A/B builds
#endif
DI float rcp_cr(float x) {
#if INCR_DIV == 2
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, y0, 1.f);
    return __builtin_fmaf(y0, e0, y0);
#else
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, y0, 1.f);
    const float y1 = __builtin_fmaf(y0, e0, y0);
    const float e1 = __builtin_fmaf(-x, y1, 1.f);
    const float y2 = __builtin_fmaf(y1, e1, y1);
    return __builtin_amdgcn_div_fixupf(y2, x, 1.f);
#endif
}
#if INCR_DIV == 0
#define QDIV(a, b) ((a) / (b))
#else
#define QDIV(a, b) ((a) * rcp_cr(b))
#endif
DI f3 mk3(float x, float y, float z) { return {x, y, z}; }
DI f3 mk3(float s) { return {s, s, s}; }
DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DI f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DI f3 operator/(f3 a, f3 b) { return {QDIV(a.x, b.x), QDIV(a.y, b.y), QDIV(a.z, b.z)}; }
DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DI f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
#if INCR_DIV == 0
DI f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
#else
DI f3 operator/(f3 a, float s) { const float r = rcp_cr(s); return {a.x * r, a.y * r, a.z * r}; }
#endif
DI f3 operator+(float s, f3 a) { return {s + a.x, s + a.y, s + a.z}; }
DI f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
DI f3 operator-(float s, f3 a) { return {s - a.x, s - a.y, s - a.z}; }
DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
DI f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
DI f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
DI f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
DI f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DI f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
DI f4 operator*(float s, f4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }

DI float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DI float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
DI f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DI f3 normalize(f3 v) { const float r = QDIV(1.f, sqrtf(dot(v, v))); return {v.x * r, v.y * r, v.z * r}; }      // HLSL normalize = x * rsqrt(dot(x, x)) (DXIL: Dot3, Rsqrt, FMul), rsqrt pinned to 1 / sqrt: DESIGN.md section 3
DI float hmax(float a, float b) { return fmaxf(a, b); }
DI float hmin(float a, float b) { return fminf(a, b); }
DI float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
DI f3 saturate(f3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
DI f3 reflect(f3 i, f3 n) { float d = dot(i, n); return i - (2.f * n) * d; }
DI float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
DI float pow4(float x) { float x2 = x * x; return x2 * x2; }

// (v,0) * M  /  (v,w) * M for row-major M with row vectors
DI f3 mul_dir(f3 v, const float* M) {
    return {(v.x * M[0] + v.y * M[4]) + v.z * M[8], (v.x * M[1] + v.y * M[5]) + v.z * M[9], (v.x * M[2] + v.y * M[6]) + v.z * M[10]};
}
DI f4 mul_vec4(f4 v, const float* M) {
    f4 r;
    r.x = ((v.x * M[0] + v.y * M[4]) + v.z * M[8]) + v.w * M[12];
    r.y = ((v.x * M[1] + v.y * M[5]) + v.z * M[9]) + v.w * M[13];
    r.z = ((v.x * M[2] + v.y * M[6]) + v.z * M[10]) + v.w * M[14];
    r.w = ((v.x * M[3] + v.y * M[7]) + v.z * M[11]) + v.w * M[15];
    return r;
}

DI uint32_t fast_div(uint32_t n, const FastDiv& f) {
    if (f.d == 1u) return n;                                  // wave-uniform
    const uint32_t q = __umulhi(f.magic, n);
    return (((n - q) >> 1) + q) >> f.more;
}
DI uint32_t fast_mod(uint32_t n, const FastDiv& f) { return n - fast_div(n, f) * f.d; }

// ---------------------------------------------------------------- CMJ sampler (uniform-sampling.hlsli:152-201)
// (the divisions by the launch constants nx = ny and N = nx * ny go through FastDiv: same integers)
DI uint32_t CMJ_Permute(uint32_t i, const FastDiv& dl, uint32_t p) {
    const uint32_t l = dl.d;
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
    return fast_mod(i + p, dl);
}
DI float CMJ_RandFloat(uint32_t i, uint32_t p) {
    i ^= p;
    i ^= i >> 17;
    i ^= i >> 10; i *= 0xb36534e5;
    i ^= i >> 12;
    i ^= i >> 21; i *= 0x93fc4795;
    i ^= 0xdf6e307f;
    i ^= i >> 17; i *= 1 | p >> 18;
    return (float)i * (1.0f / 4294967808.0f);
}
DI f2 CorrelatedMultiJitteredSampling(uint32_t sampleIdx, const FastDiv& dn, const FastDiv& dN, uint32_t pattern) {      // nx == ny == dn.d, N == dN.d
    const uint32_t nx = dn.d, ny = dn.d, N = dN.d;
    sampleIdx = CMJ_Permute(sampleIdx, dN, pattern * 0x51633e2d);
    const uint32_t row = fast_div(sampleIdx, dn);
    uint32_t sx = CMJ_Permute(sampleIdx - row * nx, dn, pattern * 0x68bc21eb);
    uint32_t sy = CMJ_Permute(row, dn, pattern * 0x02e5be93);
    float jx = CMJ_RandFloat(sampleIdx, pattern * 0x967a889b);
    float jy = CMJ_RandFloat(sampleIdx, pattern * 0x368cc8b7);
    return {QDIV((float)sx + QDIV((float)sy + jx, (float)ny), (float)nx), QDIV((float)sampleIdx + jy, (float)N)};
}
// raytracing/common.hlsli:8-14
DI f2 SamplePoint(uint32_t pixelIdx, uint32_t sampleIdx, uint32_t& setIdx, const FastDiv& divSqrtN, const FastDiv& divN, uint32_t numPixels) {
    const uint32_t permutationIdx = setIdx * numPixels + pixelIdx;
    setIdx += 1;
    return CorrelatedMultiJitteredSampling(sampleIdx, divSqrtN, divN, permutationIdx);
}

#define K_PI 3.14159265358979323846f
#define K_INV_PI 0.31830988618379067154f
#define K_PI_OVER_2 1.57079632679489661923f
#define K_PI_OVER_4 0.78539816339744830961f

DI f2 ConcentricSampleDisk(f2 u) {                        // uniform-sampling.hlsli:20-45
    f2 o = {2.f * u.x - 1.f, 2.f * u.y - 1.f};
    if (o.x == 0.f && o.y == 0.f) return {0.f, 0.f};
    float theta, r;
    if (fabsf(o.x) > fabsf(o.y)) { r = o.x; theta = K_PI_OVER_4 * QDIV(o.y, o.x); }
    else { r = o.y; theta = K_PI_OVER_2 - K_PI_OVER_4 * QDIV(o.x, o.y); }
    return {r * cosf(theta), r * sinf(theta)};
}
DI f3 CosineSampleHemisphere(f2 u) {                      // :61-66
    f2 d = ConcentricSampleDisk(u);
    float z = sqrtf(hmax(0.f, 1.f - d.x * d.x - d.y * d.y));
    return {d.x, d.y, z};
}
DI float CosineHemispherePdf(float c) { return c * K_INV_PI; }

DI float GGX(float NoH, float Roughness) {                // bxdf-sampling.hlsli:14-26
    float a = Roughness * Roughness;
    float a2 = a * a;
    NoH = hmax(NoH, 0.f);
    float NoH2 = NoH * NoH;
    float denom = (NoH2 * (a2 - 1.f) + 1.f);
    denom = K_PI * denom * denom;
    return QDIV(a2, hmax(denom, 0.0001f));
}
DI f3 SampleGGX(f2 u, float roughness) {                  // :29-42
    float a = roughness * roughness;
    float Phi = 2 * K_PI * u.x;
    float CosTheta = sqrtf(QDIV(1.f - u.y, 1.f + (a * a - 1.f) * u.y));
    float SinTheta = sqrtf(1.f - CosTheta * CosTheta);
    return {SinTheta * sinf(Phi), SinTheta * cosf(Phi), CosTheta};
}
DI float GGXPdf(float NoH, float roughness) { return GGX(NoH, roughness) * NoH; }

DI float G_SmithGGXCorrelated(float NoV, float NoL, float roughness) {   // pbr.hlsli:8-14
    float a2 = roughness * roughness;
    float GGXV = NoL * sqrtf(NoV * NoV * (1.0f - a2) + a2);
    float GGXL = NoV * sqrtf(NoL * NoL * (1.0f - a2) + a2);
    return QDIV(0.5f, hmax(GGXV + GGXL, 0.00001f));
}
DI f3 F_Schlick(float u, f3 f0) { float f = pow5(1.0f - u); return f + f0 * (1.0f - f); }
DI float G_SchlickGGX(float u, float k) { return QDIV(u, u * (1.f - k) + k); }
DI float G_Smith_Direct(float NoV, float NoL, float roughness) {
    float a2 = roughness * roughness;
    float k = (a2 + 1) * (a2 + 1) / 8.f;
    return G_SchlickGGX(NoV, k) * G_SchlickGGX(NoL, k);
}
DI float Exposure(int ev100) { return 1.0f / (ldexpf(1.f, ev100) * 1.2f); }
DI f3 ACESFilm(f3 x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return saturate((x * (a * x + b)) / (x * (c * x + d) + e));
}
// R8G8B8A8_UNORM_SRGB render-target encode + the tonemap of one R11G11B10F texel (postprocess/tonemap.hlsl:30-59)
// The encode is floor(e(c) * 255 + 0.5) with e(c) = 12.92 c below 0.0031308 and 1.055 c^(1/2.4) - 0.055 above, a non-decreasing step function of c: it is given by
// the 255 values of c where it steps.  `steps[k]` is the smallest float whose code is >= k (steps[0] = 0, steps[256] = +inf), found at context creation by bisection
// with the HOST's powf -- the one the reference's CPU restatement calls: the GPU result equals it bit for bit (the device library's powf differs from it by an ulp
// here and there: round 1-4 tests allowed +-1 LSB), and the ~150 instructions of powf per channel -- half of the TAA resolve's instruction count -- become a
// hardware log2 / exp2 estimate of the code plus two table compares.
static constexpr uint32_t kSrgbSteps = 257;
DI uint32_t linear_to_srgb8(const float* __restrict__ steps, float c) {
    if (!(c > 0.f)) c = 0.f;
    if (c > 1.f) c = 1.f;
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.f / 2.4f)) - 0.055f;     // estimate: within a fraction of a code
    int k = (int)floorf(e * 255.f + 0.5f);
    k = min(max(k, 0), 255);
    if (c < steps[k]) --k;
    else if (c >= steps[k + 1]) ++k;
    return (uint32_t)k;
}

// ---------------------------------------------------------------- Preetham sky (preetham.hlsli:36-89)

DI f3 Perez(float cosTheta, float gamma, float cosGamma, const incr_perez_distribution& pc) {
    f3 A = {pc.A[0], pc.A[1], pc.A[2]}, B = {pc.B[0], pc.B[1], pc.B[2]}, C = {pc.C[0], pc.C[1], pc.C[2]},
       D = {pc.D[0], pc.D[1], pc.D[2]}, E = {pc.E[0], pc.E[1], pc.E[2]};
    f3 e1 = {expf(B.x / cosTheta), expf(B.y / cosTheta), expf(B.z / cosTheta)};
    f3 e2 = {expf(D.x * gamma), expf(D.y * gamma), expf(D.z * gamma)};
    return (1.0f + A * e1) * ((1.0f + C * e2) + (E * cosGamma) * cosGamma);
}
DI f3 Yxy2RGB(f3 Yxy) {
    float Y = Yxy.x, x = Yxy.y, y = Yxy.z;
    float X = x * (Y / y);
    float Z = (1.0f - x - y) * (Y / y);
    f3 RGB;
    RGB.x = 3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z;
    RGB.y = -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z;
    RGB.z = 0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z;
    return RGB;
}
DI f3 CalculateSkyRadianceRGB(const SkySun& sun, f3 e, const incr_perez_distribution& pc) {
    float cosE = saturate(e.y);
    float cosG = saturate(dot(mk3(sun.s[0], sun.s[1], sun.s[2]), e));
    float gammaE = acosf(cosG);
    f3 F = Perez(cosE, gammaE, cosG, pc);
    f3 Yxy = mk3(sun.YxyZenith[0], sun.YxyZenith[1], sun.YxyZenith[2]) * (F / mk3(sun.F0[0], sun.F0[1], sun.F0[2]));
    return Yxy2RGB(Yxy);
}

// ---------------------------------------------------------------- small-float typed-UAV stores
DI uint32_t encode_small_float(float f, int mant_bits, bool has_sign, int rounding, uint32_t* sign_out) {
    uint32_t u = __float_as_uint(f);
    uint32_t sign = u >> 31;
    uint32_t absu = u & 0x7fffffffu;
    const uint32_t exp_all = 31u << mant_bits;
    *sign_out = has_sign ? sign : 0;
    if (absu > 0x7f800000u) return exp_all | ((1u << mant_bits) - 1u);
    if (!has_sign && sign) return 0;
    if (absu == 0x7f800000u) return exp_all;
    int e = (int)(absu >> 23) - 127 + 15;
    uint32_t m = absu & 0x7fffffu;
    const int shift = 23 - mant_bits;
    const uint32_t max_finite = exp_all - 1u;
    if (e >= 31) return rounding == 0 ? max_finite : exp_all;
    uint32_t r;
    if (e <= 0) {
        if (e < -mant_bits - 1) return 0;
        m |= 0x800000u;
        int s = shift + 1 - e;
        if (s > 31) return 0;
        r = m >> s;
        if (rounding == 1) {
            uint32_t rem = m & ((1u << s) - 1u), half = 1u << (s - 1);
            if (rem > half || (rem == half && (r & 1u))) r += 1;
        }
        return r;
    }
    r = ((uint32_t)e << mant_bits) | (m >> shift);
    if (rounding == 1) {
        uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r += 1;
    }
    return r;
}
DI float decode_small_float(uint32_t v, int mant_bits, uint32_t sign) {
    uint32_t e = v >> mant_bits, m = v & ((1u << mant_bits) - 1u);
    float r;
    if (e == 0) r = ldexpf((float)m, -14 - mant_bits);
    else if (e == 31) r = m ? __uint_as_float(0x7fc00000u) : __uint_as_float(0x7f800000u);
    else r = ldexpf((float)(m | (1u << mant_bits)), (int)e - 15 - mant_bits);
    return sign ? -r : r;
}
// The same conversions on the hardware's own converters (INCR_FAST_SMALL_FLOAT, default): the default store rounding of every typed
// UAV here is round-toward-zero (D3D11.3 3.2.3.3), which is exactly v_cvt_pkrtz_f16_f32 -- overflow to the largest finite value,
// denormals kept (f16 denormals are enabled on gfx950), +-inf kept -- and an R11 / R10 channel is the f16 pattern with its low 4 / 5
// mantissa bits cut off (truncating twice = truncating once), negative values store 0, NaN the all-ones pattern as encode_small_float
// does.  Decoding is v_cvt_f32_f16 on the widened pattern.  One instruction where the generic code spends 10 (decode) to 35 (encode) per
// channel: the TAA resolve decodes 13 texels per pixel, the lighting chain re-quantises three times, a cube tap decodes 12 channels.
// Bit-identical to the generic functions on every input (tests/test_gpu_round3.py::test_small_float_codec_matches_the_generic_one);
// nearest-even stores (storage_rounding = 1) keep the generic path.
#ifndef INCR_FAST_SMALL_FLOAT
#define INCR_FAST_SMALL_FLOAT 1
#endif
typedef _Float16 incr_h2 __attribute__((ext_vector_type(2)));
DI uint32_t pkrtz_bits(float a, float b) { const incr_h2 h = __builtin_bit_cast(incr_h2, __builtin_amdgcn_cvt_pkrtz(a, b)); return __builtin_bit_cast(uint32_t, h); }
DI float f16_pattern_to_f32(uint32_t h16) { const unsigned short u = (unsigned short)h16; return (float)__builtin_bit_cast(_Float16, u); }
DI uint32_t f32_to_f16_bits_generic(float f, int rounding) { uint32_t s; uint32_t v = encode_small_float(f, 10, true, rounding, &s); return (s << 15) | v; }
DI uint32_t f32_to_f16_bits(float f, int rounding) {
#if INCR_FAST_SMALL_FLOAT
    if (rounding == 0) {
        const uint32_t h = pkrtz_bits(f, 0.f) & 0xffffu;
        return f != f ? ((__float_as_uint(f) >> 16) & 0x8000u) | 0x7fffu : h;
    }
#endif
    return f32_to_f16_bits_generic(f, rounding);
}
DI float f16_bits_to_f32(uint32_t h) {
#if INCR_FAST_SMALL_FLOAT
    return f16_pattern_to_f32(h & 0xffffu);
#else
    return decode_small_float(h & 0x7fffu, 10, (h >> 15) & 1u);
#endif
}
DI uint32_t pack_r11g11b10_generic(f3 c, int rounding) {
    uint32_t s;
    uint32_t r = encode_small_float(c.x, 6, false, rounding, &s);
    uint32_t g = encode_small_float(c.y, 6, false, rounding, &s);
    uint32_t b = encode_small_float(c.z, 5, false, rounding, &s);
    return r | (g << 11) | (b << 22);
}
DI uint32_t unsigned_small_from_f16(uint32_t h16, float f, int drop) {      // one R11 (drop 4) / R10 (drop 5) channel from the truncated f16 pattern of f
    if (f != f) return (0x7fffu >> drop);
    return (h16 & 0x8000u) ? 0u : (h16 >> drop);
}
DI uint32_t pack_r11g11b10(f3 c, int rounding) {
#if INCR_FAST_SMALL_FLOAT
    if (rounding == 0) {
        const uint32_t xy = pkrtz_bits(c.x, c.y), z = pkrtz_bits(c.z, 0.f);
        return unsigned_small_from_f16(xy & 0xffffu, c.x, 4) | (unsigned_small_from_f16(xy >> 16, c.y, 4) << 11) | (unsigned_small_from_f16(z & 0xffffu, c.z, 5) << 22);
    }
#endif
    return pack_r11g11b10_generic(c, rounding);
}
DI f3 unpack_r11g11b10_generic(uint32_t p) {
    return {decode_small_float(p & 0x7ffu, 6, 0), decode_small_float((p >> 11) & 0x7ffu, 6, 0), decode_small_float(p >> 22, 5, 0)};
}
DI f3 unpack_r11g11b10(uint32_t p) {
#if INCR_FAST_SMALL_FLOAT
    return {f16_pattern_to_f32((p & 0x7ffu) << 4), f16_pattern_to_f32(((p >> 11) & 0x7ffu) << 4), f16_pattern_to_f32((p >> 22) << 5)};
#else
    return unpack_r11g11b10_generic(p);
#endif
}

// ---------------------------------------------------------------- texture unit stand-in (bilinear, wrap, mip 0)
static constexpr uint32_t kTexJointMR = 0x100u;      // internal: not an incr_texture_desc format
DI f4 decode_texel(const float* __restrict__ srgbLut, uint32_t format, uint32_t rgba) {
    const float inv255 = 1.f / 255.f;
    const uint32_t r = rgba & 0xffu, g = (rgba >> 8) & 0xffu, b = (rgba >> 16) & 0xffu, a = rgba >> 24;
    switch (format) {
    case INCR_TEX_RGBA8_SRGB:       return {srgbLut[r], srgbLut[g], srgbLut[b], (float)a * inv255};
    case INCR_TEX_RG_SNORM:         return {2.f * ((float)r * inv255) - 1.f, 2.f * ((float)g * inv255) - 1.f, 0.f, 1.f};
    case INCR_TEX_BGRA8_SWIZZLE_RG: return {(float)b * inv255, (float)g * inv255, 0.f, 1.f};
    case kTexJointMR:               return {(float)b * inv255, (float)a * inv255, 0.f, 1.f};      // second word of MaterialTexels::chain: metal in byte 2, roughness in byte 3 (the same bytes, the same operations)
    // decoded BC4 / BC5 blocks (texture_mips.hip bc_decode_kernel): the palette entry's exact numerator over 35 x 255, or 35 x 127 signed,
    // times the float nearest the reciprocal (8925 and 4445 times it are exactly 1; a division here is ten instructions at every one of
    // the samplers' decode sites, executed or not)
    case INCR_TEX_RG16_OVER_8925:   return {(float)(rgba & 0xffffu) * (1.f / 8925.f), (float)(rgba >> 16) * (1.f / 8925.f), 0.f, 1.f};
    case INCR_TEX_RG16S_OVER_4445:  return {(float)(int16_t)(rgba & 0xffffu) * (1.f / 4445.f), (float)(int16_t)(rgba >> 16) * (1.f / 4445.f), 0.f, 1.f};
    default:                        return {(float)r * inv255, (float)g * inv255, (float)b * inv255, (float)a * inv255};
    }
}
// LINEAR SAMPLE ADDRESSING (round 5; D3D11.3 functional spec 7.18.8): the scaled texel coordinate is snapped to fixed point with 8 fractional bits (round to
// nearest even), half a texel is subtracted THERE, the integer part names the left / upper texel and the 8 fractional bits are the weight: w = f / 256 and
// 1 - w = (256 - f) / 256, both exact floats.  Rounds 1-4 used the float fraction x - floor(x): further from what a texture unit does, and a weight that the
// oracle's CPU and the device had to agree on to the last ulp.  Every linear filter of the library goes through this: 2-D textures (wrap and clamp), the
// lat-long sky, the cube maps, the LUTs.  linear_split: the same on a coordinate that was clamped in fixed point.
DI void linear_split(int xi, int* i0, float* w) { *i0 = xi >> 8; *w = (float)(xi & 255) * 0.00390625f; }
DI int linear_fixed(float scaled) { return __float2int_rn(scaled * 256.f) - 128; }
DI void linear_address(float scaled, int* i0, float* w) { linear_split(linear_fixed(scaled), i0, w); }
// Bilinear, wrap, texel centre + 0.5, 8-bit fixed-point weights (linear_address), on level `level` of a texture's mip chain (levels packed one after the
// other behind level 0; level l is max(W >> l, 1) x max(H >> l, 1)).
DI f4 SampleBilinearLevel(const DeviceTexture& t, const float* __restrict__ srgbLut, uint32_t level, f2 uv) {
    int W = (int)t.width, H = (int)t.height;
    size_t off = 0;
    for (uint32_t l = 0; l < level; ++l) { off += texel_level_words((uint32_t)W, (uint32_t)H); W = W > 1 ? W >> 1 : 1; H = H > 1 ? H >> 1 : 1; }
    float fu = uv.x - floorf(uv.x), fv = uv.y - floorf(uv.y);
    if (!(fu >= 0.f && fu <= 1.f)) fu = 0.f;
    if (!(fv >= 0.f && fv <= 1.f)) fv = 0.f;
    int x0, y0; float fx, fy;
    linear_address(fu * (float)W, &x0, &fx); linear_address(fv * (float)H, &y0, &fy);
    // wrap: fu, fv are in [0, 1], so the left texel is in [-1, W - 1] -- two compares instead of four integer divisions by a run-time W
    // (~40 VALU instructions each; this function runs 3-10 times per shaded hit / G-buffer pixel)
    x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
    const int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    const uint32_t* __restrict__ px = (const uint32_t*)t.texels + off;
    const uint32_t c00 = px[texel_tile_index((uint32_t)x0, (uint32_t)y0, (uint32_t)W)], c10 = px[texel_tile_index((uint32_t)x1, (uint32_t)y0, (uint32_t)W)];
    const uint32_t c01 = px[texel_tile_index((uint32_t)x0, (uint32_t)y1, (uint32_t)W)], c11 = px[texel_tile_index((uint32_t)x1, (uint32_t)y1, (uint32_t)W)];
    f4 t00 = decode_texel(srgbLut, t.format, c00), t10 = decode_texel(srgbLut, t.format, c10);
    f4 t01 = decode_texel(srgbLut, t.format, c01), t11 = decode_texel(srgbLut, t.format, c11);
    float wx0 = 1.f - fx, wy0 = 1.f - fy;
    f4 r;
    r.x = (t00.x * wx0 + t10.x * fx) * wy0 + (t01.x * wx0 + t11.x * fx) * fy;
    r.y = (t00.y * wx0 + t10.y * fx) * wy0 + (t01.y * wx0 + t11.y * fx) * fy;
    r.z = (t00.z * wx0 + t10.z * fx) * wy0 + (t01.z * wx0 + t11.z * fx) * fy;
    r.w = (t00.w * wx0 + t10.w * fx) * wy0 + (t01.w * wx0 + t11.w * fx) * fy;
    return r;
}
// SampleLevel(s, uv, 0): what the path tracer uses (material/common.hlsli:4-5, no UV derivatives there)
DI f4 SampleLevel0(const DeviceTexture* __restrict__ textures, const float* __restrict__ srgbLut, int texIndex, f2 uv) {
    return SampleBilinearLevel(textures[texIndex], srgbLut, 0u, uv);
}

// ---- Sample(s, uv) of the raster passes (material/common.hlsli:6-8 TEX_SAMPLE = Sample outside PATH_TRACING) ----
// Pinned texture-unit rules.  The reference's sampler is g_anisoSampler = FILTER_ANISOTROPIC with the root
// signature's default MaxAnisotropy 16 (gbuffer-compute.hlsl:20, visibility-pass.hlsl:11, gbuffer-raster.hlsl:11; d3dx12_root_signature.h:277).
// D3D leaves the footprint to the implementation; this one is the public textbook form (Vulkan 1.3 spec "Texel Anisotropic
// Filtering"): rho_max / rho_min = the longer / shorter screen-space derivative in texels, N = min(ceil(rho_max / rho_min), MaxAnisotropy)
// trilinear taps at uv + longer derivative * (i / (N + 1) - 1/2), i = 1 .. N, averaged, each at LOD = log2(rho_max / min(rho_max / rho_min,
// MaxAnisotropy)).  No square root is taken: N is the smallest n with n^2 rho_min^2 >= rho_max^2 and the LOD is half the logarithm of
// max(rho_min^2, rho_max^2 / MaxAnisotropy^2) -- the same float operations on host and device.  MaxAnisotropy 1 is the isotropic rule
// of D3D11.3 functional spec 7.18.11 (LOD of the longer derivative, one tap).  The logarithm: exponent + a degree-4 polynomial in
// the mantissa (|error| < 1.2e-4); LOD clamped to the chain, its fraction truncated to 8 bits, trilinear = lerp of two bilinear taps.
DI float pinned_log2(float x) {         // x > 0, finite, normal
    const uint32_t b = __float_as_uint(x);
    const float e = (float)((int)(b >> 23) - 127);
    const float t = __uint_as_float((b & 0x7fffffu) | 0x3f800000u) - 1.f;
    const float p = 1.4425331354141235f + t * (-0.7161056399345398f + t * (0.4395408034324646f + t * (-0.2250901162624359f + t * 0.05923847481608391f)));
    return e + t * p;
}
struct Footprint { float lod; uint32_t taps; f2 axis; };      // axis: the longer derivative, in uv units (the line the taps lie on)
DI Footprint texture_footprint(const DeviceTexture& t, f2 ddx, f2 ddy, uint32_t maxAniso) {
    const float W = (float)t.width, H = (float)t.height;
    const float ax = ddx.x * W, ay = ddx.y * H, bx = ddy.x * W, by = ddy.y * H;
    const float lx = ax * ax + ay * ay, ly = bx * bx + by * by;          // squared lengths of the two derivatives, in texels
    const bool xMajor = lx >= ly;
    const float major = xMajor ? lx : ly, minor = xMajor ? ly : lx;
    Footprint f; f.lod = 0.f; f.taps = 1u; f.axis = xMajor ? ddx : ddy;
    if (!(major >= 1.17549435e-38f) || !(major <= 3.0e38f)) return f;     // zero, denormal, NaN or infinite footprint: level 0, one tap
    uint32_t n = 1u;
    while (n < maxAniso && (float)(n * n) * minor < major) ++n;
    f.taps = n;
    const float m = fmaxf(minor, major / (float)(maxAniso * maxAniso));   // squared minor axis after the ratio clamp
    if (m >= 1.17549435e-38f) f.lod = 0.5f * pinned_log2(m);
    return f;
}
DI float tap_offset(uint32_t i, uint32_t n) { return (float)i / (float)(n + 1u) - 0.5f; }      // i = 1 .. n
// The two levels a trilinear tap reads, resolved ONCE per Sample (every tap of an anisotropic line reads the same two): word offset
// and size of level floor(lod) and of the next one; without a second level to blend with (fraction 0 or end of the chain) level 1
// repeats level 0, so that the taps' loads stay unconditional.
struct LevelPair { uint32_t off0, off1; int W0, H0, W1, H1; float frac; bool two; };
DI LevelPair level_pair(const DeviceTexture& t, float lod) {
    const uint32_t numMips = t.pad[1] ? t.pad[1] : 1u;
    lod = fminf(fmaxf(lod, 0.f), (float)(numMips - 1u));
    const float l0f = floorf(lod);
    LevelPair L;
    L.frac = floorf((lod - l0f) * 256.f) * (1.f / 256.f);
    const uint32_t l0 = (uint32_t)l0f;
    L.two = !(L.frac == 0.f || l0 + 1u >= numMips);
    int W = (int)t.width, H = (int)t.height;
    uint32_t off = 0;
    for (uint32_t l = 0; l < l0; ++l) { off += texel_level_words((uint32_t)W, (uint32_t)H); W = W > 1 ? W >> 1 : 1; H = H > 1 ? H >> 1 : 1; }
    L.off0 = off; L.W0 = W; L.H0 = H;
    L.off1 = L.two ? off + texel_level_words((uint32_t)W, (uint32_t)H) : off;
    L.W1 = L.two ? (W > 1 ? W >> 1 : 1) : W; L.H1 = L.two ? (H > 1 ? H >> 1 : 1) : H;
    return L;
}
struct BilinearTap { uint32_t i00, i10, i01, i11; float fx, fy; };
DI BilinearTap bilinear_tap(int W, int H, size_t off, f2 uv) {
    float fu = uv.x - floorf(uv.x), fv = uv.y - floorf(uv.y);
    if (!(fu >= 0.f && fu <= 1.f)) fu = 0.f;
    if (!(fv >= 0.f && fv <= 1.f)) fv = 0.f;
    BilinearTap t; int x0, y0;
    linear_address(fu * (float)W, &x0, &t.fx); linear_address(fv * (float)H, &y0, &t.fy);
    x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
    const int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    const uint32_t o = (uint32_t)off;
    t.i00 = o + texel_tile_index((uint32_t)x0, (uint32_t)y0, (uint32_t)W); t.i10 = o + texel_tile_index((uint32_t)x1, (uint32_t)y0, (uint32_t)W);
    t.i01 = o + texel_tile_index((uint32_t)x0, (uint32_t)y1, (uint32_t)W); t.i11 = o + texel_tile_index((uint32_t)x1, (uint32_t)y1, (uint32_t)W);
    return t;
}
DI f4 bilinear_blend(const BilinearTap& t, const float* __restrict__ srgbLut, uint32_t format, uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11) {
    const f4 t00 = decode_texel(srgbLut, format, c00), t10 = decode_texel(srgbLut, format, c10);
    const f4 t01 = decode_texel(srgbLut, format, c01), t11 = decode_texel(srgbLut, format, c11);
    const float wx0 = 1.f - t.fx, wy0 = 1.f - t.fy;
    f4 r;
    r.x = (t00.x * wx0 + t10.x * t.fx) * wy0 + (t01.x * wx0 + t11.x * t.fx) * t.fy;
    r.y = (t00.y * wx0 + t10.y * t.fx) * wy0 + (t01.y * wx0 + t11.y * t.fx) * t.fy;
    r.z = (t00.z * wx0 + t10.z * t.fx) * wy0 + (t01.z * wx0 + t11.z * t.fx) * t.fy;
    r.w = (t00.w * wx0 + t10.w * t.fx) * wy0 + (t01.w * wx0 + t11.w * t.fx) * t.fy;
    return r;
}
// One trilinear tap = SampleBilinearLevel on both levels + the lerp.  (Fetching both levels' texels together, and two taps per round
// with sixteen loads in flight, was measured: the G-buffer kernel went from 72 to 130 VGPRs and from 0.94 to 1.33 ms.)
DI f4 bilinear_at(const DeviceTexture& t, const float* __restrict__ srgbLut, int W, int H, uint32_t off, f2 uv) {
    const BilinearTap k = bilinear_tap(W, H, off, uv);
    const uint32_t* __restrict__ px = (const uint32_t*)t.texels;
    const uint32_t c00 = px[k.i00], c10 = px[k.i10], c01 = px[k.i01], c11 = px[k.i11];
    return bilinear_blend(k, srgbLut, t.format, c00, c10, c01, c11);
}
DI f4 trilinear_at(const DeviceTexture& t, const float* __restrict__ srgbLut, const LevelPair& L, f2 uv) {
    const f4 a = bilinear_at(t, srgbLut, L.W0, L.H0, L.off0, uv);
    if (!L.two) return a;
    const f4 b = bilinear_at(t, srgbLut, L.W1, L.H1, L.off1, uv);
    return {a.x + (b.x - a.x) * L.frac, a.y + (b.y - a.y) * L.frac, a.z + (b.z - a.z) * L.frac, a.w + (b.w - a.w) * L.frac};
}
DI f4 SampleGrad(const DeviceTexture* __restrict__ textures, const float* __restrict__ srgbLut, int texIndex, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso) {
    const DeviceTexture t = textures[texIndex];
    const Footprint fp = texture_footprint(t, ddx, ddy, maxAniso);
    const LevelPair L = level_pair(t, fp.lod);
    if (fp.taps == 1u) return trilinear_at(t, srgbLut, L, uv);
    f4 sum = {0.f, 0.f, 0.f, 0.f};                       // the line of taps, summed in tap order
    for (uint32_t i = 1u; i <= fp.taps; ++i) {
        const float o = tap_offset(i, fp.taps);
        const f4 c = trilinear_at(t, srgbLut, L, {uv.x + fp.axis.x * o, uv.y + fp.axis.y * o});
        sum.x += c.x; sum.y += c.y; sum.z += c.z; sum.w += c.w;
    }
    const float n = (float)fp.taps;
    return {sum.x / n, sum.y / n, sum.z / n, sum.w / n};
}

// SampleGrad(...).w for the alpha test of the visibility scan (raster.hip fragment), a per-lane serial loop whose cost is its dependent
// round trips: the same taps, the same arithmetic on the alpha channel alone (UNORM alpha: a / 255 for the formats that carry one,
// 1 otherwise), bit for bit.
DI bool format_has_alpha(uint32_t format) { return format == INCR_TEX_RGBA8_UNORM || format == INCR_TEX_RGBA8_SRGB; }
DI float texel_alpha(uint32_t rgba) { return (float)(rgba >> 24) * (1.f / 255.f); }          // of a format that carries alpha (decode_texel's .w)
DI float bilinear_alpha(const BilinearTap& t, uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11) {
    const float wx0 = 1.f - t.fx, wy0 = 1.f - t.fy;
    return (texel_alpha(c00) * wx0 + texel_alpha(c10) * t.fx) * wy0 + (texel_alpha(c01) * wx0 + texel_alpha(c11) * t.fx) * t.fy;
}
// one tap's alpha: the eight texels of both levels are fetched together (one memory round trip per tap in a per-lane serial loop)
DI float trilinear_alpha_at(const DeviceTexture& t, const LevelPair& L, f2 uv) {
    const BilinearTap ka = bilinear_tap(L.W0, L.H0, L.off0, uv), kb = bilinear_tap(L.W1, L.H1, L.off1, uv);
    const uint32_t* __restrict__ px = (const uint32_t*)t.texels;
    const uint32_t a00 = px[ka.i00], a10 = px[ka.i10], a01 = px[ka.i01], a11 = px[ka.i11];
    const uint32_t b00 = px[kb.i00], b10 = px[kb.i10], b01 = px[kb.i01], b11 = px[kb.i11];
    const float a = bilinear_alpha(ka, a00, a10, a01, a11);
    if (!L.two) return a;
    const float b = bilinear_alpha(kb, b00, b10, b01, b11);
    return a + (b - a) * L.frac;
}
// The same tap out of the texture's ALPHA QUADS (texture_mips.hip alpha_quads_kernel; DeviceTexture::pad[2] = 1 and texels -> the quad chain): one word per level holds
// the four alpha bytes of the footprint, so a tap is two loads, and only the footprint's first index is computed.  Same bytes, same operations.
DI float quad_alpha(uint32_t q, int k) { return (float)((q >> (8 * k)) & 0xffu) * (1.f / 255.f); }
DI float bilinear_alpha_quad(const BilinearTap& t, uint32_t q) {
    const float wx0 = 1.f - t.fx, wy0 = 1.f - t.fy;
    return (quad_alpha(q, 0) * wx0 + quad_alpha(q, 1) * t.fx) * wy0 + (quad_alpha(q, 2) * wx0 + quad_alpha(q, 3) * t.fx) * t.fy;
}
DI float trilinear_alpha_quads_at(const DeviceTexture& t, const LevelPair& L, f2 uv) {
    const BilinearTap ka = bilinear_tap(L.W0, L.H0, L.off0, uv), kb = bilinear_tap(L.W1, L.H1, L.off1, uv);
    const uint32_t* __restrict__ px = (const uint32_t*)t.texels;
    const uint32_t qa = px[ka.i00], qb = px[kb.i00];
    const float a = bilinear_alpha_quad(ka, qa);
    if (!L.two) return a;
    const float b = bilinear_alpha_quad(kb, qb);
    return a + (b - a) * L.frac;
}
// clip(Sample(..).w - 0.5): true when the fragment survives.  Along an anisotropic line the decision is often known before the last
// tap and the loop stops there -- without changing it: every tap's alpha is >= 0 (convex weights of non-negative texels, and
// a + (b - a) frac >= 0 for frac < 1), so a partial sum >= n / 2 (exact in float for n <= 16) keeps fl(sum / n) >= 0.5 by the
// monotonicity of the rounded sum and division; and no tap exceeds 1 by more than a few ulp, so a partial sum that cannot reach
// 0.49999 n even if every remaining tap read 1.00001 ends below 0.5.
#ifndef INCR_ALPHA_TAP_BATCH
#define INCR_ALPHA_TAP_BATCH 2
#endif
DI bool AlphaTestGrad(const DeviceTexture& t, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso) {
    if (!format_has_alpha(t.format)) return true;          // every texel decodes to alpha 1: any filter of ones stays above 0.5
    const Footprint fp = texture_footprint(t, ddx, ddy, maxAniso);
    const LevelPair L = level_pair(t, fp.lod);
    const bool quads = t.pad[2] != 0u;          // wave-divergent at worst per triangle: the descriptor travels with it
    auto tap = [&](f2 st) { return quads ? trilinear_alpha_quads_at(t, L, st) : trilinear_alpha_at(t, L, st); };
    if (fp.taps == 1u) return !(tap(uv) - 0.5f < 0.f);
    const float n = (float)fp.taps;
    float sum = 0.f;
#if INCR_ALPHA_TAP_BATCH > 1
    if (quads) {
        // FOUR TAPS PER ROUND TRIP (round 5).  The loop below is a chain of dependent memory round trips -- a tap's texels, then the decision whether the next tap is needed -- and
        // the visibility scan is made of such chains: a band item of foliage is 32 tiles x up to 16 round trips of ~1 us on ONE wave while the rest of the grid has
        // finished (PMC, round 4: the average wave lives a third of the launch).  Here the quad words of up to four taps are fetched together and then summed in the
        // same order with the same two exits after every tap: the same decision, a quarter of the round trips.  (Taps beyond the line's count re-read the last
        // one's address and are not summed.)
        const uint32_t* __restrict__ px = (const uint32_t*)t.texels;
        for (uint32_t i = 1u; i <= fp.taps; i += (uint32_t)INCR_ALPHA_TAP_BATCH) {
            BilinearTap ka[INCR_ALPHA_TAP_BATCH], kb[INCR_ALPHA_TAP_BATCH]; uint32_t qa[INCR_ALPHA_TAP_BATCH], qb[INCR_ALPHA_TAP_BATCH];
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)INCR_ALPHA_TAP_BATCH; ++k) {
                const float o = tap_offset(min(i + k, fp.taps), fp.taps);
                const f2 st = {uv.x + fp.axis.x * o, uv.y + fp.axis.y * o};
                ka[k] = bilinear_tap(L.W0, L.H0, L.off0, st); kb[k] = bilinear_tap(L.W1, L.H1, L.off1, st);
            }
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)INCR_ALPHA_TAP_BATCH; ++k) { qa[k] = px[ka[k].i00]; qb[k] = px[kb[k].i00]; }
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)INCR_ALPHA_TAP_BATCH; ++k) {
                if (i + k > fp.taps) break;
                const float a = bilinear_alpha_quad(ka[k], qa[k]);
                sum += L.two ? a + (bilinear_alpha_quad(kb[k], qb[k]) - a) * L.frac : a;
                if (sum >= 0.5f * n) return true;
                if (sum + (float)(fp.taps - (i + k)) * 1.00001f < 0.49999f * n) return false;
            }
        }
        return !(sum / n - 0.5f < 0.f);
    }
#endif
    for (uint32_t i = 1u; i <= fp.taps; ++i) {
        const float o = tap_offset(i, fp.taps);
        sum += tap({uv.x + fp.axis.x * o, uv.y + fp.axis.y * o});
        if (sum >= 0.5f * n) return true;
        if (sum + (float)(fp.taps - i) * 1.00001f < 0.49999f * n) return false;
    }
    return !(sum / n - 0.5f < 0.f);
}

// material/common.hlsli:10-115
struct MatProps { f3 emissive, basecolor; float opacity; bool bHasNormalmap; f3 normalmap; float metallic, roughness, ao, aoblend; };
template <class Sampler> DI MatProps EvaluateMaterialPropertiesT(const incr_material& mat, f2 uv, Sampler sample) {
    MatProps o;
    o.emissive = {mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]};
    if (mat.emissiveTextureIndex != -1) { f4 t = sample(mat.emissiveTextureIndex, uv); o.emissive = o.emissive * mk3(t.x, t.y, t.z); }
    o.basecolor = {mat.baseColorFactor[0], mat.baseColorFactor[1], mat.baseColorFactor[2]};
    o.opacity = 1.f;
    if (mat.baseColorTextureIndex != -1) {
        f4 t = sample(mat.baseColorTextureIndex, uv);
        o.basecolor = o.basecolor * mk3(t.x, t.y, t.z);
        o.opacity = t.w;
    }
    o.bHasNormalmap = mat.normalTextureIndex != -1;
    o.normalmap = {0.f, 0.f, 1.f};
    if (o.bHasNormalmap) {
        f4 t = sample(mat.normalTextureIndex, uv);
        f2 nxy = {t.x, t.y};
        float nz = sqrtf(1.f - dot(nxy, nxy));
        o.normalmap = {nxy.x, nxy.y, nz};
    }
    o.metallic = mat.metallicFactor;
    o.roughness = mat.roughnessFactor;
    if (mat.metallicRoughnessTextureIndex != -1) {
        f4 t = sample(mat.metallicRoughnessTextureIndex, uv);
        o.metallic = t.x; o.roughness = t.y;
    }
    o.aoblend = mat.aoStrength;
    o.ao = 1.f;
    if (mat.aoTextureIndex != -1) o.ao = sample(mat.aoTextureIndex, uv).x;
    return o;
}

// PATH_TRACING: SampleLevel(s, uv, 0).  Same arithmetic as EvaluateMaterialPropertiesT with the level-0 sampler, but the memory
// operations of the three textures nearly every material binds (base colour, normal, metallic-roughness) are issued in three
// rounds -- all descriptors, then all twelve texels, then the decode -- instead of texture by texture: the shade kernel runs at
// 3 waves per SIMD and spent two thirds of its cycles waiting, with six DEPENDENT round trips per hit (descriptor -> texels,
// three times); now there are two.  Emissive and AO textures (rare) keep the plain path.
struct Level0Tap { const uint32_t* __restrict__ px; uint32_t format; int W, i00, i10, i01, i11; float fx, fy; };
DI Level0Tap level0_tap(const DeviceTexture& t, f2 uv) {
    Level0Tap a;
    const int W = (int)t.width, H = (int)t.height;
    float fu = uv.x - floorf(uv.x), fv = uv.y - floorf(uv.y);
    if (!(fu >= 0.f && fu <= 1.f)) fu = 0.f;
    if (!(fv >= 0.f && fv <= 1.f)) fv = 0.f;
    int x0, y0;                                            // wrap without integer division, as in SampleBilinearLevel
    linear_address(fu * (float)W, &x0, &a.fx); linear_address(fv * (float)H, &y0, &a.fy);
    x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
    const int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    a.px = (const uint32_t*)t.texels; a.format = t.format; a.W = W;
    a.i00 = (int)texel_tile_index((uint32_t)x0, (uint32_t)y0, (uint32_t)W); a.i10 = (int)texel_tile_index((uint32_t)x1, (uint32_t)y0, (uint32_t)W);
    a.i01 = (int)texel_tile_index((uint32_t)x0, (uint32_t)y1, (uint32_t)W); a.i11 = (int)texel_tile_index((uint32_t)x1, (uint32_t)y1, (uint32_t)W);
    return a;
}
DI f4 level0_finish(const Level0Tap& a, const float* __restrict__ srgbLut, uint32_t c00, uint32_t c10, uint32_t c01, uint32_t c11) {
    const f4 t00 = decode_texel(srgbLut, a.format, c00), t10 = decode_texel(srgbLut, a.format, c10);
    const f4 t01 = decode_texel(srgbLut, a.format, c01), t11 = decode_texel(srgbLut, a.format, c11);
    const float fx = a.fx, fy = a.fy, wx0 = 1.f - fx, wy0 = 1.f - fy;
    f4 r;
    r.x = (t00.x * wx0 + t10.x * fx) * wy0 + (t01.x * wx0 + t11.x * fx) * fy;
    r.y = (t00.y * wx0 + t10.y * fx) * wy0 + (t01.y * wx0 + t11.y * fx) * fy;
    r.z = (t00.z * wx0 + t10.z * fx) * wy0 + (t01.z * wx0 + t11.z * fx) * fy;
    r.w = (t00.w * wx0 + t10.w * fx) * wy0 + (t01.w * wx0 + t11.w * fx) * fy;
    return r;
}
// The interleaved form (device_types.h MaterialTexels): ONE tap computation and four 8-byte loads for the base colour, the normal and the
// metallic-roughness sample of a hit.  Texel bytes, decode and filter arithmetic are those of level0_tap / level0_finish channel by
// channel, so the result equals the texture-by-texture path bit for bit (tests/test_gpu_round3.py).
DI MatProps EvaluateMaterialPropertiesInterleaved(const incr_material& mat, const MaterialTexels mt, const DeviceTexture* __restrict__ textures, const float* __restrict__ lut, f2 uv) {
    const int W = (int)mt.width, H = (int)mt.height;
    float fu = uv.x - floorf(uv.x), fv = uv.y - floorf(uv.y);
    if (!(fu >= 0.f && fu <= 1.f)) fu = 0.f;
    if (!(fv >= 0.f && fv <= 1.f)) fv = 0.f;
    int x0, y0; float fx, fy;
    linear_address(fu * (float)W, &x0, &fx); linear_address(fv * (float)H, &y0, &fy);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    x0 = x0 < 0 ? x0 + W : x0; y0 = y0 < 0 ? y0 + H : y0;
    const int x1 = x0 + 1 == W ? 0 : x0 + 1, y1 = y0 + 1 == H ? 0 : y0 + 1;
    const uint2* __restrict__ px = (const uint2*)mt.texels;
    const uint2 c00 = px[material_texel_index((uint32_t)x0, (uint32_t)y0, (uint32_t)W)], c10 = px[material_texel_index((uint32_t)x1, (uint32_t)y0, (uint32_t)W)];
    const uint2 c01 = px[material_texel_index((uint32_t)x0, (uint32_t)y1, (uint32_t)W)], c11 = px[material_texel_index((uint32_t)x1, (uint32_t)y1, (uint32_t)W)];
    auto lerp4 = [&](float a, float b, float c, float d) { return (a * wx0 + b * fx) * wy0 + (c * wx0 + d * fx) * fy; };
    MatProps o;
    o.emissive = {mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]};
    if (mat.emissiveTextureIndex != -1) { f4 e = SampleLevel0(textures, lut, mat.emissiveTextureIndex, uv); o.emissive = o.emissive * mk3(e.x, e.y, e.z); }
    {
        const f4 t00 = decode_texel(lut, mt.baseFormat, c00.x), t10 = decode_texel(lut, mt.baseFormat, c10.x), t01 = decode_texel(lut, mt.baseFormat, c01.x), t11 = decode_texel(lut, mt.baseFormat, c11.x);
        const f3 b = {lerp4(t00.x, t10.x, t01.x, t11.x), lerp4(t00.y, t10.y, t01.y, t11.y), lerp4(t00.z, t10.z, t01.z, t11.z)};
        o.basecolor = mk3(mat.baseColorFactor[0], mat.baseColorFactor[1], mat.baseColorFactor[2]) * b;
        o.opacity = lerp4(t00.w, t10.w, t01.w, t11.w);
    }
    const float inv255 = 1.f / 255.f;
    {   // INCR_TEX_RG_SNORM: 2 * c / 255 - 1 on the R and G bytes (decode_texel)
        auto sn = [&](uint32_t w, int k) { return 2.f * ((float)((w >> (8 * k)) & 0xffu) * inv255) - 1.f; };
        const f2 nxy = {lerp4(sn(c00.y, 0), sn(c10.y, 0), sn(c01.y, 0), sn(c11.y, 0)), lerp4(sn(c00.y, 1), sn(c10.y, 1), sn(c01.y, 1), sn(c11.y, 1))};
        o.bHasNormalmap = true;
        o.normalmap = {nxy.x, nxy.y, sqrtf(1.f - dot(nxy, nxy))};
    }
    {   // INCR_TEX_BGRA8_SWIZZLE_RG: metal = B / 255, roughness = G / 255 (stored as bytes 2 and 3 of the second word)
        auto un = [&](uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu) * inv255; };
        o.metallic = lerp4(un(c00.y, 2), un(c10.y, 2), un(c01.y, 2), un(c11.y, 2));
        o.roughness = lerp4(un(c00.y, 3), un(c10.y, 3), un(c01.y, 3), un(c11.y, 3));
    }
    o.aoblend = mat.aoStrength;
    o.ao = 1.f;
    if (mat.aoTextureIndex != -1) o.ao = SampleLevel0(textures, lut, mat.aoTextureIndex, uv).x;
    return o;
}
DI MatProps EvaluateMaterialProperties(const incr_material& mat, const DeviceTexture* __restrict__ textures, const float* __restrict__ lut, f2 uv) {
    const int idx[3] = {mat.baseColorTextureIndex, mat.normalTextureIndex, mat.metallicRoughnessTextureIndex};
    DeviceTexture t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { t[k] = DeviceTexture{nullptr, 1u, 1u, 0u, {0u, 0u, 0u}}; if (idx[k] != -1) t[k] = textures[idx[k]]; }      // round 1: descriptors
    Level0Tap a[3]; uint32_t c[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                                                                                              // round 2: texels
        a[k] = level0_tap(t[k], uv);
        c[k][0] = c[k][1] = c[k][2] = c[k][3] = 0u;
        if (idx[k] != -1) { c[k][0] = a[k].px[a[k].i00]; c[k][1] = a[k].px[a[k].i10]; c[k][2] = a[k].px[a[k].i01]; c[k][3] = a[k].px[a[k].i11]; }
    }
    MatProps o;                                                                                                                                // round 3: material/common.hlsli:10-115
    o.emissive = {mat.emissiveFactor[0], mat.emissiveFactor[1], mat.emissiveFactor[2]};
    if (mat.emissiveTextureIndex != -1) { f4 e = SampleLevel0(textures, lut, mat.emissiveTextureIndex, uv); o.emissive = o.emissive * mk3(e.x, e.y, e.z); }
    o.basecolor = {mat.baseColorFactor[0], mat.baseColorFactor[1], mat.baseColorFactor[2]};
    o.opacity = 1.f;
    if (idx[0] != -1) {
        const f4 b = level0_finish(a[0], lut, c[0][0], c[0][1], c[0][2], c[0][3]);
        o.basecolor = o.basecolor * mk3(b.x, b.y, b.z);
        o.opacity = b.w;
    }
    o.bHasNormalmap = idx[1] != -1;
    o.normalmap = {0.f, 0.f, 1.f};
    if (o.bHasNormalmap) {
        const f4 n = level0_finish(a[1], lut, c[1][0], c[1][1], c[1][2], c[1][3]);
        const f2 nxy = {n.x, n.y};
        const float nz = sqrtf(1.f - dot(nxy, nxy));
        o.normalmap = {nxy.x, nxy.y, nz};
    }
    o.metallic = mat.metallicFactor;
    o.roughness = mat.roughnessFactor;
    if (idx[2] != -1) {
        const f4 m = level0_finish(a[2], lut, c[2][0], c[2][1], c[2][2], c[2][3]);
        o.metallic = m.x; o.roughness = m.y;
    }
    o.aoblend = mat.aoStrength;
    o.ao = 1.f;
    if (mat.aoTextureIndex != -1) o.ao = SampleLevel0(textures, lut, mat.aoTextureIndex, uv).x;
    return o;
}
// Sample(g_anisoSampler, uv) of a material's textures.  Base colour, normal and metallic-roughness maps nearly always share one size:
// their footprints, levels, tap positions, texel indices and bilinear weights are then the same numbers, computed once per tap for
// the three (the G-buffer kernel is VALU-bound on exactly this address arithmetic: 2 x ~60 instructions per trilinear tap and texture,
// up to 16 taps); each texture still sees SampleGrad's arithmetic operation for operation.
struct JointSamples { f4 base, normal, mr; };
// FB / FN / FM: the three formats as compile-time constants (the usual sRGB base colour + SNORM normals + swizzled metallic-roughness), or
// -1 = read the descriptor's: the per-texel format switch of decode_texel then folds away, and with it the channels nobody reads
template <int F> DI f4 bilinear_from(const DeviceTexture& t, const float* __restrict__ lut, const BilinearTap& k) {
    const uint32_t* __restrict__ px = (const uint32_t*)t.texels;
    const uint32_t c00 = px[k.i00], c10 = px[k.i10], c01 = px[k.i01], c11 = px[k.i11];
    return bilinear_blend(k, lut, F < 0 ? t.format : (uint32_t)F, c00, c10, c01, c11);
}
DI f4 lerp_levels(f4 a, f4 b, float frac) { return {a.x + (b.x - a.x) * frac, a.y + (b.y - a.y) * frac, a.z + (b.z - a.z) * frac, a.w + (b.w - a.w) * frac}; }
template <int FB, int FN, int FM>
DI JointSamples SampleGradJointT(const DeviceTexture& tb, const DeviceTexture& tn, const DeviceTexture& tm, const float* __restrict__ lut, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso) {
    const Footprint fp = texture_footprint(tb, ddx, ddy, maxAniso);
    const LevelPair L = level_pair(tb, fp.lod);
    JointSamples s; s.base = {0.f, 0.f, 0.f, 0.f}; s.normal = s.base; s.mr = s.base;
    for (uint32_t i = 1u; i <= fp.taps; ++i) {
        const float o = fp.taps == 1u ? 0.f : tap_offset(i, fp.taps);
        f2 st = uv;
        if (fp.taps != 1u) st = {uv.x + fp.axis.x * o, uv.y + fp.axis.y * o};
        const BilinearTap ka = bilinear_tap(L.W0, L.H0, L.off0, st);
        f4 cb = bilinear_from<FB>(tb, lut, ka), cn = bilinear_from<FN>(tn, lut, ka), cm = bilinear_from<FM>(tm, lut, ka);
        if (L.two) {
            const BilinearTap kb = bilinear_tap(L.W1, L.H1, L.off1, st);
            cb = lerp_levels(cb, bilinear_from<FB>(tb, lut, kb), L.frac); cn = lerp_levels(cn, bilinear_from<FN>(tn, lut, kb), L.frac); cm = lerp_levels(cm, bilinear_from<FM>(tm, lut, kb), L.frac);
        }
        if (fp.taps == 1u) { s.base = cb; s.normal = cn; s.mr = cm; return s; }
        s.base.x += cb.x; s.base.y += cb.y; s.base.z += cb.z; s.base.w += cb.w;
        s.normal.x += cn.x; s.normal.y += cn.y; s.normal.z += cn.z; s.normal.w += cn.w;
        s.mr.x += cm.x; s.mr.y += cm.y; s.mr.z += cm.z; s.mr.w += cm.w;
    }
    const float n = (float)fp.taps;
    s.base = {s.base.x / n, s.base.y / n, s.base.z / n, s.base.w / n};
    s.normal = {s.normal.x / n, s.normal.y / n, s.normal.z / n, s.normal.w / n};
    s.mr = {s.mr.x / n, s.mr.y / n, s.mr.z / n, s.mr.w / n};
    return s;
}
// The same Sample out of the material's INTERLEAVED chain (MaterialTexels::chain; sRGB base colour + SNORM normals + swizzled metallic-roughness of one size and level count):
// a tap's four texels of the three maps are four 8-byte loads; decode, weights and summation order are SampleGradJointT's operation for operation.
DI JointSamples SampleGradJointChain(const uint2* __restrict__ chain, const DeviceTexture& tb, const float* __restrict__ lut, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso) {
    const Footprint fp = texture_footprint(tb, ddx, ddy, maxAniso);
    const LevelPair L = level_pair(tb, fp.lod);
    JointSamples s; s.base = {0.f, 0.f, 0.f, 0.f}; s.normal = s.base; s.mr = s.base;
    auto tap = [&](const BilinearTap& k, f4& cb, f4& cn, f4& cm) {
        const uint2 c00 = chain[k.i00], c10 = chain[k.i10], c01 = chain[k.i01], c11 = chain[k.i11];
        cb = bilinear_blend(k, lut, (uint32_t)INCR_TEX_RGBA8_SRGB, c00.x, c10.x, c01.x, c11.x);
        cn = bilinear_blend(k, lut, (uint32_t)INCR_TEX_RG_SNORM, c00.y, c10.y, c01.y, c11.y);
        cm = bilinear_blend(k, lut, kTexJointMR, c00.y, c10.y, c01.y, c11.y);
    };
    for (uint32_t i = 1u; i <= fp.taps; ++i) {
        const float o = fp.taps == 1u ? 0.f : tap_offset(i, fp.taps);
        f2 st = uv;
        if (fp.taps != 1u) st = {uv.x + fp.axis.x * o, uv.y + fp.axis.y * o};
        const BilinearTap ka = bilinear_tap(L.W0, L.H0, L.off0, st);
        f4 cb, cn, cm; tap(ka, cb, cn, cm);
        if (L.two) {
            const BilinearTap kb = bilinear_tap(L.W1, L.H1, L.off1, st);
            f4 db, dn, dm; tap(kb, db, dn, dm);
            cb = lerp_levels(cb, db, L.frac); cn = lerp_levels(cn, dn, L.frac); cm = lerp_levels(cm, dm, L.frac);
        }
        if (fp.taps == 1u) { s.base = cb; s.normal = cn; s.mr = cm; return s; }
        s.base.x += cb.x; s.base.y += cb.y; s.base.z += cb.z; s.base.w += cb.w;
        s.normal.x += cn.x; s.normal.y += cn.y; s.normal.z += cn.z; s.normal.w += cn.w;
        s.mr.x += cm.x; s.mr.y += cm.y; s.mr.z += cm.z; s.mr.w += cm.w;
    }
    const float n = (float)fp.taps;
    s.base = {s.base.x / n, s.base.y / n, s.base.z / n, s.base.w / n};
    s.normal = {s.normal.x / n, s.normal.y / n, s.normal.z / n, s.normal.w / n};
    s.mr = {s.mr.x / n, s.mr.y / n, s.mr.z / n, s.mr.w / n};
    return s;
}
DI JointSamples SampleGradJoint(const DeviceTexture& tb, const DeviceTexture& tn, const DeviceTexture& tm, const float* __restrict__ lut, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso) {
    if (tb.format == INCR_TEX_RGBA8_SRGB && tn.format == INCR_TEX_RG_SNORM && tm.format == INCR_TEX_BGRA8_SWIZZLE_RG)
        return SampleGradJointT<INCR_TEX_RGBA8_SRGB, INCR_TEX_RG_SNORM, INCR_TEX_BGRA8_SWIZZLE_RG>(tb, tn, tm, lut, uv, ddx, ddy, maxAniso);
    if (tb.format == INCR_TEX_RGBA8_SRGB && tn.format == INCR_TEX_RG16S_OVER_4445 && tm.format == INCR_TEX_RG16_OVER_8925)          // the same three slots handed over block-compressed
        return SampleGradJointT<INCR_TEX_RGBA8_SRGB, INCR_TEX_RG16S_OVER_4445, INCR_TEX_RG16_OVER_8925>(tb, tn, tm, lut, uv, ddx, ddy, maxAniso);
    return SampleGradJointT<-1, -1, -1>(tb, tn, tm, lut, uv, ddx, ddy, maxAniso);
}
DI MatProps EvaluateMaterialPropertiesGrad(const incr_material& mat, const DeviceTexture* __restrict__ textures, const float* __restrict__ lut, f2 uv, f2 ddx, f2 ddy, uint32_t maxAniso,
                                           const void* chain = nullptr) {   // raster: Sample; chain = the material's MaterialTexels::chain where the scene built one
    const int ib = mat.baseColorTextureIndex, in = mat.normalTextureIndex, im = mat.metallicRoughnessTextureIndex;
    if (chain) {          // implies the three maps, one size, one level count, the usual formats (incr_scene_create)
        const DeviceTexture tb = textures[ib];
        const JointSamples js = SampleGradJointChain((const uint2*)chain, tb, lut, uv, ddx, ddy, maxAniso);
        return EvaluateMaterialPropertiesT(mat, uv, [&](int tex, f2 st) {
            return tex == ib ? js.base : (tex == in ? js.normal : (tex == im ? js.mr : SampleGrad(textures, lut, tex, st, ddx, ddy, maxAniso)));
        });
    }
    if (ib != -1 && in != -1 && im != -1) {
        const DeviceTexture tb = textures[ib], tn = textures[in], tm = textures[im];
        if (tb.width == tn.width && tb.width == tm.width && tb.height == tn.height && tb.height == tm.height && tb.pad[1] == tn.pad[1] && tb.pad[1] == tm.pad[1]) {
            const JointSamples js = SampleGradJoint(tb, tn, tm, lut, uv, ddx, ddy, maxAniso);
            return EvaluateMaterialPropertiesT(mat, uv, [&](int tex, f2 st) {
                return tex == ib ? js.base : (tex == in ? js.normal : (tex == im ? js.mr : SampleGrad(textures, lut, tex, st, ddx, ddy, maxAniso)));
            });
        }
    }
    return EvaluateMaterialPropertiesT(mat, uv, [&](int tex, f2 st) { return SampleGrad(textures, lut, tex, st, ddx, ddy, maxAniso); });
}

// exp(x) for x <= 0 with a fixed evaluation (HLSL leaves exp's precision open; the vMF prefilter's exp feeds UNORM8 texels that
// oracle and product must agree on bit for bit): 2^(x log2 e) = 2^floor * P(frac), P of degree 6 (|error| < 1.7e-7 on [0, 1))
DI float pinned_exp_neg(float x) {
    if (!(x > -87.f)) return 0.f;                        // underflow, -inf, NaN
    const float t = x * 1.44269502162933349609375f;
    const float fi = floorf(t), f = t - fi;
    const float p = 1.0f + f * (0.693146824836731f + f * (0.24023109674453735f + f * (0.055478911846876144f + f * (0.009686185978353024f
                  + f * (0.001238241558894515f + f * 0.000218712622881867f)))));
    return __uint_as_float(__float_as_uint(p) + (uint32_t)((int)fi << 23));
}

DI uint32_t tonemap_texel(const float* __restrict__ srgbSteps, uint32_t hdrBits, float exposure, int viewmode) {
    const f3 h = unpack_r11g11b10(hdrBits);
    f3 l;
    if (viewmode != 0 && viewmode != 1) l = h; else l = ACESFilm(h * exposure);
    return linear_to_srgb8(srgbSteps, l.x) | (linear_to_srgb8(srgbSteps, l.y) << 8) | (linear_to_srgb8(srgbSteps, l.z) << 16) | 0xff000000u;
}

} // namespace incr
