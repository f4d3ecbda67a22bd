// TEST INFRASTRUCTURE ONLY -- CPU oracle for the Incarnation render hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Scalar float32 helpers with a pinned evaluation order.  Compiled with
// -ffp-contract=off so that no multiply-add is fused: +,-,*,/ and sqrt are the
// IEEE correctly-rounded operations, the same ones the HIP kernels use.
//
// PARITY UNPINNED: the reference (Windows / D3D12 / HLSL / DXR) cannot run here and
// ships no tests or golden vectors; this oracle is a
// restatement of the HLSL, cited function by function.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static inline float3 f3(float x, float y, float z) { return {x, y, z}; }
static inline float3 f3(float s) { return {s, s, s}; }
static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// DIVISION in shader arithmetic is pinned as a * rcp(b) with rcp the correctly rounded reciprocal (round 4): HLSL's a / b is DXIL FDiv, which
// GPUs lower to a reciprocal and a multiplication -- never to an IEEE division -- and a vector divided by a scalar to ONE reciprocal.  Sites:
// the float3 operators below, SamplePoint, ConcentricSampleDisk, GGX, SampleGGX, G_SmithGGXCorrelated, G_SchlickGGX, the light attenuation
// terms and the BRDF-sampling pdf (oracle_pt.cpp).  Everything else (geometry, texture addressing, the raster set-up) keeps the IEEE quotient.
static inline float rcp(float x) { return 1.f / x; }
static inline float qdiv(float a, float b) { return a * rcp(b); }
static inline float3 operator/(float3 a, float3 b) { return {qdiv(a.x, b.x), qdiv(a.y, b.y), qdiv(a.z, b.z)}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator/(float3 a, float s) { const float r = rcp(s); return {a.x * r, a.y * r, a.z * r}; }
static inline float3 operator+(float3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline float3 operator+(float s, float3 a) { return {s + a.x, s + a.y, s + a.z}; }
static inline float3 operator-(float s, float3 a) { return {s - a.x, s - a.y, s - a.z}; }
static inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
static inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
static inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }

static inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
static inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
static inline float2 operator*(float s, float2 a) { return {s * a.x, s * a.y}; }
static inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline float4 operator-(float4 a, float4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline float4 operator*(float s, float4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }

// dot: ((x*x' + y*y') + z*z')  -- pinned order
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
static inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// HLSL normalize(v).  DXC lowers the intrinsic to dot + Rsqrt + FMul (x * rsqrt(dot(x, x))); D3D leaves rsqrt's last bit to the GPU, so
// it is pinned to the correctly rounded sqrt and reciprocal: v * (1 / sqrt(dot(v, v))).  (Rounds 1-2 pinned v / sqrt(dot(v, v)): three
// divisions where the shader has three multiplies.)
static inline float3 normalize(float3 v) { const float r = 1.f / sqrtf(dot(v, v)); return {v.x * r, v.y * r, v.z * r}; }
// HLSL max/min return the non-NaN operand; saturate(NaN) = 0.
static inline float hmax(float a, float b) { return fmaxf(a, b); }
static inline float hmin(float a, float b) { return fminf(a, b); }
static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
// LINEAR SAMPLE ADDRESSING (D3D11.3 functional spec 7.18.8; pinned in round 5): the scaled texel coordinate snapped to fixed point with 8
// fractional bits (round to nearest even: lrintf under the default rounding mode), half a texel subtracted there; integer part = the left / upper texel, the
// 8 fractional bits = the weight f / 256 (and 1 - w = (256 - f) / 256: both exact).  Every linear filter of the oracle goes through it.
static inline void linear_split(int xi, int* i0, float* w) { *i0 = xi >> 8; *w = (float)(xi & 255) * 0.00390625f; }
static inline int linear_fixed(float scaled) { return (int)lrintf(scaled * 256.f) - 128; }
static inline void linear_address(float scaled, int* i0, float* w) { linear_split(linear_fixed(scaled), i0, w); }
static inline float3 saturate(float3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
// HLSL reflect(i, n) = i - 2 * n * dot(i, n)
static inline float3 reflect(float3 i, float3 n) { float d = dot(i, n); return i - (2.f * n) * d; }
// pow(x, 5.0) and pow(x, 4): pinned as repeated multiplication (DESIGN.md "pinned fixed-function choices")
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float pow4(float x) { float x2 = x * x; return x2 * x2; }

// ---------------------------------------------------------------- matrices (row-major, row vectors)
struct Mat4 { float m[16]; };
static inline Mat4 mat_identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
// C = A * B, c_ij = ((a_i0*b_0j + a_i1*b_1j) + a_i2*b_2j) + a_i3*b_3j
static inline Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i * 4 + j] = ((a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j]) +
                              a.m[i * 4 + 2] * b.m[2 * 4 + j]) + a.m[i * 4 + 3] * b.m[3 * 4 + j];
    return r;
}
// mul(float4(v, w), M).xyz
static inline float3 mul_point(float3 v, const Mat4& M) {
    return {((v.x * M.m[0] + v.y * M.m[4]) + v.z * M.m[8]) + M.m[12],
            ((v.x * M.m[1] + v.y * M.m[5]) + v.z * M.m[9]) + M.m[13],
            ((v.x * M.m[2] + v.y * M.m[6]) + v.z * M.m[10]) + M.m[14]};
}
static inline float3 mul_dir(float3 v, const Mat4& M) {
    return {(v.x * M.m[0] + v.y * M.m[4]) + v.z * M.m[8],
            (v.x * M.m[1] + v.y * M.m[5]) + v.z * M.m[9],
            (v.x * M.m[2] + v.y * M.m[6]) + v.z * M.m[10]};
}
static inline float4 mul_vec4(float4 v, const Mat4& M) {
    float4 r;
    r.x = ((v.x * M.m[0] + v.y * M.m[4]) + v.z * M.m[8]) + v.w * M.m[12];
    r.y = ((v.x * M.m[1] + v.y * M.m[5]) + v.z * M.m[9]) + v.w * M.m[13];
    r.z = ((v.x * M.m[2] + v.y * M.m[6]) + v.z * M.m[10]) + v.w * M.m[14];
    r.w = ((v.x * M.m[3] + v.y * M.m[7]) + v.z * M.m[11]) + v.w * M.m[15];
    return r;
}
// General 4x4 inverse by cofactors, evaluated in double and rounded once to float.
// (SimpleMath Matrix::Invert -> XMMatrixInverse is float cofactor expansion; the rounding of a
//  closed-source-order float expansion cannot be reproduced bit for bit, so double + one rounding is pinned.)
static inline Mat4 mat_inverse(const Mat4& M) {
    double a[16]; for (int i = 0; i < 16; ++i) a[i] = M.m[i];
    double inv[16];
    inv[0] = a[5]*a[10]*a[15] - a[5]*a[11]*a[14] - a[9]*a[6]*a[15] + a[9]*a[7]*a[14] + a[13]*a[6]*a[11] - a[13]*a[7]*a[10];
    inv[4] = -a[4]*a[10]*a[15] + a[4]*a[11]*a[14] + a[8]*a[6]*a[15] - a[8]*a[7]*a[14] - a[12]*a[6]*a[11] + a[12]*a[7]*a[10];
    inv[8] = a[4]*a[9]*a[15] - a[4]*a[11]*a[13] - a[8]*a[5]*a[15] + a[8]*a[7]*a[13] + a[12]*a[5]*a[11] - a[12]*a[7]*a[9];
    inv[12] = -a[4]*a[9]*a[14] + a[4]*a[10]*a[13] + a[8]*a[5]*a[14] - a[8]*a[6]*a[13] - a[12]*a[5]*a[10] + a[12]*a[6]*a[9];
    inv[1] = -a[1]*a[10]*a[15] + a[1]*a[11]*a[14] + a[9]*a[2]*a[15] - a[9]*a[3]*a[14] - a[13]*a[2]*a[11] + a[13]*a[3]*a[10];
    inv[5] = a[0]*a[10]*a[15] - a[0]*a[11]*a[14] - a[8]*a[2]*a[15] + a[8]*a[3]*a[14] + a[12]*a[2]*a[11] - a[12]*a[3]*a[10];
    inv[9] = -a[0]*a[9]*a[15] + a[0]*a[11]*a[13] + a[8]*a[1]*a[15] - a[8]*a[3]*a[13] - a[12]*a[1]*a[11] + a[12]*a[3]*a[9];
    inv[13] = a[0]*a[9]*a[14] - a[0]*a[10]*a[13] - a[8]*a[1]*a[14] + a[8]*a[2]*a[13] + a[12]*a[1]*a[10] - a[12]*a[2]*a[9];
    inv[2] = a[1]*a[6]*a[15] - a[1]*a[7]*a[14] - a[5]*a[2]*a[15] + a[5]*a[3]*a[14] + a[13]*a[2]*a[7] - a[13]*a[3]*a[6];
    inv[6] = -a[0]*a[6]*a[15] + a[0]*a[7]*a[14] + a[4]*a[2]*a[15] - a[4]*a[3]*a[14] - a[12]*a[2]*a[7] + a[12]*a[3]*a[6];
    inv[10] = a[0]*a[5]*a[15] - a[0]*a[7]*a[13] - a[4]*a[1]*a[15] + a[4]*a[3]*a[13] + a[12]*a[1]*a[7] - a[12]*a[3]*a[5];
    inv[14] = -a[0]*a[5]*a[14] + a[0]*a[6]*a[13] + a[4]*a[1]*a[14] - a[4]*a[2]*a[13] - a[12]*a[1]*a[6] + a[12]*a[2]*a[5];
    inv[3] = -a[1]*a[6]*a[11] + a[1]*a[7]*a[10] + a[5]*a[2]*a[11] - a[5]*a[3]*a[10] - a[9]*a[2]*a[7] + a[9]*a[3]*a[6];
    inv[7] = a[0]*a[6]*a[11] - a[0]*a[7]*a[10] - a[4]*a[2]*a[11] + a[4]*a[3]*a[10] + a[8]*a[2]*a[7] - a[8]*a[3]*a[6];
    inv[11] = -a[0]*a[5]*a[11] + a[0]*a[7]*a[9] + a[4]*a[1]*a[11] - a[4]*a[3]*a[9] - a[8]*a[1]*a[7] + a[8]*a[3]*a[5];
    inv[15] = a[0]*a[5]*a[10] - a[0]*a[6]*a[9] - a[4]*a[1]*a[10] + a[4]*a[2]*a[9] + a[8]*a[1]*a[6] - a[8]*a[2]*a[5];
    double det = a[0]*inv[0] + a[1]*inv[4] + a[2]*inv[8] + a[3]*inv[12];
    Mat4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = (float)(inv[i] / det);
    return r;
}
static inline float mat_det3(const Mat4& M) {
    const float* a = M.m;
    return (a[0] * (a[5] * a[10] - a[6] * a[9]) - a[1] * (a[4] * a[10] - a[6] * a[8])) + a[2] * (a[4] * a[9] - a[5] * a[8]);
}

// ---------------------------------------------------------------- small-float storage formats
// D3D11.3 functional spec 3.2.3.3: narrowing float->float conversions round toward zero
// (rounding = 0), hardware-style round-to-nearest-even selectable (rounding = 1).
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// Encode f32 -> (5-bit exponent, mant_bits mantissa) with optional sign bit.  Returns the unsigned field (without sign).
static inline uint32_t encode_small_float(float f, int mant_bits, bool has_sign, int rounding, uint32_t* sign_out) {
    uint32_t u = f2u(f);
    uint32_t sign = u >> 31;
    uint32_t absu = u & 0x7fffffffu;
    const uint32_t exp_all = 31u << mant_bits;
    if (sign_out) *sign_out = has_sign ? sign : 0;
    if (absu > 0x7f800000u) return exp_all | ((1u << mant_bits) - 1u);      // NaN
    if (!has_sign && sign) return 0;                                       // negative (incl. -inf) -> 0
    if (absu == 0x7f800000u) return exp_all;                               // +-inf
    int e = (int)(absu >> 23) - 127 + 15;
    uint32_t m = absu & 0x7fffffu;
    const int shift = 23 - mant_bits;
    const uint32_t max_finite = exp_all - 1u;
    if (e >= 31) return rounding == 0 ? max_finite : exp_all;              // overflow: RTZ -> max finite, RNE -> inf
    uint32_t r;
    if (e <= 0) {                                                          // denormal in the target
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
        if (rem > half || (rem == half && (r & 1u))) r += 1;               // may carry into exponent (-> inf), correct
    }
    return r;
}
static inline float decode_small_float(uint32_t v, int mant_bits, uint32_t sign) {
    uint32_t e = v >> mant_bits, m = v & ((1u << mant_bits) - 1u);
    float r;
    if (e == 0) r = ldexpf((float)m, -14 - mant_bits);
    else if (e == 31) r = m ? NAN : INFINITY;
    else r = ldexpf((float)(m | (1u << mant_bits)), (int)e - 15 - mant_bits);
    return sign ? -r : r;
}
static inline uint16_t f32_to_f16(float f, int rounding) {
    uint32_t s; uint32_t v = encode_small_float(f, 10, true, rounding, &s);
    return (uint16_t)((s << 15) | v);
}
static inline float f16_to_f32(uint16_t h) { return decode_small_float(h & 0x7fffu, 10, h >> 15); }
// DXGI_FORMAT_R11G11B10_FLOAT: R bits 0-10 (6-bit mantissa), G bits 11-21, B bits 22-31 (5-bit mantissa)
static inline uint32_t pack_r11g11b10(float3 c, int rounding) {
    uint32_t r = encode_small_float(c.x, 6, false, rounding, nullptr);
    uint32_t g = encode_small_float(c.y, 6, false, rounding, nullptr);
    uint32_t b = encode_small_float(c.z, 5, false, rounding, nullptr);
    return r | (g << 11) | (b << 22);
}
static inline float3 unpack_r11g11b10(uint32_t p) {
    return {decode_small_float(p & 0x7ffu, 6, 0), decode_small_float((p >> 11) & 0x7ffu, 6, 0), decode_small_float(p >> 22, 5, 0)};
}

// exp(x) for x <= 0 with a fixed evaluation (HLSL leaves exp's precision open; the vMF prefilter's exp feeds UNORM8 texels that
// oracle and product must agree on bit for bit): 2^(x log2 e) = 2^floor * P(frac), P of degree 6 (|error| < 1.7e-7 on [0, 1))
static inline float pinned_exp_neg(float x) {
    if (!(x > -87.f)) return 0.f;                        // underflow, -inf, NaN
    const float t = x * 1.44269502162933349609375f;
    const float fi = floorf(t), f = t - fi;
    const float p = 1.0f + f * (0.693146824836731f + f * (0.24023109674453735f + f * (0.055478911846876144f + f * (0.009686185978353024f
                  + f * (0.001238241558894515f + f * 0.000218712622881867f)))));
    uint32_t b; memcpy(&b, &p, 4); b += (uint32_t)((int)fi << 23); float r; memcpy(&r, &b, 4); return r;
}

} // namespace orc
