// host_math.h -- float32 host math with a fixed evaluation order (compiled with -ffp-contract=off).
// Row-major matrices, row vectors (DirectX SimpleMath / HLSL `mul(v, M)` under -Zpr;
// reference: source/demo-dll/src/shadercompiler.cpp:12-14).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/incr_types.h"

namespace incr {

struct V3 { float x, y, z; };
static inline V3 v3(float x, float y, float z) { return {x, y, z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 normalize3(V3 v) { float l = sqrtf(dot3(v, v)); return {v.x / l, v.y / l, v.z / l}; }

struct M44 { float m[16]; };
static inline M44 m44_from(const incr_matrix& s) { M44 r; memcpy(r.m, s.m, 64); return r; }
static inline M44 m44_identity() { M44 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }

// C = A * B with c_ij = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j
static inline M44 m44_mul(const M44& a, const M44& b) {
    M44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a.m[i * 4 + 0] * b.m[j] + a.m[i * 4 + 1] * b.m[4 + j];
            s = s + a.m[i * 4 + 2] * b.m[8 + j];
            s = s + a.m[i * 4 + 3] * b.m[12 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
// (p, 1) * M
static inline V3 m44_transform_point(V3 p, const M44& M) {
    V3 r;
    r.x = ((p.x * M.m[0] + p.y * M.m[4]) + p.z * M.m[8]) + M.m[12];
    r.y = ((p.x * M.m[1] + p.y * M.m[5]) + p.z * M.m[9]) + M.m[13];
    r.z = ((p.x * M.m[2] + p.y * M.m[6]) + p.z * M.m[10]) + M.m[14];
    return r;
}
// determinant of the upper-left 3x3 (sign tells whether an instance transform mirrors)
static inline float m44_det3(const M44& M) {
    const float* a = M.m;
    float c0 = a[5] * a[10] - a[6] * a[9];
    float c1 = a[4] * a[10] - a[6] * a[8];
    float c2 = a[4] * a[9] - a[5] * a[8];
    return (a[0] * c0 - a[1] * c1) + a[2] * c2;
}
// Inverse through the adjugate, accumulated in double precision and rounded to float once
// (stands in for SimpleMath Matrix::Invert, render-jobs/path-tracing.inl:199,201).
static inline M44 m44_inverse(const M44& src) {
    double m[16];
    for (int i = 0; i < 16; ++i) m[i] = (double)src.m[i];
    // 2x2 sub-determinants of the lower two rows and the upper two rows
    double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    double inv = 1.0 / det;
    double r[16];
    r[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * inv;
    r[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * inv;
    r[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * inv;
    r[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * inv;
    r[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * inv;
    r[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * inv;
    r[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * inv;
    r[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * inv;
    r[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * inv;
    r[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * inv;
    r[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * inv;
    r[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * inv;
    r[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * inv;
    r[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * inv;
    r[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * inv;
    r[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * inv;
    M44 out;
    for (int i = 0; i < 16; ++i) out.m[i] = (float)r[i];
    return out;
}

} // namespace incr
