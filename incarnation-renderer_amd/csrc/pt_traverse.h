// pt_traverse.h -- BVH4 traversal with LDS-staged per-lane stacks, ray/triangle test, camera rays.
// Shared by the path-tracing kernels (pathtrace.hip: lane-per-pixel megakernel; pathtrace_wave.hip: persistent wavefront kernel)
// and by incr_trace_rays.  See pathtrace.hip for the reference citations.
#pragma once
#include "pt_device.h"

namespace incr {

static constexpr int kBlock = 256;
static constexpr int kStackLds = 24;      // LDS-resident entries per lane (24 KB per workgroup)
static constexpr int kStackOvf = 72;      // private overflow (4-wide SAH trees stay far below this)

struct Hit { float t, u, v; uint32_t tri; };

// Per-lane traversal stack: the first NLDS entries live in LDS ([entry][lane], stride STRIDE dwords: lanes of a
// wave hit 64 consecutive banks), deeper entries in a private (scratch) array that only pathological rays touch.
template <int NLDS, int STRIDE>
struct LaneStackT {
    uint32_t* lds;            // &stack[0][tid]
    uint32_t ovf[kStackOvf + (kStackLds - NLDS)];
    int sp;
    static constexpr int kCap = kStackOvf + kStackLds;
    DI void reset() { sp = 0; }
    DI void push(uint32_t v) {
        if constexpr (NLDS > 0) { if (sp < NLDS) { lds[sp * STRIDE] = v; ++sp; return; } }
        if (sp < kCap) { ovf[sp - NLDS] = v; ++sp; }      // deeper than kCap is unreachable for a valid tree: drop rather than corrupt memory
    }
    DI uint32_t pop() {
        --sp;
        if constexpr (NLDS > 0) { if (sp < NLDS) return lds[sp * STRIDE]; }
        return ovf[sp - NLDS];
    }
};
using LaneStack = LaneStackT<kStackLds, kBlock>;

DI bool tri_intersect(const TriIsect* __restrict__ tris, uint32_t ti, f3 o, f3 d, float tmin, float tmax,
                      float* t, float* u, float* v, uint32_t* flatId, uint32_t* flags) {
    const float4* __restrict__ p = (const float4*)(tris + ti);
    const float4 a = p[0], b = p[1], c = p[2];
    const f3 v0 = {a.x, a.y, a.z}, e1 = {a.w, b.x, b.y}, e2 = {b.z, b.w, c.x};
    const f3 pv = cross(d, e2);
    const float det = dot(e1, pv);
    if (!(det > 0.f)) return false;                       // RAY_FLAG_CULL_BACK_FACING_TRIANGLES
    const float inv = 1.f / det;
    const f3 sv = o - v0;
    const float uu = dot(sv, pv) * inv;
    if (!(uu >= 0.f && uu <= 1.f)) return false;
    const f3 q = cross(sv, e1);
    const float vv = dot(d, q) * inv;
    if (!(vv >= 0.f && uu + vv <= 1.f)) return false;
    const float tt = dot(e2, q) * inv;
    if (!(tt > tmin && tt < tmax)) return false;
    *t = tt; *u = uu; *v = vv;
    *flatId = __float_as_uint(c.y); *flags = __float_as_uint(c.z);
    return true;
}

// ahsMain (pathtracing.hlsl:270-302): alpha test of non-opaque geometry
DI bool any_hit_accepts(const PtParams& P, uint32_t ti, float u, float v) {
    const TriShade* __restrict__ sh = P.shade + ti;
    const f2 uv0 = {sh->uv[0][0], sh->uv[0][1]}, uv1 = {sh->uv[1][0], sh->uv[1][1]}, uv2 = {sh->uv[2][0], sh->uv[2][1]};
    const f2 uv = uv0 + u * (uv1 - uv0) + v * (uv2 - uv0);
    const int tex = P.materials[sh->materialIndex].baseColorTextureIndex;
    if (tex != -1) {
        const float alpha = SampleLevel0(P.textures, P.srgbLut, tex, uv).w;
        if (alpha < 0.5f) return false;
    }
    return true;
}

#define CSWAP(i, j) { const bool sw = key[j] < key[i]; const float tk = sw ? key[j] : key[i]; const float tk2 = sw ? key[i] : key[j]; \
                      const uint32_t tc = sw ? ch[j] : ch[i]; const uint32_t tc2 = sw ? ch[i] : ch[j]; key[i] = tk; key[j] = tk2; ch[i] = tc; ch[j] = tc2; }

// Tests the 4 child boxes of `node`; returns children sorted near->far in ch[0..n).
DI int node_children(const BvhNode4* __restrict__ nodes, uint32_t node, f3 o, f3 invd, float tmin, float tmax, uint32_t ch[4]) {
    const float4* __restrict__ p = (const float4*)(nodes + node);
    const float4 lox = p[0], loy = p[1], loz = p[2], hix = p[3], hiy = p[4], hiz = p[5];
    const uint4 cc = *(const uint4*)(p + 6);
    const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, ly[4] = {loy.x, loy.y, loy.z, loy.w}, lz[4] = {loz.x, loz.y, loz.z, loz.w};
    const float hx[4] = {hix.x, hix.y, hix.z, hix.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
    float key[4];
    ch[0] = cc.x; ch[1] = cc.y; ch[2] = cc.z; ch[3] = cc.w;
    int n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = (lx[k] - o.x) * invd.x, bx = (hx[k] - o.x) * invd.x;
        const float ay = (ly[k] - o.y) * invd.y, by = (hy[k] - o.y) * invd.y;
        const float az = (lz[k] - o.z) * invd.z, bz = (hz[k] - o.z) * invd.z;
        float t0 = tmin, t1 = tmax;
        t0 = fmaxf(t0, fminf(ax, bx) * 0.9999990f - 1e-30f); t1 = fminf(t1, fmaxf(ax, bx) * 1.0000010f + 1e-30f);
        t0 = fmaxf(t0, fminf(ay, by) * 0.9999990f - 1e-30f); t1 = fminf(t1, fmaxf(ay, by) * 1.0000010f + 1e-30f);
        t0 = fmaxf(t0, fminf(az, bz) * 0.9999990f - 1e-30f); t1 = fminf(t1, fmaxf(az, bz) * 1.0000010f + 1e-30f);
        const bool hit = (t0 <= t1) && (ch[k] != kEmptyChild);
        key[k] = hit ? t0 : __uint_as_float(0x7f800000u);
        n += hit ? 1 : 0;
    }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
    return n;
}

// TraceRay(RAY_FLAG_CULL_BACK_FACING_TRIANGLES): closest hit, equal-t ties to the smaller flattened id.
template <class ST> DI bool trace_closest(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax, Hit* hit) {
    const f3 invd = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    st.reset();
    bool found = false; float best = tmax; uint32_t bestFlat = 0xffffffffu, bestTri = 0; float bu = 0.f, bv = 0.f;
    uint32_t cur = 0;                    // root node
    for (;;) {
        if (!(cur & kLeafBit)) {
            uint32_t ch[4];
            const int n = node_children(P.nodes, cur, o, invd, tmin, best, ch);
            if (n > 0) {
                if (n > 3) st.push(ch[3]);
                if (n > 2) st.push(ch[2]);
                if (n > 1) st.push(ch[1]);
                cur = ch[0];
                continue;
            }
        } else {
            const uint32_t first = (cur & ~kLeafBit) >> 2, cnt = (cur & 3u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t ti = first + i;
                float t, u, v; uint32_t flat, flags;
                if (!tri_intersect(P.tris, ti, o, d, tmin, tmax, &t, &u, &v, &flat, &flags)) continue;
                if (found && !(t < best || (t == best && flat < bestFlat))) continue;
                if (!found && !(t < best)) continue;
                if (flags & kTriMirrored) { const float tmp = u; u = v; v = tmp; }
                if (!(flags & kTriOpaque) && !any_hit_accepts(P, ti, u, v)) continue;
                found = true; best = t; bestFlat = flat; bestTri = ti; bu = u; bv = v;
            }
        }
        if (st.sp == 0) break;
        cur = st.pop();
    }
    if (found) { hit->t = best; hit->u = bu; hit->v = bv; hit->tri = bestTri; }
    return found;
}

// RayQuery<CULL_BACK_FACING> + one Proceed(): occluded iff any front-facing triangle lies in (tmin, tmax)
// (lighting/common.hlsli:102-115, 161-174).
template <class ST> DI bool trace_any(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax) {
    const f3 invd = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    st.reset();
    uint32_t cur = 0;
    for (;;) {
        if (!(cur & kLeafBit)) {
            uint32_t ch[4];
            const int n = node_children(P.nodes, cur, o, invd, tmin, tmax, ch);
            if (n > 0) {
                if (n > 3) st.push(ch[3]);
                if (n > 2) st.push(ch[2]);
                if (n > 1) st.push(ch[1]);
                cur = ch[0];
                continue;
            }
        } else {
            const uint32_t first = (cur & ~kLeafBit) >> 2, cnt = (cur & 3u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                float t, u, v; uint32_t flat, flags;
                if (tri_intersect(P.tris, first + i, o, d, tmin, tmax, &t, &u, &v, &flat, &flags)) return true;
            }
        }
        if (st.sp == 0) break;
        cur = st.pop();
    }
    return false;
}

struct Counters { uint32_t camera, bounce, skyShadow, lightShadow, hits; };

struct Ray { f3 o, d; float tmin, tmax; };

// raytracing/common.hlsli:25-53
DI Ray GenerateCameraRay(const PtParams& P, f2 index, f2 randomSample) {
    const f2 xy = {index.x + 0.5f, index.y + 0.5f};
    f2 screenPos = {xy.x / (float)P.W * 2.0f - 1.0f, xy.y / (float)P.H * 2.0f - 1.0f};
    screenPos.y = -screenPos.y;
    const float* cm = P.cameraMatrix;
    const f3 cameraPos = {cm[12], cm[13], cm[14]};
    const f4 world = mul_vec4({screenPos.x, screenPos.y, 0.0001f, 1.f}, P.projectionToWorld);
    const f3 wxyz = {world.x / world.w, world.y / world.w, world.z / world.w};
    const f3 primaryRayDir = normalize(wxyz - cameraPos);
    const f3 focalPoint = cameraPos + P.focalLength * primaryRayDir;
    const f3 cameraRight = {cm[0], cm[1], cm[2]};
    const f3 cameraUp = {cm[4], cm[5], cm[6]};
    const f2 offset = ConcentricSampleDisk(randomSample);
    const f3 rayOrigin = cameraPos + (P.aperture * offset.x) * cameraRight + (P.aperture * offset.y) * cameraUp;
    Ray r;
    r.o = rayOrigin;
    r.d = normalize(focalPoint - rayOrigin);
    r.tmin = 0.001f; r.tmax = 10000.0f;
    return r;
}

DI int64_t shard_global_row(const PtParams& P, uint32_t localRow) {
    if (P.rowBlockHeight == 0) return localRow < P.H ? (int64_t)localRow : -1;
    const uint32_t k = localRow / P.rowBlockHeight, r = localRow % P.rowBlockHeight;
    const uint64_t y = (uint64_t)(P.rowBlockFirst + (uint64_t)k * P.rowBlockStride) * P.rowBlockHeight + r;
    return y < P.H ? (int64_t)y : -1;
}

} // namespace incr
