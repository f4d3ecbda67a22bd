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
// KEYED stacks also keep each entry's box-entry distance so that closest-hit traversal can drop popped nodes
// that lie beyond the current best hit without fetching them.
template <int NLDS, int STRIDE, bool KEYED = false>
struct LaneStackT {
    static constexpr bool kKeyed = KEYED;
    static constexpr int kCap = kStackOvf + kStackLds;
    uint32_t* lds;            // &stack[0][tid]
    float* ldsKey;            // &keys[0][tid]   (KEYED only)
    uint32_t ovf[kCap - NLDS];
    float ovfKey[KEYED ? kCap - NLDS : 1];
    int sp;
    DI void reset() { sp = 0; }
    DI void push(uint32_t v, float key = 0.f) {
        if constexpr (NLDS > 0) {
            if (sp < NLDS) { lds[sp * STRIDE] = v; if constexpr (KEYED) ldsKey[sp * STRIDE] = key; ++sp; return; }
        }
        if (sp < kCap) { ovf[sp - NLDS] = v; if constexpr (KEYED) ovfKey[sp - NLDS] = key; ++sp; }   // deeper is unreachable for a valid tree
    }
    DI uint32_t pop(float* key = nullptr) {
        --sp;
        if constexpr (NLDS > 0) {
            if (sp < NLDS) { if constexpr (KEYED) *key = ldsKey[sp * STRIDE]; return lds[sp * STRIDE]; }
        }
        if constexpr (KEYED) *key = ovfKey[sp - NLDS];
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

// Per-ray traversal constants: reciprocal direction and, per axis, which float4 of the node holds the NEAR planes
// (lo planes when the direction component is >= 0, hi planes otherwise): the slab test needs no per-child min/max.
struct RaySetup { f3 o, invd; uint32_t nearX, nearY, nearZ; };
DI RaySetup ray_setup(f3 o, f3 d) {
    RaySetup r;
    r.o = o;
    r.invd = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    r.nearX = d.x < 0.f ? 3u : 0u; r.nearY = d.y < 0.f ? 4u : 1u; r.nearZ = d.z < 0.f ? 5u : 2u;
    return r;
}

// Tests the 4 child boxes of `node`; returns children sorted near->far in ch[0..n) and their entry distances in key[].
// Slab test t = (plane - o) * invd with the far distance widened by 1 + 2*gamma(3) (Ize 2013 / pbrt robust bounds test):
// conservative, so the exact triangle test -- not the box test -- decides every hit.
DI int node_children(const BvhNode4* __restrict__ nodes, uint32_t node, const RaySetup& r, float tmin, float tmax, uint32_t ch[4], float key[4]) {
    const float4* __restrict__ p = (const float4*)(nodes + node);
    const float4 nx4 = p[r.nearX], ny4 = p[r.nearY], nz4 = p[r.nearZ];
    const float4 fx4 = p[3u - r.nearX], fy4 = p[5u - r.nearY], fz4 = p[7u - r.nearZ];
    const uint4 cc = *(const uint4*)(p + 6);
    const float nx[4] = {nx4.x, nx4.y, nx4.z, nx4.w}, ny[4] = {ny4.x, ny4.y, ny4.z, ny4.w}, nz[4] = {nz4.x, nz4.y, nz4.z, nz4.w};
    const float fx[4] = {fx4.x, fx4.y, fx4.z, fx4.w}, fy[4] = {fy4.x, fy4.y, fy4.z, fy4.w}, fz[4] = {fz4.x, fz4.y, fz4.z, fz4.w};
    ch[0] = cc.x; ch[1] = cc.y; ch[2] = cc.z; ch[3] = cc.w;
    int n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = (nx[k] - r.o.x) * r.invd.x, ay = (ny[k] - r.o.y) * r.invd.y, az = (nz[k] - r.o.z) * r.invd.z;
        const float bx = (fx[k] - r.o.x) * r.invd.x, by = (fy[k] - r.o.y) * r.invd.y, bz = (fz[k] - r.o.z) * r.invd.z;
        const float t0 = fmaxf(fmaxf(fmaxf(ax, ay), az), tmin);                       // NaN planes (0 * inf) are ignored by max/min
        const float t1 = fminf(fminf(fminf(bx, by), bz), tmax) * 1.0000004f;
        const bool hit = (t0 <= t1) && (ch[k] != kEmptyChild);
        key[k] = hit ? t0 : __uint_as_float(0x7f800000u);
        n += hit ? 1 : 0;
    }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
    return n;
}

// ---- traversal as a resumable state machine: one call = one node (4 box tests) or one leaf (<= 4 triangles) ----
// The wavefront kernels interleave these steps with lane refill so that a wave never idles on its slowest ray.
struct TravClosest {
    RaySetup rs; f3 d; float tmin, tmax;
    float best, bu, bv; uint32_t bestFlat, bestTri, cur; bool found;
};
DI void trav_closest_begin(TravClosest& T, f3 o, f3 d, float tmin, float tmax) {
    T.rs = ray_setup(o, d); T.d = d; T.tmin = tmin; T.tmax = tmax;
    T.best = tmax; T.bu = 0.f; T.bv = 0.f; T.bestFlat = 0xffffffffu; T.bestTri = 0; T.cur = 0; T.found = false;
}
// returns true when the traversal is complete (T.found / T.best / T.bu / T.bv / T.bestTri hold the result)
template <class ST> DI bool trav_closest_step(const PtParams& P, ST& st, TravClosest& T) {
    if (!(T.cur & kLeafBit)) {
        uint32_t ch[4]; float key[4];
        const int n = node_children(P.nodes, T.cur, T.rs, T.tmin, T.best, ch, key);
        if (n > 0) {
            if (n > 3) st.push(ch[3], key[3]);
            if (n > 2) st.push(ch[2], key[2]);
            if (n > 1) st.push(ch[1], key[1]);
            T.cur = ch[0];
            return false;
        }
    } else {
        const uint32_t first = (T.cur & ~kLeafBit) >> 2, cnt = (T.cur & 3u) + 1u;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t ti = first + i;
            float t, u, v; uint32_t flat, flags;
            if (!tri_intersect(P.tris, ti, T.rs.o, T.d, T.tmin, T.tmax, &t, &u, &v, &flat, &flags)) continue;
            if (T.found && !(t < T.best || (t == T.best && flat < T.bestFlat))) continue;   // equal-t ties: smaller flattened id wins
            if (!T.found && !(t < T.best)) continue;
            if (flags & kTriMirrored) { const float tmp = u; u = v; v = tmp; }
            if (!(flags & kTriOpaque) && !any_hit_accepts(P, ti, u, v)) continue;
            T.found = true; T.best = t; T.bestFlat = flat; T.bestTri = ti; T.bu = u; T.bv = v;
        }
    }
    // pop; keyed stacks skip subtrees that start beyond the best hit (strictly: equal-t ties must still be visited)
    while (st.sp > 0) {
        if constexpr (ST::kKeyed) { float k; T.cur = st.pop(&k); if (k > T.best) continue; }
        else T.cur = st.pop();
        return false;
    }
    return true;
}

// TraceRay(RAY_FLAG_CULL_BACK_FACING_TRIANGLES): closest hit
template <class ST> DI bool trace_closest(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax, Hit* hit) {
    TravClosest T;
    trav_closest_begin(T, o, d, tmin, tmax);
    st.reset();
    while (!trav_closest_step(P, st, T)) {}
    if (T.found) { hit->t = T.best; hit->u = T.bu; hit->v = T.bv; hit->tri = T.bestTri; }
    return T.found;
}

struct TravAny { RaySetup rs; f3 d; float tmin, tmax; uint32_t cur; bool occluded; };
DI void trav_any_begin(TravAny& T, f3 o, f3 d, float tmin, float tmax) {
    T.rs = ray_setup(o, d); T.d = d; T.tmin = tmin; T.tmax = tmax; T.cur = 0; T.occluded = false;
}
template <class ST> DI bool trav_any_step(const PtParams& P, ST& st, TravAny& T) {
    if (!(T.cur & kLeafBit)) {
        uint32_t ch[4]; float key[4];
        const int n = node_children(P.nodes, T.cur, T.rs, T.tmin, T.tmax, ch, key);
        if (n > 0) {
            if (n > 3) st.push(ch[3]);
            if (n > 2) st.push(ch[2]);
            if (n > 1) st.push(ch[1]);
            T.cur = ch[0];
            return false;
        }
    } else {
        const uint32_t first = (T.cur & ~kLeafBit) >> 2, cnt = (T.cur & 3u) + 1u;
        for (uint32_t i = 0; i < cnt; ++i) {
            float t, u, v; uint32_t flat, flags;
            if (tri_intersect(P.tris, first + i, T.rs.o, T.d, T.tmin, T.tmax, &t, &u, &v, &flat, &flags)) { T.occluded = true; return true; }
        }
    }
    if (st.sp == 0) return true;
    T.cur = st.pop();
    return false;
}

// RayQuery<CULL_BACK_FACING> + one Proceed(): occluded iff any front-facing triangle lies in (tmin, tmax)
// (lighting/common.hlsli:102-115, 161-174).
template <class ST> DI bool trace_any(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax) {
    TravAny T;
    trav_any_begin(T, o, d, tmin, tmax);
    st.reset();
    while (!trav_any_step(P, st, T)) {}
    return T.occluded;
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
