// pt_traverse.h -- 8-wide quantised BVH traversal with LDS-staged per-lane stacks, ray/triangle test, camera rays.
// Shared by every kernel that traces: the wavefront path tracer (pathtrace_wf.hip), incr_trace_rays (pt_queries.hip) and the raster
// passes that issue ray queries (raster.hip: sun shadow packets, HBAO, forward lighting).
//
// Stand-in for the fixed-function DXR traversal: TraceRay(RAY_FLAG_CULL_BACK_FACING_TRIANGLES)
// closest hit with the ahsMain alpha test on non-opaque geometry, and RayQuery + one Proceed() for shadow rays.
//
// Traversal is a RESUMABLE STATE MACHINE with exactly ONE memory round trip per step: the loads of a step are issued
// together before any branching.  Measured on MI355X: with per-leaf / per-triangle loops inside the step, a wave paid
// up to ~10 serialised round trips per iteration (~8000 cycles per step, waves 70 % waiting).  trav_closest_step2 /
// trav_any_step2 (what every kernel uses) test one 8-wide node (5 x dwordx4) AND one queued triangle (3 x dwordx4) in the same
// step; trav_closest_step, the first generation (node OR triangle), is kept for the statistics build (tools/trav_stats.py).
#pragma once
#include "pt_device.h"

namespace incr {

static constexpr int kBlock = 256;
static constexpr int kStackLds = 24;      // stack entries per lane of the lane-per-pixel kernels (= kMaxBvhDepth - 1)

struct Hit { float t, u, v; uint32_t tri; };

// Per-lane traversal stack, LDS only ([entry][lane], stride STRIDE dwords: the lanes of a wave hit 64 consecutive banks).
// An entry names ALL pending inner children of one node -- childBase << 9 | reverse << 8 | slot mask (device_types.h
// SLOT ORDER) -- so the stack holds at most one entry per tree level: N >= maxDepth - 1 can never overflow and there is
// no private (scratch) spill array.  Round 1 pushed one entry per child: 8 LDS entries overflowed for ~10 % of the rays
// (p90 depth 8, max 22 on Sponza-stat), i.e. on nearly every wave step, and cost 20 M scratch-store instructions per
// launch; per node the depth is p99 5, max 7 (tools/trav_sim).  The host picks N from incr_scene's maxDepth.
// KEYED stacks also keep, per entry, the smallest box-entry distance of the group: closest-hit traversal drops a
// whole group that lies beyond the current best hit without fetching any of its nodes.
typedef float incr_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const incr_v4f* LdsNodes;      // native vector type: one ds_read_b128 per element
template <int NLDS, int STRIDE, bool KEYED = false, bool TOP = false>
struct LaneStackT {
    static constexpr bool kKeyed = KEYED;
    static constexpr bool kTop = TOP;          // the workgroup keeps the first `topCount` nodes (the hottest, scene_build.cpp NODE ORDER) in LDS
    static constexpr int kCap = NLDS;
    LdsNodes topNodes; uint32_t topCount;
    DI void bind_top(const float4* ldsNodes, uint32_t count) { topNodes = (LdsNodes)ldsNodes; topCount = count; }
    // explicit LDS address space: through a generic pointer the pushes/pops compile to flat_* instructions
    __attribute__((address_space(3))) uint32_t* lds;      // &stack[0][tid]
    // KEYED only: &keys[0][tid].  A key is kept as the UPPER 16 BITS of its float (round 5): entry distances are positive, so truncation rounds down and the key stays a
    // lower bound -- a group is dropped a little later at worst, the hit found is the same -- and the closest-hit kernel's stack takes 6 bytes per entry instead of 8: the
    // LDS it gives back holds 87 hot nodes per workgroup instead of 23 (pathtrace_wf.hip TravCfg).
    __attribute__((address_space(3))) uint16_t* ldsKey;
    DI void bind(uint32_t* stackBase, uint16_t* keyBase) {
        lds = (__attribute__((address_space(3))) uint32_t*)stackBase;
        ldsKey = (__attribute__((address_space(3))) uint16_t*)keyBase;
    }
    int sp;
    DI void reset() { sp = 0; }
    DI void push(uint32_t v, float key = 0.f) {
        if (sp < NLDS) { lds[sp * STRIDE] = v; if constexpr (KEYED) ldsKey[sp * STRIDE] = (uint16_t)(__float_as_uint(key) >> 16); ++sp; }      // sp == NLDS: unreachable for a checked tree
    }
    DI uint32_t top() const { return lds[(sp - 1) * STRIDE]; }
    DI float topKey() const { return __uint_as_float((uint32_t)ldsKey[(sp - 1) * STRIDE] << 16); }
    DI void setTop(uint32_t v) { lds[(sp - 1) * STRIDE] = v; }
    DI void drop() { --sp; }
};
// Private-memory flavour for the rare inline queries of kernels that own no LDS stack (pathtrace_wf.hip accumulate).
template <int N>
struct PrivStackT {
    static constexpr bool kKeyed = false;
    static constexpr bool kTop = false;
    LdsNodes topNodes; uint32_t topCount;
    uint32_t e[N]; int sp;
    DI void reset() { sp = 0; }
    DI void push(uint32_t v, float = 0.f) { if (sp < N) e[sp++] = v; }
    DI uint32_t top() const { return e[sp - 1]; }
    DI float topKey() const { return 0.f; }
    DI void setTop(uint32_t v) { e[sp - 1] = v; }
    DI void drop() { --sp; }
};
using LaneStack = LaneStackT<kStackLds, kBlock>;
DI uint32_t stack_entry(uint32_t childBase, bool reverse, uint32_t slotMask) { return childBase << 9 | (reverse ? 0x100u : 0u) | slotMask; }

// Moeller-Trumbore on the stored (v0, e1, e2); every operation unfused and in a pinned order: the hit predicate and t, u, v are part of
// the parity contract (bit-exact against the CPU oracle).  a, b, c = the three dwordx4 of a TriIsect record.
// The PREDICATE "the ray hits the triangle inside (tmin, tmax)" is evaluated on the UNSCALED numerators (round 4; rounds 1-3 divided first):
//     det = e1 . (d x e2) > 0          (RAY_FLAG_CULL_BACK_FACING_TRIANGLES)
//     U = (o - v0) . (d x e2),  V = d . ((o - v0) x e1),  T = e2 . ((o - v0) x e1)
//     0 <= U <= det,  0 <= V,  U + V <= det,  tmin det < T < tmax det
// -- the same triangle up to the last ulp of its edges, and what a shadow query needs is only this: an any-hit test costs no division
// (~12 instructions of the ~62 of a triangle step in kernels that are VALU-issue bound).  A closest-hit query divides once the
// predicate holds: t = T / det, u = U / det, v = V / det as products with the correctly rounded 1 / det, the values of rounds 1-3.
DI bool tri_hit(const float4 a, const float4 b, const float4 c, f3 o, f3 d, float tmin, float tmax, float* det, float* U, float* V, float* Tn) {
    const f3 v0 = {a.x, a.y, a.z}, e1 = {a.w, b.x, b.y}, e2 = {b.z, b.w, c.x};
    const f3 pv = cross(d, e2);
    const float dt = dot(e1, pv);
    if (!(dt > 0.f)) return false;
    const f3 sv = o - v0;
    const float uu = dot(sv, pv);
    if (!(uu >= 0.f && uu <= dt)) return false;
    const f3 q = cross(sv, e1);
    const float vv = dot(d, q);
    if (!(vv >= 0.f && uu + vv <= dt)) return false;
    const float tt = dot(e2, q);
    if (!(tt > tmin * dt && tt < tmax * dt)) return false;
    *det = dt; *U = uu; *V = vv; *Tn = tt;
    return true;
}
DI bool tri_intersect(const float4 a, const float4 b, const float4 c, f3 o, f3 d, float tmin, float tmax,
                      float* t, float* u, float* v, uint32_t* flatId, uint32_t* flags) {
    float det, U, V, Tn;
    if (!tri_hit(a, b, c, o, d, tmin, tmax, &det, &U, &V, &Tn)) return false;
    const float inv = 1.f / det;
    *t = Tn * inv; *u = U * inv; *v = V * inv;
    *flatId = __float_as_uint(c.y); *flags = __float_as_uint(c.z);
    return true;
}

// ahsMain (pathtracing.hlsl:270-302): alpha test of non-opaque geometry
DI bool any_hit_accepts(const PtParams& P, uint32_t flat, float u, float v) {
    const TriShade* __restrict__ sh = P.shade + flat;          // shading records are indexed by flattened triangle id
    const f2 uv0 = {sh->uv[0][0], sh->uv[0][1]}, uv1 = {sh->uv[1][0], sh->uv[1][1]}, uv2 = {sh->uv[2][0], sh->uv[2][1]};
    const f2 uv = uv0 + u * (uv1 - uv0) + v * (uv2 - uv0);
    const int tex = P.materials[sh->materialIndex].baseColorTextureIndex;
    if (tex != -1) {
        const float alpha = SampleLevel0(P.textures, P.srgbLut, tex, uv).w;
        if (alpha < 0.5f) return false;
    }
    return true;
}

// ---------------------------------------------------------------- 8-wide quantised BVH: node test
// Per-ray constants.  Child box plane along an axis: origin + q * scale, so t(q) = q * a + b with
// a = scale * invd and b = (origin - o) * invd: ONE fma per plane (these boxes only cull, they never decide a hit,
// so fused arithmetic is fine here; the exact triangle test stays unfused).
struct RaySetup { f3 o, invd; };
// 1 / d by v_rcp_f32 (1 ulp) instead of the exact division (12 instructions per component, 36 per started ray in kernels that are
// VALU-issue bound): the boxes only cull, and node8_test's margins carry the extra ulp (a and b both scale by 1 + e, |e| <= 2^-23:
// |dt| <= e (255 |a| + |b|)).  d = 0, denormal or NaN give inf / NaN planes, which the box test ignores (conservative).
DI RaySetup ray_setup(f3 o, f3 d) { return {o, {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)}}; }

struct Node8 {              // decoded header of one node + the hit mask / entry distances of its children
    uint32_t imask, childBase, triBase, leafTris, sortAxis;
    uint32_t hitMask;       // bit k: the ray interval overlaps child k's (conservative) box
    float tn[8];            // entry distance per child (valid where hitMask is set)
};

DI uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DI float ubyte_f(uint32_t packed, int k) { return (float)((packed >> (8 * k)) & 0xffu); }      // v_cvt_f32_ubyteK

// Conservative slab test of the 8 quantised child boxes (n0..n4 = the five dwordx4 of a BvhNode8); margins below.
// The exact triangle test -- never the box test -- decides a hit.
DI void node8_test(const float4 n0, const float4 n1, const float4 n2, const float4 n3, const float4 n4,
                   const RaySetup& r, float tmin, float tmax, Node8& N) {
    const uint32_t em = __float_as_uint(n0.w);
    N.imask = em >> 24;
    N.childBase = __float_as_uint(n1.x); N.triBase = __float_as_uint(n1.y);
    N.leafTris = __float_as_uint(n1.z); N.sortAxis = __float_as_uint(n1.w) & 3u;
    const float sx = __uint_as_float((em & 0xffu) << 23), sy = __uint_as_float(((em >> 8) & 0xffu) << 23), sz = __uint_as_float(((em >> 16) & 0xffu) << 23);
    const float ax = sx * r.invd.x, ay = sy * r.invd.y, az = sz * r.invd.z;
    const float bx = (n0.x - r.o.x) * r.invd.x, by = (n0.y - r.o.y) * r.invd.y, bz = (n0.z - r.o.z) * r.invd.z;
    // margins: 0.02 quantum + 2^-22 |b| for the rounding of a, b and the fma, plus the 1 + 4e-7 widening of the far distance
    // (Ize 2013) folded in -- |t| <= 255 |a| + |b| on every plane -- so that no child pays a multiply for it, plus one ulp of
    // 1 / d (ray_setup: 1.2e-7 (255 |a| + |b|) = 3.1e-5 quantum + 1.2e-7 |b|)
    const float mx = __builtin_fmaf(fabsf(bx), 7.7e-7f, 0.0203f * fabsf(ax)), my = __builtin_fmaf(fabsf(by), 7.7e-7f, 0.0203f * fabsf(ay)), mz = __builtin_fmaf(fabsf(bz), 7.7e-7f, 0.0203f * fabsf(az));
    const float tmaxW = tmax * 1.0000004f;
    const float bnx = bx - mx, bfx = bx + mx, bny = by - my, bfy = by + my, bnz = bz - mz, bfz = bz + mz;
    // near planes: lo bytes when the direction component is >= 0, hi bytes otherwise
    const bool px = ax >= 0.f, py = ay >= 0.f, pz = az >= 0.f;
    const uint32_t qlox0 = __float_as_uint(n2.x), qlox1 = __float_as_uint(n2.y), qloy0 = __float_as_uint(n2.z), qloy1 = __float_as_uint(n2.w);
    const uint32_t qloz0 = __float_as_uint(n3.x), qloz1 = __float_as_uint(n3.y), qhix0 = __float_as_uint(n3.z), qhix1 = __float_as_uint(n3.w);
    const uint32_t qhiy0 = __float_as_uint(n4.x), qhiy1 = __float_as_uint(n4.y), qhiz0 = __float_as_uint(n4.z), qhiz1 = __float_as_uint(n4.w);
    const uint32_t nx0 = px ? qlox0 : qhix0, nx1 = px ? qlox1 : qhix1, fx0 = px ? qhix0 : qlox0, fx1 = px ? qhix1 : qlox1;
    const uint32_t ny0 = py ? qloy0 : qhiy0, ny1 = py ? qloy1 : qhiy1, fy0 = py ? qhiy0 : qloy0, fy1 = py ? qhiy1 : qloy1;
    const uint32_t nz0 = pz ? qloz0 : qhiz0, nz1 = pz ? qloz1 : qhiz1, fz0 = pz ? qhiz0 : qloz0, fz1 = pz ? qhiz1 : qloz1;
    uint32_t hit = 0u;
#pragma unroll
    for (int k = 0; k < INCR_BVH_WIDTH; ++k) {
        const int j = k & 3;
        const uint32_t nxp = k < 4 ? nx0 : nx1, nyp = k < 4 ? ny0 : ny1, nzp = k < 4 ? nz0 : nz1;
        const uint32_t fxp = k < 4 ? fx0 : fx1, fyp = k < 4 ? fy0 : fy1, fzp = k < 4 ? fz0 : fz1;
        const float tnx = __builtin_fmaf(ubyte_f(nxp, j), ax, bnx), tny = __builtin_fmaf(ubyte_f(nyp, j), ay, bny), tnz = __builtin_fmaf(ubyte_f(nzp, j), az, bnz);
        const float tfx = __builtin_fmaf(ubyte_f(fxp, j), ax, bfx), tfy = __builtin_fmaf(ubyte_f(fyp, j), ay, bfy), tfz = __builtin_fmaf(ubyte_f(fzp, j), az, bfz);
        const float t0 = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), tmin);           // NaN planes (0 * inf) are ignored by max/min
        const float t1 = fminf(fminf(fminf(tfx, tfy), tfz), tmaxW);
        N.tn[k] = t0;
        hit |= (t0 <= t1) ? (1u << k) : 0u;
    }
    // A slot without a child may come out as "hit": it is not an inner node (imask) and its nibble of leafTris is 0, so it adds
    // neither a node to visit nor a triangle to test -- no separate existence mask is needed (dropping it: +3 % rays/s).
    // (v_pk_fma_f32 for the plane fmas -- child pairs in round 1, the near/far pair of one child in round 2, 24 packed for 48
    // scalar fmas and 5 % fewer VALU instructions in the kernel -- measured -7 % and -9 %: a packed fp32 fma issues no faster
    // than the two it replaces on this part, and the register pairs cost occupancy or spills.  Likewise v_perm_b32 building f16
    // pairs (1024 + q) for v_fma_mix_f32 -- 24 perms + 48 mixed fmas for 48 converts + 48 fmas: shadow 9.1 -> 9.8 ms.)
    N.hitMask = hit;
}
DI uint32_t node8_child(const Node8& N, uint32_t k) { return N.childBase + k; }      // SLOT ORDER: inner children sit in slots 0 .. n-1
// bit i set <=> triangle record triBase + i belongs to a leaf child whose box the ray overlaps (device_types.h LEAF SLOTS): every
// hit bit widened to its nibble (three shift-or-mask rounds and a multiplication by 15), ANDed with the node's triangle mask.
// Inner and empty slots have empty nibbles, so their hit bits need no masking.
DI uint32_t spread_nibbles(uint32_t x) {          // bit k (k < 8) -> bits 4 k .. 4 k + 3
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    // bit 4 k -> nibble k: two shift-ors, inline so that the optimiser cannot fold them into the x * 15 it makes of (x << 4) - x (measured: shadow 7.16 -> 7.05 ms per frame alone)
    uint32_t y, r;
    asm("v_lshl_or_b32 %0, %1, 1, %1" : "=v"(y) : "v"(x));
    asm("v_lshl_or_b32 %0, %1, 2, %1" : "=v"(r) : "v"(y));
    return r;
}
DI uint32_t node8_tri_mask(const Node8& N) { return spread_nibbles(N.hitMask) & N.leafTris; }
// the same for a WAVE-UNIFORM hit mask (the packet walks): plain C, so that the value stays in scalar registers (the inline assembly above would move it to vector ones and
// with it the packet's triangle loop and its loads)
DI uint32_t node8_tri_mask_uniform(uint32_t hitMask, uint32_t leafTris) {
    uint32_t x = hitMask;
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return ((x << 4) - x) & leafTris;
}

// One fetch per step: 80 bytes at the node, or 48 at the pending triangle (the last two dwordx4 are only issued by lanes on a
// node step).  Measured on MI355X (tools/microbench/gather_lanes.hip): a divergent dwordx4 gather costs the vector L1 one
// slot per ACTIVE lane plus a fixed ~22 lanes' worth per wave instruction -- 137 G 80-B records/s chip-wide with all lanes
// on, the unit this traversal is bound by (idle lanes are nearly free, instructions are not); the same gather out of LDS
// runs at 461 G records/s.  So the hottest nodes (index < topCount; ~40 % of all node visits for the top three levels of
// Sponza-stat) are served from a per-workgroup LDS copy when the stack type says so.
// Record addresses in 32 bits.  `base + index * 80` as 64-bit arithmetic compiles to v_mad_u64_u32, which issues at HALF rate on gfx950 (5.5 against 2.7 cycles per SIMD slot,
// tools/microbench/issue_rate.hip; v_mul_lo_u32 is full rate) and leaves a 64-bit address pair in vector registers; a traversal iteration made two of them plus a separate
// multiplication for the LDS copy of the node.  A record's BYTE OFFSET fits 32 bits (2^23 nodes x 80 B; the triangle records of a scene the 32-bit indices can name): one
// v_mul_lo_u32, shared by the LDS and the HBM path, and the loads take the scalar-base + 32-bit-vector-offset form (global_load ... v_off, s[base:base+1]).
DI uint32_t node_byte_offset(uint32_t node) { return node * 80u; }
DI uint32_t tri_byte_offset(uint32_t tri) { return tri * 48u; }
DI const float4* node_ptr(const PtParams& P, uint32_t node) { return (const float4*)((const char*)P.nodes + (size_t)node_byte_offset(node)); }
DI const float4* tri_ptr(const PtParams& P, uint32_t tri) { return (const float4*)((const char*)P.tris + (size_t)tri_byte_offset(tri)); }
// (Same box, ms per 4K frame with every kernel alone, shadow / extend: 64-bit addresses 7.36 / 6.93; these 7.05 / 6.89; the offsets as inline shift-adds 7.25 / 6.89:
//  profiles/r04_ab_lane_fill.txt.)
// (Round 5, measured and removed: the LDS branch and the memory branch below write the same registers, so the compiler makes every ds_read of the LDS branch wait for ALL the
//  memory loads of the other lanes -- s_waitcnt vmcnt(0) in front of it: a lane served from LDS pays the memory latency and then an LDS round trip.  Issuing the LDS reads FIRST, then
//  the triangle loads, then the other lanes' node loads removes that wait, but the register allocation it produces puts the ds_read results where the refill's visibility store keeps
//  its address, and the store's hazard becomes a vmcnt(0) in front of the step that also waits for the refill's queue loads -- the software pipelining of wf_shadow_kernel is gone:
//  any-hit 7.77-7.86 ms per frame alone against 7.34-7.62, closest-hit unchanged, frame 16.70-16.76 against 16.52-16.58 ms: profiles/r05_ab_fetch_order.txt.)
struct Fetch5 { float4 a, b, c, d, e; };
template <class ST> DI Fetch5 fetch_step(const PtParams& P, const ST& st, uint32_t cur, uint32_t triBase, uint32_t triMask) {
    Fetch5 f;
    if constexpr (ST::kTop) {
        if (!triMask && cur < st.topCount) {
            LdsNodes q = (LdsNodes)((__attribute__((address_space(3))) const char*)st.topNodes + node_byte_offset(cur));
            const incr_v4f a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
            f.a = make_float4(a.x, a.y, a.z, a.w); f.b = make_float4(b.x, b.y, b.z, b.w); f.c = make_float4(c.x, c.y, c.z, c.w);
            f.d = make_float4(d.x, d.y, d.z, d.w); f.e = make_float4(e.x, e.y, e.z, e.w);
            return f;
        }
    }
    const uint32_t ti = triBase + (uint32_t)__builtin_ctz(triMask | 0x80000000u);      // 32-bit sum: triBase is biased modulo 2^32 (device_types.h LEAF SLOTS)
    const float4* __restrict__ p = triMask ? tri_ptr(P, ti) : node_ptr(P, cur);
    f.a = p[0]; f.b = p[1]; f.c = p[2];
    if (!triMask) { f.d = p[3]; f.e = p[4]; } else { f.d = f.c; f.e = f.c; }
    return f;
}

// ---------------------------------------------------------------- closest hit
struct TravClosest {
    RaySetup rs; f3 d; float tmin, tmax;
    float best, bu, bv; uint32_t bestFlat; bool found;      // bestFlat = the hit's flattened triangle id = its TriShade index (device_types.h)
    uint32_t cur; float curKey;          // next node to visit (valid when haveCur)
    uint32_t triBase, triMask;           // pending triangle tests of the node visited last
    uint32_t negDir;                     // bit a: the ray points down axis a (selects the walk direction through a node's sorted slots)
    bool haveCur;
#ifdef INCR_TRAV_STATS
    uint32_t nNodes = 0, nTris = 0;      // developer build only: traversal statistics through incr_trace_rays
#endif
};
DI void trav_closest_begin(TravClosest& T, f3 o, f3 d, float tmin, float tmax) {
    T.rs = ray_setup(o, d); T.d = d; T.tmin = tmin; T.tmax = tmax;
    T.best = tmax; T.bu = 0.f; T.bv = 0.f; T.bestFlat = 0xffffffffu; T.found = false;
    T.cur = 0; T.curKey = 0.f; T.haveCur = true; T.triBase = 0; T.triMask = 0;
    T.negDir = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
}
// returns true when the traversal is complete (T.found / T.best / T.bu / T.bv / T.bestFlat hold the result)
template <class ST> DI bool trav_closest_step(const PtParams& P, ST& st, TravClosest& T) {
    const Fetch5 f = fetch_step(P, st, T.cur, T.triBase, T.triMask);
    if (T.triMask) {
        // ---- one pending triangle (fetch_step read the record of the lowest set bit) ----
        T.triMask &= T.triMask - 1u;
#ifdef INCR_TRAV_STATS
        T.nTris++;
#endif
        float t, u, v; uint32_t flat, flags;
        if (tri_intersect(f.a, f.b, f.c, T.rs.o, T.d, T.tmin, T.tmax, &t, &u, &v, &flat, &flags)) {
            // closest hit; equal-t ties go to the smaller flattened id (traversal-order independent)
            const bool better = T.found ? (t < T.best || (t == T.best && flat < T.bestFlat)) : (t < T.best);
            if (better) {
                if (flags & kTriMirrored) { const float tmp = u; u = v; v = tmp; }
                if ((flags & kTriOpaque) || any_hit_accepts(P, flat, u, v)) { T.found = true; T.best = t; T.bestFlat = flat; T.bu = u; T.bv = v; }
            }
        }
        if (T.triMask) return false;
    } else {
        // ---- one 8-wide node ----
#ifdef INCR_TRAV_STATS
        T.nNodes++;
#endif
        Node8 N;
        node8_test(f.a, f.b, f.c, f.d, f.e, T.rs, T.tmin, T.best, N);
        T.triBase = N.triBase;
        T.triMask = node8_tri_mask(N);
        // inner children: the nearest is visited next (after this node's triangles); the others wait on the stack as ONE
        // entry keyed by the smallest of their entry distances and leave it in slot order -- the builder sorted the slots
        // along the node's longest axis, `reverse` walks them from the ray's side.  Against exact far-to-near pushes this
        // visits 4-7 % more nodes on Sponza-stat, and saves the selection sort (up to 7 rounds of 8 compares per node).
        const uint32_t inner = N.hitMask & N.imask;
        T.haveCur = inner != 0u;
        if (inner) {
            // smallest and second-smallest entry distance among the overlapped inner children in 5 instructions per slot: the
            // distances are positive floats, so their bit patterns order like unsigned integers; the low three mantissa bits
            // carry the slot number (the distance rounds DOWN by < 8 ulp: keys stay lower bounds), slots that are not
            // overlapped inner children get the all-ones key.  (Two compare-and-select scans cost 98 VALU instructions per
            // node step against 46 -- and the traversal kernels are VALU-bound.)
            uint32_t m1 = 0xffffffffu, m2 = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < INCR_BVH_WIDTH; ++k) {
                const uint32_t fill = (uint32_t)__builtin_amdgcn_sbfe((int)~inner, (uint32_t)k, 1u);      // 0 or ~0
                const uint32_t key = ((__float_as_uint(N.tn[k]) & ~7u) | (uint32_t)k) | fill;
                m2 = umed3(m1, m2, key);
                m1 = min(m1, key);
            }
            const uint32_t nearK = m1 & 7u;
            const uint32_t rest = inner & ~(1u << nearK);
            if (rest) {
                const uint32_t axis = N.sortAxis;
                st.push(stack_entry(N.childBase, (T.negDir >> axis) & 1u, rest), __uint_as_float(m2 & ~7u));
            }
            T.cur = N.childBase + nearK; T.curKey = __uint_as_float(m1 & ~7u);
        }
        if (T.triMask) return false;
    }
    // ---- choose the next node: the nearest child if it still starts before the best hit, else the stack ----
    if (T.haveCur) { T.haveCur = false; if (T.curKey <= T.best) return false; }      // equal-t ties must still be visited
    while (st.sp > 0) {
        uint32_t e = st.top();
        if constexpr (ST::kKeyed) { if (st.topKey() > T.best) { st.drop(); continue; } }
        const uint32_t m = e & 0xffu;
        const uint32_t k = (e & 0x100u) ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
        T.cur = (e >> 9) + k;
        e &= ~(1u << k);
        if (e & 0xffu) st.setTop(e); else st.drop();
        return false;
    }
    return true;
}
// NOTE on haveCur: after the node branch sets cur/haveCur, the "choose" block consumes it exactly once.  When the
// triangles of the node come first, haveCur stays set until the last pending triangle has been tested (the fetch of a
// triangle step does not use T.cur).

// Node step and triangle step of a lane in the SAME iteration (wf_extend_kernel).  The triangles a node step finds wait in a two-entry
// queue (triBase/triMask = front, triBase1/triMask1 = back) and are tested one per iteration while the lane goes on to its next node; a
// lane only stalls its node steps while both entries are taken.  Closest hit does not depend on the order in which triangles are
// tested.  Both branches of the loop body then run with more of the wave's lanes: the wave model (tools/trav_sim wave_model_overlap)
// said -17 ... -21 % instructions per step, the kernel went from 2.26 to 1.95 ms per launch (extend 8.6 -> 7.7 ms per frame).
struct TravClosest2 : TravClosest { uint32_t triBase1, triMask1; };
// (GATED TRIANGLE ROUNDS -- the triangle branch of an iteration taken only when at least G lanes of the wave have a triangle waiting, or no lane can take a node step --
// were measured in round 4: 4K frame 16.69 ms ungated, 17.35 / 17.71 / 18.54 at G = 16 / 24 / 32: the triangles a lane has not tested yet are hits it has not found yet,
// an any-hit ray that would have ended goes on walking and a closest-hit ray goes on with too large a tmax.  profiles/r04_ab_lane_fill.txt.)
template <class ST> DI bool trav_closest_step2(const PtParams& P, ST& st, TravClosest2& T) {
    const bool wantTri = T.triMask != 0u;
    const bool wantNode = T.haveCur && T.triMask1 == 0u;
    // ---- both fetches first
    float4 ta, tb, tc;                   // deliberately not initialised: only read under the condition they are loaded under (zeroing cost 12 v_mov per iteration)
    uint32_t ti = 0;
    if (wantTri) { ti = T.triBase + (uint32_t)__builtin_ctz(T.triMask); const float4* __restrict__ tp = tri_ptr(P, ti); ta = tp[0]; tb = tp[1]; tc = tp[2]; }
    Fetch5 f;                            // as above (20 v_mov)
    if (wantNode) f = fetch_step(P, st, T.cur, 0u, 0u);
    // ---- one pending triangle
    if (wantTri) {
        T.triMask &= T.triMask - 1u;
        float t, u, v; uint32_t flat, flags;
        if (tri_intersect(ta, tb, tc, T.rs.o, T.d, T.tmin, T.tmax, &t, &u, &v, &flat, &flags)) {
            const bool better = T.found ? (t < T.best || (t == T.best && flat < T.bestFlat)) : (t < T.best);
            if (better) {
                if (flags & kTriMirrored) { const float tmp = u; u = v; v = tmp; }
                if ((flags & kTriOpaque) || any_hit_accepts(P, flat, u, v)) { T.found = true; T.best = t; T.bestFlat = flat; T.bu = u; T.bv = v; }
            }
        }
        if (!T.triMask) { T.triBase = T.triBase1; T.triMask = T.triMask1; T.triMask1 = 0u; }
    }
    // ---- one node
    if (wantNode) {
        Node8 N;
        node8_test(f.a, f.b, f.c, f.d, f.e, T.rs, T.tmin, T.best, N);
        const uint32_t tm = node8_tri_mask(N);
        if (tm) { if (!T.triMask) { T.triBase = N.triBase; T.triMask = tm; } else { T.triBase1 = N.triBase; T.triMask1 = tm; } }
        const uint32_t inner = N.hitMask & N.imask;
        T.haveCur = false;
        if (inner) {
            uint32_t m1 = 0xffffffffu, m2 = 0xffffffffu;
#pragma unroll
            for (int k = 0; k < INCR_BVH_WIDTH; ++k) {
                const uint32_t fill = (uint32_t)__builtin_amdgcn_sbfe((int)~inner, (uint32_t)k, 1u);
                const uint32_t key = ((__float_as_uint(N.tn[k]) & ~7u) | (uint32_t)k) | fill;
                m2 = umed3(m1, m2, key);
                m1 = min(m1, key);
            }
            const uint32_t nearK = m1 & 7u;
            const uint32_t rest = inner & ~(1u << nearK);
            if (rest) {
                const uint32_t axis = N.sortAxis;
                st.push(stack_entry(N.childBase, (T.negDir >> axis) & 1u, rest), __uint_as_float(m2 & ~7u));
            }
            T.cur = N.childBase + nearK; T.curKey = __uint_as_float(m1 & ~7u);
            T.haveCur = T.curKey <= T.best;                          // the node's own triangles are still queued: best may shrink later
        }
        // next node: the nearest child, else the stack
        while (!T.haveCur && st.sp > 0) {
            uint32_t e = st.top();
            if constexpr (ST::kKeyed) { if (st.topKey() > T.best) { st.drop(); continue; } }
            const uint32_t m = e & 0xffu;
            const uint32_t k = (e & 0x100u) ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
            T.cur = (e >> 9) + k;
            e &= ~(1u << k);
            if (e & 0xffu) st.setTop(e); else st.drop();
            T.haveCur = true;
        }
    }
    return !T.haveCur && !T.triMask;
}

// TraceRay(RAY_FLAG_CULL_BACK_FACING_TRIANGLES): closest hit
template <class ST> DI bool trace_closest(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax, Hit* hit) {
    TravClosest2 T;
    trav_closest_begin(T, o, d, tmin, tmax);          // haveCur: the root is the first node to visit
    T.triBase1 = 0; T.triMask1 = 0;
    st.reset();
    while (!trav_closest_step2(P, st, T)) {}
    if (T.found) { hit->t = T.best; hit->u = T.bu; hit->v = T.bv; hit->tri = T.bestFlat; }
    return T.found;
}

// ---------------------------------------------------------------- any hit (shadow rays)
struct TravAny { RaySetup rs; f3 d; float tmin, tmax; uint32_t cur, triBase, triMask; bool occluded, haveCur; };
DI void trav_any_begin(TravAny& T, f3 o, f3 d, float tmin, float tmax) {
    T.rs = ray_setup(o, d); T.d = d; T.tmin = tmin; T.tmax = tmax; T.cur = 0; T.triBase = 0; T.triMask = 0; T.occluded = false; T.haveCur = false;
}
// The same overlap for any-hit queries (wf_shadow_kernel).  A hit ends the query, so a triangle waiting in the queue costs the node
// steps made while it waits when it turns out to be the occluder -- measured, the fuller lanes win: shadow 8.9 -> 7.9 ms per frame.
struct TravAny2 : TravAny { uint32_t triBase1, triMask1; };
template <class ST> DI bool trav_any_step2(const PtParams& P, ST& st, TravAny2& T) {
    const bool wantTri = T.triMask != 0u;
    const bool wantNode = T.haveCur && T.triMask1 == 0u;
    float4 ta, tb, tc;                   // deliberately not initialised: only read under the condition they are loaded under (zeroing cost 12 v_mov per iteration)
    if (wantTri) { const uint32_t ti = T.triBase + (uint32_t)__builtin_ctz(T.triMask); const float4* __restrict__ tp = tri_ptr(P, ti); ta = tp[0]; tb = tp[1]; tc = tp[2]; }      // 32-bit sum: triBase is biased modulo 2^32
    Fetch5 f;                            // as above (20 v_mov)
    if (wantNode) f = fetch_step(P, st, T.cur, 0u, 0u);
#ifdef INCR_DIAG_VMEM      // sensitivity diagnostic: every traversal fetch issued twice (the second hits the line the first brought in)
    { incr_v4f da, db, dc, dd, de, dta, dtb, dtc;
      if (wantTri) { const uint32_t ti = T.triBase + (uint32_t)__builtin_ctz(T.triMask); const volatile incr_v4f* tp = (const volatile incr_v4f*)tri_ptr(P, ti); dta = tp[0]; dtb = tp[1]; dtc = tp[2]; asm volatile("" :: "v"(dta.x), "v"(dtb.x), "v"(dtc.x)); }
      if (wantNode && !(T.cur < st.topCount)) { const volatile incr_v4f* np = (const volatile incr_v4f*)node_ptr(P, T.cur); da = np[0]; db = np[1]; dc = np[2]; dd = np[3]; de = np[4]; asm volatile("" :: "v"(da.x), "v"(db.x), "v"(dc.x), "v"(dd.x), "v"(de.x)); } }
#endif
#ifdef INCR_DIAG_VALU      // sensitivity diagnostic: INCR_DIAG_VALU x 10 extra vector instructions per iteration
    { float dv = T.rs.o.x;
#pragma unroll
      for (int q = 0; q < INCR_DIAG_VALU; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_max_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_max_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_max_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_max_f32 %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_max_f32 %0, %0, %0" : "+v"(dv)); }
#endif
    if (wantTri) {
        T.triMask &= T.triMask - 1u;
        float det, U, V, Tn;
        if (tri_hit(ta, tb, tc, T.rs.o, T.d, T.tmin, T.tmax, &det, &U, &V, &Tn)) { T.occluded = true; return true; }
        if (!T.triMask) { T.triBase = T.triBase1; T.triMask = T.triMask1; T.triMask1 = 0u; }
    }
    if (wantNode) {
        Node8 N;
        node8_test(f.a, f.b, f.c, f.d, f.e, T.rs, T.tmin, T.tmax, N);
        const uint32_t tm = node8_tri_mask(N);
        if (tm) { if (!T.triMask) { T.triBase = N.triBase; T.triMask = tm; } else { T.triBase1 = N.triBase; T.triMask1 = tm; } }
        const uint32_t inner = N.hitMask & N.imask;
        T.haveCur = inner != 0u;
        if (inner) {
            const uint32_t rest = inner & (inner - 1u);
            if (rest) st.push(stack_entry(N.childBase, false, rest));
            T.cur = N.childBase + (uint32_t)__builtin_ctz(inner);
        } else if (st.sp > 0) {
            uint32_t e = st.top();
            T.cur = (e >> 9) + (uint32_t)__builtin_ctz(e & 0xffu);
            e &= e - 1u;
            if (e & 0xffu) st.setTop(e); else st.drop();
            T.haveCur = true;
        }
    }
    return !T.haveCur && !T.triMask;
}

// RayQuery<CULL_BACK_FACING> + one Proceed(): occluded iff any front-facing triangle lies in (tmin, tmax)
// (lighting/common.hlsli:102-115, 161-174).
template <class ST> DI bool trace_any(const PtParams& P, ST& st, f3 o, f3 d, float tmin, float tmax) {
    TravAny2 T;
    trav_any_begin(T, o, d, tmin, tmax);
    T.haveCur = true; T.triBase1 = 0; T.triMask1 = 0;
    st.reset();
    while (!trav_any_step2(P, st, T)) {}
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
