// trav_sim.cpp -- developer tool (not product, not oracle): CPU model of the GPU traversal state machines of
// csrc/pt_traverse.h over the REAL 8-wide BVH that scene_build.cpp produces, to compare stack disciplines and child
// orders by step counts and stack depths before spending GPU time.  Built by tools/trav_sim/run.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../incarnation-renderer_amd/csrc/scene_build.h"

using namespace incr;

namespace {
struct Ray { float o[3], d[3], tmin, tmax; };
struct NodeT { uint32_t hit; float tn[8]; };

static void node_test(const BvhNode8& n, const Ray& r, const float invd[3], float tmax, NodeT& N) {
    const uint8_t eb[3] = {n.ex, n.ey, n.ez};
    const float org[3] = {n.ox, n.oy, n.oz};
    const uint8_t* qlo[3] = {n.qlox, n.qloy, n.qloz}; const uint8_t* qhi[3] = {n.qhix, n.qhiy, n.qhiz};
    float a[3], bn[3], bf[3]; bool pos[3];
    for (int k = 0; k < 3; ++k) {
        uint32_t bits = (uint32_t)eb[k] << 23; float sc; memcpy(&sc, &bits, 4);
        a[k] = sc * invd[k];
        const float b = (org[k] - r.o[k]) * invd[k];
        const float m = 0.02f * fabsf(a[k]) + 2.4e-7f * fabsf(b);
        bn[k] = b - m; bf[k] = b + m; pos[k] = a[k] >= 0.f;
    }
    N.hit = 0;
    for (int c = 0; c < 8; ++c) {
        float t0 = r.tmin, t1 = tmax;
        for (int k = 0; k < 3; ++k) {
            const float qn = (float)(pos[k] ? qlo[k][c] : qhi[k][c]), qf = (float)(pos[k] ? qhi[k][c] : qlo[k][c]);
            const float tn = fmaf(qn, a[k], bn[k]), tf = fmaf(qf, a[k], bf[k]);
            t0 = fmaxf(t0, tn); t1 = fminf(t1, tf);      // fmaxf/fminf ignore NaN like v_max/v_min
        }
        t1 *= 1.0000004f;
        N.tn[c] = t0;
        if (t0 <= t1) N.hit |= 1u << c;
    }
}
static uint32_t tri_mask(const BvhNode8& n, uint32_t hit) {          // device_types.h LEAF SLOTS: hit bits widened to nibbles
    uint32_t m = 0;
    for (int k = 0; k < 8; ++k) if ((hit >> k) & 1u) m |= 15u << (4 * k);
    return m & n.leafTris;
}
static bool tri_isect(const TriIsect& T, const Ray& r, float tmax, float* t) {
    const float e1[3] = {T.e1x, T.e1y, T.e1z}, e2[3] = {T.e2x, T.e2y, T.e2z}, v0[3] = {T.v0x, T.v0y, T.v0z};
    const float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (!(det > 0.f)) return false;
    const float inv = 1.f / det;
    const float sv[3] = {r.o[0] - v0[0], r.o[1] - v0[1], r.o[2] - v0[2]};
    const float u = (sv[0] * pv[0] + sv[1] * pv[1] + sv[2] * pv[2]) * inv;
    if (!(u >= 0.f && u <= 1.f)) return false;
    const float q[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
    const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * inv;
    if (!(v >= 0.f && u + v <= 1.f)) return false;
    const float tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (!(tt > r.tmin && tt < tmax)) return false;
    *t = tt; return true;
}
static uint32_t child_index(const BvhNode8& n, uint32_t k) { return n.childBase + (uint32_t)__builtin_popcount(n.imask & ((1u << k) - 1u)); }

// per-ray step traces for the wave-scheduling model: one byte per node visit = the triangle tests that followed it
struct Trace { std::vector<uint32_t> start; std::vector<uint8_t> tris; void begin() { start.push_back((uint32_t)tris.size()); } };
static Trace* g_trace = nullptr;
static std::vector<uint8_t> g_nodeDepth; static double g_visitsByDepth[2][32];
static std::vector<double> g_visitsByIndex[2];          // round 5: how many node visits do the K lowest node indices (NODE ORDER: the LDS copy of the traversal kernels) take?
struct Stats { double rays = 0, nodes = 0, tris = 0, pushes = 0, hits = 0; std::vector<uint32_t> depthHist = std::vector<uint32_t>(128, 0); };
static void note(Stats& s, uint32_t nodes, uint32_t tris, uint32_t pushes, uint32_t maxDepth, bool hit) {
    s.rays++; s.nodes += nodes; s.tris += tris; s.pushes += pushes; s.hits += hit; s.depthHist[std::min(maxDepth, 127u)]++;
}

struct Result { bool found; float t; uint32_t tri; };

// ---- closest hit -------------------------------------------------------------------------------------
// variant 0: per-child keyed entries pushed far-to-near (round 1).
// variant 1: nearest child first; the other overlapped inner children as ONE group entry (childBase | mask) with the
//            minimum entry distance of the group as its key; children of a popped group leave in slot order.
// variant 2: as 1, slot order reversed when the ray points down the node's sort axis (stored in bits of the entry).
// variant 3: groups only, no nearest-first (slot order with reversal).
static Result closest(const HostBvh& B, const std::vector<uint8_t>& sortAxis, const Ray& r, int variant, Stats& S) {
    const float invd[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]};
    float best = r.tmax; bool found = false; uint32_t bestTri = 0, bestFlat = 0xffffffffu;
    uint32_t nodes = 0, tris = 0, pushes = 0, maxDepth = 0;
    struct E { uint32_t node; float key; uint32_t base, mask; bool rev; };
    std::vector<E> st;
    auto test_tris = [&](const BvhNode8& n, uint32_t tm) {
        while (tm) {
            const uint32_t ti = n.triBase + (uint32_t)__builtin_ctz(tm); tm &= tm - 1u; ++tris;
            float t;
            if (tri_isect(B.tris[ti], r, r.tmax, &t)) {
                const uint32_t flat = B.tris[ti].flatId;
                const bool better = found ? (t < best || (t == best && flat < bestFlat)) : (t < best);
                if (better) { found = true; best = t; bestTri = ti; bestFlat = flat; }
            }
        }
    };
    uint32_t cur = 0; bool have = true;
    if (g_trace) g_trace->begin();
    for (;;) {
        if (!have) {
            bool got = false;
            while (!st.empty()) {
                E& e = st.back();
                if (variant == 0) { const E x = e; st.pop_back(); if (x.key > best) continue; cur = x.node; got = true; break; }
                if (e.key > best || e.mask == 0) { st.pop_back(); continue; }
                const uint32_t k = e.rev ? 31u - (uint32_t)__builtin_clz(e.mask) : (uint32_t)__builtin_ctz(e.mask);
                e.mask &= ~(1u << k);
                cur = e.base + k;          // requires inner-first slot order (see reorder())
                if (e.mask == 0) st.pop_back();
                got = true; break;
            }
            if (!got) break;
        }
        have = false;
        const BvhNode8& n = B.nodes8[cur]; ++nodes; if (!g_nodeDepth.empty()) { g_visitsByDepth[0][g_nodeDepth[cur]]++; g_visitsByIndex[0][cur]++; }
        NodeT N; node_test(n, r, invd, best, N);
        if (g_trace) g_trace->tris.push_back((uint8_t)__builtin_popcount(tri_mask(n, N.hit)));
        test_tris(n, tri_mask(n, N.hit));
        uint32_t inner = N.hit & n.imask;
        if (!inner) continue;
        if (variant == 0) {
            uint32_t nearK = 8; float nearT = INFINITY;
            for (int k = 0; k < 8; ++k) if (((inner >> k) & 1u) && N.tn[k] < nearT) { nearT = N.tn[k]; nearK = k; }
            uint32_t rest = inner & ~(1u << nearK);
            while (rest) {
                float farT = -1.f; uint32_t farK = 0;
                for (int k = 0; k < 8; ++k) if (((rest >> k) & 1u) && N.tn[k] > farT) { farT = N.tn[k]; farK = k; }
                st.push_back({child_index(n, farK), farT, 0, 0, false}); ++pushes;
                rest &= ~(1u << farK);
            }
            maxDepth = std::max<uint32_t>(maxDepth, st.size());
            if (nearT <= best) { cur = child_index(n, nearK); have = true; }
        } else {
            const int ax = sortAxis[cur];
            const bool rev = (variant >= 2) && r.d[ax] < 0.f;
            uint32_t rest = inner;
            if (variant != 3) {
                uint32_t nearK = 8; float nearT = INFINITY;
                for (int k = 0; k < 8; ++k) if (((inner >> k) & 1u) && N.tn[k] < nearT) { nearT = N.tn[k]; nearK = k; }
                rest &= ~(1u << nearK);
                if (nearT <= best) { cur = n.childBase + nearK; have = true; }
            }
            if (rest) {
                float mk = INFINITY;
                for (int k = 0; k < 8; ++k) if ((rest >> k) & 1u) mk = std::min(mk, N.tn[k]);
                st.push_back({0, mk, n.childBase, rest, rev}); ++pushes;
                maxDepth = std::max<uint32_t>(maxDepth, st.size());
            }
        }
    }
    note(S, nodes, tris, pushes, maxDepth, found);
    return {found, best, bestTri};
}

// ---- any hit ----------------------------------------------------------------------------------------
// variant 0: one entry per overlapped inner child (round 1).  variant 1: one group entry per node.
static std::vector<uint8_t> g_areaOrder;      // per node: its 8 slots by descending child-box area (any-hit experiment)
static void build_area_order(const HostBvh& B) {
    g_areaOrder.assign(B.nodes8.size() * 8, 0);
    for (size_t i = 0; i < B.nodes8.size(); ++i) {
        const BvhNode8& n = B.nodes8[i];
        float sc[3]; const uint8_t eb[3] = {n.ex, n.ey, n.ez};
        for (int a = 0; a < 3; ++a) { uint32_t bits = (uint32_t)eb[a] << 23; memcpy(&sc[a], &bits, 4); }
        float area[8]; int ord[8];
        for (int k = 0; k < 8; ++k) {
            const float dx = ((int)n.qhix[k] - (int)n.qlox[k]) * sc[0], dy = ((int)n.qhiy[k] - (int)n.qloy[k]) * sc[1], dz = ((int)n.qhiz[k] - (int)n.qloz[k]) * sc[2];
            area[k] = (dx < 0 || dy < 0 || dz < 0) ? -1.f : dx * dy + dy * dz + dz * dx; ord[k] = k;
        }
        std::stable_sort(ord, ord + 8, [&](int a, int b) { return area[a] > area[b]; });
        for (int k = 0; k < 8; ++k) g_areaOrder[i * 8 + k] = (uint8_t)ord[k];
    }
}
static bool anyhit_area(const HostBvh& B, const Ray& r, Stats& S) {
    const float invd[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]};
    uint32_t nodes = 0, tris = 0, pushes = 0, maxDepth = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;      // (node whose children wait, mask)
    uint32_t cur = 0; bool have = true, occluded = false;
    for (;;) {
        if (!have) {
            if (st.empty()) break;
            auto& e = st.back();
            uint32_t k = 8;
            for (int j = 0; j < 8; ++j) { const uint32_t c = g_areaOrder[(size_t)e.first * 8 + j]; if ((e.second >> c) & 1u) { k = c; break; } }
            cur = B.nodes8[e.first].childBase + k; e.second &= ~(1u << k);
            if (!e.second) st.pop_back();
        }
        have = false;
        const uint32_t me = cur;
        const BvhNode8& n = B.nodes8[cur]; ++nodes;
        NodeT N; node_test(n, r, invd, r.tmax, N);
        uint32_t tm = tri_mask(n, N.hit);
        while (tm && !occluded) { const uint32_t ti = n.triBase + (uint32_t)__builtin_ctz(tm); tm &= tm - 1u; ++tris; float t; if (tri_isect(B.tris[ti], r, r.tmax, &t)) occluded = true; }
        if (occluded) break;
        uint32_t inner = N.hit & n.imask;
        if (!inner) continue;
        uint32_t k0 = 8;
        for (int j = 0; j < 8; ++j) { const uint32_t c = g_areaOrder[(size_t)me * 8 + j]; if ((inner >> c) & 1u) { k0 = c; break; } }
        inner &= ~(1u << k0);
        if (inner) { st.push_back({me, inner}); ++pushes; }
        cur = n.childBase + k0; have = true;
        maxDepth = std::max<uint32_t>(maxDepth, st.size());
    }
    note(S, nodes, tris, pushes, maxDepth, occluded);
    return occluded;
}
static uint32_t g_occTri = 0xffffffffu;
static bool anyhit(const HostBvh& B, const std::vector<uint8_t>& sortAxis, const Ray& r, int variant, Stats& S) {
    const float invd[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]};
    uint32_t nodes = 0, tris = 0, pushes = 0, maxDepth = 0;
    std::vector<uint32_t> st;        // variant 0: node indices; variant 1: base << 8 | mask
    uint32_t cur = 0; bool have = true, occluded = false;
    if (g_trace) g_trace->begin();
    for (;;) {
        if (!have) {
            if (st.empty()) break;
            if (variant == 0) { cur = st.back(); st.pop_back(); }
            else {
                uint32_t& e = st.back();
                const bool rev = (e >> 31) & 1u;
                const uint32_t m = e & 0xffu;
                const uint32_t k = rev ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
                cur = ((e & 0x7fffffffu) >> 8) + k;
                e &= ~(1u << k);
                if ((e & 0xffu) == 0) st.pop_back();
            }
        }
        have = false;
        const BvhNode8& n = B.nodes8[cur]; ++nodes; if (!g_nodeDepth.empty()) { g_visitsByDepth[1][g_nodeDepth[cur]]++; g_visitsByIndex[1][cur]++; }
        NodeT N; node_test(n, r, invd, r.tmax, N);
        uint32_t tm = tri_mask(n, N.hit);
        const uint32_t trisBefore = tris;
        while (tm && !occluded) { const uint32_t ti = n.triBase + (uint32_t)__builtin_ctz(tm); tm &= tm - 1u; ++tris; float t; if (tri_isect(B.tris[ti], r, r.tmax, &t)) { occluded = true; g_occTri = ti; } }
        if (g_trace) g_trace->tris.push_back((uint8_t)(tris - trisBefore));
        if (occluded) break;
        uint32_t inner = N.hit & n.imask;
        if (!inner) continue;
        if (variant == 0) {
            const uint32_t k0 = (uint32_t)__builtin_ctz(inner); inner &= inner - 1u;
            while (inner) { const uint32_t k = (uint32_t)__builtin_ctz(inner); inner &= inner - 1u; st.push_back(child_index(n, k)); ++pushes; }
            cur = child_index(n, k0); have = true;
        } else {
            const bool rev = variant == 3 && r.d[sortAxis[cur]] < 0.f;
            uint32_t k0 = rev ? 31u - (uint32_t)__builtin_clz(inner) : (uint32_t)__builtin_ctz(inner);
            if (variant == 2) { float nt = INFINITY; for (int k = 0; k < 8; ++k) if (((inner >> k) & 1u) && N.tn[k] < nt) { nt = N.tn[k]; k0 = k; } }
            inner &= ~(1u << k0);
            if (inner) { st.push_back(n.childBase << 8 | inner | (rev ? 0x80000000u : 0u)); ++pushes; }
            cur = n.childBase + k0; have = true;
        }
        maxDepth = std::max<uint32_t>(maxDepth, st.size());
    }
    note(S, nodes, tris, pushes, maxDepth, occluded);
    return occluded;
}

// Slot order: inner children first (slot k of an inner child <=> node childBase + k), optionally sorted along the
// node's longest axis by child-box centre (sortAxis[node] is recorded for the reversal trick).
static void reorder(HostBvh& B, std::vector<uint8_t>& sortAxis, bool sortByAxis) {
    // the builder emits this order itself since round 2 (scene_build.cpp SLOT ORDER): only the axis table is left to fill
    (void)sortByAxis;
    sortAxis.assign(B.nodes8.size(), 0);
    for (size_t ni = 0; ni < B.nodes8.size(); ++ni) sortAxis[ni] = B.nodes8[ni].sortAxis;
}

// ---- wave-scheduling model ---------------------------------------------------------------------------
// Persistent 64-lane waves over a ray list (chunks of `chunk` rays from a shared cursor, refill when >= `refill` lanes idle).
// A lane's work is its trace: N, T x t0, N, T x t1, ...  Cost model in VALU instructions per wave iteration:
// policy 0 (if-if, pt_traverse.h today): every live lane takes its next step, cost = cC + cN [any node lane] + cT [any tri lane].
// policy 1 (phased): one kind of step per iteration -- the kind with more lanes waiting relative to `bias`; lanes of the other kind idle.
// policy 2 (threshold): both kinds when each has >= thr lanes, otherwise only the larger.
struct WaveModel { int policy; int chunk, refill; float cC, cN, cT, bias; int thr; };
static void wave_model(const char* name, const Trace& tr, size_t nWaves, const WaveModel& M) {
    const size_t nRays = tr.start.size();
    struct Lane { bool live = false; uint32_t pos = 0, end = 0; uint32_t trisLeft = 0; bool nodeNext = true; };
    struct Wave { Lane l[64]; size_t next = 0, end = 0; bool dry = false, done = false; };
    std::vector<Wave> waves(nWaves);
    size_t cursor = 0; double cost = 0, iters = 0, steps = 0, nodeLanes = 0, triLanes = 0, nodeIters = 0, triIters = 0;
    size_t alive = nWaves;
    while (alive) {
        for (Wave& w : waves) {
            if (w.done) continue;
            int live = 0; for (auto& L : w.l) live += L.live;
            if (!w.dry && (64 - live >= M.refill || live == 0)) {
                for (auto& L : w.l) {
                    if (L.live) continue;
                    if (w.next >= w.end) { if (cursor >= nRays) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)M.chunk, nRays); cursor = w.end; }
                    const size_t r = w.next++;
                    L.live = true; L.pos = tr.start[r]; L.end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); L.trisLeft = 0; L.nodeNext = true;
                    if (L.pos == L.end) L.live = false; else ++live;
                }
            }
            if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
            int nN = 0, nT = 0;
            for (auto& L : w.l) if (L.live) { if (L.trisLeft) ++nT; else ++nN; }
            bool doN = nN > 0, doT = nT > 0;
            if (M.policy == 1 && nN && nT) { if ((float)nN * M.bias >= (float)nT) doT = false; else doN = false; }
            if (M.policy == 2 && nN && nT) { if (nN < M.thr && nN <= nT) doN = false; else if (nT < M.thr && nT < nN) doT = false; }
            cost += M.cC + (doN ? M.cN : 0.f) + (doT ? M.cT : 0.f); iters++;
            if (doN) { nodeIters++; nodeLanes += nN; } if (doT) { triIters++; triLanes += nT; }
            for (auto& L : w.l) {
                if (!L.live) continue;
                if (L.trisLeft) { if (!doT) continue; --L.trisLeft; ++steps; }
                else { if (!doN) continue; L.trisLeft = tr.tris[L.pos++]; ++steps; }
                if (!L.trisLeft && L.pos >= L.end) L.live = false;
            }
        }
    }
    printf("  %-16s policy %d bias %.2f thr %2d refill %2d: cost/step %6.1f  iters %.3gM  node iters %.0f%% @ %.1f lanes, tri iters %.0f%% @ %.1f lanes\n", name, M.policy, M.bias, M.thr, M.refill,
           cost / steps, iters / 1e6, 100 * nodeIters / iters, nodeLanes / std::max(nodeIters, 1.0), 100 * triIters / iters, triLanes / std::max(triIters, 1.0));
}

// policy 3 (overlap): a lane's node steps run every iteration; the triangles a node step finds go into a per-lane queue of Q
// (triBase, mask) entries and are tested one per iteration ALONGSIDE the node steps, in iterations where >= thr lanes have
// one pending or some lane can do nothing else (queue full / no node left).  Triangle order does not change the result.
static void wave_model_overlap(const char* name, const Trace& tr, size_t nWaves, const WaveModel& M, int Q) {
    const size_t nRays = tr.start.size();
    struct Lane { bool live = false; uint32_t pos = 0, end = 0; uint32_t q[8]; int qn = 0; };
    struct Wave { Lane l[64]; size_t next = 0, end = 0; bool dry = false, done = false; };
    std::vector<Wave> waves(nWaves);
    size_t cursor = 0; double cost = 0, iters = 0, steps = 0, nodeLanes = 0, triLanes = 0, nodeIters = 0, triIters = 0, stalls = 0;
    size_t alive = nWaves;
    while (alive) {
        for (Wave& w : waves) {
            if (w.done) continue;
            int live = 0; for (auto& L : w.l) live += L.live;
            if (!w.dry && (64 - live >= M.refill || live == 0)) {
                for (auto& L : w.l) {
                    if (L.live) continue;
                    if (w.next >= w.end) { if (cursor >= nRays) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)M.chunk, nRays); cursor = w.end; }
                    const size_t r = w.next++;
                    L.live = true; L.pos = tr.start[r]; L.end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); L.qn = 0;
                    if (L.pos == L.end) L.live = false; else ++live;
                }
            }
            if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
            int nN = 0, nT = 0, must = 0;
            for (auto& L : w.l) if (L.live) {
                const bool canNode = L.pos < L.end && L.qn < Q;
                if (canNode) ++nN;
                if (L.qn) { ++nT; if (!canNode) ++must; }
                if (L.pos < L.end && L.qn >= Q) ++stalls;
            }
            const bool doN = nN > 0, doT = nT > 0 && (nT >= M.thr || must >= (int)M.bias || nN == 0);
            cost += M.cC + (doN ? M.cN : 0.f) + (doT ? M.cT : 0.f); iters++;
            if (doN) { nodeIters++; nodeLanes += nN; } if (doT) { triIters++; triLanes += nT; }
            for (auto& L : w.l) {
                if (!L.live) continue;
                const bool canNode = L.pos < L.end && L.qn < Q;
                if (doT && L.qn) { ++steps; if (--L.q[0] == 0) { for (int i = 1; i < L.qn; ++i) L.q[i - 1] = L.q[i]; --L.qn; } }
                if (canNode) { ++steps; const uint32_t t = tr.tris[L.pos++]; if (t) L.q[L.qn++] = t; }
                if (L.pos >= L.end && L.qn == 0) L.live = false;
            }
        }
    }
    printf("  %-16s overlap Q %d thr %2d must %2.0f: cost/step %6.1f  iters %.3gM  node iters %.0f%% @ %.1f lanes, tri iters %.0f%% @ %.1f lanes, stalled lane-iters %.1f%%\n", name, Q, M.thr, M.bias,
           cost / steps, iters / 1e6, 100 * nodeIters / iters, nodeLanes / std::max(nodeIters, 1.0), 100 * triIters / iters, triLanes / std::max(triIters, 1.0), 100 * stalls / (iters * 64));
}

// policy 7 (drain spill): the overlapped step of policy 3, but a wave whose cursor has run dry and that holds fewer than `spillAt` live
// rays WRITES THEIR STATE OUT (cS instructions) and leaves; the spilled rays of all waves are then continued by a second launch whose
// waves start full (64 spilled rays each, refilled from the spill list like a ray list, cR instructions per refill on top of cC).
// Levels > 1 spill again from the resume launch.  Answers: how many wave iterations does the drain of a persistent launch cost, and how
// many of them does compaction across waves give back?
struct SpillRay { uint32_t pos, end; uint32_t q[8]; int qn; };
static void wave_model_spill(const char* name, const Trace& tr, size_t nWaves, const WaveModel& M, int Q, int spillAt, int levels, float cS, float cR) {
    const size_t nRays = tr.start.size();
    std::vector<SpillRay> in(nRays);
    for (size_t r = 0; r < nRays; ++r) { in[r].pos = tr.start[r]; in[r].end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); in[r].qn = 0; }
    double cost = 0, iters = 0, steps = 0, drainIters = 0, drainLive = 0; size_t spilledTotal = 0;
    for (int level = 0; level <= levels && !in.empty(); ++level) {
        std::vector<SpillRay> out;
        struct Lane { bool live = false; SpillRay r; };
        struct Wave { Lane l[64]; size_t next = 0, end = 0; bool dry = false, done = false; };
        const size_t nW = level == 0 ? nWaves : std::min(nWaves, (in.size() + 63) / 64);
        std::vector<Wave> waves(nW);
        size_t cursor = 0, alive = nW;
        const int chunk = level == 0 ? M.chunk : 64;
        const bool maySpill = level < levels;
        while (alive) {
            for (Wave& w : waves) {
                if (w.done) continue;
                int live = 0; for (auto& L : w.l) live += L.live;
                if (!w.dry && (64 - live >= M.refill || live == 0)) {
                    if (level > 0) cost += cR;
                    for (auto& L : w.l) {
                        if (L.live) continue;
                        if (w.next >= w.end) { if (cursor >= in.size()) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)chunk, in.size()); cursor = w.end; }
                        L.r = in[w.next++]; L.live = !(L.r.pos >= L.r.end && L.r.qn == 0);
                        if (L.live) ++live;
                    }
                }
                if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
                if (w.dry && maySpill && live < spillAt) {
                    for (auto& L : w.l) if (L.live) { out.push_back(L.r); L.live = false; }
                    cost += cS; w.done = true; --alive; continue;
                }
                int nN = 0, nT = 0;
                for (auto& L : w.l) if (L.live) { if (L.r.pos < L.r.end && L.r.qn < Q) ++nN; if (L.r.qn) ++nT; }
                cost += M.cC + (nN ? M.cN : 0.f) + (nT ? M.cT : 0.f); iters++;
                if (w.dry) { drainIters++; drainLive += live; }
                for (auto& L : w.l) {
                    if (!L.live) continue;
                    const bool canNode = L.r.pos < L.r.end && L.r.qn < Q;
                    if (L.r.qn) { ++steps; if (--L.r.q[0] == 0) { for (int i = 1; i < L.r.qn; ++i) L.r.q[i - 1] = L.r.q[i]; --L.r.qn; } }
                    if (canNode) { ++steps; const uint32_t t = tr.tris[L.r.pos++]; if (t) L.r.q[L.r.qn++] = t; }
                    if (L.r.pos >= L.r.end && L.r.qn == 0) L.live = false;
                }
            }
        }
        spilledTotal += out.size();
        in.swap(out);
    }
    printf("  %-8s spill at < %2d live, %d level(s), cS %3.0f cR %3.0f: cost/step %6.2f  iters %.4gM  drain iters %.1f%% @ %.1f live  spilled rays %.2f%%\n", name, spillAt, levels, cS, cR,
           cost / steps, iters / 1e6, 100 * drainIters / iters, drainLive / std::max(drainIters, 1.0), 100.0 * spilledTotal / nRays);
}

// policy 4 (sorted workgroup): the 256 rays of a workgroup live in LDS; every iteration the rays on a node step are handed to the
// first waves and the rays on a triangle step to the last ones, so a wave runs one kind of step (the wave on the boundary both);
// waves that get no ray skip the iteration.  cX = what a wave pays for fetching / storing ray state and building the next lists.
static void wave_model_sorted(const char* name, const Trace& tr, size_t nBlocks, const WaveModel& M, float cX) {
    const size_t nRays = tr.start.size();
    struct Lane { bool live = false; uint32_t pos = 0, end = 0; uint32_t trisLeft = 0; };
    struct Block { Lane l[256]; size_t next = 0, end = 0; bool dry = false, done = false; };
    std::vector<Block> blocks(nBlocks);
    size_t cursor = 0; double cost = 0, iters = 0, steps = 0;
    size_t alive = nBlocks;
    while (alive) {
        for (Block& w : blocks) {
            if (w.done) continue;
            int live = 0; for (auto& L : w.l) live += L.live;
            if (!w.dry && (256 - live >= M.refill || live == 0)) {
                for (auto& L : w.l) {
                    if (L.live) continue;
                    if (w.next >= w.end) { if (cursor >= nRays) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)M.chunk, nRays); cursor = w.end; }
                    const size_t r = w.next++;
                    L.live = true; L.pos = tr.start[r]; L.end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); L.trisLeft = 0;
                    if (L.pos == L.end) L.live = false; else ++live;
                }
            }
            if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
            int nN = 0, nT = 0;
            for (auto& L : w.l) if (L.live) { if (L.trisLeft) ++nT; else ++nN; }
            // waves: items [0, nN) node, [nN, nN + nT) tri, 64 per wave
            const int total = nN + nT, nWaves = (total + 63) / 64;
            for (int wv = 0; wv < nWaves; ++wv) {
                const int lo = wv * 64, hi = std::min(lo + 64, total);
                const bool hasN = lo < nN, hasT = hi > nN;
                cost += M.cC + cX + (hasN ? M.cN : 0.f) + (hasT ? M.cT : 0.f);
            }
            iters += 1;
            for (auto& L : w.l) {
                if (!L.live) continue;
                if (L.trisLeft) --L.trisLeft; else L.trisLeft = tr.tris[L.pos++];
                ++steps;
                if (!L.trisLeft && L.pos >= L.end) L.live = false;
            }
        }
    }
    printf("  %-16s sorted workgroup, exchange %3.0f refill %3d: cost/step %6.1f  block iters %.3gM\n", name, cX, M.refill, cost / steps, iters / 1e6);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (float)((rng_state >> 40) * (1.0 / 16777216.0)); }

static void print_stats(const char* name, const Stats& s) {
    uint64_t tot = 0; for (uint32_t h : s.depthHist) tot += h;
    auto pct = [&](double p) { uint64_t acc = 0; for (size_t d = 0; d < s.depthHist.size(); ++d) { acc += s.depthHist[d]; if ((double)acc >= p * (double)tot) return (int)d; } return 127; };
    int mx = 0; for (size_t d = 0; d < s.depthHist.size(); ++d) if (s.depthHist[d]) mx = (int)d;
    printf("  %-26s rays %8.0f  nodes/ray %6.2f  tris/ray %6.2f  pushes/ray %6.2f  hit %.3f  depth p50 %d p90 %d p99 %d p99.9 %d max %d\n",
           name, s.rays, s.nodes / s.rays, s.tris / s.rays, s.pushes / s.rays, s.hits / s.rays, pct(0.5), pct(0.9), pct(0.99), pct(0.999), mx);
}
} // namespace

// policy 5 (per-wave ray pool): R rays of a wave live in LDS (state + stack); every iteration the rays that can make a node step
// are compacted onto the 64 lanes (first 64 in slot order starting at a rotating offset), then the rays with a queued triangle
// (queue of Q entries per ray, as the overlapped step).  cX = list building + state load / store per phase that runs.
static void wave_model_pool(const char* name, const Trace& tr, size_t nWaves, const WaveModel& M, int R, int Q, float cX, int triThr) {
    const size_t nRays = tr.start.size();
    struct Ray { bool live = false; uint32_t pos = 0, end = 0; uint32_t q[8]; int qn = 0; };
    struct Wave { std::vector<Ray> l; size_t next = 0, end = 0; bool dry = false, done = false; int rot = 0; };
    std::vector<Wave> waves(nWaves); for (auto& w : waves) w.l.resize(R);
    size_t cursor = 0; double cost = 0, iters = 0, steps = 0, nodeLanes = 0, triLanes = 0, nodeIters = 0, triIters = 0;
    size_t alive = nWaves;
    while (alive) {
        for (Wave& w : waves) {
            if (w.done) continue;
            int live = 0; for (auto& L : w.l) live += L.live;
            if (!w.dry && (R - live >= M.refill || live == 0)) {
                for (auto& L : w.l) {
                    if (L.live) continue;
                    if (w.next >= w.end) { if (cursor >= nRays) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)M.chunk, nRays); cursor = w.end; }
                    const size_t r = w.next++;
                    L.live = true; L.pos = tr.start[r]; L.end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); L.qn = 0;
                    if (L.pos == L.end) L.live = false; else ++live;
                }
            }
            if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
            // tri phase first (state after it feeds the node phase's queue test)
            int nT = 0, must = 0, nNwant = 0;
            for (auto& L : w.l) if (L.live) { if (L.qn) { ++nT; if (!(L.pos < L.end && L.qn < Q)) ++must; } if (L.pos < L.end && L.qn < Q) ++nNwant; }
            const bool doT = nT > 0 && (nT >= triThr || must > 0 || nNwant == 0);
            int doneT = 0;
            if (doT) {
                for (int i = 0; i < R && doneT < 64; ++i) { Ray& L = w.l[(i + w.rot) % R]; if (!L.live || !L.qn) continue; ++doneT; ++steps; if (--L.q[0] == 0) { for (int k = 1; k < L.qn; ++k) L.q[k - 1] = L.q[k]; --L.qn; } if (L.pos >= L.end && L.qn == 0) L.live = false; }
            }
            int doneN = 0;
            for (int i = 0; i < R && doneN < 64; ++i) { Ray& L = w.l[(i + w.rot) % R]; if (!L.live || !(L.pos < L.end && L.qn < Q)) continue; ++doneN; ++steps; const uint32_t t = tr.tris[L.pos++]; if (t) L.q[L.qn++] = t; if (L.pos >= L.end && L.qn == 0) L.live = false; }
            w.rot = (w.rot + 17) % R;
            cost += M.cC + (doneN ? M.cN + cX : 0.f) + (doneT ? M.cT + cX : 0.f); iters++;
            if (doneN) { nodeIters++; nodeLanes += doneN; } if (doneT) { triIters++; triLanes += doneT; }
        }
    }
    printf("  %-8s pool R %3d Q %d cX %3.0f triThr %2d refill %2d: cost/step %6.2f  iters %.3gM  node iters %.0f%% @ %.1f lanes, tri iters %.0f%% @ %.1f lanes\n", name, R, Q, cX, triThr, M.refill,
           cost / steps, iters / 1e6, 100 * nodeIters / iters, nodeLanes / std::max(nodeIters, 1.0), 100 * triIters / iters, triLanes / std::max(triIters, 1.0));
}
// policy 6 (two rays per lane, in registers): a lane's node step goes to whichever of its two rays can make one, likewise its
// triangle step; cX = the selects per phase.
static void wave_model_dual(const char* name, const Trace& tr, size_t nWaves, const WaveModel& M, int K, int Q, float cX) {
    const size_t nRays = tr.start.size();
    struct Ray { bool live = false; uint32_t pos = 0, end = 0; uint32_t q[8]; int qn = 0; };
    struct Wave { std::vector<Ray> l; size_t next = 0, end = 0; bool dry = false, done = false; };
    const int R = 64 * K;
    std::vector<Wave> waves(nWaves); for (auto& w : waves) w.l.resize(R);
    size_t cursor = 0; double cost = 0, iters = 0, steps = 0, nodeLanes = 0, triLanes = 0, nodeIters = 0, triIters = 0;
    size_t alive = nWaves;
    while (alive) {
        for (Wave& w : waves) {
            if (w.done) continue;
            int live = 0; for (auto& L : w.l) live += L.live;
            if (!w.dry && (R - live >= M.refill || live == 0)) {
                for (auto& L : w.l) {
                    if (L.live) continue;
                    if (w.next >= w.end) { if (cursor >= nRays) { w.dry = true; break; } w.next = cursor; w.end = std::min(cursor + (size_t)M.chunk, nRays); cursor = w.end; }
                    const size_t r = w.next++;
                    L.live = true; L.pos = tr.start[r]; L.end = r + 1 < nRays ? tr.start[r + 1] : (uint32_t)tr.tris.size(); L.qn = 0;
                    if (L.pos == L.end) L.live = false; else ++live;
                }
            }
            if (live == 0) { if (w.dry) { w.done = true; --alive; } continue; }
            int doneT = 0, doneN = 0;
            for (int lane = 0; lane < 64; ++lane) {
                for (int k = 0; k < K; ++k) { Ray& L = w.l[lane * K + k]; if (!L.live || !L.qn) continue; ++doneT; ++steps; if (--L.q[0] == 0) { for (int i = 1; i < L.qn; ++i) L.q[i - 1] = L.q[i]; --L.qn; } if (L.pos >= L.end && L.qn == 0) L.live = false; break; }
                for (int k = 0; k < K; ++k) { Ray& L = w.l[lane * K + k]; if (!L.live || !(L.pos < L.end && L.qn < Q)) continue; ++doneN; ++steps; const uint32_t t = tr.tris[L.pos++]; if (t) L.q[L.qn++] = t; if (L.pos >= L.end && L.qn == 0) L.live = false; break; }
            }
            cost += M.cC + (doneN ? M.cN + cX : 0.f) + (doneT ? M.cT + cX : 0.f); iters++;
            if (doneN) { nodeIters++; nodeLanes += doneN; } if (doneT) { triIters++; triLanes += doneT; }
        }
    }
    printf("  %-8s dual K %d Q %d cX %3.0f refill %2d: cost/step %6.2f  iters %.3gM  node iters %.0f%% @ %.1f lanes, tri iters %.0f%% @ %.1f lanes\n", name, K, Q, cX, M.refill,
           cost / steps, iters / 1e6, 100 * nodeIters / iters, nodeLanes / std::max(nodeIters, 1.0), 100 * triIters / iters, triLanes / std::max(triIters, 1.0));
}

extern "C" int trav_sim(const incr_scene_desc* desc, const float* cam /* pos, right, up, look */, float tanHalfFovY, float aspect, int W, int H) {
    HostScene hs; copy_scene_desc(*desc, hs);
    const char* why = "";
    if (!validate_scene(hs, &why)) { printf("scene rejected: %s\n", why); return 1; }
    HostBvh base; build_world_bvh(hs, base);
    printf("triangles %zu nodes8 %zu maxDepth %u\n", base.tris.size(), base.nodes8.size(), base.maxDepth);
    // lights
    struct PL { float p[3], range; };
    std::vector<PL> pls; float sun[3] = {0.3f, 0.8f, 0.5f};
    for (size_t i = 0; i < hs.lightIndices.size(); ++i) {
        const incr_light& L = hs.globalLights[(size_t)hs.lightIndices[i]];
        const float* m = hs.lightTransforms[(size_t)hs.lightIndices[i]].m;
        if (L.type == INCR_LIGHT_POINT) pls.push_back({{m[12], m[13], m[14]}, L.range});
        else if (L.type == INCR_LIGHT_DIRECTIONAL) { sun[0] = -m[8]; sun[1] = -m[9]; sun[2] = -m[10]; }
    }
    printf("point lights %zu\n", pls.size());
    const bool fast = getenv("X_FAST") != nullptr;
    for (int cfg = fast ? 2 : 0; cfg < 3; ++cfg) {
        HostBvh B = base; std::vector<uint8_t> sortAxis(B.nodes8.size(), 0);
        if (cfg >= 1) reorder(B, sortAxis, cfg == 2);
        if (getenv("X_AREA")) build_area_order(B);
        if (const char* e = check_bvh(B)) { printf("cfg %d: BVH check failed: %s\n", cfg, e); return 2; }
        printf("slot order %s\n", cfg == 0 ? "as built" : cfg == 1 ? "inner first" : "inner first + sorted along longest axis");
        const int nv = cfg == 0 ? 1 : 4;
        for (int v = 0; v < nv; ++v) {
            const int variant = cfg == 0 ? 0 : v;
            if (cfg == 1 && v >= 2) continue;            // reversal needs the sorted order
            if (fast && v != 2) continue;
            rng_state = 0x9E3779B97F4A7C15ull;
            Stats sp, sb, ss, ss0, ssOcc;
            const bool occCache = getenv("X_OCC") != nullptr;
            std::vector<uint32_t> occTri(3 * 256, 0xffffffffu); double occStat[3][3] = {};
            Trace trP, trB, trS; const bool model = cfg == 2 && variant == 2;
            g_nodeDepth.clear(); memset(g_visitsByDepth, 0, sizeof g_visitsByDepth);
            if (cfg == 2 && variant == 2) { g_visitsByIndex[0].assign(B.nodes8.size(), 0.0); g_visitsByIndex[1].assign(B.nodes8.size(), 0.0); g_nodeDepth.assign(B.nodes8.size(), 0); std::vector<std::pair<uint32_t,uint8_t>> q; q.push_back({0u, 1}); while (!q.empty()) { auto [ni, d] = q.back(); q.pop_back(); g_nodeDepth[ni] = d; for (int k = 0; k < 8; ++k) if ((B.nodes8[ni].imask >> k) & 1u) q.push_back({B.nodes8[ni].childBase + (uint32_t)k, (uint8_t)(d + 1)}); } }
            double checksum = 0;
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
                const float sx = ((x + 0.5f) / W * 2.f - 1.f) * tanHalfFovY * aspect, sy = (1.f - (y + 0.5f) / H * 2.f) * tanHalfFovY;
                Ray r; float l = 0;
                for (int k = 0; k < 3; ++k) { r.o[k] = cam[k]; r.d[k] = cam[9 + k] + sx * cam[3 + k] + sy * cam[6 + k]; l += r.d[k] * r.d[k]; }
                l = 1.f / sqrtf(l); for (int k = 0; k < 3; ++k) r.d[k] *= l;
                r.tmin = 0.001f; r.tmax = 10000.f;
                Stats* S = &sp;
                for (int bounce = 0; bounce < 3; ++bounce) {
                    g_trace = model ? (bounce == 0 ? &trP : &trB) : nullptr;
                    const Result h = closest(B, sortAxis, r, variant, *S);
                    g_trace = nullptr;
                    S = &sb;
                    if (!h.found) break;
                    checksum += h.t;
                    const TriIsect& T = B.tris[h.tri];
                    float n[3] = {T.e1y * T.e2z - T.e1z * T.e2y, T.e1z * T.e2x - T.e1x * T.e2z, T.e1x * T.e2y - T.e1y * T.e2x};
                    float nl = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    if (n[0] * r.d[0] + n[1] * r.d[1] + n[2] * r.d[2] > 0.f) nl = -nl;
                    for (int k = 0; k < 3; ++k) n[k] *= nl;
                    float p[3]; for (int k = 0; k < 3; ++k) p[k] = r.o[k] + h.t * r.d[k];
                    // shadow rays: sun, in-range point lights, one cosine sky sample
                    auto shadow = [&](const float d[3], float tmax, int lightKey = -1) {
                        Ray s; for (int k = 0; k < 3; ++k) { s.o[k] = p[k]; s.d[k] = d[k]; } s.tmin = 0.1f; s.tmax = tmax;
                        if (occCache && lightKey >= 0) {          // X_OCC: last-occluder cache per light (and per bounce), rays in pixel order
                            uint32_t& c = occTri[(size_t)bounce * 256 + (size_t)lightKey];
                            occStat[bounce][0]++;
                            float tt; if (c != 0xffffffffu && tri_isect(B.tris[c], s, s.tmax, &tt)) { occStat[bounce][1]++; return; }
                            g_occTri = 0xffffffffu;
                            const bool o = anyhit(B, sortAxis, s, 1, ssOcc);
                            if (o) { c = g_occTri; occStat[bounce][2]++; }
                            return;
                        }
                        g_trace = model ? &trS : nullptr;
                        anyhit(B, sortAxis, s, (fast || model) ? 1 : variant, ss);
                        if (getenv("X_AREA")) { g_trace = nullptr; anyhit_area(B, s, ss0); }
                        g_trace = nullptr;
                    };
                    if (sun[0] * n[0] + sun[1] * n[1] + sun[2] * n[2] > 0.f) shadow(sun, 1000.f, 0);
                    for (const PL& L : pls) {
                        float d[3] = {L.p[0] - p[0], L.p[1] - p[1], L.p[2] - p[2]};
                        const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        if (dist >= L.range) continue;
                        for (int k = 0; k < 3; ++k) d[k] /= dist;
                        if (d[0] * n[0] + d[1] * n[1] + d[2] * n[2] > 0.f) shadow(d, 1000.f, 1 + (int)(&L - pls.data()));     // the reference traces to TMax 1000
                    }
                    // cosine direction about n
                    float t1[3] = {fabsf(n[0]) > 0.5f ? 0.f : 1.f, fabsf(n[0]) > 0.5f ? 1.f : 0.f, 0.f};
                    float bx[3] = {n[1] * t1[2] - n[2] * t1[1], n[2] * t1[0] - n[0] * t1[2], n[0] * t1[1] - n[1] * t1[0]};
                    float bl = 1.f / sqrtf(bx[0] * bx[0] + bx[1] * bx[1] + bx[2] * bx[2]); for (int k = 0; k < 3; ++k) bx[k] *= bl;
                    float by[3] = {n[1] * bx[2] - n[2] * bx[1], n[2] * bx[0] - n[0] * bx[2], n[0] * bx[1] - n[1] * bx[0]};
                    auto cosdir = [&](float out[3]) {
                        const float u1 = rnd(), u2 = rnd(), rr = sqrtf(u1), ph = 6.2831853f * u2, z = sqrtf(fmaxf(0.f, 1.f - u1));
                        for (int k = 0; k < 3; ++k) out[k] = rr * cosf(ph) * bx[k] + rr * sinf(ph) * by[k] + z * n[k];
                    };
                    float sd[3]; cosdir(sd); shadow(sd, 1000.f);
                    float nd[3]; cosdir(nd);
                    for (int k = 0; k < 3; ++k) { r.o[k] = p[k]; r.d[k] = nd[k]; }
                }
            }
            printf(" variant %d (checksum %.6f)\n", variant, checksum);
            if (cfg == 2 && variant == 2) {
                for (int w = 0; w < 2; ++w) { double tot = 0, acc = 0; for (int d = 0; d < 32; ++d) tot += g_visitsByDepth[w][d];
                    printf("  %s node visits by depth (cumulative share):", w ? "any-hit" : "closest"); for (int d = 1; d < 12; ++d) { acc += g_visitsByDepth[w][d]; printf(" d%d %.3f", d, acc / tot); } printf("\n");
                    printf("  %s node visits taken by the K lowest indices:", w ? "any-hit" : "closest"); double a2 = 0; size_t i2 = 0;
                    for (size_t K : {8, 24, 64, 120, 200, 288, 400, 576, 800, 1100, 1536, 2048, 4096}) { for (; i2 < K && i2 < g_visitsByIndex[w].size(); ++i2) a2 += g_visitsByIndex[w][i2]; printf(" %zu: %.3f", K, a2 / tot); } printf("\n"); }

            }
            print_stats("closest primary", sp); print_stats("closest bounce", sb); print_stats("any-hit shadow", ss);
            if (occCache) { print_stats("any-hit light rays behind the occluder cache", ssOcc);
                for (int b = 0; b < 3; ++b) printf("  occluder cache, bounce %d: light rays %.0f  cache hits %.3f  occluded by traversal %.3f\n", b, occStat[b][0], occStat[b][1] / std::max(occStat[b][0], 1.0), occStat[b][2] / std::max(occStat[b][0], 1.0)); }
            if (getenv("X_AREA")) print_stats("any-hit, area order", ss0);
            if (model) {
                struct { const char* n; const Trace* t; size_t perWave; int chunk, refill; float cC, cN, cT; } sets[3] = {
                    {"primary", &trP, 1150, 64, 16, 100, 300, 150}, {"bounce", &trB, 700, 64, 16, 100, 300, 150}, {"shadow", &trS, 1700, 128, 12, 80, 260, 110}};
                for (auto& S : sets) {
                    const size_t nW = std::max<size_t>(1, S.t->start.size() / S.perWave);
                    wave_model(S.n, *S.t, nW, {0, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, 0});
                    for (float bias : {0.5f, 1.f, 2.f}) wave_model(S.n, *S.t, nW, {1, S.chunk, S.refill, S.cC, S.cN, S.cT, bias, 0});
                    for (int thr : {8, 16, 24}) wave_model(S.n, *S.t, nW, {2, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, thr});
                    wave_model_overlap(S.n, *S.t, nW, {3, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, 1}, 2);
                    if (getenv("X_THR")) for (int Q : {2, 3, 4}) for (int thr : {1, 16, 24, 32, 40, 48}) wave_model_overlap(S.n, *S.t, nW, {3, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, thr}, Q);
                    if (getenv("X_SPILL")) {
                        wave_model_spill(S.n, *S.t, nW, {3, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, 1}, 2, 0, 0, 0.f, 0.f);
                        for (int lv : {1, 2}) for (int at : {16, 24, 32, 40, 48}) wave_model_spill(S.n, *S.t, nW, {3, S.chunk, S.refill, S.cC, S.cN, S.cT, 1.f, 1}, 2, at, lv, 150.f, 150.f);
                    }
                    if (getenv("X_POOL")) {
                        for (int R : {96, 128, 192}) for (float cX : {20.f, 40.f}) for (int tt : {1, 24, 40}) wave_model_pool(S.n, *S.t, std::max<size_t>(1, nW * 64 / R), {5, S.chunk, 16, S.cC, S.cN, S.cT, 1.f, 0}, R, 2, cX, tt);
                        for (int K : {2, 3}) for (float cX : {15.f, 25.f}) wave_model_dual(S.n, *S.t, std::max<size_t>(1, nW / K), {6, S.chunk, 16, S.cC, S.cN, S.cT, 1.f, 0}, K, 2, cX);
                    }
                    for (float cX : {40.f, 80.f, 120.f}) for (int rf : {32, 64}) wave_model_sorted(S.n, *S.t, std::max<size_t>(1, nW / 4), {4, S.chunk * 4, rf, S.cC, S.cN, S.cT, 1.f, 0}, cX);
                }
            }
        }
    }
    return 0;
}
