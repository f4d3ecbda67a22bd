// meshletize.cpp -- host-side meshlet builder: the load-time step that defines the (meshlet, triangle) ids the
// visibility buffer encodes.  Implements MeshUtils::Meshletize (reference: source/demo-dll/src/mesh-utils.cpp:640-835
// with its helpers :114-553; the reference credits the Microsoft D3D12MeshShaders sample) with the reference's exact
// decision sequence -- position-hash point reps, hash-chained directed edges with best-normal pairing, the candidate
// stack re-scored and re-sorted after every accepted triangle, REVERSED winding on insert ({i2, i1, i0}, :682-687),
// 10:10:10 packing (inc/mesh-utils.h:8-14) -- so that meshlet ids, vertex lists and packed triangles equal those of the
// step-for-step restatement in oracle/oracle_meshlet.cpp bit for bit (tests/test_meshletize.py), which is how the
// "bit-exact visibility ids" of the north star are anchored.  The toolchain-defined pieces (DirectXMath float order,
// XMVectorLog = log2, 64-bit size_t hash, tie order of std::sort) are pinned as documented in that file.
//
// What differs from a literal transcription is bookkeeping only: vertex re-use and slot look-ups are O(1) per vertex
// (a per-vertex slot count / last slot for the current meshlet instead of scanning its vertex list per candidate), the
// candidate and done sets are bitmaps, and the sort is an insertion sort over an almost sorted list.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/incr.h"

namespace {

struct P3 { float x, y, z; };
static inline P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float dot(P3 a, P3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }       // XMVector3Dot, SSE2 order
static inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline P3 normalized(P3 v) {                                                         // XMVector3Normalize
    const float len = std::sqrt(dot(v, v));
    if (len == 0.f) return {0.f, 0.f, 0.f};
    if (!(len <= FLT_MAX)) { const float q = std::nanf(""); return {q, q, q}; }
    return {v.x / len, v.y / len, v.z / len};
}
static inline P3 face_normal(P3 p0, P3 p1, P3 p2) { return normalized(cross(sub(p0, p1), sub(p0, p2))); }   // ComputeNormal :238-248

struct Sphere { P3 c; float r; };

// MinimumBoundingSphere :114-180: extreme pair along the widest axis, then one growing pass (radius^2 is NOT refreshed
// inside the pass, as in the reference).
static Sphere bounding_sphere(const P3* pts, uint32_t n) {
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (uint32_t i = 1; i < n; ++i) {
        const float* v = &pts[i].x;
        for (int a = 0; a < 3; ++a) {
            if (v[a] < (&pts[lo[a]].x)[a]) lo[a] = i;
            if (v[a] > (&pts[hi[a]].x)[a]) hi[a] = i;
        }
    }
    uint32_t axis = 0; float best = 0.f;
    for (uint32_t a = 0; a < 3; ++a) { const P3 d = sub(pts[hi[a]], pts[lo[a]]); const float d2 = dot(d, d); if (d2 > best) { best = d2; axis = a; } }
    const P3 p1 = pts[lo[axis]], p2 = pts[hi[axis]];
    Sphere s; s.c = {(p1.x + p2.x) * 0.5f, (p1.y + p2.y) * 0.5f, (p1.z + p2.z) * 0.5f};
    const P3 d12 = sub(p2, p1); s.r = std::sqrt(dot(d12, d12)) * 0.5f;
    const float r2 = s.r * s.r;
    for (uint32_t i = 0; i < n; ++i) {
        const P3 d = sub(pts[i], s.c); const float d2 = dot(d, d);
        if (d2 > r2) {
            const float dist = std::sqrt(d2);
            const float k = (s.r / dist) * 0.5f + 0.5f, omk = 1.f - k;
            s.c = {s.c.x * k + pts[i].x * omk, s.c.y * k + pts[i].y * omk, s.c.z * k + pts[i].z * omk};
            s.r = (s.r + dist) * 0.5f;
        }
    }
    return s;
}

static inline uint64_t position_hash(const P3& p) {      // hash<XMFLOAT3> = CRCHash of the three dwords, size_t = 64 bit (:198-211, :553-556)
    uint32_t w[3]; memcpy(w, &p, 12);
    uint64_t h = 0;
    for (int i = 0; i < 3; ++i) { const uint32_t high = (uint32_t)(h & 0xf8000000u); h = (h << 5) ^ (uint64_t)(high >> 27); h ^= (uint64_t)w[i]; }
    return h;
}

struct Meshlet { std::vector<uint32_t> verts; std::vector<uint32_t> packedTris; Sphere bounds{{0, 0, 0}, 0}; };

// BuildAdjacencyList :285-487.  adj[f*3 + e] = face across edge e or ~0u.  The pairing depends on the order of the hash
// chains (LIFO per bucket `pointRep % (vertexCount / 3)`) and on entries being unlinked as they are matched, so the chains
// are kept exactly: `next` indices instead of pointers, ~0u = end of chain.
static void build_adjacency(const uint32_t* idx, uint32_t triCount, const P3* pos, uint32_t vertexCount, std::vector<uint32_t>& adj) {
    const uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> rep(vertexCount);
    { std::unordered_map<uint64_t, uint32_t> first; first.reserve(vertexCount);
      for (uint32_t i = 0; i < vertexCount; ++i) { const auto ins = first.emplace(position_hash(pos[i]), i); rep[i] = ins.first->second; } }
    const uint32_t buckets = vertexCount / 3;
    struct Edge { uint32_t i0, i1, i2, face, next; };
    std::vector<Edge> edges((size_t)triCount * 3);
    std::vector<uint32_t> head(buckets, kNone);
    for (uint32_t f = 0; f < triCount; ++f)
        for (uint32_t e = 0; e < 3; ++e) {
            Edge& E = edges[(size_t)f * 3 + e];
            E.i0 = rep[idx[f * 3 + e]]; E.i1 = rep[idx[f * 3 + (e + 1) % 3]]; E.i2 = rep[idx[f * 3 + (e + 2) % 3]];
            E.face = f; E.next = head[E.i0 % buckets]; head[E.i0 % buckets] = f * 3 + e;
        }
    auto edge_normal = [&](uint32_t a, uint32_t b, uint32_t c) { return normalized(cross(sub(pos[a], pos[b]), sub(pos[b], pos[c]))); };
    auto unlink = [&](uint32_t bucket, uint32_t prev, uint32_t cur) { if (prev != kNone) edges[prev].next = edges[cur].next; else head[bucket] = edges[cur].next; };
    adj.assign((size_t)triCount * 3, kNone);
    for (uint32_t f = 0; f < triCount; ++f)
        for (uint32_t pt = 0; pt < 3; ++pt) {
            if (adj[(size_t)f * 3 + pt] != kNone) continue;
            const uint32_t i0 = rep[idx[f * 3 + (pt + 1) % 3]], i1 = rep[idx[f * 3 + pt]], i2 = rep[idx[f * 3 + (pt + 2) % 3]];   // the edge, reversed
            const uint32_t bucket = i0 % buckets;
            uint32_t found = kNone, foundPrev = kNone;
            for (uint32_t cur = head[bucket], prev = kNone; cur != kNone; prev = cur, cur = edges[cur].next)
                if (edges[cur].i1 == i1 && edges[cur].i0 == i0) { found = cur; foundPrev = prev; break; }
            const P3 n0 = edge_normal(i1, i0, i2);
            float bestDot = -2.0f;
            for (uint32_t cur = found, prev = foundPrev; cur != kNone; prev = cur, cur = edges[cur].next) {
                const Edge& C = edges[cur];
                if (bestDot == -2.0f || (C.i1 == i1 && C.i0 == i0)) {
                    const float d = dot(n0, edge_normal(C.i0, C.i1, C.i2));
                    if (d > bestDot) { found = cur; foundPrev = prev; bestDot = d; }
                }
            }
            if (found == kNone) continue;
            const uint32_t other = edges[found].face;
            unlink(bucket, foundPrev, found);
            adj[(size_t)f * 3 + pt] = other;
            const uint32_t bucket2 = i1 % buckets;                       // this face's own half of the edge leaves its chain too
            for (uint32_t cur = head[bucket2], prev = kNone; cur != kNone; prev = cur, cur = edges[cur].next)
                if (edges[cur].face == f && edges[cur].i1 == i0 && edges[cur].i0 == i1) { unlink(bucket2, prev, cur); break; }
            bool linked = false;
            for (uint32_t q = 0; q < pt; ++q) if (adj[(size_t)f * 3 + q] == other) { linked = true; adj[(size_t)f * 3 + pt] = kNone; break; }
            if (!linked) {
                uint32_t e2 = 0;
                for (; e2 < 3; ++e2) { const uint32_t k = idx[other * 3 + e2]; if (k == kNone) continue; if (rep[k] == i0) break; }
                if (e2 < 3) adj[(size_t)other * 3 + e2] = f;
            }
        }
}

// The meshlet under construction plus O(1) views of its vertex list: slots[v] = how many slots hold vertex v (a degenerate
// triangle stores its repeated vertex twice, as in the reference), last[v] = the highest such slot.
struct Builder {
    Meshlet* cur = nullptr;
    std::vector<uint32_t> slots, last, touched;
    void reset(Meshlet* m) { for (uint32_t v : touched) slots[v] = 0; touched.clear(); cur = m; }
    uint32_t reuse(const uint32_t tri[3]) const { return slots[tri[0]] + slots[tri[1]] + slots[tri[2]]; }       // ComputeReuse :220-236
    bool add(const uint32_t tri[3], uint32_t maxVerts, uint32_t maxPrims) {                                     // AddToMeshlet :490-539
        Meshlet& m = *cur;
        if (m.verts.size() == maxVerts || m.packedTris.size() == maxPrims) return false;
        const uint32_t fresh = 3u - reuse(tri);                          // unsigned, like the reference's --newCount per matching slot
        if ((uint64_t)m.verts.size() + (uint64_t)fresh > maxVerts) return false;   // size_t + uint32 there: a wrapped `fresh` (repeated vertices) rejects
        uint32_t local[3];
        for (int j = 0; j < 3; ++j) local[j] = slots[tri[j]] ? last[tri[j]] : 0xffffffffu;      // look-ups see the list BEFORE this triangle's pushes
        for (int j = 0; j < 3; ++j) if (local[j] == 0xffffffffu) {
            local[j] = (uint32_t)m.verts.size(); m.verts.push_back(tri[j]);
            if (!slots[tri[j]]) touched.push_back(tri[j]);
            slots[tri[j]]++; last[tri[j]] = local[j];
        }
        m.packedTris.push_back((local[0] & 1023u) | ((local[1] & 1023u) << 10) | ((local[2] & 1023u) << 20));   // i0:10 | i1:10 | i2:10
        return true;
    }
    bool full(uint32_t maxVerts, uint32_t maxPrims) const { return cur->verts.size() == maxVerts || cur->packedTris.size() == maxPrims; }
};

static float score(const Builder& b, const Sphere& s, P3 axis, const uint32_t tri[3], const P3 v[3]) {      // ComputeScore :251-283
    const float reuseScore = 1.f - ((float)b.reuse(tri) / 3.0f);
    float maxSq = 0.f;
    for (int j = 0; j < 3; ++j) { const P3 d = sub(s.c, v[j]); maxSq = std::fmax(maxSq, dot(d, d)); }
    const float locScore = std::log2(maxSq / (s.r * s.r) + 1.f);          // XMVectorLog = base-2 logarithm
    const float oriScore = (-dot(face_normal(v[0], v[1], v[2]), axis) + 1.f) / 2.0f;
    return (0.334f * reuseScore + 0.333f * locScore) + 0.333f * oriScore;
}

struct Cand { uint32_t tri; float score; };

static void meshletize(uint32_t maxVerts, uint32_t maxPrims, const uint32_t* idx, uint32_t indexCount, const P3* pos, uint32_t vertexCount,
                       std::vector<Meshlet>& out) {
    const uint32_t triCount = indexCount / 3;
    out.clear();
    if (!triCount || vertexCount < 3) return;                             // vertexCount / 3 buckets: the reference divides by zero below 3 vertices
    std::vector<uint32_t> adj;
    build_adjacency(idx, triCount, pos, vertexCount, adj);
    std::vector<uint8_t> done(triCount, 0), queued(triCount, 0);
    std::vector<uint32_t> queuedList;                                     // members of `queued`, for the clears
    auto clear_queued = [&]() { for (uint32_t t : queuedList) queued[t] = 0; queuedList.clear(); };
    auto enqueue = [&](uint32_t t) { if (!queued[t]) { queued[t] = 1; queuedList.push_back(t); } };
    std::vector<Cand> cand;                                               // a stack: the best (lowest score) candidate is at the back
    std::vector<P3> mpos, mnrm;
    out.reserve(triCount / 32 + 2);
    out.emplace_back();
    Builder b; b.slots.assign(vertexCount, 0); b.last.assign(vertexCount, 0); b.reset(&out.back());
    Sphere sphere{{0, 0, 0}, 0}; P3 axis{0, 0, 0};
    uint32_t seed = 0;
    cand.push_back({0u, 0.f}); enqueue(0);
    auto next_meshlet = [&]() { mpos.clear(); mnrm.clear(); clear_queued(); };
    while (!cand.empty()) {
        const uint32_t t = cand.back().tri; cand.pop_back();
        const uint32_t tri[3] = {idx[t * 3 + 2], idx[t * 3 + 1], idx[t * 3]};          // reversed winding on insert
        if (b.add(tri, maxVerts, maxPrims)) {
            done[t] = 1;
            const P3 p[3] = {pos[tri[0]], pos[tri[1]], pos[tri[2]]};
            mpos.push_back(p[0]); mpos.push_back(p[1]); mpos.push_back(p[2]);
            mnrm.push_back(face_normal(p[0], p[1], p[2]));
            sphere = bounding_sphere(mpos.data(), (uint32_t)mpos.size());
            b.cur->bounds = sphere;
            axis = normalized(bounding_sphere(mnrm.data(), (uint32_t)mnrm.size()).c);
            for (int e = 0; e < 3; ++e) {
                const uint32_t n = adj[(size_t)t * 3 + e];
                if (n == 0xffffffffu || done[n] || queued[n]) continue;
                cand.push_back({n, FLT_MAX}); enqueue(n);
            }
            for (Cand& c : cand) {
                const uint32_t ci[3] = {idx[c.tri * 3], idx[c.tri * 3 + 1], idx[c.tri * 3 + 2]};
                const P3 cv[3] = {pos[ci[0]], pos[ci[1]], pos[ci[2]]};
                c.score = score(b, sphere, axis, ci, cv);
            }
            if (b.full(maxVerts, maxPrims)) {
                // meshlet full: the candidate at the back seeds the next meshlet, the others return to the pool
                next_meshlet();
                if (!cand.empty()) { const Cand keep = cand.back(); cand.assign(1, keep); enqueue(keep.tri); }
                out.emplace_back(); b.reset(&out.back());
            } else {
                for (size_t i = 1; i < cand.size(); ++i) {                 // descending by score, ties keep their order
                    const Cand x = cand[i]; size_t j = i;
                    while (j > 0 && x.score > cand[j - 1].score) { cand[j] = cand[j - 1]; --j; }
                    cand[j] = x;
                }
            }
        } else if (cand.empty()) {                                          // did not fit and nothing else is waiting: start a new meshlet (t is not
            next_meshlet();                                                 // marked done, so the seed search below finds it again)
            out.emplace_back(); b.reset(&out.back());
        }
        if (cand.empty()) {
            while (seed < triCount && done[seed]) ++seed;
            if (seed == triCount) break;
            cand.push_back({seed, 0.f}); enqueue(seed);
        }
    }
    if (!out.empty() && out.back().packedTris.empty()) out.pop_back();
}

} // namespace

struct incr_meshlet_set {
    std::vector<incr_gpu_meshlet> meshlets;
    std::vector<uint32_t> vertexIndices, packedTriangles;
};

extern "C" {

int incr_meshletize(const uint32_t* indices, uint32_t index_count, const float* positions, uint32_t vertex_count,
                    uint32_t position_stride_bytes, uint32_t max_verts, uint32_t max_prims, incr_meshlet_set** out_set) {
    if (!out_set) return INCR_ERR_INVALID_ARGUMENT;
    *out_set = nullptr;
    if ((index_count && (!indices || !positions)) || index_count % 3 || position_stride_bytes < 12 || max_verts < 3 || max_verts > 256 || max_prims < 1 || max_prims > 128)
        return INCR_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < index_count; ++i) if (indices[i] >= vertex_count) return INCR_ERR_INVALID_ARGUMENT;
    std::vector<P3> pos(vertex_count);
    for (uint32_t i = 0; i < vertex_count; ++i) memcpy(&pos[i], (const uint8_t*)positions + (size_t)i * position_stride_bytes, 12);
    std::vector<Meshlet> ms;
    meshletize(max_verts, max_prims, indices, index_count, pos.data(), vertex_count, ms);
    incr_meshlet_set* set = new incr_meshlet_set();
    for (const Meshlet& m : ms) {
        incr_gpu_meshlet g; memset(&g, 0, sizeof g);
        g.vertexBegin = (uint32_t)set->vertexIndices.size(); g.vertexCount = (uint32_t)m.verts.size();
        g.triangleBegin = (uint32_t)set->packedTriangles.size(); g.triangleCount = (uint32_t)m.packedTris.size();
        g.boundingSphere[0] = m.bounds.c.x; g.boundingSphere[1] = m.bounds.c.y; g.boundingSphere[2] = m.bounds.c.z; g.boundingSphere[3] = m.bounds.r;
        g.meshIndex = g.positionAccessor = g.uvAccessor = g.normalAccessor = g.tangentAccessor = g.materialIndex = -1;    // filled by the caller (scene.cpp:519-533)
        set->meshlets.push_back(g);
        set->vertexIndices.insert(set->vertexIndices.end(), m.verts.begin(), m.verts.end());
        set->packedTriangles.insert(set->packedTriangles.end(), m.packedTris.begin(), m.packedTris.end());
    }
    *out_set = set;
    return INCR_OK;
}
uint32_t incr_meshlet_set_counts(const incr_meshlet_set* set, uint32_t* num_vertex_indices, uint32_t* num_triangles) {
    if (!set) return 0;
    if (num_vertex_indices) *num_vertex_indices = (uint32_t)set->vertexIndices.size();
    if (num_triangles) *num_triangles = (uint32_t)set->packedTriangles.size();
    return (uint32_t)set->meshlets.size();
}
int incr_meshlet_set_copy(const incr_meshlet_set* set, incr_gpu_meshlet* meshlets, uint32_t* vertex_indices, uint32_t* packed_triangles) {
    if (!set) return INCR_ERR_INVALID_ARGUMENT;
    if (meshlets && !set->meshlets.empty()) memcpy(meshlets, set->meshlets.data(), set->meshlets.size() * sizeof(incr_gpu_meshlet));
    if (vertex_indices && !set->vertexIndices.empty()) memcpy(vertex_indices, set->vertexIndices.data(), set->vertexIndices.size() * 4);
    if (packed_triangles && !set->packedTriangles.empty()) memcpy(packed_triangles, set->packedTriangles.data(), set->packedTriangles.size() * 4);
    return INCR_OK;
}
void incr_meshlet_set_free(incr_meshlet_set* set) { delete set; }

} // extern "C"
