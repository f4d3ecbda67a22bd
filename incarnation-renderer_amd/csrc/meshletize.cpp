// meshletize.cpp -- host-side meshlet builder: the load-time step that defines the (meshlet, triangle) ids the
// visibility buffer encodes.  Restates the ALGORITHM of MeshUtils::Meshletize
// (reference: source/demo-dll/src/mesh-utils.cpp:640-835 with its helpers :114-553; the reference credits the
// Microsoft D3D12MeshShaders sample): greedy growth of meshlets of <= maxVerts unique vertices and <= maxPrims
// triangles along triangle adjacency, candidates ranked by a score that mixes vertex re-use, distance to the
// meshlet's bounding sphere and deviation from its mean normal; triangles are inserted with REVERSED winding
// ({i2, i1, i0}, :682-687) and packed 10:10:10 (inc/mesh-utils.h:8-14), which the shaders undo by reading the packed
// fields high to low (shaders/geo-raster/visibility-pass.hlsl:78).
//
// The result is deterministic for a given input, but it is not claimed to be identical to the reference's output:
// that depends on std::sort tie order, std::unordered_set iteration and a position hash used as identity there.
#include <algorithm>
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
static inline P3 add(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline P3 mul(P3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline P3 normalized(P3 a) { const float l = std::sqrt(dot(a, a)); return l > 0.f ? mul(a, 1.f / l) : a; }
static inline P3 face_normal(P3 p0, P3 p1, P3 p2) { return normalized(cross(sub(p0, p1), sub(p0, p2))); }

struct Sphere { P3 c; float r; };

// Ritter-style bounding sphere: start from the extreme pair along the widest axis, then grow (mesh-utils.cpp:114-180).
static Sphere bounding_sphere(const P3* pts, size_t n) {
    size_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 1; i < n; ++i) {
        const float v[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int a = 0; a < 3; ++a) {
            const float l = a == 0 ? pts[lo[a]].x : a == 1 ? pts[lo[a]].y : pts[lo[a]].z;
            const float h = a == 0 ? pts[hi[a]].x : a == 1 ? pts[hi[a]].y : pts[hi[a]].z;
            if (v[a] < l) lo[a] = i;
            if (v[a] > h) hi[a] = i;
        }
    }
    int axis = 0; float best = 0.f;
    for (int a = 0; a < 3; ++a) { const P3 d = sub(pts[hi[a]], pts[lo[a]]); const float d2 = dot(d, d); if (d2 > best) { best = d2; axis = a; } }
    const P3 p1 = pts[lo[axis]], p2 = pts[hi[axis]];
    Sphere s; s.c = mul(add(p1, p2), 0.5f);
    const P3 d12 = sub(p2, p1); s.r = std::sqrt(dot(d12, d12)) * 0.5f;
    for (size_t i = 0; i < n; ++i) {
        const P3 d = sub(pts[i], s.c); const float d2 = dot(d, d);
        if (d2 > s.r * s.r) {
            const float dist = std::sqrt(d2);
            const float k = (s.r / dist) * 0.5f + 0.5f;
            s.c = add(mul(s.c, k), mul(pts[i], 1.f - k));
            s.r = (s.r + dist) * 0.5f;
        }
    }
    return s;
}

struct PosKey { uint32_t a, b, c; bool operator==(const PosKey& o) const { return a == o.a && b == o.b && c == o.c; } };
struct PosHash { size_t operator()(const PosKey& k) const { uint64_t h = k.a; h = h * 0x9E3779B97F4A7C15ull ^ k.b; h = h * 0x9E3779B97F4A7C15ull ^ k.c; return (size_t)(h ^ (h >> 29)); } };

struct Meshlet { std::vector<uint32_t> verts; std::vector<uint32_t> packedTris; Sphere bounds{{0, 0, 0}, 0}; };

// Triangle adjacency across shared edges of WELDED positions: adj[f*3 + e] = face on the other side of edge e
// (vertices e, e+1) or ~0u.  When several faces share an edge the one whose normal agrees best is taken; every
// half-edge is paired at most once.
static void build_adjacency(const uint32_t* idx, uint32_t triCount, const P3* pos, uint32_t vertexCount, std::vector<uint32_t>& adj) {
    std::vector<uint32_t> rep(vertexCount);
    std::unordered_map<PosKey, uint32_t, PosHash> first; first.reserve(vertexCount * 2);
    for (uint32_t i = 0; i < vertexCount; ++i) {
        PosKey k; memcpy(&k, &pos[i], 12);
        auto it = first.find(k);
        if (it == first.end()) { first.emplace(k, i); rep[i] = i; } else rep[i] = it->second;
    }
    struct Half { uint32_t face, edge; bool used; };
    std::unordered_map<uint64_t, std::vector<uint32_t>> byEdge; byEdge.reserve((size_t)triCount * 3);
    std::vector<Half> halves((size_t)triCount * 3);
    auto key = [](uint32_t a, uint32_t b) { return ((uint64_t)a << 32) | b; };
    for (uint32_t f = 0; f < triCount; ++f)
        for (uint32_t e = 0; e < 3; ++e) {
            const uint32_t a = rep[idx[f * 3 + e]], b = rep[idx[f * 3 + (e + 1) % 3]];
            halves[(size_t)f * 3 + e] = {f, e, false};
            byEdge[key(a, b)].push_back(f * 3 + e);
        }
    adj.assign((size_t)triCount * 3, 0xffffffffu);
    std::vector<P3> normals(triCount);
    for (uint32_t f = 0; f < triCount; ++f) normals[f] = face_normal(pos[idx[f * 3]], pos[idx[f * 3 + 1]], pos[idx[f * 3 + 2]]);
    for (uint32_t f = 0; f < triCount; ++f)
        for (uint32_t e = 0; e < 3; ++e) {
            Half& me = halves[(size_t)f * 3 + e];
            if (me.used) continue;
            const uint32_t a = rep[idx[f * 3 + e]], b = rep[idx[f * 3 + (e + 1) % 3]];
            if (a == b) continue;
            auto it = byEdge.find(key(b, a));          // the neighbour runs the edge in the opposite direction
            if (it == byEdge.end()) continue;
            int bestH = -1; float bestDot = -2.f;
            for (uint32_t h : it->second) {
                Half& o = halves[h];
                if (o.used || o.face == f) continue;
                const float d = dot(normals[f], normals[o.face]);
                if (d > bestDot) { bestDot = d; bestH = (int)h; }
            }
            if (bestH < 0) continue;
            Half& o = halves[(size_t)bestH];
            me.used = o.used = true;
            adj[(size_t)f * 3 + e] = o.face;
            adj[(size_t)o.face * 3 + o.edge] = f;
        }
}

static bool try_add(Meshlet& m, const uint32_t tri[3], uint32_t maxVerts, uint32_t maxPrims) {   // AddToMeshlet :494-543
    if (m.verts.size() == maxVerts || m.packedTris.size() == maxPrims) return false;
    uint32_t local[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, fresh = 3;
    for (uint32_t i = 0; i < m.verts.size(); ++i)
        for (int j = 0; j < 3; ++j) if (m.verts[i] == tri[j] && local[j] == 0xffffffffu) { local[j] = i; --fresh; }
    if (m.verts.size() + fresh > maxVerts) return false;
    for (int j = 0; j < 3; ++j) if (local[j] == 0xffffffffu) {
        // a degenerate triangle can name the same new vertex twice: reuse the slot just created
        for (int q = 0; q < j; ++q) if (tri[q] == tri[j]) local[j] = local[q];
        if (local[j] == 0xffffffffu) { local[j] = (uint32_t)m.verts.size(); m.verts.push_back(tri[j]); }
    }
    m.packedTris.push_back((local[0] & 1023u) | ((local[1] & 1023u) << 10) | ((local[2] & 1023u) << 20));   // i0:10 | i1:10 | i2:10
    return true;
}

static float score(const Meshlet& m, const Sphere& s, P3 axis, const uint32_t tri[3], const P3 v[3]) {      // ComputeScore :251-280
    uint32_t reuse = 0;
    for (uint32_t u : m.verts) for (int j = 0; j < 3; ++j) if (u == tri[j]) ++reuse;
    const float reuseScore = 1.f - (float)reuse / 3.f;
    float maxSq = 0.f;
    for (int j = 0; j < 3; ++j) { const P3 d = sub(s.c, v[j]); maxSq = std::max(maxSq, dot(d, d)); }
    const float locScore = std::log(maxSq / (s.r * s.r) + 1.f);
    const float oriScore = (1.f - dot(face_normal(v[0], v[1], v[2]), axis)) * 0.5f;
    return 0.334f * reuseScore + 0.333f * locScore + 0.333f * oriScore;
}

static void meshletize(uint32_t maxVerts, uint32_t maxPrims, const uint32_t* idx, uint32_t indexCount, const P3* pos, uint32_t vertexCount,
                       std::vector<Meshlet>& out) {
    const uint32_t triCount = indexCount / 3;
    out.clear();
    if (!triCount) return;
    std::vector<uint32_t> adj;
    build_adjacency(idx, triCount, pos, vertexCount, adj);
    std::vector<uint8_t> done(triCount, 0), queued(triCount, 0);
    std::vector<std::pair<uint32_t, float>> cand;     // sorted by descending score: the best (lowest) is taken from the back
    std::vector<P3> mpos, mnrm;
    out.emplace_back();
    Sphere sphere{{0, 0, 0}, 0}; P3 axis{0, 0, 1};
    uint32_t seed = 0;
    cand.push_back({0u, 0.f}); queued[0] = 1;
    while (!cand.empty()) {
        const uint32_t t = cand.back().first; cand.pop_back();
        const uint32_t tri[3] = {idx[t * 3 + 2], idx[t * 3 + 1], idx[t * 3]};          // reversed winding on insert
        Meshlet* cur = &out.back();
        if (try_add(*cur, tri, maxVerts, maxPrims)) {
            done[t] = 1;
            const P3 p[3] = {pos[tri[0]], pos[tri[1]], pos[tri[2]]};
            mpos.push_back(p[0]); mpos.push_back(p[1]); mpos.push_back(p[2]);
            mnrm.push_back(face_normal(p[0], p[1], p[2]));
            sphere = bounding_sphere(mpos.data(), mpos.size());
            cur->bounds = sphere;
            axis = normalized(bounding_sphere(mnrm.data(), mnrm.size()).c);
            for (int e = 0; e < 3; ++e) {
                const uint32_t n = adj[(size_t)t * 3 + e];
                if (n == 0xffffffffu || done[n] || queued[n]) continue;
                cand.push_back({n, FLT_MAX}); queued[n] = 1;
            }
            for (auto& c : cand) {
                const uint32_t ci[3] = {idx[c.first * 3], idx[c.first * 3 + 1], idx[c.first * 3 + 2]};
                const P3 cv[3] = {pos[ci[0]], pos[ci[1]], pos[ci[2]]};
                c.second = score(*cur, sphere, axis, ci, cv);
            }
            if (cur->verts.size() == maxVerts || cur->packedTris.size() == maxPrims) {
                // meshlet full: the last candidate seeds the next meshlet, everything else goes back to the pool
                mpos.clear(); mnrm.clear();
                for (auto& c : cand) queued[c.first] = 0;
                if (!cand.empty()) { const auto keep = cand.back(); cand.assign(1, keep); queued[keep.first] = 1; }
                out.emplace_back();
            } else {
                std::stable_sort(cand.begin(), cand.end(), [](const std::pair<uint32_t, float>& a, const std::pair<uint32_t, float>& b) { return a.second > b.second; });
            }
        } else if (cand.empty()) {
            mpos.clear(); mnrm.clear();
            out.emplace_back();
            // the triangle that did not fit is still pending: it becomes the seed below (done[t] is still 0)
            queued[t] = 0;
        } else {
            queued[t] = 0;      // could not be added to this meshlet; it stays available for a later one
        }
        if (cand.empty()) {
            while (seed < triCount && done[seed]) ++seed;
            if (seed == triCount) break;
            cand.push_back({seed, 0.f}); queued[seed] = 1;
        }
    }
    while (!out.empty() && out.back().packedTris.empty()) out.pop_back();
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
