// oracle_meshlet.cpp -- TEST INFRASTRUCTURE (CPU oracle).  Only tests/, tools/make_golden.py and __graft_entry__.smoke()
// may use it; the product's meshletizer is incarnation-renderer_amd/csrc/meshletize.cpp and never links this file.
//
// Step-for-step restatement of MeshUtils::Meshletize (reference: source/demo-dll/src/mesh-utils.cpp:640-835) and of its
// helpers MinimumBoundingSphere (:114-180), CRCHash / hash<XMFLOAT3> (:198-211, :553-556), CompareScores (:214-217),
// ComputeReuse (:220-236), ComputeNormal (:238-248), ComputeScore (:251-283), BuildAdjacencyList (:285-487),
// AddToMeshlet (:490-539), IsMeshletFull (:541-548), as called by FScene::GenerateMeshlets with MAX_VERTS 64,
// MAX_PRIMITIVES 126 (src/scene.cpp:1545-1551).  The reference credits the Microsoft D3D12MeshShaders sample.
//
// PARITY UNPINNED (as the rest of oracle/): the reference ships no meshlet fixtures.  Behaviour the reference leaves to
// its toolchain, and how it is pinned here (the product pins the same):
//   * DirectXMath (not in the reference tree): XMVector3Dot / LengthSq = (x*x + y*y) + z*z in float32 (the SSE2 path:
//     mul, add_ss y, add_ss z); XMVector3Normalize = v / sqrt(dot) with a zero-length vector -> 0 and an infinite length
//     -> NaN; XMVectorLog = base-2 logarithm (XMVectorLog is DirectXMath's alias of XMVectorLog2); evaluated with libm
//     log2f, not DirectXMath's polynomial; XMVector3Greater compares x, y, z (all three hold the same scalar here).
//   * size_t is 64 bit (x64 build): CRCHash keeps the reference's 32-bit rotate masks on a 64-bit accumulator.
//   * std::unordered_map<size_t, T> keyed by the HASH of the position (:303-318): two different positions with the same
//     hash share a point rep, exactly as there; only find / insert are used, so bucket order never matters.
//   * std::unordered_set<uint32_t> candidateCheck: only count / insert / clear -> no iteration order involved.
//   * std::sort(candidates, CompareScores) (:790) is NOT stable and MSVC's tie order cannot be known here: ties between
//     equal scores keep their previous relative order (stable insertion sort below).  Scores are recomputed for every
//     candidate right before the sort, so ties need exactly equal floats.
//   * hashSize = vertexCount / 3 (:322) is 0 for fewer than 3 vertices (a division by zero there): such input yields no
//     meshlets here.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 cross3(F3 a, F3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline F3 normalize3(F3 v) {
    const float len = std::sqrt(dot3(v, v));
    if (len == 0.f) return {0.f, 0.f, 0.f};
    if (!(len <= 3.402823466e38f)) { const float q = std::nanf(""); return {q, q, q}; }
    return {v.x / len, v.y / len, v.z / len};
}

struct InlineMeshlet {                                  // FInlineMeshlet (inc/mesh-utils.h)
    std::vector<uint32_t> uniqueVertexIndices;
    std::vector<uint32_t> primitiveIndices;             // i0 | i1 << 10 | i2 << 20
    F4 boundingSphere = {0.f, 0.f, 0.f, 0.f};
};

// :114-180
F4 MinimumBoundingSphere(const F3* points, uint32_t count) {
    uint32_t minAxis[3] = {0, 0, 0}, maxAxis[3] = {0, 0, 0};
    for (uint32_t i = 1; i < count; ++i) {
        const float* point = (const float*)(points + i);
        for (uint32_t j = 0; j < 3; ++j) {
            const float* mn = (const float*)(&points[minAxis[j]]);
            const float* mx = (const float*)(&points[maxAxis[j]]);
            minAxis[j] = point[j] < mn[j] ? i : minAxis[j];
            maxAxis[j] = point[j] > mx[j] ? i : maxAxis[j];
        }
    }
    float distSqMax = 0.f; uint32_t axis = 0;
    for (uint32_t i = 0; i < 3u; ++i) {
        const F3 d = sub(points[maxAxis[i]], points[minAxis[i]]);
        const float distSq = dot3(d, d);
        if (distSq > distSqMax) { distSqMax = distSq; axis = i; }
    }
    const F3 p1 = points[minAxis[axis]], p2 = points[maxAxis[axis]];
    F3 center = {(p1.x + p2.x) * 0.5f, (p1.y + p2.y) * 0.5f, (p1.z + p2.z) * 0.5f};
    const F3 d12 = sub(p2, p1);
    float radius = std::sqrt(dot3(d12, d12)) * 0.5f;
    float radiusSq = radius * radius;                   // computed ONCE (:158): the loop below never refreshes it, as there
    for (uint32_t i = 0; i < count; ++i) {
        const F3 point = points[i];
        const F3 pc = sub(point, center);
        const float distSq = dot3(pc, pc);
        if (distSq > radiusSq) {
            const float dist = std::sqrt(distSq);
            const float k = (radius / dist) * 0.5f + 0.5f;
            const float omk = 1.f - k;
            center = {center.x * k + point.x * omk, center.y * k + point.y * omk, center.z * k + point.z * omk};
            radius = (radius + dist) * 0.5f;
        }
    }
    return {center.x, center.y, center.z, radius};
}

// :198-211 with size_t = uint64_t; hash<XMFLOAT3> = CRCHash over the three dwords (:553-556)
uint64_t CRCHash(const uint32_t* dwords, uint32_t dwordCount) {
    uint64_t h = 0;
    for (uint32_t i = 0; i < dwordCount; ++i) {
        const uint32_t highOrd = (uint32_t)(h & 0xf8000000u);
        h = h << 5;
        h = h ^ (uint64_t)(highOrd >> 27);
        h = h ^ (uint64_t)dwords[i];
    }
    return h;
}

uint32_t ComputeReuse(const InlineMeshlet& m, const uint32_t tri[3]) {      // :220-236
    uint32_t count = 0;
    for (uint32_t i = 0; i < (uint32_t)m.uniqueVertexIndices.size(); ++i)
        for (uint32_t j = 0; j < 3u; ++j) if (m.uniqueVertexIndices[i] == tri[j]) ++count;
    return count;
}
F3 ComputeNormal(const F3 tri[3]) { return normalize3(cross3(sub(tri[0], tri[1]), sub(tri[0], tri[2]))); }      // :238-248

float ComputeScore(const InlineMeshlet& m, F4 sphere, F3 normal, const uint32_t triIndices[3], const F3 triVerts[3]) {      // :251-283
    const float reuseWeight = 0.334f, locWeight = 0.333f, oriWeight = 0.333f;
    const uint32_t reuse = ComputeReuse(m, triIndices);
    const float reuseScore = 1.f - ((float)reuse / 3.0f);
    float maxSq = 0.f;
    for (uint32_t i = 0; i < 3u; ++i) {
        const F3 v = {sphere.x - triVerts[i].x, sphere.y - triVerts[i].y, sphere.z - triVerts[i].z};
        maxSq = std::fmax(maxSq, dot3(v, v));           // XMVectorMax: a NaN operand loses (maxps(a, b) returns b when unordered)
    }
    const float r2 = sphere.w * sphere.w;
    const float locScore = std::log2(maxSq / r2 + 1.f);
    const F3 n = ComputeNormal(triVerts);
    const float d = dot3(n, normal);
    const float oriScore = (-d + 1.f) / 2.0f;
    return (reuseWeight * reuseScore + locWeight * locScore) + oriWeight * oriScore;
}

struct EdgeEntry { uint32_t i0, i1, i2, Face; EdgeEntry* Next; };

// :285-487
void BuildAdjacencyList(const uint32_t* indices, uint32_t indexCount, const F3* positions, uint32_t vertexCount, uint32_t* adjacency) {
    const uint32_t triCount = indexCount / 3;
    std::vector<uint32_t> pointRep(vertexCount);
    std::unordered_map<uint64_t, uint32_t> uniquePositionMap;
    uniquePositionMap.reserve(vertexCount);
    for (uint32_t i = 0; i < vertexCount; ++i) {
        const uint64_t hash = CRCHash((const uint32_t*)&positions[i], 3);
        auto it = uniquePositionMap.find(hash);
        if (it != uniquePositionMap.end()) pointRep[i] = it->second;
        else { uniquePositionMap.insert({hash, i}); pointRep[i] = i; }
    }
    const uint32_t hashSize = vertexCount / 3;
    std::vector<EdgeEntry*> hashTable(hashSize, nullptr);
    std::vector<EdgeEntry> entries((size_t)triCount * 3);
    uint32_t entryIndex = 0;
    for (uint32_t iFace = 0; iFace < triCount; ++iFace) {
        const uint32_t index = iFace * 3;
        for (uint32_t iEdge = 0; iEdge < 3; ++iEdge) {
            EdgeEntry& entry = entries[entryIndex++];
            entry.i0 = pointRep[indices[index + (iEdge % 3)]];
            entry.i1 = pointRep[indices[index + ((iEdge + 1) % 3)]];
            entry.i2 = pointRep[indices[index + ((iEdge + 2) % 3)]];
            const uint32_t key = entry.i0 % hashSize;
            entry.Next = hashTable[key];
            entry.Face = iFace;
            hashTable[key] = &entry;
        }
    }
    std::memset(adjacency, 0xff, (size_t)indexCount * sizeof(uint32_t));
    for (uint32_t iFace = 0; iFace < triCount; ++iFace) {
        const uint32_t index = iFace * 3;
        for (uint32_t point = 0; point < 3; ++point) {
            if (adjacency[iFace * 3 + point] != 0xffffffffu) continue;
            // look for the edge directed the other way
            const uint32_t i0 = pointRep[indices[index + ((point + 1) % 3)]];
            const uint32_t i1 = pointRep[indices[index + (point % 3)]];
            const uint32_t i2 = pointRep[indices[index + ((point + 2) % 3)]];
            const uint32_t key = i0 % hashSize;
            EdgeEntry* found = nullptr; EdgeEntry* foundPrev = nullptr;
            for (EdgeEntry* current = hashTable[key], *prev = nullptr; current != nullptr; prev = current, current = current->Next)
                if (current->i1 == i1 && current->i0 == i0) { found = current; foundPrev = prev; break; }
            // this face's normal
            const F3 n0 = normalize3(cross3(sub(positions[i1], positions[i0]), sub(positions[i0], positions[i2])));
            // best edge-sharing candidate by normal agreement
            float bestDot = -2.0f;
            for (EdgeEntry* current = found, *prev = foundPrev; current != nullptr; prev = current, current = current->Next) {
                if (bestDot == -2.0f || (current->i1 == i1 && current->i0 == i0)) {
                    const F3 n1 = normalize3(cross3(sub(positions[current->i0], positions[current->i1]), sub(positions[current->i1], positions[current->i2])));
                    const float d = dot3(n0, n1);
                    if (d > bestDot) { found = current; foundPrev = prev; bestDot = d; }
                }
            }
            if (found && found->Face != 0xffffffffu) {
                if (foundPrev != nullptr) foundPrev->Next = found->Next; else hashTable[key] = found->Next;
                adjacency[iFace * 3 + point] = found->Face;
                // remove this face's own (reverse) edge entry
                const uint32_t key2 = i1 % hashSize;
                for (EdgeEntry* current = hashTable[key2], *prev = nullptr; current != nullptr; prev = current, current = current->Next) {
                    if (current->Face == iFace && current->i1 == i0 && current->i0 == i1) {
                        if (prev != nullptr) prev->Next = current->Next; else hashTable[key2] = current->Next;
                        break;
                    }
                }
                bool linked = false;
                for (uint32_t point2 = 0; point2 < point; ++point2) {
                    if (found->Face == adjacency[iFace * 3 + point2]) { linked = true; adjacency[iFace * 3 + point] = 0xffffffffu; break; }
                }
                if (!linked) {
                    uint32_t edge2 = 0;
                    for (; edge2 < 3; ++edge2) {
                        const uint32_t k = indices[found->Face * 3 + edge2];
                        if (k == 0xffffffffu) continue;
                        if (pointRep[k] == i0) break;
                    }
                    if (edge2 < 3) adjacency[found->Face * 3 + edge2] = iFace;
                }
            }
        }
    }
}

bool AddToMeshlet(uint32_t maxVerts, uint32_t maxPrims, InlineMeshlet& m, const uint32_t tri[3]) {      // :490-539
    if (m.uniqueVertexIndices.size() == maxVerts) return false;
    if (m.primitiveIndices.size() == maxPrims) return false;
    const uint32_t Undef = 0xffffffffu;
    uint32_t idx[3] = {Undef, Undef, Undef};
    uint32_t newCount = 3;
    for (uint32_t i = 0; i < m.uniqueVertexIndices.size(); ++i)
        for (uint32_t j = 0; j < 3; ++j) if (m.uniqueVertexIndices[i] == tri[j]) { idx[j] = i; --newCount; }
    if (m.uniqueVertexIndices.size() + newCount > maxVerts) return false;
    for (uint32_t j = 0; j < 3; ++j) if (idx[j] == Undef) { idx[j] = (uint32_t)m.uniqueVertexIndices.size(); m.uniqueVertexIndices.push_back(tri[j]); }
    m.primitiveIndices.push_back((idx[0] & 0x3ffu) | ((idx[1] & 0x3ffu) << 10) | ((idx[2] & 0x3ffu) << 20));
    return true;
}
bool IsMeshletFull(uint32_t maxVerts, uint32_t maxPrims, const InlineMeshlet& m) { return m.uniqueVertexIndices.size() == maxVerts || m.primitiveIndices.size() == maxPrims; }

struct Candidate { uint32_t first; float second; };
void SortCandidates(std::vector<Candidate>& c) {        // std::sort(..., a.second > b.second) with ties kept in place
    for (size_t i = 1; i < c.size(); ++i) {
        const Candidate x = c[i]; size_t j = i;
        while (j > 0 && x.second > c[j - 1].second) { c[j] = c[j - 1]; --j; }
        c[j] = x;
    }
}

// :640-835
void Meshletize(uint32_t maxVerts, uint32_t maxPrims, const uint32_t* indices, uint32_t indexCount, const F3* positions, uint32_t vertexCount,
                std::vector<InlineMeshlet>& output) {
    output.clear();
    const uint32_t triCount = indexCount / 3;
    if (triCount == 0 || vertexCount < 3) return;
    std::vector<uint32_t> adjacency(indexCount);
    BuildAdjacencyList(indices, indexCount, positions, vertexCount, adjacency.data());
    output.emplace_back();
    InlineMeshlet* curr = &output.back();
    std::vector<bool> checklist(triCount);
    std::vector<F3> m_positions, normals;
    std::vector<Candidate> candidates;
    std::unordered_set<uint32_t> candidateCheck;
    F4 psphere = {0, 0, 0, 0}; F3 normal = {0, 0, 0};
    uint32_t triIndex = 0;
    candidates.push_back({triIndex, 0.0f});
    candidateCheck.insert(triIndex);
    while (!candidates.empty()) {
        const uint32_t index = candidates.back().first;
        candidates.pop_back();
        const uint32_t tri[3] = {indices[index * 3 + 2], indices[index * 3 + 1], indices[index * 3]};
        if (AddToMeshlet(maxVerts, maxPrims, *curr, tri)) {
            checklist[index] = true;
            const F3 points[3] = {positions[tri[0]], positions[tri[1]], positions[tri[2]]};
            m_positions.push_back(points[0]); m_positions.push_back(points[1]); m_positions.push_back(points[2]);
            normals.push_back(ComputeNormal(points));
            psphere = MinimumBoundingSphere(m_positions.data(), (uint32_t)m_positions.size());
            curr->boundingSphere = psphere;
            const F4 nsphere = MinimumBoundingSphere(normals.data(), (uint32_t)normals.size());
            normal = normalize3({nsphere.x, nsphere.y, nsphere.z});
            const uint32_t adjIndex = index * 3;
            const uint32_t adj[3] = {adjacency[adjIndex], adjacency[adjIndex + 1], adjacency[adjIndex + 2]};
            for (uint32_t i = 0; i < 3u; ++i) {
                if (adj[i] == 0xffffffffu) continue;
                if (checklist[adj[i]]) continue;
                if (candidateCheck.count(adj[i])) continue;
                candidates.push_back({adj[i], 3.402823466e38f});
                candidateCheck.insert(adj[i]);
            }
            for (uint32_t i = 0; i < (uint32_t)candidates.size(); ++i) {
                const uint32_t candidate = candidates[i].first;
                const uint32_t triIndices[3] = {indices[candidate * 3], indices[candidate * 3 + 1], indices[candidate * 3 + 2]};
                const F3 triVerts[3] = {positions[triIndices[0]], positions[triIndices[1]], positions[triIndices[2]]};
                candidates[i].second = ComputeScore(*curr, psphere, normal, triIndices, triVerts);
            }
            if (IsMeshletFull(maxVerts, maxPrims, *curr)) {
                m_positions.clear(); normals.clear(); candidateCheck.clear();
                if (!candidates.empty()) { candidates[0] = candidates.back(); candidates.resize(1); candidateCheck.insert(candidates[0].first); }
                output.emplace_back();
                curr = &output.back();
            } else SortCandidates(candidates);
        } else {
            if (candidates.empty()) {
                m_positions.clear(); normals.clear(); candidateCheck.clear();
                output.emplace_back();
                curr = &output.back();
            }
        }
        if (candidates.empty()) {
            while (triIndex < triCount && checklist[triIndex]) ++triIndex;
            if (triIndex == triCount) break;
            candidates.push_back({triIndex, 0.0f});
            candidateCheck.insert(triIndex);
        }
    }
    if (output.back().primitiveIndices.empty()) output.pop_back();
}

} // namespace

// C interface for oracle/binding.py: two calls (size, then copy).  positions: vertex_count x 3 floats.
struct orc_meshlets { std::vector<InlineMeshlet> m; };
extern "C" {
orc_meshlets* orc_meshletize(const uint32_t* indices, uint32_t index_count, const float* positions, uint32_t vertex_count, uint32_t max_verts, uint32_t max_prims,
                             uint32_t* num_meshlets, uint32_t* num_vertex_indices, uint32_t* num_triangles) {
    orc_meshlets* r = new orc_meshlets();
    for (uint32_t i = 0; i < index_count / 3 * 3; ++i) if (indices[i] >= vertex_count) { *num_meshlets = *num_vertex_indices = *num_triangles = 0; return r; }
    Meshletize(max_verts, max_prims, indices, index_count / 3 * 3, (const F3*)positions, vertex_count, r->m);
    uint32_t nv = 0, nt = 0;
    for (const InlineMeshlet& m : r->m) { nv += (uint32_t)m.uniqueVertexIndices.size(); nt += (uint32_t)m.primitiveIndices.size(); }
    *num_meshlets = (uint32_t)r->m.size(); *num_vertex_indices = nv; *num_triangles = nt;
    return r;
}
// per meshlet: counts[4k..] = {vertexBegin, vertexCount, triangleBegin, triangleCount}, spheres[4k..] = bounding sphere
void orc_meshlets_copy(const orc_meshlets* r, uint32_t* counts, float* spheres, uint32_t* vertex_indices, uint32_t* packed_triangles) {
    uint32_t vb = 0, tb = 0;
    for (size_t k = 0; k < r->m.size(); ++k) {
        const InlineMeshlet& m = r->m[k];
        counts[4 * k] = vb; counts[4 * k + 1] = (uint32_t)m.uniqueVertexIndices.size(); counts[4 * k + 2] = tb; counts[4 * k + 3] = (uint32_t)m.primitiveIndices.size();
        spheres[4 * k] = m.boundingSphere.x; spheres[4 * k + 1] = m.boundingSphere.y; spheres[4 * k + 2] = m.boundingSphere.z; spheres[4 * k + 3] = m.boundingSphere.w;
        for (uint32_t v : m.uniqueVertexIndices) vertex_indices[vb++] = v;
        for (uint32_t t : m.primitiveIndices) packed_triangles[tb++] = t;
    }
}
void orc_meshlets_free(orc_meshlets* r) { delete r; }
}
