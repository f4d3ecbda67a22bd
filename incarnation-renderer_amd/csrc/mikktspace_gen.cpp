// mikktspace_gen.cpp -- tangents for primitives that ship none (host side of the scene ingest).
//
// Reference: MeshUtils::FixupMeshes (source/demo-dll/src/mesh-utils.cpp:560-638) runs MikkTSpace's genTangSpaceDefault on every
// primitive whose material has a normal map and whose attributes lack TANGENT; its callbacks read POSITION / NORMAL / TEXCOORD_0
// through the index buffer and store float4(tangent, sign) at the INDEXED vertex (:28-112), so of the faces around a vertex the
// last one in face order decides what the vertex keeps.  The library is an empty submodule of the reference tree (ext/MikkTSpace);
// this is an implementation of its published algorithm (mikktspace.c, M. Mikkelsen, 2011; triangles, angular threshold 180 degrees)
// with this library's own data structures -- corners welded through a hash map instead of the cell-and-partition merge, a fan
// walk with an explicit stack instead of recursion, structure-of-arrays state -- arranged so that every float operation happens
// on the same operands in the same order as in mikktspace.c: the result equals the step-by-step restatement in oracle/oracle_mikkt.cpp
// bit for bit (tests/test_mikktspace.py) and the tangents the Damaged Helmet asset ships to 2e-7.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/incr.h"

namespace {

struct V3f { float x, y, z; };
inline V3f sub(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3f add(V3f a, V3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3f mul(float s, V3f v) { return {s * v.x, s * v.y, s * v.z}; }
inline float dot(V3f a, V3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float len(V3f v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
inline bool nz(float f) { return fabsf(f) > FLT_MIN; }
inline bool nz(V3f v) { return nz(v.x) || nz(v.y) || nz(v.z); }
inline V3f unit(V3f v) { return mul(1 / len(v), v); }
inline V3f unit_if(V3f v) { return nz(v) ? unit(v) : v; }
inline V3f reject(V3f v, V3f n) { return sub(v, mul(dot(n, v), n)); }          // v minus its part along n

struct Mesh {
    const float* pos; const float* nrm; const float* uv; const uint32_t* idx;
    V3f P(uint32_t corner) const { const float* p = pos + (size_t)idx[corner] * 3; return {p[0], p[1], p[2]}; }
    V3f N(uint32_t corner) const { const float* p = nrm + (size_t)idx[corner] * 3; return {p[0], p[1], p[2]}; }
    const float* T(uint32_t corner) const { return uv + (size_t)idx[corner] * 2; }
};

// corners with the same position, normal and texture coordinate (float ==: -0 equals +0, a NaN equals nothing) get one label
struct WeldKey { uint32_t w[8]; bool operator==(const WeldKey& o) const { return !memcmp(w, o.w, sizeof w); } };
struct WeldHash { size_t operator()(const WeldKey& k) const { uint64_t h = 1469598103934665603ull; for (uint32_t v : k.w) { h ^= v; h *= 1099511628211ull; } return (size_t)h; } };
static void weld(const Mesh& M, uint32_t corners, std::vector<uint32_t>& label) {
    label.resize(corners);
    std::unordered_map<WeldKey, uint32_t, WeldHash> seen;
    seen.reserve(corners);
    for (uint32_t c = 0; c < corners; ++c) {
        const V3f p = M.P(c), n = M.N(c); const float* t = M.T(c);
        const float f[8] = {p.x, p.y, p.z, n.x, n.y, n.z, t[0], t[1]};
        WeldKey k; bool nan = false;
        for (int i = 0; i < 8; ++i) { const float v = f[i] == 0.f ? 0.f : f[i]; nan = nan || v != v; memcpy(&k.w[i], &v, 4); }
        if (nan) { label[c] = c; continue; }
        label[c] = seen.emplace(k, c).first->second;
    }
}

enum : uint8_t { kDegenerate = 1, kAnyGroup = 4, kOrientPreserving = 8 };

struct Tris {          // per triangle, in PROCESSING order (good triangles first, as mikktspace.c's DegenPrologue leaves them)
    std::vector<uint32_t> face;                 // original face
    std::vector<uint32_t> v;                    // 3 labels
    std::vector<uint8_t> flag;
    std::vector<V3f> os, ot; std::vector<float> magS, magT;
    std::vector<int32_t> nb;                    // 3 neighbours across the edges (v0 v1), (v1 v2), (v2 v0)
    std::vector<int32_t> group;                 // 3: the group each corner belongs to (-1 none)
};
struct Frame { V3f os; float magS; V3f ot; float magT; int counter; bool orient; };

static Frame frame_of_subgroup(const Mesh& M, const Tris& T, const int* members, int count, uint32_t rep) {
    Frame r = {{0, 0, 0}, 0, {0, 0, 0}, 0, 0, false};
    float angleSum = 0;
    for (int m = 0; m < count; ++m) {
        const int f = members[m];
        if (T.flag[(size_t)f] & kAnyGroup) continue;
        const uint32_t* tv = &T.v[(size_t)f * 3];
        const int i = tv[0] == rep ? 0 : (tv[1] == rep ? 1 : (tv[2] == rep ? 2 : -1));
        if (i < 0) continue;
        const V3f n = M.N(tv[i]);
        const V3f os = unit_if(reject(T.os[(size_t)f], n)), ot = unit_if(reject(T.ot[(size_t)f], n));
        const V3f p0 = M.P(tv[i > 0 ? i - 1 : 2]), p1 = M.P(tv[i]), p2 = M.P(tv[i < 2 ? i + 1 : 0]);
        const V3f e1 = unit_if(reject(sub(p0, p1), n)), e2 = unit_if(reject(sub(p2, p1), n));
        float c = dot(e1, e2); c = c > 1 ? 1 : (c < -1 ? -1 : c);
        const float angle = (float)acos(c);            // the corner's angle weights the face
        r.os = add(r.os, mul(angle, os)); r.ot = add(r.ot, mul(angle, ot));
        r.magS += angle * T.magS[(size_t)f]; r.magT += angle * T.magT[(size_t)f];
        angleSum += angle;
    }
    r.os = unit_if(r.os); r.ot = unit_if(r.ot);
    if (angleSum > 0) { r.magS /= angleSum; r.magT /= angleSum; }
    return r;
}

} // namespace

extern "C" int incr_generate_tangents(const float* positions, const float* normals, const float* uvs, const uint32_t* indices,
                                      uint32_t vertex_count, uint32_t triangle_count, float* out_tangents) {
    if (!positions || !normals || !uvs || !indices || !out_tangents) return INCR_ERR_INVALID_ARGUMENT;
    if (triangle_count > 0x1fffffffu) return INCR_ERR_INVALID_ARGUMENT;
    const uint32_t corners = triangle_count * 3;
    for (uint32_t c = 0; c < corners; ++c) if (indices[c] >= vertex_count) return INCR_ERR_INVALID_ARGUMENT;
    memset(out_tangents, 0, (size_t)vertex_count * 16);
    if (!triangle_count) return INCR_OK;
    const Mesh M = {positions, normals, uvs, indices};
    std::vector<uint32_t> label;
    weld(M, corners, label);

    // processing order: a degenerate triangle (two corners at one position) trades places with the next good one behind it, as long
    // as there is one -- mikktspace.c's DegenPrologue; the order matters because faces are later summed in ascending processing number
    const uint32_t total = triangle_count;
    std::vector<uint32_t> order(total); std::vector<uint8_t> degenerate(total, 0);
    uint32_t good = 0;
    for (uint32_t t = 0; t < total; ++t) {
        order[t] = t;
        const V3f a = M.P(t * 3), b = M.P(t * 3 + 1), c = M.P(t * 3 + 2);
        auto same = [](V3f p, V3f q) { return p.x == q.x && p.y == q.y && p.z == q.z; };
        degenerate[t] = same(a, b) || same(a, c) || same(b, c);
        good += !degenerate[t];
    }
    {
        uint32_t search = 1, t = 0; bool finding = true;
        while (t < good && finding) {
            if (!degenerate[order[t]]) { if (search < t + 2) search = t + 2; }
            else {
                while (search < total && degenerate[order[search]]) ++search;
                if (search < total) { std::swap(order[t], order[search]); ++search; } else { ++search; finding = false; }
            }
            if (finding) ++t;
        }
    }
    Tris T;
    T.face = order;
    T.v.resize((size_t)total * 3); T.flag.assign(total, 0);
    for (uint32_t t = 0; t < total; ++t) {
        for (int i = 0; i < 3; ++i) T.v[(size_t)t * 3 + i] = label[order[t] * 3 + i];
        if (degenerate[order[t]]) T.flag[t] |= kDegenerate;
    }
    // first-order derivatives of the good triangles
    T.os.assign(good, {0, 0, 0}); T.ot.assign(good, {0, 0, 0}); T.magS.assign(good, 0.f); T.magT.assign(good, 0.f);
    T.nb.assign((size_t)good * 3, -1); T.group.assign((size_t)good * 3, -1);
    for (uint32_t t = 0; t < good; ++t) {
        T.flag[t] |= kAnyGroup;
        const uint32_t* tv = &T.v[(size_t)t * 3];
        const V3f v1 = M.P(tv[0]), v2 = M.P(tv[1]), v3 = M.P(tv[2]);
        const float* t1 = M.T(tv[0]); const float* t2 = M.T(tv[1]); const float* t3 = M.T(tv[2]);
        const float t21x = t2[0] - t1[0], t21y = t2[1] - t1[1], t31x = t3[0] - t1[0], t31y = t3[1] - t1[1];
        const V3f d1 = sub(v2, v1), d2 = sub(v3, v1);
        const float area2 = t21x * t31y - t21y * t31x;
        const V3f os = sub(mul(t31y, d1), mul(t21y, d2)), ot = add(mul(-t31x, d1), mul(t21x, d2));
        if (area2 > 0) T.flag[t] |= kOrientPreserving;
        if (nz(area2)) {
            const float absArea = fabsf(area2), lenS = len(os), lenT = len(ot);
            const float s = (T.flag[t] & kOrientPreserving) ? 1.0f : -1.0f;
            if (nz(lenS)) T.os[t] = mul(s / lenS, os);
            if (nz(lenT)) T.ot[t] = mul(s / lenT, ot);
            T.magS[t] = lenS / absArea; T.magT[t] = lenT / absArea;
            if (nz(T.magS[t]) && nz(T.magT[t])) T.flag[t] &= (uint8_t)~kAnyGroup;
        }
    }
    // neighbours: edges sorted by (low label, high label, triangle); an edge takes the first later entry of the same pair that runs
    // the other way round and is still free
    {
        struct Edge { uint32_t a, b, t; };
        std::vector<Edge> edges((size_t)good * 3);
        for (uint32_t t = 0; t < good; ++t)
            for (int i = 0; i < 3; ++i) { const uint32_t p = T.v[(size_t)t * 3 + i], q = T.v[(size_t)t * 3 + (i < 2 ? i + 1 : 0)]; edges[(size_t)t * 3 + i] = {std::min(p, q), std::max(p, q), t}; }
        std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.t < y.t); });
        auto directed = [&](uint32_t t, uint32_t a, uint32_t b, uint32_t* from, uint32_t* to) {      // which edge of t joins a and b, and its direction
            const uint32_t* tv = &T.v[(size_t)t * 3];
            if (tv[0] == a || tv[0] == b) { if (tv[1] == a || tv[1] == b) { *from = tv[0]; *to = tv[1]; return 0; } *from = tv[2]; *to = tv[0]; return 2; }
            *from = tv[1]; *to = tv[2]; return 1;
        };
        for (size_t i = 0; i < edges.size(); ++i) {
            uint32_t fa, ta; const int ea = directed(edges[i].t, edges[i].a, edges[i].b, &fa, &ta);
            if (T.nb[(size_t)edges[i].t * 3 + ea] != -1) continue;
            for (size_t j = i + 1; j < edges.size() && edges[j].a == edges[i].a && edges[j].b == edges[i].b; ++j) {
                uint32_t fb, tb; const int eb = directed(edges[j].t, edges[j].a, edges[j].b, &fb, &tb);
                if (fa == tb && ta == fb && T.nb[(size_t)edges[j].t * 3 + eb] == -1) { T.nb[(size_t)edges[i].t * 3 + ea] = (int32_t)edges[j].t; T.nb[(size_t)edges[j].t * 3 + eb] = (int32_t)edges[i].t; break; }
            }
        }
    }
    // groups: the fan of faces around a welded vertex that can share its tangent space (same orientation, connected across edges)
    struct Group { uint32_t rep; bool orient; uint32_t first, count; };
    std::vector<Group> groups; std::vector<int> groupFaces; groupFaces.reserve((size_t)good * 3);
    {
        std::vector<uint32_t> stack;
        for (uint32_t f = 0; f < good; ++f)
            for (int i = 0; i < 3; ++i) {
                if ((T.flag[f] & kAnyGroup) || T.group[(size_t)f * 3 + i] != -1) continue;
                const int g = (int)groups.size();
                groups.push_back({T.v[(size_t)f * 3 + i], (T.flag[f] & kOrientPreserving) != 0, (uint32_t)groupFaces.size(), 0u});
                T.group[(size_t)f * 3 + i] = g; groupFaces.push_back((int)f);
                // depth-first, the neighbour across the edge leaving the vertex before the one across the edge arriving at it
                stack.clear();
                const int32_t l = T.nb[(size_t)f * 3 + i], r = T.nb[(size_t)f * 3 + (i > 0 ? i - 1 : 2)];
                if (r >= 0) stack.push_back((uint32_t)r);
                if (l >= 0) stack.push_back((uint32_t)l);
                while (!stack.empty()) {
                    const uint32_t t = stack.back(); stack.pop_back();
                    const uint32_t* tv = &T.v[(size_t)t * 3];
                    const uint32_t rep = groups[(size_t)g].rep;
                    const int k = tv[0] == rep ? 0 : (tv[1] == rep ? 1 : (tv[2] == rep ? 2 : -1));
                    if (k < 0 || T.group[(size_t)t * 3 + k] != -1) continue;          // already in this or another group
                    if (T.flag[t] & kAnyGroup) {
                        // a triangle without a texture-space area takes the orientation of the first group that reaches it
                        if (T.group[(size_t)t * 3] == -1 && T.group[(size_t)t * 3 + 1] == -1 && T.group[(size_t)t * 3 + 2] == -1)
                            T.flag[t] = (uint8_t)((T.flag[t] & ~kOrientPreserving) | (groups[(size_t)g].orient ? kOrientPreserving : 0));
                    }
                    if (((T.flag[t] & kOrientPreserving) != 0) != groups[(size_t)g].orient) continue;
                    T.group[(size_t)t * 3 + k] = g; groupFaces.push_back((int)t);
                    const int32_t l2 = T.nb[(size_t)t * 3 + k], r2 = T.nb[(size_t)t * 3 + (k > 0 ? k - 1 : 2)];
                    if (r2 >= 0) stack.push_back((uint32_t)r2);
                    if (l2 >= 0) stack.push_back((uint32_t)l2);
                }
                groups[(size_t)g].count = (uint32_t)groupFaces.size() - groups[(size_t)g].first;
            }
    }
    // tangent frames per corner (slot = 3 x original face + corner)
    std::vector<Frame> frames((size_t)total * 3, Frame{{1, 0, 0}, 1, {0, 1, 0}, 1, 0, false});
    const float thresCos = (float)cos((180.0f * (float)3.14159265358979323846) / 180.0f);      // genTangSpaceDefault: 180 degrees
    {
        std::vector<int> members; std::vector<std::vector<int>> subgroups; std::vector<Frame> subFrames;
        for (size_t g = 0; g < groups.size(); ++g) {
            const Group& G = groups[g];
            const int* faces = &groupFaces[G.first];
            subgroups.clear(); subFrames.clear();
            for (uint32_t a = 0; a < G.count; ++a) {
                const int f = faces[a];
                const int i = T.group[(size_t)f * 3] == (int)g ? 0 : (T.group[(size_t)f * 3 + 1] == (int)g ? 1 : 2);
                const V3f n = M.N(T.v[(size_t)f * 3 + i]);
                const V3f os = unit_if(reject(T.os[(size_t)f], n)), ot = unit_if(reject(T.ot[(size_t)f], n));
                members.clear();
                for (uint32_t b = 0; b < G.count; ++b) {
                    const int t = faces[b];
                    const V3f os2 = unit_if(reject(T.os[(size_t)t], n)), ot2 = unit_if(reject(T.ot[(size_t)t], n));
                    const bool any = ((T.flag[(size_t)f] | T.flag[(size_t)t]) & kAnyGroup) != 0;
                    const float cs = dot(os, os2), ct = dot(ot, ot2);
                    if (any || T.face[(size_t)f] == T.face[(size_t)t] || (cs > thresCos && ct > thresCos)) members.push_back(t);
                }
                std::sort(members.begin(), members.end());
                size_t s = 0;
                while (s < subgroups.size() && subgroups[s] != members) ++s;
                if (s == subgroups.size()) { subgroups.push_back(members); subFrames.push_back(frame_of_subgroup(M, T, members.data(), (int)members.size(), G.rep)); }
                Frame& out = frames[(size_t)T.face[(size_t)f] * 3 + i];
                if (out.counter == 1) {
                    const Frame& b = subFrames[s];
                    const bool equal = out.magS == b.magS && out.magT == b.magT && out.os.x == b.os.x && out.os.y == b.os.y && out.os.z == b.os.z && out.ot.x == b.ot.x && out.ot.y == b.ot.y && out.ot.z == b.ot.z;
                    if (!equal) { out.magS = 0.5f * (out.magS + b.magS); out.magT = 0.5f * (out.magT + b.magT); out.os = unit_if(add(out.os, b.os)); out.ot = unit_if(add(out.ot, b.ot)); }
                    out.counter = 2;
                } else { out = subFrames[s]; out.counter = 1; }
                out.orient = G.orient;
            }
        }
    }
    // a degenerate triangle's corners take the frame of the first good corner on the same welded vertex
    for (uint32_t t = good; t < total; ++t)
        for (int i = 0; i < 3; ++i) {
            const uint32_t want = T.v[(size_t)t * 3 + i];
            for (size_t j = 0; j < (size_t)good * 3; ++j) if (T.v[j] == want) { frames[(size_t)T.face[t] * 3 + i] = frames[(size_t)T.face[j / 3] * 3 + j % 3]; break; }
        }
    // SetTSpaceBasic (mesh-utils.cpp:105-112): float4(tangent, +-1) at the indexed vertex, faces in order
    for (uint32_t c = 0; c < corners; ++c) {
        const Frame& fr = frames[c];
        float* dst = out_tangents + (size_t)indices[c] * 4;
        dst[0] = fr.os.x; dst[1] = fr.os.y; dst[2] = fr.os.z; dst[3] = fr.orient ? 1.0f : -1.0f;
    }
    return INCR_OK;
}
