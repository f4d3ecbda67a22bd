// scene_build.cpp -- host half of the scene path: flatten -> SAH BVH2 -> BVH4 -> baked shading records.
// See scene_build.h.  Compiled with -ffp-contract=off: the world-space vertex positions are part of the
// parity contract (tests compare hit t/u/v bit for bit against the oracle's flattening).
#include "scene_build.h"
#include "test_hooks.h"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cfloat>
#include <cstdio>
#include <limits>
#include <numeric>

namespace incr {

void copy_scene_desc(const incr_scene_desc& d, HostScene& s) {
    s.buffers.resize(d.num_buffers);
    for (uint32_t i = 0; i < d.num_buffers; ++i) {
        const uint8_t* p = (const uint8_t*)d.buffers[i];
        s.buffers[i].assign(p, p + d.buffer_sizes[i]);
        s.buffers[i].resize(s.buffers[i].size() + 16, 0);   // slack for the dword-aligned 16-bit index unpack
    }
    s.bufferViews.assign(d.buffer_views, d.buffer_views + d.num_buffer_views);
    s.accessors.assign(d.accessors, d.accessors + d.num_accessors);
    s.primitiveCounts.assign(d.primitive_counts, d.primitive_counts + d.num_meshes);
    s.primitiveOffset.assign(d.num_meshes + 1, 0);
    for (uint32_t i = 0; i < d.num_meshes; ++i) s.primitiveOffset[i + 1] = s.primitiveOffset[i] + s.primitiveCounts[i];
    s.meshTransforms.resize(d.num_meshes);
    for (uint32_t i = 0; i < d.num_meshes; ++i) s.meshTransforms[i] = m44_from(d.mesh_transforms[i]);
    s.meshVisibility.assign(d.num_meshes, 1u);
    if (d.mesh_visibility) s.meshVisibility.assign(d.mesh_visibility, d.mesh_visibility + d.num_meshes);
    s.primitives.assign(d.primitives, d.primitives + d.num_primitives);
    s.materials.assign(d.materials, d.materials + d.num_materials);
    s.meshlets.assign(d.meshlets, d.meshlets + d.num_meshlets);
    s.meshletVertexIndices.assign(d.meshlet_vertex_indices, d.meshlet_vertex_indices + d.num_meshlet_vertex_indices);
    s.meshletPackedTriangles.assign(d.meshlet_packed_triangles, d.meshlet_packed_triangles + d.num_meshlet_triangles);
    s.globalLights.assign(d.global_lights, d.global_lights + d.num_global_lights);
    s.lightIndices.assign(d.light_indices, d.light_indices + d.num_scene_lights);
    s.lightTransforms.resize(d.num_scene_lights);
    for (uint32_t i = 0; i < d.num_scene_lights; ++i) s.lightTransforms[i] = m44_from(d.light_transforms[i]);
    s.rootTransform = m44_from(d.root_transform);
    memcpy(s.sunDir, d.sun_dir, 12);
    s.maxAnisotropy = d.max_anisotropy ? d.max_anisotropy : 16u;
    s.decalPrimitives.clear(); s.decalTransforms.clear();
    if (d.num_decal_primitives && d.decal_primitives && d.decal_transforms) {
        s.decalPrimitives.assign(d.decal_primitives, d.decal_primitives + d.num_decal_primitives);
        s.decalTransforms.resize(d.num_decal_primitives);
        for (uint32_t i = 0; i < d.num_decal_primitives; ++i) s.decalTransforms[i] = m44_from(d.decal_transforms[i]);
    }
}

// ---- accessor fetchers: the rules of shaders/common/mesh-material.hlsli:45-170, resolved on the host ----
struct AccessorRef { const uint8_t* base; uint32_t stride; uint64_t avail; };
static bool resolve_accessor(const HostScene& s, int accessorIndex, AccessorRef* out) {
    if (accessorIndex < 0 || (size_t)accessorIndex >= s.accessors.size()) return false;   // "== -1 -> 0"
    const incr_mesh_accessor& a = s.accessors[(size_t)accessorIndex];
    if (a.bufferViewIndex < 0 || (size_t)a.bufferViewIndex >= s.bufferViews.size()) return false;
    const incr_mesh_buffer_view& v = s.bufferViews[(size_t)a.bufferViewIndex];
    if (v.bufferSrvIndex < 0 || (size_t)v.bufferSrvIndex >= s.buffers.size()) return false;
    const std::vector<uint8_t>& b = s.buffers[(size_t)v.bufferSrvIndex];
    const uint64_t off = (uint64_t)a.byteOffset + v.byteOffset;
    if (off > b.size()) return false;
    out->base = b.data() + off; out->stride = a.byteStride; out->avail = b.size() - off;
    return true;
}
static inline bool in_range(const AccessorRef& r, uint64_t index, uint32_t bytes) { return index * r.stride + bytes <= r.avail; }

static void fetch_uint3(const AccessorRef* r, uint32_t index, uint32_t out[3]) {          // GetUint3 :45-90
    out[0] = out[1] = out[2] = 0;
    if (!r) return;
    if (r->stride == 4) { if (in_range(*r, index + 2, 4)) memcpy(out, r->base + (size_t)index * 4, 12); }
    else if (r->stride == 2) { if (in_range(*r, index + 2, 2)) { uint16_t t[3]; memcpy(t, r->base + (size_t)index * 2, 6); out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; } }
}
static void fetch_floats(const AccessorRef* r, uint32_t index, int ncomp, float* out) {     // GetFloat2/3/4 :92-170
    for (int i = 0; i < ncomp; ++i) out[i] = 0.f;
    if (!r) return;
    bool ok = false;
    if (ncomp == 4) ok = r->stride == 16;
    else if (ncomp == 3) ok = r->stride == 12 || r->stride == 16;
    else if (ncomp == 2) ok = r->stride == 8 || r->stride == 12 || r->stride == 16;
    if (ok && in_range(*r, index, (uint32_t)ncomp * 4)) memcpy(out, r->base + (size_t)index * r->stride, (size_t)ncomp * 4);
}

bool validate_scene(const HostScene& s, const char** why) {
    static thread_local char msg[160];
    *why = msg;
    if (s.maxAnisotropy < 1u || s.maxAnisotropy > 16u) { snprintf(msg, sizeof msg, "max_anisotropy %u: D3D12_REQ_MAXANISOTROPY is 16", s.maxAnisotropy); return false; }
    if (s.primitiveOffset.back() != s.primitives.size()) { snprintf(msg, sizeof msg, "sum(primitive_counts)=%u != num_primitives=%zu", s.primitiveOffset.back(), s.primitives.size()); return false; }
    for (size_t i = 0; i < s.primitives.size(); ++i) {
        const incr_gpu_primitive& p = s.primitives[i];
        if (p.materialIndex < 0 || (size_t)p.materialIndex >= s.materials.size()) { snprintf(msg, sizeof msg, "primitive %zu: materialIndex %d out of range", i, p.materialIndex); return false; }
        if (p.meshIndex < 0 || (size_t)p.meshIndex >= s.meshTransforms.size()) { snprintf(msg, sizeof msg, "primitive %zu: meshIndex %d out of range", i, p.meshIndex); return false; }
        if (p.indexCount < 0 || p.indicesPerTriangle != 3) { snprintf(msg, sizeof msg, "primitive %zu: indexCount/indicesPerTriangle invalid", i); return false; }
        AccessorRef r;
        if (!resolve_accessor(s, p.indexAccessor, &r) || !resolve_accessor(s, p.positionAccessor, &r)) { snprintf(msg, sizeof msg, "primitive %zu: index/position accessor unresolved", i); return false; }
    }
    for (size_t i = 0; i < s.lightIndices.size(); ++i)
        if (s.lightIndices[i] < 0 || (size_t)s.lightIndices[i] >= s.globalLights.size()) { snprintf(msg, sizeof msg, "light_indices[%zu] out of range", i); return false; }
    *why = "";
    return true;
}

// ---- SAH BVH2 over triangle boxes ------------------------------------------------------------------
// SAH weights of the wide-BVH collapse: a wide-node visit against one triangle test (tools/trav_sim measures the step counts)
#ifndef INCR_BVH_COST_NODE
#define INCR_BVH_COST_NODE 1.0f
#endif
#ifndef INCR_LIGHT_GRID_CELLS
#define INCR_LIGHT_GRID_CELLS 64
#endif
#ifndef INCR_BVH_COST_TRI
#define INCR_BVH_COST_TRI 1.0f
#endif
namespace {
struct Box { float lo[3], hi[3]; };
static inline void box_reset(Box& b) { for (int a = 0; a < 3; ++a) { b.lo[a] = FLT_MAX; b.hi[a] = -FLT_MAX; } }
static inline void box_grow(Box& b, const Box& o) { for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); } }
static inline float box_area(const Box& b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
}
struct Node2 { Box box; uint32_t left, right, first, count; uint32_t spanFirst, spanCount; };   // count > 0: leaf over order[first .. first+count); span*: the subtree's run of `order` (every node)

struct Builder {
    const Box* tb; const V3* cen;
    std::vector<uint32_t> order;
    std::vector<Node2> nodes;
    // 32 bins: 1 % fewer node visits than 16 on Sponza-stat; traversal cost 0.5 / 2, leaf limits 2 / 3 and splitting pairs all
    // moved the step count by < 2 % either way (tools/trav_sim)
    static constexpr int kBins = 32;
    static constexpr int bins = kBins;

    // over the items [first, first + count) of tb / cen: `order` holds their indices, build(0, count) is the root
    Builder(const Box* tb_, const V3* cen_, uint32_t first, uint32_t count) : tb(tb_), cen(cen_) {
        order.resize(count); std::iota(order.begin(), order.end(), first);
        nodes.reserve(count ? 2 * (size_t)count : 1);
    }
    static inline float axis_of(const V3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

    // depth-capped: past sahDepth levels the split is the object median along the longest centroid axis, so the
    // recursion (and the tree) stays logarithmic however the SAH behaves on pathological input
    uint32_t sahDepth = 48;
    uint32_t build(uint32_t first, uint32_t count, uint32_t depth = 0) {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.push_back({});
        Box bb, cb; box_reset(bb); box_reset(cb);
        for (uint32_t i = first; i < first + count; ++i) {
            const uint32_t t = order[i];
            box_grow(bb, tb[t]);
            const float c[3] = {cen[t].x, cen[t].y, cen[t].z};
            for (int a = 0; a < 3; ++a) { cb.lo[a] = std::min(cb.lo[a], c[a]); cb.hi[a] = std::max(cb.hi[a], c[a]); }
        }
        nodes[me].box = bb; nodes[me].spanFirst = first; nodes[me].spanCount = count;
        if (count <= 1) { nodes[me].first = first; nodes[me].count = count; return me; }
        // binned SAH over the three axes
        float bestCost = FLT_MAX; int bestAxis = -1, bestSplit = -1;
        for (int a = 0; a < 3; ++a) {
            const float lo = cb.lo[a], ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            Box binsB[kBins]; uint32_t cnt[kBins];
            for (int b = 0; b < bins; ++b) { box_reset(binsB[b]); cnt[b] = 0; }
            const float scale = (float)bins / ext;
            for (uint32_t i = first; i < first + count; ++i) {
                const uint32_t t = order[i];
                int b = (int)((axis_of(cen[t], a) - lo) * scale); b = std::min(std::max(b, 0), bins - 1);
                box_grow(binsB[b], tb[t]); cnt[b]++;
            }
            float rightArea[kBins]; uint32_t rightCnt[kBins];
            Box acc; box_reset(acc); uint32_t n = 0;
            for (int b = bins - 1; b > 0; --b) { box_grow(acc, binsB[b]); n += cnt[b]; rightArea[b] = box_area(acc); rightCnt[b] = n; }
            box_reset(acc); n = 0;
            for (int b = 0; b < bins - 1; ++b) {
                box_grow(acc, binsB[b]); n += cnt[b];
                if (n == 0 || rightCnt[b + 1] == 0) continue;
                const float cost = box_area(acc) * (float)n + rightArea[b + 1] * (float)rightCnt[b + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestSplit = b; }
            }
        }
        uint32_t mid = first;
        const float leafCost = box_area(bb) * (float)count;
        (void)leafCost;      // the binary tree goes down to single triangles: which subtrees become leaves is the collapse's decision
        if (depth < sahDepth && bestAxis >= 0) {
            const float lo = cb.lo[bestAxis], scale = (float)bins / (cb.hi[bestAxis] - cb.lo[bestAxis]);
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = (int)((axis_of(cen[t], bestAxis) - lo) * scale); b = std::min(std::max(b, 0), bins - 1);
                return b <= bestSplit;
            });
            mid = (uint32_t)(it - order.begin());
        }
        if (mid == first || mid == first + count) {
            // all centroids coincide (or past the SAH depth): object median along the longest centroid axis
            mid = first + count / 2;
            int a = 0; float ext = cb.hi[0] - cb.lo[0];
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > ext) { ext = cb.hi[k] - cb.lo[k]; a = k; }
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) {
                const float cx = axis_of(cen[x], a), cy = axis_of(cen[y], a);
                return cx < cy || (cx == cy && x < y);
            });
        }
        const uint32_t l = build(first, mid - first, depth + 1);
        const uint32_t r = build(mid, first + count - mid, depth + 1);
        nodes[me].left = l; nodes[me].right = r; nodes[me].count = 0;
        return me;
    }
};

// ---- 8-wide subtrees ----------------------------------------------------------------------------------------------------------
// The world-space tree is TWO-LEVEL IN ONE ARRAY: every mesh owns a subtree built once over its own triangles (the reference's BLAS
// per mesh, scene.cpp:685-772), and a small top tree over the mesh roots (its TLAS, rebuilt every frame by update-tlas.inl:24-89)
// joins them -- same node format, same traversal, no per-instance ray transform.  When meshes move, their subtrees are re-transformed
// and refitted on the device (bvh_update.hip) and the top tree is REBUILT from the new mesh boxes (build_top_tree: a thousand boxes,
// a fraction of a millisecond on the host), so the structure is as good as a fresh build however far the instances travel.
struct WideTree { std::vector<BvhNode8> nodes; std::vector<uint32_t> depth; std::vector<uint32_t> leafOrder; uint32_t maxDepth = 0; Box box; };

// BVH2 `b` (root = node 0 over b.order[0 .. count)) -> 8-wide quantised nodes with local indices
static void collapse_to_wide(const Builder& b, size_t count, WideTree& W) {
    W.nodes.clear(); W.depth.clear(); W.leafOrder.clear();
    // 3. collapse BVH2 -> 8-wide quantised nodes.  Which binary subtrees become the (up to 8) children of a wide node, and which
    //    become leaves of <= kMaxLeafTris triangles, is chosen by dynamic programming over the SAH cost (Ylitie, Karras, Laine
    //    2017, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", section 4.1):
    //      C(n, 1) = min( leaf: A_n * P_n * cTri   [P_n <= kMaxLeafTris],   wide node: A_n * cNode + D(n, 8) )
    //      C(n, i) = min( D(n, i), C(n, i - 1) ),   D(n, j) = min over 0 < k < j of C(left, k) + C(right, j - k)
    //    (C(n, i): cheapest way to present subtree n as at most i children of some wide node).  Round 1 opened the child with the
    //    largest area until a node had 8: on Sponza-stat that left the average node 3.8 children of 8 -- and a node step costs the
    //    same ~300 VALU instructions however many of its slots are in use; the traversal kernels are VALU-bound.
    //    Triangles are laid out in NODE order: the leaf children of one node8 own one contiguous run.
    W.nodes.clear(); W.maxDepth = 0;
    if (count) {
        const size_t n2 = b.nodes.size();
        // INCR_BVH_WIDTH (device_types.h): children per node the traversal tests, 8 unless an A/B build narrows the tree; a narrower node's
        // step is cheaper (the slab tests are 0.65 of it)
        constexpr int WIDTH = INCR_BVH_WIDTH;
        constexpr float cNode = INCR_BVH_COST_NODE * (0.35f + 0.65f * (float)WIDTH / 8.f), cTri = INCR_BVH_COST_TRI;
        struct Dp { float c[8]; uint8_t split[9]; uint8_t wideSplit; bool leaf; };      // wideSplit = k of D(n, WIDTH), the node's own children (split[WIDTH] may be cleared below)      // c[i], i = 1..7; split[j] = k of D(n, j) (j = 2..8), 0 = "use C(n, j - 1)"
        std::vector<Dp> dp(n2);
        for (size_t i = n2; i-- > 0;) {                              // children come after their parent in `nodes`: reverse order is bottom-up
            const Node2& nd2 = b.nodes[i]; Dp& d = dp[i];
            float A = box_area(nd2.box);
            if (!(A >= 0.f)) A = 0.f;                                // NaN boxes
            A = std::min(A, 1e28f);                                  // keeps every sum of costs finite (<= 2^23 triangles)
            const float leafC = nd2.spanCount <= kMaxLeafTris ? A * (float)nd2.spanCount * cTri : FLT_MAX;
            memset(d.split, 0, sizeof d.split); d.wideSplit = 0;
            if (nd2.count) { d.leaf = true; for (int k = 1; k < 8; ++k) d.c[k] = leafC; continue; }
            const Dp& L = dp[nd2.left]; const Dp& R = dp[nd2.right];
            float D[9];
            for (int j = 2; j <= 8; ++j) {
                float bestC = FLT_MAX; int bestK = 1;
                for (int k = 1; k < j; ++k) { if (k > 7 || j - k > 7) continue; const float c = L.c[k] + R.c[j - k]; if (c < bestC) { bestC = c; bestK = k; } }
                D[j] = bestC; d.split[j] = (uint8_t)bestK;
            }
            const float wide = A * cNode + D[WIDTH];
            d.wideSplit = d.split[WIDTH];
            d.leaf = nd2.spanCount <= kMaxLeafTris && leafC <= wide;
            d.c[1] = std::min(leafC, wide);
            for (int j = 2; j < 8; ++j) { if (D[j] < d.c[j - 1]) d.c[j] = D[j]; else { d.c[j] = d.c[j - 1]; d.split[j] = 0; } }
        }
        // the (<= j) children subtree x contributes to a wide node
        struct Pick { uint32_t node; int j; };
        auto collect = [&](uint32_t x, int j, uint32_t* kids, int& nk) {
            Pick stack[32]; int sp = 0; stack[sp++] = {x, j};
            while (sp) {
                Pick pk = stack[--sp];
                const Node2& c = b.nodes[pk.node];
                while (pk.j > 1 && !c.count && dp[pk.node].split[pk.j] == 0) --pk.j;
                if (pk.j == 1 || c.count) { kids[nk++] = pk.node; continue; }
                const int k = dp[pk.node].split[pk.j];
                stack[sp++] = {c.right, pk.j - k}; stack[sp++] = {c.left, k};
            }
        };
        struct Item8 { uint32_t n2, n8, depth; };
        std::vector<Item8> work8;
        W.nodes.push_back({});
        work8.push_back({0u, 0u, 1u});
        while (!work8.empty()) {
            const Item8 it = work8.back(); work8.pop_back();
            W.maxDepth = std::max(W.maxDepth, it.depth);
            if (W.depth.size() < W.nodes.size()) W.depth.resize(W.nodes.size(), 0u);
            W.depth[it.n8] = it.depth;
            uint32_t kids[8]; int nk = 0;
            const Node2& root2 = b.nodes[it.n2];
            if (root2.count || (it.n2 == 0 && dp[0].leaf)) kids[nk++] = it.n2;      // degenerate: the whole tree is one leaf
            else { const int k = dp[it.n2].wideSplit; collect(root2.left, k, kids, nk); collect(root2.right, WIDTH - k, kids, nk); }
            auto is_leaf = [&](uint32_t x) { return dp[x].leaf; };
            BvhNode8 nd; memset(&nd, 0, sizeof nd);
            const Box& nb = b.nodes[it.n2].box;
            // SLOT ORDER (device_types.h): inner children first, sorted by box centre along the node's longest axis; then leaves
            int sortAxis = 0;
            for (int a = 1; a < 3; ++a) if (nb.hi[a] - nb.lo[a] > nb.hi[sortAxis] - nb.lo[sortAxis]) sortAxis = a;
            std::stable_sort(kids, kids + nk, [&](uint32_t x, uint32_t y) {
                const Node2& cx = b.nodes[x]; const Node2& cy = b.nodes[y];
                if (is_leaf(x) != is_leaf(y)) return !is_leaf(x);
                if (is_leaf(x)) return false;                                // leaves keep their collapse order
                return cx.box.lo[sortAxis] + cx.box.hi[sortAxis] < cy.box.lo[sortAxis] + cy.box.hi[sortAxis];
            });
            nd.ox = nb.lo[0]; nd.oy = nb.lo[1]; nd.oz = nb.lo[2];
            float scale[3]; uint8_t ebits[3];
            for (int a = 0; a < 3; ++a) {
                const float ext = std::max(nb.hi[a] - nb.lo[a], 0.f);
                int e = 1;
                if (ext > 0.f) { e = (int)std::ceil(std::log2((double)ext / 255.0)) + 127; e = std::min(std::max(e, 1), 254); }
                for (;;) {     // 255 quanta must cover the extent in float arithmetic
                    uint32_t bits = (uint32_t)e << 23; float sc; memcpy(&sc, &bits, 4);
                    if (nb.lo[a] + 255.f * sc >= nb.hi[a] || e >= 254) { scale[a] = sc; break; }
                    ++e;
                }
                ebits[a] = (uint8_t)e;
            }
            nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
            nd.childBase = (uint32_t)W.nodes.size();
            uint32_t numInner = 0;
            for (int k = 0; k < nk; ++k) if (!is_leaf(kids[k])) ++numInner;
            nd.triBase = (uint32_t)W.leafOrder.size() - 4u * numInner;          // LEAF SLOTS (device_types.h): the leaf in slot k owns records triBase + 4 k .. + 3
            W.nodes.resize(W.nodes.size() + numInner);
            uint32_t innerSlot = 0;
            uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
            const float org[3] = {nd.ox, nd.oy, nd.oz};
            for (int k = 0; k < 8; ++k) {
                if (k >= nk) { for (int a = 0; a < 3; ++a) { qlo[a][k] = 255; qhi[a][k] = 0; } continue; }      // empty slot: inverted box
                const Node2& c = b.nodes[kids[k]];
                for (int a = 0; a < 3; ++a) {
                    int lo = (int)std::floor((c.box.lo[a] - org[a]) / scale[a]), hi = (int)std::ceil((c.box.hi[a] - org[a]) / scale[a]);
                    lo = std::min(std::max(lo, 0), 255); hi = std::min(std::max(hi, 0), 255);
                    while (lo > 0 && org[a] + (float)lo * scale[a] > c.box.lo[a]) --lo;          // conservative under float decode
                    while (hi < 255 && org[a] + (float)hi * scale[a] < c.box.hi[a]) ++hi;
                    qlo[a][k] = (uint8_t)lo; qhi[a][k] = (uint8_t)hi;
                }
                if (is_leaf(kids[k])) {
                    nd.leafTris |= ((1u << c.spanCount) - 1u) << (4 * k);
                    for (uint32_t t = 0; t < 4u; ++t) W.leafOrder.push_back(t < c.spanCount ? b.order[c.spanFirst + t] : kTriHole);
                } else { nd.imask |= (uint8_t)(1u << k); work8.push_back({kids[k], nd.childBase + innerSlot, it.depth + 1}); ++innerSlot; }
            }
            nd.sortAxis = (uint8_t)sortAxis;
            W.nodes[it.n8] = nd;
        }
    } else {
        BvhNode8 e; memset(&e, 0, sizeof e);
        for (int k = 0; k < 8; ++k) { e.qlox[k] = e.qloy[k] = e.qloz[k] = 255; }
        e.ex = e.ey = e.ez = 1;
        W.nodes.push_back(e);
    }

}

static void reorder_best_first(WideTree& W) {
    // NODE ORDER: best-first by box area -- the root, then repeatedly the children block of the largest node placed so far.
    // Index order then approximates visit frequency (the top three levels take ~40 % of all node visits on Sponza-stat,
    // tools/trav_sim), so "index < K" names the K hottest nodes: the traversal kernels keep those in LDS (pt_traverse.h).
    W.depth.resize(W.nodes.size(), 1u);
    if (W.nodes.size() > 1) {
        const size_t nn = W.nodes.size();
        std::vector<float> area(nn, 0.f);
        for (size_t i = 0; i < nn; ++i) {
            const BvhNode8& nd = W.nodes[i];
            const uint8_t eb[3] = {nd.ex, nd.ey, nd.ez}; float ext[3];
            for (int a = 0; a < 3; ++a) { uint32_t bits = (uint32_t)eb[a] << 23; float sc; memcpy(&sc, &bits, 4); ext[a] = 255.f * sc; }
            area[i] = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];          // area of the node's quantisation frame: monotone down the tree
        }
        std::vector<uint32_t> newIndex(nn, 0u);
        std::vector<std::pair<float, uint32_t>> heap;                                // (area, old index), max-heap
        uint32_t next = 1; heap.push_back({area[0], 0u});
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end()); const uint32_t o = heap.back().second; heap.pop_back();
            const BvhNode8& nd = W.nodes[o];
            for (uint32_t k = 0; k < 8; ++k) if ((nd.imask >> k) & 1u) {
                const uint32_t c = nd.childBase + k;
                newIndex[c] = next++;
                heap.push_back({area[c], c}); std::push_heap(heap.begin(), heap.end());
            }
        }
        std::vector<BvhNode8> reordered(nn); std::vector<uint32_t> depth2(nn, 1u);
        for (size_t i = 0; i < nn; ++i) {
            BvhNode8 nd = W.nodes[i];
            if (nd.imask) nd.childBase = newIndex[nd.childBase];
            reordered[newIndex[i]] = nd; depth2[newIndex[i]] = W.depth[i];
        }
        W.nodes.swap(reordered); W.depth.swap(depth2);
    }
}

// A mesh's subtree over the flattened triangles [first, first + count).  The traversal kernels carry one stack entry per tree level
// (kMaxBvhDepth).  The SAH-optimal collapse takes at least one binary level per wide level, so a skewed binary tree (a long strip, a
// teapot in a stadium) can come out deeper than its budget: the build then repeats with the SAH switched off below a shallower binary
// depth -- object-median splits are balanced, ceil(log2 n) <= 23 levels for 2^23 triangles, and the wide tree is never deeper than
// the binary one.
static void build_wide_subtree(const Box* tb, const V3* cen, uint32_t first, uint32_t count, uint32_t depthBudget, WideTree& W) {
    uint32_t depthLimit = depthBudget;
    if (const char* e = test_hook_env("INCR_BVH_DEPTH_LIMIT")) depthLimit = std::min<uint32_t>(depthLimit, (uint32_t)std::max(atoi(e), 1));      // test hook: exercises the rebuilds on ordinary scenes
    box_reset(W.box);
    for (const uint32_t sahDepth : {48u, 12u, 0u}) {
        Builder b(tb, cen, first, count);
        b.sahDepth = sahDepth;
        if (count) { b.build(0, count); W.box = b.nodes[0].box; }
        collapse_to_wide(b, count, W);
        if (W.maxDepth <= depthLimit) break;
    }
    reorder_best_first(W);
}

// origin, exponents and the eight quantised child boxes of a node whose box is `nb` (conservative under the traversal's float decode)
static void quantise_children(BvhNode8& nd, const Box& nb, const Box* child, int nk) {
    nd.ox = nb.lo[0]; nd.oy = nb.lo[1]; nd.oz = nb.lo[2];
    float scale[3]; uint8_t ebits[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = std::max(nb.hi[a] - nb.lo[a], 0.f);
        int e = 1;
        if (ext > 0.f) { e = (int)std::ceil(std::log2((double)ext / 255.0)) + 127; e = std::min(std::max(e, 1), 254); }
        for (;;) { uint32_t bits = (uint32_t)e << 23; float sc; memcpy(&sc, &bits, 4); if (nb.lo[a] + 255.f * sc >= nb.hi[a] || e >= 254) { scale[a] = sc; break; } ++e; }
        ebits[a] = (uint8_t)e;
    }
    nd.ex = ebits[0]; nd.ey = ebits[1]; nd.ez = ebits[2];
    uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
    const float org[3] = {nd.ox, nd.oy, nd.oz};
    for (int k = 0; k < 8; ++k) {
        const bool empty = k >= nk || !(child[k].lo[0] <= child[k].hi[0] && child[k].lo[1] <= child[k].hi[1] && child[k].lo[2] <= child[k].hi[2]);
        for (int a = 0; a < 3; ++a) {
            if (empty) { qlo[a][k] = 255; qhi[a][k] = 0; continue; }           // nothing to hit below: inverted box
            int lo = (int)std::floor((child[k].lo[a] - org[a]) / scale[a]), hi = (int)std::ceil((child[k].hi[a] - org[a]) / scale[a]);
            lo = std::min(std::max(lo, 0), 255); hi = std::min(std::max(hi, 0), 255);
            while (lo > 0 && org[a] + (float)lo * scale[a] > child[k].lo[a]) --lo;
            while (hi < 255 && org[a] + (float)hi * scale[a] < child[k].hi[a]) ++hi;
            qlo[a][k] = (uint8_t)lo; qhi[a][k] = (uint8_t)hi;
        }
    }
}
} // namespace

// The TOP REGION nodes8[0 .. topCapacity): the top tree over the meshes' boxes (binned SAH over the boxes, children opened by area
// until a node has eight) with ONE COPY OF EVERY MESH ROOT as its leaves-that-are-inner-nodes, children of a node contiguous (SLOT
// ORDER), the root at index 0.  meshBoxes: 6 floats per mesh (lo.xyz, hi.xyz; lo > hi = nothing to hit).  Also rewrites the dynamic
// tail of the refit order (mesh roots, then top nodes deepest first) and maxDepth.  Called at scene creation (exact boxes) and by
// incr_update_bvh (conservative boxes of the moved meshes: the device refit that follows tightens every box).
void rebuild_top_tree(HostBvh& out, const float* meshBoxes) {
    const size_t numMeshes = out.meshRoot.size();
    std::vector<uint32_t> items;
    for (size_t m = 0; m < numMeshes; ++m) if (out.meshHasTris[m]) items.push_back((uint32_t)m);
    std::fill(out.nodes8.begin(), out.nodes8.begin() + out.topCapacity, BvhNode8{});
    std::vector<uint32_t> depth(out.topCapacity, 0u);
    std::vector<uint32_t> roots, tops;            // region indices
    uint32_t used = 1;
    auto box_of = [&](uint32_t m) { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = meshBoxes[m * 6 + a]; b.hi[a] = meshBoxes[m * 6 + 3 + a]; } return b; };
    if (items.empty()) {
        BvhNode8 e; memset(&e, 0, sizeof e);
        for (int k = 0; k < 8; ++k) { e.qlox[k] = e.qloy[k] = e.qloz[k] = 255; }
        e.ex = e.ey = e.ez = 1;
        out.nodes8[0] = e; depth[0] = 1; tops.push_back(0);
    } else if (items.size() == 1) {
        out.nodes8[0] = out.meshRoot[items[0]]; depth[0] = 1; roots.push_back(0); out.meshRootIndex[items[0]] = 0;
    } else {
        std::vector<Box> tb(items.size()); std::vector<V3> cen(items.size());
        Box all; box_reset(all);
        for (size_t i = 0; i < items.size(); ++i) { tb[i] = box_of(items[i]); if (tb[i].lo[0] <= tb[i].hi[0]) box_grow(all, tb[i]); }
        for (size_t i = 0; i < items.size(); ++i) {
            if (!(tb[i].lo[0] <= tb[i].hi[0] && tb[i].lo[1] <= tb[i].hi[1] && tb[i].lo[2] <= tb[i].hi[2])) {      // hidden / empty mesh: a point where it disturbs the SAH least
                for (int a = 0; a < 3; ++a) tb[i].lo[a] = tb[i].hi[a] = all.lo[0] <= all.hi[0] ? 0.5f * (all.lo[a] + all.hi[a]) : 0.f;
            }
            cen[i] = v3(0.5f * (tb[i].lo[0] + tb[i].hi[0]), 0.5f * (tb[i].lo[1] + tb[i].hi[1]), 0.5f * (tb[i].lo[2] + tb[i].hi[2]));
        }
        for (const uint32_t sahDepth : {48u, 0u}) {
            Builder b(tb.data(), cen.data(), 0u, (uint32_t)items.size());
            b.sahDepth = sahDepth;
            b.build(0, (uint32_t)items.size());
            std::fill(depth.begin(), depth.end(), 0u); roots.clear(); tops.clear(); used = 1;
            struct Item { uint32_t n2, at, depth; };
            std::vector<Item> work; work.push_back({0u, 0u, 1u});
            uint32_t topDepth = 0;
            while (!work.empty()) {
                const Item it = work.back(); work.pop_back();
                const Node2& x = b.nodes[it.n2];
                uint32_t kids[8]; int nk = 0;
                kids[nk++] = x.left; kids[nk++] = x.right;
                while (nk < INCR_BVH_WIDTH) {          // open the largest inner child until the node has eight
                    int best = -1; float bestA = -1.f;
                    for (int k = 0; k < nk; ++k) if (!b.nodes[kids[k]].count) { const float A = box_area(b.nodes[kids[k]].box); if (A > bestA) { bestA = A; best = k; } }
                    if (best < 0) break;
                    const Node2& c = b.nodes[kids[best]];
                    kids[best] = c.left; kids[nk++] = c.right;
                }
                int sortAxis = 0;
                for (int a = 1; a < 3; ++a) if (x.box.hi[a] - x.box.lo[a] > x.box.hi[sortAxis] - x.box.lo[sortAxis]) sortAxis = a;
                std::stable_sort(kids, kids + nk, [&](uint32_t p, uint32_t q) {
                    const Box& bp = b.nodes[p].box; const Box& bq = b.nodes[q].box;
                    return bp.lo[sortAxis] + bp.hi[sortAxis] < bq.lo[sortAxis] + bq.hi[sortAxis];
                });
                BvhNode8 nd; memset(&nd, 0, sizeof nd);
                Box cb[8];
                for (int k = 0; k < nk; ++k) { cb[k] = b.nodes[kids[k]].box; if (b.nodes[kids[k]].count) cb[k] = box_of(items[b.order[b.nodes[kids[k]].first]]); }      // a mesh's own box (empty for a hidden mesh)
                quantise_children(nd, x.box, cb, nk);
                nd.imask = (uint8_t)((1u << nk) - 1u); nd.sortAxis = (uint8_t)sortAxis; nd.childBase = used;
                used += (uint32_t)nk;
                for (int k = 0; k < nk; ++k) {
                    const Node2& c = b.nodes[kids[k]];
                    const uint32_t at = nd.childBase + (uint32_t)k;
                    depth[at] = it.depth + 1;
                    if (c.count) { const uint32_t m = items[b.order[c.first]]; out.nodes8[at] = out.meshRoot[m]; out.meshRootIndex[m] = at; roots.push_back(at); }
                    else work.push_back({kids[k], at, it.depth + 1});
                }
                out.nodes8[it.at] = nd; depth[it.at] = it.depth; tops.push_back(it.at);
                topDepth = std::max(topDepth, it.depth);
            }
            if (topDepth <= 8u) break;             // (median splits over <= 2^23 meshes: <= 23 binary levels, <= 8 levels of eight)
        }
    }
    out.topCount = used;
    // dynamic tail of the refit order + the tree's depth
    out.refitNodes.resize(out.refitStaticNodes); out.refitRanges.resize(2 * (size_t)out.refitStaticRanges);
    out.maxDepth = 1;
    if (!roots.empty()) { out.refitRanges.push_back((uint32_t)out.refitNodes.size()); out.refitRanges.push_back((uint32_t)roots.size()); out.refitNodes.insert(out.refitNodes.end(), roots.begin(), roots.end()); }
    for (size_t m = 0; m < numMeshes; ++m) if (out.meshHasTris[m]) out.maxDepth = std::max(out.maxDepth, depth[out.meshRootIndex[m]] + out.meshLocalDepth[m] - 1u);
    uint32_t maxTop = 0; for (uint32_t t : tops) maxTop = std::max(maxTop, depth[t]);
    out.maxDepth = std::max(out.maxDepth, maxTop);
    for (uint32_t d = maxTop; d >= 1u; --d) {
        const uint32_t first = (uint32_t)out.refitNodes.size();
        for (uint32_t t : tops) if (depth[t] == d) out.refitNodes.push_back(t);
        if (out.refitNodes.size() > first) { out.refitRanges.push_back(first); out.refitRanges.push_back((uint32_t)out.refitNodes.size() - first); }
    }
}

void refresh_mesh_instances(const HostScene& s, HostBvh& out, std::vector<MeshInstanceInfo>& info) {
    const size_t nm = s.meshTransforms.size();
    out.meshLocalToWorld.resize(nm); info.resize(nm);
    uint32_t inst = 0;
    Box world; box_reset(world);
    for (size_t m = 0; m < nm; ++m) {
        out.meshLocalToWorld[m] = m44_mul(s.meshTransforms[m], s.rootTransform);
        const bool visible = s.meshVisibility[m] != 0;
        info[m].instance = visible ? inst++ : 0xffffffffu;                 // update-tlas.inl:20-48: one instance per VISIBLE mesh
        info[m].mirrored = m44_det3(out.meshLocalToWorld[m]) < 0.f ? 1u : 0u;
        if (visible && out.meshObjBox.size() == nm * 6 && out.meshObjBox[m * 6] <= out.meshObjBox[m * 6 + 3]) {
            const float* b = &out.meshObjBox[m * 6];
            for (int c = 0; c < 8; ++c) {
                const V3 p = m44_transform_point(v3(b[(c & 1) ? 3 : 0], b[(c & 2) ? 4 : 1], b[(c & 4) ? 5 : 2]), out.meshLocalToWorld[m]);
                const float q[3] = {p.x, p.y, p.z};
                for (int a = 0; a < 3; ++a) if (q[a] == q[a]) { world.lo[a] = std::min(world.lo[a], q[a]); world.hi[a] = std::max(world.hi[a], q[a]); }
            }
        }
    }
    out.numInstances = inst;
    const bool any = world.lo[0] <= world.hi[0];
    for (int a = 0; a < 3; ++a) { out.sceneLo[a] = any ? world.lo[a] : 0.f; out.sceneHi[a] = any ? world.hi[a] : 0.f; }
}

namespace {
// Work sharing for the load-time loops below: `count` items in chunks of `grain`, handed out by an atomic cursor to up to 64 host threads
// (the reference builds its acceleration structures on the GPU, scene.cpp:747-843; here the build is host work spread over the cores)
template <class F> void parallel_for(size_t count, size_t grain, F&& fn) {
    const size_t chunks = (count + grain - 1) / std::max<size_t>(grain, 1);
    const size_t nThreads = std::min<size_t>(std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 64), chunks);
    std::atomic<size_t> next{0};
    // An exception must not leave a worker thread (std::terminate), and a thread that failed to start must not leave running ones unjoined: a worker keeps the
    // first exception of its chunks and stops the hand-out; the threads that did start are always joined; the exception is rethrown on the calling thread, where
    // incr_scene_create turns it into an error code.
    std::exception_ptr failure; std::mutex failureLock;
    auto worker = [&]() {
        try { for (;;) { const size_t c = next.fetch_add(1); if (c >= chunks) break; fn(c * grain, std::min(count, (c + 1) * grain)); } }
        catch (...) { std::lock_guard<std::mutex> g(failureLock); if (!failure) failure = std::current_exception(); next.store(chunks); }
    };
    if (nThreads <= 1) { worker(); if (failure) std::rethrow_exception(failure); return; }
    struct JoinAll { std::vector<std::thread> th; ~JoinAll() { for (std::thread& t : th) if (t.joinable()) t.join(); } } pool;
    pool.th.reserve(nThreads);
    try { for (size_t t = 0; t < nThreads; ++t) pool.th.emplace_back(worker); }
    catch (...) { std::lock_guard<std::mutex> g(failureLock); if (!failure) failure = std::current_exception(); next.store(chunks); }
    for (std::thread& t : pool.th) t.join();
    if (failure) std::rethrow_exception(failure);
}
struct StageTimer {          // INCR_SCENE_TIMING=1: wall time of the stages of build_world_bvh on stderr (tools/scene_create_time.py)
    const bool on = test_hook_env("INCR_SCENE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) { if (!on) return; const auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "  build_world_bvh %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; }
};
}
void build_world_bvh(const HostScene& s, HostBvh& out) {
    StageTimer timer;
    // 1. flatten (update-tlas.inl:20-48: one instance per VISIBLE mesh, transform = meshTransform * rootTransform).
    //    Hidden meshes stay in the tree as unhittable (NaN) triangles: incr_update_bvh can then show / hide a mesh by
    //    re-transforming and refitting on the device, without a rebuild; ids count visible meshes only, as DXR's do.
    struct Flat { V3 v0, e1, e2; V3 obj[3]; uint32_t inst, geom, prim, flags, mesh, primGlobal; uint32_t idx[3]; bool hidden; };
    RecordVec<Flat> flat;
    const size_t numMeshes = s.meshTransforms.size();
    out.meshObjBox.assign(numMeshes * 6, 0.f);
    for (size_t m = 0; m < numMeshes; ++m) for (int a = 0; a < 3; ++a) { out.meshObjBox[m * 6 + a] = FLT_MAX; out.meshObjBox[m * 6 + 3 + a] = -FLT_MAX; }
    out.meshLocalToWorld.resize(numMeshes);
    for (size_t m = 0; m < numMeshes; ++m) out.meshLocalToWorld[m] = m44_mul(s.meshTransforms[m], s.rootTransform);
    std::vector<uint32_t> meshFirst(numMeshes + 1, 0u);          // flattened triangles of mesh m: [meshFirst[m], meshFirst[m + 1])
    std::vector<uint32_t> meshInst(numMeshes, 0xffffffffu);
    uint32_t inst = 0;
    for (uint32_t mesh = 0; mesh < (uint32_t)numMeshes; ++mesh) {
        uint32_t ntri = 0;
        for (uint32_t g = 0; g < s.primitiveCounts[mesh]; ++g) ntri += (uint32_t)s.primitives[s.primitiveOffset[mesh] + g].indexCount / 3;
        meshFirst[mesh + 1] = meshFirst[mesh] + ntri;
        if (s.meshVisibility[mesh]) meshInst[mesh] = inst++;
    }
    out.numInstances = inst;
    flat.resize(meshFirst[numMeshes]);
    parallel_for(numMeshes, 1, [&](size_t m0, size_t m1) {
        for (uint32_t mesh = (uint32_t)m0; mesh < (uint32_t)m1; ++mesh) {
            size_t w = meshFirst[mesh];
            const bool hidden = !s.meshVisibility[mesh];
            const M44& l2w = out.meshLocalToWorld[mesh];
            const bool mirrored = m44_det3(l2w) < 0.f;
            for (uint32_t g = 0; g < s.primitiveCounts[mesh]; ++g) {
                const uint32_t pg = s.primitiveOffset[mesh] + g;
                const incr_gpu_primitive& prim = s.primitives[pg];
                const bool opaque = s.materials[(size_t)prim.materialIndex].alphaMode == INCR_ALPHA_OPAQUE;    // scene.cpp:736
                AccessorRef ia, pa; const bool hi = resolve_accessor(s, prim.indexAccessor, &ia), hp = resolve_accessor(s, prim.positionAccessor, &pa);
                const uint32_t ntri = (uint32_t)prim.indexCount / 3;
                for (uint32_t p = 0; p < ntri; ++p) {
                    Flat& f = flat[w++];
                    fetch_uint3(hi ? &ia : nullptr, p * 3, f.idx);
                    V3 v[3];
                    for (int k = 0; k < 3; ++k) {
                        float pos[3]; fetch_floats(hp ? &pa : nullptr, f.idx[k], 3, pos);
                        f.obj[k] = v3(pos[0], pos[1], pos[2]);
                        v[k] = m44_transform_point(f.obj[k], l2w);
                        for (int a = 0; a < 3; ++a) if (pos[a] == pos[a]) {
                            out.meshObjBox[mesh * 6 + a] = std::min(out.meshObjBox[mesh * 6 + a], pos[a]);
                            out.meshObjBox[mesh * 6 + 3 + a] = std::max(out.meshObjBox[mesh * 6 + 3 + a], pos[a]);
                        }
                    }
                    f.v0 = v[0];
                    f.e1 = (mirrored ? v[2] : v[1]) - v[0];
                    f.e2 = (mirrored ? v[1] : v[2]) - v[0];
                    f.inst = hidden ? 0xffffffffu : meshInst[mesh]; f.geom = g; f.prim = p; f.mesh = mesh; f.primGlobal = pg; f.hidden = hidden;
                    f.flags = (opaque ? kTriOpaque : 0u) | (mirrored ? kTriMirrored : 0u);
                }
            }
        }
    });
    timer.lap("flatten");

    // triangle boxes
    const size_t n = flat.size();
    RecordVec<Box> tbv(n); RecordVec<V3> cenv(n);
    Box* const tb = tbv.data(); V3* const cen = cenv.data();
    Box scene; box_reset(scene);
    {
        std::mutex sceneLock;
        parallel_for(n, 65536, [&](size_t i0, size_t i1) {
            Box part; box_reset(part);
            for (size_t i = i0; i < i1; ++i) {
                const Flat& f = flat[i];
                const V3 a = f.v0, b = f.v0 + f.e1, c = f.v0 + f.e2;
                const float xs[3] = {a.x, b.x, c.x}, ys[3] = {a.y, b.y, c.y}, zs[3] = {a.z, b.z, c.z};
                tb[i].lo[0] = std::min(xs[0], std::min(xs[1], xs[2])); tb[i].hi[0] = std::max(xs[0], std::max(xs[1], xs[2]));
                tb[i].lo[1] = std::min(ys[0], std::min(ys[1], ys[2])); tb[i].hi[1] = std::max(ys[0], std::max(ys[1], ys[2]));
                tb[i].lo[2] = std::min(zs[0], std::min(zs[1], zs[2])); tb[i].hi[2] = std::max(zs[0], std::max(zs[1], zs[2]));
                // NaN / inf vertices: give the triangle an empty box so it can never be hit but keeps its slot
                for (int k = 0; k < 3; ++k) if (!(tb[i].lo[k] >= -FLT_MAX && tb[i].hi[k] <= FLT_MAX)) { tb[i].lo[0] = tb[i].lo[1] = tb[i].lo[2] = 0.f; tb[i].hi[0] = tb[i].hi[1] = tb[i].hi[2] = 0.f; }
                cen[i] = v3(0.5f * (tb[i].lo[0] + tb[i].hi[0]), 0.5f * (tb[i].lo[1] + tb[i].hi[1]), 0.5f * (tb[i].lo[2] + tb[i].hi[2]));
                if (!f.hidden) box_grow(part, tb[i]);
            }
            std::lock_guard<std::mutex> g(sceneLock);
            if (part.lo[0] <= part.hi[0]) box_grow(scene, part);          // min / max: the union does not depend on the order of the parts
        });
    }
    { const bool any = n && scene.lo[0] <= scene.hi[0]; for (int a = 0; a < 3; ++a) { out.sceneLo[a] = any ? scene.lo[a] : 0.f; out.sceneHi[a] = any ? scene.hi[a] : 0.f; } }
    timer.lap("triangle boxes");
    // 2. one 8-wide subtree per mesh over its run of the flattened triangles (the reference's BLAS per mesh), built in parallel
    std::vector<WideTree> sub(numMeshes);
    {
        std::vector<uint32_t> todo;
        for (uint32_t m = 0; m < (uint32_t)numMeshes; ++m) if (meshFirst[m + 1] > meshFirst[m]) todo.push_back(m);
        // the subtree builders allocate (Builder::order / nodes: ~1 GB over a Bistro-scale scene): through parallel_for, whose workers hand an exception back to this thread
        parallel_for(todo.size(), 1, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; ++i) { const uint32_t m = todo[i]; build_wide_subtree(tb, cen, meshFirst[m], meshFirst[m + 1] - meshFirst[m], kMaxBvhDepth - 8u, sub[m]); }
        });
    }
    timer.lap("subtrees (parallel)");
    // 3. one node array: the TOP REGION (top tree + a copy of every mesh root, rebuild_top_tree) in front, then the subtrees below
    //    their roots, mesh after mesh; triangle record slots in the same order.
    out.meshRoot.assign(numMeshes, BvhNode8{}); out.meshHasTris.assign(numMeshes, 0); out.meshRootIndex.assign(numMeshes, 0u); out.meshLocalDepth.assign(numMeshes, 1u);
    uint32_t numItems = 0;
    for (size_t m = 0; m < numMeshes; ++m) if (meshFirst[m + 1] > meshFirst[m]) { out.meshHasTris[m] = 1; ++numItems; }
    out.topCapacity = 2u * numItems + 2u;
    std::vector<uint32_t> leafOrder;          // leaf-order position -> flattened triangle id (holes: kTriHole)
    std::vector<uint32_t> blockStart(numMeshes, 0u), slotStart(numMeshes, 0u);
    uint32_t maxLocal = 1;
    {
        size_t nodesEnd = out.topCapacity, slotsEnd = 0;
        for (size_t m = 0; m < numMeshes; ++m) {
            if (!out.meshHasTris[m]) continue;
            blockStart[m] = (uint32_t)nodesEnd; slotStart[m] = (uint32_t)slotsEnd;
            nodesEnd += sub[m].nodes.size() - 1; slotsEnd += sub[m].leafOrder.size();
            out.meshLocalDepth[m] = std::max(sub[m].maxDepth, 1u);
            maxLocal = std::max(maxLocal, out.meshLocalDepth[m]);
        }
        out.nodes8.assign(out.topCapacity, BvhNode8{});
        out.nodes8.resize(nodesEnd);
        leafOrder.resize(slotsEnd);
    }
    parallel_for(numMeshes, 1, [&](size_t m0, size_t m1) {          // every mesh copies its subtree into its own range
        for (size_t m = m0; m < m1; ++m) {
            if (!out.meshHasTris[m]) continue;
            const WideTree& W = sub[m];
            const uint32_t slotOffset = slotStart[m];
            auto fix = [&](BvhNode8 nd) { if (nd.imask) nd.childBase = blockStart[m] + nd.childBase - 1u; nd.triBase += slotOffset; return nd; };      // (a node's children have local indices >= 1)
            out.meshRoot[m] = fix(W.nodes[0]);
            for (size_t i = 1; i < W.nodes.size(); ++i) out.nodes8[blockStart[m] + i - 1] = fix(W.nodes[i]);
            std::copy(W.leafOrder.begin(), W.leafOrder.end(), leafOrder.begin() + slotOffset);
        }
    });
    // refit order, static part: the subtrees below the roots, deepest local level first (children before parents)
    out.refitNodes.clear(); out.refitRanges.clear();
    for (uint32_t d = maxLocal; d >= 2u; --d) {
        const uint32_t first = (uint32_t)out.refitNodes.size();
        for (size_t m = 0; m < numMeshes; ++m) {
            if (!out.meshHasTris[m]) continue;
            const WideTree& W = sub[m];
            for (size_t i = 1; i < W.nodes.size(); ++i) if (W.depth[i] == d) out.refitNodes.push_back(blockStart[m] + (uint32_t)i - 1u);
        }
        if (out.refitNodes.size() > first) { out.refitRanges.push_back(first); out.refitRanges.push_back((uint32_t)out.refitNodes.size() - first); }
    }
    out.refitStaticNodes = (uint32_t)out.refitNodes.size(); out.refitStaticRanges = (uint32_t)out.refitRanges.size() / 2u;
    {
        std::vector<float> meshBoxes(numMeshes * 6, 0.f);
        for (size_t m = 0; m < numMeshes; ++m) for (int a = 0; a < 3; ++a) {
            const bool any = out.meshHasTris[m] && !s.meshVisibility[m] ? false : out.meshHasTris[m] != 0;
            meshBoxes[m * 6 + a] = any ? sub[m].box.lo[a] : 1.f; meshBoxes[m * 6 + 3 + a] = any ? sub[m].box.hi[a] : -1.f;
        }
        rebuild_top_tree(out, meshBoxes.data());
    }

    timer.lap("node array + refit order + top tree");
    // 4. emit triangles + shading records in leaf order
    const size_t slots = leafOrder.size();                 // triangles + holes (device_types.h LEAF SLOTS)
    out.numTriangles = (uint32_t)n;
    out.tris.resize(slots); out.shade.resize(n); out.triObj.resize(slots);
    // RecordVec leaves its elements uninitialised (a zero fill of ~200 MB was a serial stage): every byte that is uploaded must be written below.  The records have no
    // padding -- the named fields are all there is -- and every slot writes all of them, holes included.
    static_assert(sizeof(TriIsect) == 12 * 4 && sizeof(TriObject) == 12 * 4 && sizeof(TriShade) == (9 + 12 + 6 + 5) * 4, "records without padding: all fields are written per slot");
    parallel_for(slots, 32768, [&](size_t s0, size_t s1) {          // every slot writes its own records (shading records: one per flattened id)
    for (size_t i = s0; i < s1; ++i) {
        if (leafOrder[i] == kTriHole) {
            const float qnan = std::numeric_limits<float>::quiet_NaN();
            TriObject& ob = out.triObj[i]; memset(&ob, 0, sizeof ob); ob.mesh = kTriHole; ob.flatId = kTriHole;
            TriIsect& t = out.tris[i];
            t.v0x = t.v0y = t.v0z = t.e1x = t.e1y = t.e1z = t.e2x = t.e2y = t.e2z = qnan;
            t.flatId = kTriHole; t.flags = 0; t.pad = 0;
            continue;
        }
        const Flat& f = flat[leafOrder[i]];
        TriObject& ob = out.triObj[i];
        ob.p0x = f.obj[0].x; ob.p0y = f.obj[0].y; ob.p0z = f.obj[0].z; ob.p1x = f.obj[1].x; ob.p1y = f.obj[1].y; ob.p1z = f.obj[1].z;
        ob.p2x = f.obj[2].x; ob.p2y = f.obj[2].y; ob.p2z = f.obj[2].z; ob.mesh = f.mesh; ob.flags = f.flags & kTriOpaque; ob.flatId = leafOrder[i];
        TriIsect& t = out.tris[i];
        if (f.hidden) {        // unhittable: every comparison of the triangle test fails on NaN
            const float qnan = std::numeric_limits<float>::quiet_NaN();
            t.v0x = t.v0y = t.v0z = t.e1x = t.e1y = t.e1z = t.e2x = t.e2y = t.e2z = qnan;
            t.flatId = leafOrder[i]; t.flags = f.flags; t.pad = 0;
        } else {
        t.v0x = f.v0.x; t.v0y = f.v0.y; t.v0z = f.v0.z;
        t.e1x = f.e1.x; t.e1y = f.e1.y; t.e1z = f.e1.z;
        t.e2x = f.e2.x; t.e2y = f.e2.y; t.e2z = f.e2.z;
        t.flatId = leafOrder[i]; t.flags = f.flags; t.pad = 0;
        }
        TriShade& sh = out.shade[leafOrder[i]];
        const incr_gpu_primitive& prim = s.primitives[f.primGlobal];
        AccessorRef na, ta, ua;
        const bool hn = resolve_accessor(s, prim.normalAccessor, &na), ht = resolve_accessor(s, prim.tangentAccessor, &ta), hu = resolve_accessor(s, prim.uvAccessor, &ua);
        for (int k = 0; k < 3; ++k) {
            fetch_floats(hn ? &na : nullptr, f.idx[k], 3, sh.n[k]);
            fetch_floats(ht ? &ta : nullptr, f.idx[k], 4, sh.t[k]);
            fetch_floats(hu ? &ua : nullptr, f.idx[k], 2, sh.uv[k]);
        }
        sh.materialIndex = (uint32_t)prim.materialIndex;
        sh.meshIndex = (uint32_t)prim.meshIndex;
        sh.instance = f.inst; sh.geometry = f.geom; sh.primitive = f.prim;
    
    }
    });
    timer.lap("triangle + shading records");
}

const char* check_bvh(const HostBvh& bvh) {
    static thread_local char msg[160];
    const size_t n = bvh.tris.size();                       // record slots: triangles + holes
    if (bvh.nodes8.empty()) return "no root node";
    if (bvh.shade.size() != bvh.numTriangles) return "shading records != triangles";
    if (bvh.numTriangles > n) return "more triangles than records";
    if (bvh.nodes8.size() > kMaxBvhNodes) return "more nodes than a traversal stack entry can address (2^23)";
    if (bvh.maxDepth > kMaxBvhDepth) return "tree deeper than the traversal stack (25 levels)";
    if (n > kMaxTriRecords) return "more triangle records than a 32-bit byte offset can address (89.4 M)";
    std::vector<uint8_t> seen(n, 0), seenFlat(n, 0);
    size_t holes = 0;
    for (size_t i = 0; i < n; ++i) {
        if (bvh.tris[i].flatId == kTriHole) { ++holes; if (bvh.tris[i].v0x == bvh.tris[i].v0x) return "hole record is not NaN"; continue; }
        if (bvh.tris[i].flatId >= bvh.numTriangles || seenFlat[bvh.tris[i].flatId]) return "flatId is not a permutation"; seenFlat[bvh.tris[i].flatId] = 1;
    }
    if (n - holes != bvh.numTriangles) return "triangle count does not match the records";
    struct Item { uint32_t node; Box parent; bool hasParent; };
    std::vector<Item> st; st.push_back({0u, {}, false});
    std::vector<uint8_t> visited(bvh.nodes8.size(), 0);
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node >= bvh.nodes8.size()) return "child index out of range";
        if (visited[it.node]) return "node referenced twice";
        visited[it.node] = 1;
        const BvhNode8& nd = bvh.nodes8[it.node];
        const uint8_t ebits[3] = {nd.ex, nd.ey, nd.ez};
        float scale[3];
        for (int a = 0; a < 3; ++a) { uint32_t bits = (uint32_t)ebits[a] << 23; memcpy(&scale[a], &bits, 4); }
        const float org[3] = {nd.ox, nd.oy, nd.oz};
        const uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; const uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
        uint32_t innerSlot = 0;
        if (nd.imask & (nd.imask + 1u)) return "inner children are not in the lowest slots";
        if (nd.sortAxis > 2) return "sort axis out of range";
        for (int k = 0; k < 8; ++k) {
            const uint32_t nib = (nd.leafTris >> (4 * k)) & 15u;
            const bool inner = (nd.imask >> k) & 1u, leaf = !inner && nib != 0;
            if (inner && nib) return "inner child with triangles";
            if (nib & (nib + 1u)) return "leaf triangles are not the low bits of their nibble";
            if (!inner && !leaf) continue;
            Box bx;
            for (int a = 0; a < 3; ++a) { bx.lo[a] = org[a] + (float)qlo[a][k] * scale[a]; bx.hi[a] = org[a] + (float)qhi[a][k] * scale[a]; }
            if (leaf) {
                const uint32_t first = nd.triBase + 4u * (uint32_t)k, cnt = (uint32_t)__builtin_popcount(nib);
                if (cnt == 0 || cnt > kMaxLeafTris) return "leaf count out of range";
                for (uint32_t t = first + cnt; t < first + 4u; ++t) { if (t >= n || bvh.tris[t].flatId != kTriHole) return "leaf slot beyond the count is not a hole"; seen[t] = 1; }
                for (uint32_t t = first; t < first + cnt; ++t) {
                    if (t >= n) return "leaf range out of bounds";
                    if (seen[t]) return "triangle in two leaves";
                    seen[t] = 1;
                    const TriIsect& T = bvh.tris[t];
                    const float vx[3] = {T.v0x, T.v0x + T.e1x, T.v0x + T.e2x}, vy[3] = {T.v0y, T.v0y + T.e1y, T.v0y + T.e2y}, vz[3] = {T.v0z, T.v0z + T.e1z, T.v0z + T.e2z};
                    for (int v = 0; v < 3; ++v)
                        if (vx[v] < bx.lo[0] || vx[v] > bx.hi[0] || vy[v] < bx.lo[1] || vy[v] > bx.hi[1] || vz[v] < bx.lo[2] || vz[v] > bx.hi[2]) {
                            if (vx[v] == vx[v] && vy[v] == vy[v] && vz[v] == vz[v]) { snprintf(msg, sizeof msg, "triangle %u vertex outside its quantised leaf box", t); return msg; }
                        }
                }
            } else {
                st.push_back({nd.childBase + innerSlot, bx, true});
                ++innerSlot;
            }
        }
        // this node's own box (origin .. origin + 255 quanta) must lie inside the quantised box its parent stored for it,
        // up to one quantum of the parent (the parent's box is the conservative one that traversal tests)
        (void)it.parent; (void)it.hasParent;
    }
    for (size_t i = 0; i < n; ++i) if (!seen[i]) return "triangle not referenced by any leaf";
    for (size_t i = bvh.topCapacity; i < visited.size(); ++i) if (!visited[i]) return "unreachable node";          // (the top region keeps spare slots for rebuild_top_tree)
    for (size_t i = 0; i < bvh.topCount && i < visited.size(); ++i) if (!visited[i]) return "unreachable node in the top region";
    return nullptr;
}

void build_light_grid(const HostScene& s, const HostBvh& bvh, LightGrid& g) {
    float ext[3], maxExt = 0.f;
    for (int a = 0; a < 3; ++a) { ext[a] = std::max(bvh.sceneHi[a] - bvh.sceneLo[a], 0.f); maxExt = std::max(maxExt, ext[a]); }
    if (!(maxExt > 0.f) || !(maxExt < FLT_MAX)) maxExt = 1.f;
    // 64 cells along the longest axis (24 until round 3): a cell lists the lights whose range sphere touches it, so the candidates a
    // vertex walks in wf_shadow_gen_kernel exceed its true lights by the cell's diagonal -- with 4 m ranges in a 30 m atrium, 2.0 x at 1.25 m
    // cells and 1.3 x at 0.47 m.  Lists are exact either way (a skipped light contributes exactly 0).
    float cell = maxExt / (float)INCR_LIGHT_GRID_CELLS;
    const float pad = 1e-4f * maxExt;
    for (;;) {       // at most 2^18 cells
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) cells *= std::min(96.0, std::max(1.0, std::ceil((double)ext[a] / cell)));
        if (cells <= 262144.0) break;
        cell *= 1.26f;
    }
    for (int a = 0; a < 3; ++a) {
        g.dim[a] = (uint32_t)std::min(96.f, std::max(1.f, std::ceil(ext[a] / cell)));
        g.lo[a] = bvh.sceneLo[a] - pad;
        const float size = (ext[a] + 2.f * pad) / (float)g.dim[a];
        g.invCell[a] = size > 0.f ? 1.f / size : 0.f;
    }
    const uint32_t numCells = g.dim[0] * g.dim[1] * g.dim[2];
    const int lightCount = (int)s.globalLights.size();                          // path-tracing.inl:198
    // two passes over (light, touched cells): count, prefix, fill -- ascending light order inside every cell, no per-cell allocations
    g.cellOffsets.assign(numCells + 2, 0);
    auto for_each_cell_of_light = [&](int li, auto&& visit) {
        visit(numCells);                                                         // the catch-all cell lists every light
        const int lightId = (size_t)li < s.lightIndices.size() ? s.lightIndices[(size_t)li] : 0;
        if (lightId < 0 || (size_t)lightId >= s.globalLights.size()) return;     // zero light: never contributes
        const incr_light& L = s.globalLights[(size_t)lightId];
        const bool haveT = (size_t)lightId < s.lightTransforms.size();
        float pos[3] = {0.f, 0.f, 0.f};
        if (haveT) { const float* m = s.lightTransforms[(size_t)lightId].m; pos[0] = m[12]; pos[1] = m[13]; pos[2] = m[14]; }
        bool everywhere = false;
        float radius = 0.f;
        if (L.type == INCR_LIGHT_DIRECTIONAL) everywhere = true;
        else if (L.type == INCR_LIGHT_POINT) { if (L.range > 0.f) radius = L.range * 1.001f + pad; else everywhere = true; }
        else if (L.type == INCR_LIGHT_SPOT) { if (L.range > 0.f) return; everywhere = true; }
        else return;                                                             // Sphere/Disk/Rect: radianceIn stays 0
        if (!(radius == radius) || !(pos[0] == pos[0] && pos[1] == pos[1] && pos[2] == pos[2])) everywhere = true;   // NaN data: be conservative
        // only the cells the range sphere's bounding box touches are candidates (the grid is rebuilt by every incr_update_bvh)
        uint32_t c0[3] = {0u, 0u, 0u}, c1[3] = {g.dim[0] - 1u, g.dim[1] - 1u, g.dim[2] - 1u};
        if (!everywhere) for (int a = 0; a < 3; ++a) {
            const float lo = (pos[a] - radius - 2.f * pad - g.lo[a]) * g.invCell[a] - 1.f, hi = (pos[a] + radius + 2.f * pad - g.lo[a]) * g.invCell[a] + 1.f;
            if (!(lo == lo) || !(hi == hi)) continue;
            c0[a] = (uint32_t)std::min(std::max(lo, 0.f), (float)(g.dim[a] - 1u));
            c1[a] = (uint32_t)std::min(std::max(hi, 0.f), (float)(g.dim[a] - 1u));
        }
        for (uint32_t z = c0[2]; z <= c1[2]; ++z) for (uint32_t y = c0[1]; y <= c1[1]; ++y) for (uint32_t x = c0[0]; x <= c1[0]; ++x) {
            bool overlap = everywhere;
            if (!overlap) {
                const uint32_t c[3] = {x, y, z};
                float d2 = 0.f;
                for (int a = 0; a < 3; ++a) {
                    const float size = g.invCell[a] > 0.f ? 1.f / g.invCell[a] : 0.f;
                    const float lo = g.lo[a] + size * (float)c[a] - pad, hi = g.lo[a] + size * (float)(c[a] + 1) + pad;
                    const float d = pos[a] < lo ? lo - pos[a] : (pos[a] > hi ? pos[a] - hi : 0.f);
                    d2 += d * d;
                }
                overlap = d2 <= radius * radius;
            }
            if (overlap) visit((z * g.dim[1] + y) * g.dim[0] + x);
        }
    };
    for (int li = 0; li < lightCount; ++li) for_each_cell_of_light(li, [&](uint32_t c) { g.cellOffsets[c + 1]++; });
    for (uint32_t c = 0; c <= numCells; ++c) g.cellOffsets[c + 1] += g.cellOffsets[c];
    g.cellLights.assign(g.cellOffsets[numCells + 1], 0u);
    std::vector<uint32_t> fill(g.cellOffsets.begin(), g.cellOffsets.end() - 1);
    for (int li = 0; li < lightCount; ++li) for_each_cell_of_light(li, [&](uint32_t c) { g.cellLights[fill[c]++] = (uint32_t)li; });
}

} // namespace incr
