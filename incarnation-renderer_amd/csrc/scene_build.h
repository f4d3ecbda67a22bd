// scene_build.h -- host-side stand-in for FScene::CreateAccelerationStructures + UpdateTLASPass
// (reference: source/demo-dll/src/scene.cpp:672-848, src/render-jobs/update-tlas.inl:3-101):
// flattens every visible mesh instance to world space, builds a SAH BVH, collapses it to 8-wide nodes with
// quantised child boxes and bakes one 128-byte shading record per triangle.
#pragma once
#include <memory>
#include <vector>
#include "device_types.h"
#include "host_math.h"

namespace incr {

struct HostScene {
    // deep copies of the caller's arrays (the desc only borrows)
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<incr_mesh_buffer_view> bufferViews;
    std::vector<incr_mesh_accessor> accessors;
    std::vector<uint32_t> primitiveCounts, primitiveOffset;
    std::vector<M44> meshTransforms;
    std::vector<uint32_t> meshVisibility;
    std::vector<incr_gpu_primitive> primitives;
    std::vector<incr_material> materials;
    std::vector<incr_gpu_meshlet> meshlets;
    std::vector<uint32_t> meshletVertexIndices, meshletPackedTriangles;
    std::vector<incr_light> globalLights;
    std::vector<int32_t> lightIndices;
    std::vector<M44> lightTransforms;
    M44 rootTransform;
    float sunDir[3];
    uint32_t maxAnisotropy;          // incr_scene_desc::max_anisotropy, 0 resolved to 16
    // FScene::m_sceneMeshDecals: drawn by the decal G-buffer pass only (not in `primitives`, not in the BVH)
    std::vector<incr_gpu_primitive> decalPrimitives;
    std::vector<M44> decalTransforms;      // per decal primitive
};

// std::vector whose resize() leaves new elements UNINITIALISED: the three big record arrays of a scene (0.8 GB on the Bistro-scale scene) are
// filled slot by slot by parallel workers right after they are sized; value-initialising them first was a serial memset of the lot.
template <class T> struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using RecordVec = std::vector<T, DefaultInitAllocator<T>>;

struct HostBvh {
    std::vector<BvhNode8> nodes8;       // node 0 = root (8-wide, quantised child boxes)
    RecordVec<TriIsect> tris;           // leaf order, four record slots per leaf child: triangles + holes (device_types.h LEAF SLOTS)
    RecordVec<TriShade> shade;          // indexed by flattened triangle id (numTriangles records)
    uint32_t numTriangles = 0;          // records that are triangles
    std::vector<M44> meshLocalToWorld;  // per mesh: meshTransform * rootTransform (shader: mul(localToWorld, sceneRotation))
    uint32_t numInstances = 0;
    uint32_t maxDepth = 0;
    float sceneLo[3], sceneHi[3];
    // ---- inputs of the device-side update (bvh_update.hip: re-transform + refit, UpdateTLASPass stand-in) ----
    RecordVec<TriObject> triObj;        // leaf order: object-space vertices + mesh / flags / flattened id
    // two-level layout (scene_build.cpp): nodes8[0 .. topCapacity) is the TOP REGION -- the top tree over the meshes and one copy of
    // every mesh root, rewritten by rebuild_top_tree -- and the meshes' subtrees below their roots follow and never move
    uint32_t topCapacity = 0, topCount = 0;
    std::vector<BvhNode8> meshRoot;          // per mesh: its root node with global child / triangle indices (the top region holds the live copy)
    std::vector<uint8_t> meshHasTris;
    std::vector<uint32_t> meshRootIndex;     // where the mesh's root copy sits in the top region
    std::vector<uint32_t> meshLocalDepth;    // levels of the mesh's subtree, root included
    // device refit order (bvh_update.hip): node indices, children before parents -- the static part (subtrees below the roots, deepest
    // local level first) and the dynamic tail (mesh roots, then top nodes deepest first); refitRanges = (first, count) pairs in launch order
    std::vector<uint32_t> refitNodes, refitRanges;
    uint32_t refitStaticNodes = 0, refitStaticRanges = 0;
    std::vector<float> meshObjBox;      // 6 floats per mesh: object-space bounds of its positions (lo.xyz, hi.xyz); lo > hi when empty
};

// per-mesh state the update kernels read: DXR instance index (0xffffffff = hidden) and the mirror flag of its transform
struct MeshInstanceInfo { uint32_t instance, mirrored; };
// recompute meshLocalToWorld, the instance numbering and a conservative world box from the host scene's current transforms / visibility
void refresh_mesh_instances(const HostScene& s, HostBvh& bvh, std::vector<MeshInstanceInfo>& info);
// the top tree over the meshes' boxes (6 floats per mesh: lo.xyz, hi.xyz; lo > hi = nothing to hit) into nodes8[0 .. topCapacity)
void rebuild_top_tree(HostBvh& bvh, const float* meshBoxes);

// World-space light grid for the path tracer's next-event estimation.  A cell lists (ascending loop index) every
// light of the chsMain light loop (pathtracing.hlsl:230-236) that can contribute a NON-ZERO radiance somewhere in
// the cell: directional lights and unbounded point/spot lights everywhere, ranged point lights where the range
// sphere (padded) overlaps, ranged spot lights nowhere (Q9: their radial attenuation is always 0).  Skipping a
// light whose GetDirectRadiance is exactly 0 leaves the float sum unchanged, so the grid is an exact acceleration.
// The last cell (index dim[0]*dim[1]*dim[2]) lists ALL lights: the fallback for non-finite path attenuation.
struct LightGrid {
    float lo[3], invCell[3];
    uint32_t dim[3];
    std::vector<uint32_t> cellOffsets;   // numCells + 2 entries
    std::vector<uint32_t> cellLights;    // light loop indices
};
void build_light_grid(const HostScene& s, const HostBvh& bvh, LightGrid& out);

void copy_scene_desc(const incr_scene_desc& d, HostScene& out);
// returns false with a message when the desc is inconsistent (index out of range etc.)
bool validate_scene(const HostScene& s, const char** why);
void build_world_bvh(const HostScene& s, HostBvh& out);
// NULL when every invariant holds, else a description of the first violation
const char* check_bvh(const HostBvh& bvh);

} // namespace incr
