// device_types.h -- HBM-resident layouts of the flattened scene (shared by host builder and HIP kernels).
//
// MI355X-first layout (DESIGN.md "Data layout in HBM"): the reference's accessor -> bufferView -> buffer
// indirection (shaders/common/mesh-material.hlsli:5-191, three dependent loads per attribute) is resolved
// ONCE at scene creation into records sized to whole 128-byte cache lines, so that one closest-hit costs
// one line of geometry per candidate triangle (48 B) and one line of shading inputs (128 B).
#pragma once
#include <stdint.h>

namespace incr {

// Children per node that the builder fills and the traversal tests.  8 is the product; 4 and 6 exist for the A/B measurement of narrower
// trees under the VALU bound: same 80-byte node, the upper slots stay empty.
#ifndef INCR_BVH_WIDTH
#define INCR_BVH_WIDTH 8
#endif
// 8-wide BVH node with child boxes quantised to 8 bits relative to the node's own box (after Ylitie, Karras, Laine:
// "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017), 80 B = 5 x dwordx4 per lane
// for 8 children (the 4-wide float node costs 7 loads for 4 children).  Traversal on MI355X is bound by dependent
// node fetches and vector-L1 line look-ups, not by ALU, so the decode arithmetic is free and the halved step count
// and smaller footprint are what matter.  child box plane = origin + q * 2^(e-127); boxes are conservative.
struct alignas(16) BvhNode8 {
    float ox, oy, oz;            // node box minimum
    uint8_t ex, ey, ez, imask;   // per-axis scale exponent (IEEE biased), bit k of imask: child k is an inner node
    uint32_t childBase;          // inner child in slot k is node childBase + k (see SLOT ORDER)
    uint32_t triBase;            // the triangle of bit i of `leafTris` is record triBase + i (see LEAF SLOTS)
    uint32_t leafTris;           // bit 4 k + j: the leaf child in slot k owns a j-th triangle (count <= 4: the low `count` bits of nibble k);
                                 // nibbles of inner and empty slots are 0
    uint8_t sortAxis, pad[3];    // the axis (0..2) the inner children are sorted along (SLOT ORDER)
    uint8_t qlox[8], qloy[8], qloz[8];
    uint8_t qhix[8], qhiy[8], qhiz[8];
};
// SLOT ORDER (traversal relies on it): the inner children occupy slots 0 .. n-1 (imask == (1 << n) - 1), sorted by box
// centre along the node's longest axis (the "sort axis"); leaf children follow, empty slots last.  A traversal stack
// entry can then name ALL pending inner children of a node in one word: childBase << 9 | reverse << 8 | slot mask --
// the stack never holds more than one entry per tree level, so it fits LDS for any tree (kMaxBvhDepth).
// LEAF SLOTS: the leaf child in slot k owns the four triangle records triBase + 4 k .. triBase + 4 k + 3 (triBase is biased by
// -4 x the number of inner children, modulo 2^32, so that the first leaf's records follow the previous node's), of which the first
// `count` are triangles and the rest are HOLES (NaN records no ray can hit, flatId 0xffffffff; TriObject::mesh 0xffffffff).  The
// traversal's "which triangles of this node does the ray have to test" is then a bit operation on the 8-bit child hit mask --
// every hit bit widened to a nibble, ANDed with leafTris: 8 VALU instructions -- where (count, offset) bytes per child cost a
// divergent loop over the hit leaves (~14 instructions per round, 3-4 rounds per wave: a tenth of the VALU-bound traversal step).
static constexpr uint32_t kMaxBvhNodes = 1u << 23;     // childBase must fit 23 bits of a stack entry
static constexpr uint32_t kMaxBvhDepth = 25;           // deepest tree the traversal kernels carry a stack for (24 entries)
static constexpr uint32_t kMaxTriRecords = 0xffffffffu / 48u;      // the traversal addresses a triangle record by a 32-bit BYTE offset (pt_traverse.h tri_ptr): 89.4 M record slots
static_assert(sizeof(BvhNode8) == 80, "node8 must be 5 x 16 bytes");

#ifndef INCR_MAX_LEAF_TRIS
#define INCR_MAX_LEAF_TRIS 4
#endif
static constexpr uint32_t kMaxLeafTris = INCR_MAX_LEAF_TRIS;
static_assert(INCR_MAX_LEAF_TRIS >= 1 && INCR_MAX_LEAF_TRIS <= 4, "a leaf child owns one nibble of BvhNode8::leafTris");
static constexpr uint32_t kTriHole = 0xffffffffu;      // TriIsect::flatId / TriObject::mesh of a hole record

// Intersection record, leaf order, 48 B (3 x dwordx4).  World-space v0 and edges (Moeller-Trumbore).
struct alignas(16) TriIsect {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; uint32_t flatId; uint32_t flags; uint32_t pad;   // flags: bit0 opaque geometry, bit1 mirrored instance (u,v swapped)
};
static constexpr uint32_t kTriOpaque = 1u, kTriMirrored = 2u;

// Object-space source of a TriIsect record, leaf order, 48 B: what incr_update_bvh re-transforms on the device when a mesh
// transform, the root transform or a visibility flag changes (bvh_update.hip).
struct alignas(16) TriObject {
    float p0x, p0y, p0z, p1x;
    float p1y, p1z, p2x, p2y;
    float p2z; uint32_t mesh; uint32_t flags; uint32_t flatId;      // flags: kTriOpaque only (mirroring is a property of the transform)
};
static_assert(sizeof(TriObject) == 48, "tri source record");

// Shading record, indexed by FLATTENED TRIANGLE ID (TriIsect::flatId -- what a closest-hit query reports), 128 B: object-space vertex
// attributes exactly as the accessor fetchers return them.  (Not leaf order: the leaf-order arrays carry hole records, and a hit
// reads one record at random whatever the order.)
struct alignas(128) TriShade {
    float n[3][3];          // GetFloat3(normalAccessor)   (0 when accessor == -1)
    float t[3][4];          // GetFloat4(tangentAccessor)
    float uv[3][2];         // GetFloat2(uvAccessor)
    uint32_t materialIndex;
    uint32_t meshIndex;     // FGpuPrimitive::m_meshIndex -> localToWorld table
    uint32_t instance;      // DXR InstanceIndex()
    uint32_t geometry;      // DXR GeometryIndex()
    uint32_t primitive;     // DXR PrimitiveIndex()
};
static_assert(sizeof(TriIsect) == 48, "tri record");
static_assert(sizeof(TriShade) == 128, "shade record must be one cache line");

struct DeviceTexture {      // 32 B
    const uint8_t* texels;  // RGBA8, every mip level, TILED: 8 x 4 texel tiles of 128 bytes (see texel_tile_index)
    uint32_t width, height;
    uint32_t format;        // INCR_TEX_*
    uint32_t pad[3];        // [0] every alpha is 255, [1] number of mip levels
};
// Texture storage the samplers read: each level is a grid of 8 x 4 texel TILES, one 128-byte cache line per tile, tiles row-major,
// levels one after the other (a level occupies ceil(w / 8) * ceil(h / 4) whole tiles).  A bilinear footprint (2 x 2 texels) then
// touches 1.4 lines on average instead of the 2 a row-major image costs (its two rows are a whole image row apart) -- the shading
// kernel's incoherent bounce hits were bound by exactly those line fetches (12 texels of 4 bytes per hit, 42 % L2 misses).
#if defined(__HIPCC__)
#define INCR_HD __host__ __device__ inline
#else
#define INCR_HD inline
#endif
INCR_HD uint32_t texel_tiles_x(uint32_t w) { return (w + 7u) >> 3; }
INCR_HD uint32_t texel_level_words(uint32_t w, uint32_t h) { return texel_tiles_x(w) * ((h + 3u) >> 2) * 32u; }      // texels of a level incl. tile padding
INCR_HD uint32_t texel_tile_index(uint32_t x, uint32_t y, uint32_t w) { return ((y >> 2) * texel_tiles_x(w) + (x >> 3)) * 32u + ((y & 3u) << 3) + (x & 7u); }

// Level 0 of a material's base-colour, normal and metallic-roughness textures INTERLEAVED texel by texel (8 bytes: RGBA8 base colour,
// RG8 normal, {B, G} of the metallic-roughness texel), when the three share one size -- glTF materials nearly always do.  The path
// tracer's chsMain samples all three at the same uv with SampleLevel(.., 0) (material/common.hlsli:28-71): one bilinear footprint
// then costs 4 loads of 8 bytes out of 4 x 4-texel tiles of 128 bytes (1.56 cache lines on average) instead of 12 loads out of three
// textures (3 x 1.4 lines), and the tap arithmetic runs once.  Same texel bytes, same decode, same filter arithmetic: results are
// bit-identical to sampling the three textures one by one.  texels == NULL: the material is sampled texture by texture.
struct MaterialTexels {
    const void* texels;       // uint2 per texel, tiles of 4 x 4 (material_texel_index)
    uint32_t width, height;
    uint32_t baseFormat;      // INCR_TEX_* of the base-colour texture (normal = RG_SNORM, metallic-roughness = BGRA8_SWIZZLE_RG by construction)
    uint32_t pad;
    const void* chain;        // the same interleave for the WHOLE mip chain, in the textures' own tiled word order (texel_tile_index + the level offsets of the three
                              // textures, which share one size and level count): entry i = {base word i, normal.rg | metal << 16 | roughness << 24}.  What the raster passes'
                              // anisotropic Sample reads: four 8-byte loads per tap and level instead of twelve 4-byte ones out of three textures.  NULL where not built.
};
INCR_HD uint32_t material_tiles_x(uint32_t w) { return (w + 3u) >> 2; }
INCR_HD uint64_t material_texel_count(uint32_t w, uint32_t h) { return (uint64_t)material_tiles_x(w) * ((h + 3u) >> 2) * 16u; }
INCR_HD uint32_t material_texel_index(uint32_t x, uint32_t y, uint32_t w) { return ((y >> 2) * material_tiles_x(w) + (x >> 2)) * 16u + ((y & 3u) << 2) + (x & 3u); }

} // namespace incr
