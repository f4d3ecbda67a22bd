// device_types.h -- HBM-resident layouts of the flattened scene (shared by host builder and HIP kernels).
//
// MI355X-first layout (DESIGN.md "Data layout in HBM"): the reference's accessor -> bufferView -> buffer
// indirection (shaders/common/mesh-material.hlsli:5-191, three dependent loads per attribute) is resolved
// ONCE at scene creation into records sized to whole 128-byte cache lines, so that one closest-hit costs
// one line of geometry per candidate triangle (48 B) and one line of shading inputs (128 B).
#pragma once
#include <stdint.h>

namespace incr {

// 4-wide BVH node, 128 B = one L2 line.  Child boxes in SoA so a lane loads them with 6 dwordx4 + 1 dwordx4.
struct alignas(128) BvhNode4 {
    float lox[4], loy[4], loz[4];
    float hix[4], hiy[4], hiz[4];
    uint32_t child[4];      // inner: node index;  leaf: kLeafBit | (firstTri << 2) | (count-1);  empty: kEmptyChild
    uint32_t pad[4];
};
static constexpr uint32_t kLeafBit = 0x80000000u;
static constexpr uint32_t kEmptyChild = 0x7fffffffu;
static constexpr uint32_t kMaxLeafTris = 4;

// Intersection record, leaf order, 48 B (3 x dwordx4).  World-space v0 and edges (Moeller-Trumbore).
struct alignas(16) TriIsect {
    float v0x, v0y, v0z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; uint32_t flatId; uint32_t flags; uint32_t pad;   // flags: bit0 opaque geometry, bit1 mirrored instance (u,v swapped)
};
static constexpr uint32_t kTriOpaque = 1u, kTriMirrored = 2u;

// Shading record, leaf order, 128 B: object-space vertex attributes exactly as the accessor fetchers return them.
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
static_assert(sizeof(BvhNode4) == 128, "node must be one cache line");
static_assert(sizeof(TriIsect) == 48, "tri record");
static_assert(sizeof(TriShade) == 128, "shade record must be one cache line");

struct DeviceTexture {      // 32 B
    const uint8_t* texels;  // RGBA8, row-major
    uint32_t width, height;
    uint32_t format;        // INCR_TEX_*
    uint32_t pad[3];
};

} // namespace incr
