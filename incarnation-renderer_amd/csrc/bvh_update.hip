// bvh_update.hip -- device half of incr_update_bvh, the stand-in for RenderJob::UpdateTLASPass::Execute
// (reference: source/demo-dll/src/render-jobs/update-tlas.inl:3-101, rebuilt there every frame from the visible meshes'
// transform = meshTransform * rootTransform).
//
// The reference's TLAS rebuild is cheap because its BLASes are static.  This library keeps ONE world-space tree
// (one fetch per traversal step, no per-instance ray transform), so a transform / visibility / root-transform change is
// served by two stream-ordered kernels over data that already lives in HBM -- no host rebuild, no stream synchronise:
//   1. retransform: every triangle record is re-derived from its object-space source (TriObject, 48 B) with the mesh's
//      current localToWorld, in the SAME unfused operation order as the host flattening (scene_build.cpp): the records
//      are bit-identical to those of a scene created with the new transforms.  Triangles of hidden meshes become NaN
//      (unhittable), instance ids are renumbered over the visible meshes as DXR's InstanceIndex() would be.
//   2. refit: node boxes bottom-up, one launch per tree level (deepest first), one lane per node: the box of a leaf child
//      from its triangles, of an inner child from the level below, then the node's origin / exponents and the eight
//      quantised child boxes with the builder's conservative rounding.  Topology (child order, triangle order) is kept,
//      so hit results equal those of a fresh build -- the closest hit is order-independent, any-hit is a boolean.
// Both are HBM-streaming: 48 B in + 48 B out (+4 B instance id) per triangle, 80 B in/out + 32 B box per node.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_types.h"
#include "scene_build.h"

namespace incr {

struct NodeBox { float lo[3], pad0, hi[3], pad1; };     // 32 B

__global__ __launch_bounds__(256) void bvh_retransform_kernel(uint32_t n, const TriObject* __restrict__ src, const float* __restrict__ meshLocalToWorld,
                                                              const MeshInstanceInfo* __restrict__ meshInfo, TriIsect* __restrict__ tris, TriShade* __restrict__ shade) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4* sp = (const float4*)(src + i);
    const float4 a = sp[0], b = sp[1], c = sp[2];
    const uint32_t mesh = __float_as_uint(c.y), flags = __float_as_uint(c.z);
    if (mesh == kTriHole) return;                          // hole slot of a leaf (device_types.h LEAF SLOTS): stays the NaN record it was built as
    const MeshInstanceInfo mi = meshInfo[mesh];
    const float* __restrict__ M = meshLocalToWorld + (size_t)mesh * 16;
    float4 o0, o1, o2;
    if (mi.instance == 0xffffffffu) {
        const float qnan = __uint_as_float(0x7fc00000u);
        o0 = make_float4(qnan, qnan, qnan, qnan); o1 = o0; o2 = make_float4(qnan, c.w, __uint_as_float(flags | (mi.mirrored ? kTriMirrored : 0u)), 0.f);
    } else {
        const float px[3] = {a.x, a.w, b.z}, py[3] = {a.y, b.x, b.w}, pz[3] = {a.z, b.y, c.x};
        float vx[3], vy[3], vz[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {       // m44_transform_point, host_math.h: ((x m0 + y m4) + z m8) + m12
            vx[k] = ((px[k] * M[0] + py[k] * M[4]) + pz[k] * M[8]) + M[12];
            vy[k] = ((px[k] * M[1] + py[k] * M[5]) + pz[k] * M[9]) + M[13];
            vz[k] = ((px[k] * M[2] + py[k] * M[6]) + pz[k] * M[10]) + M[14];
        }
        const int k1 = mi.mirrored ? 2 : 1, k2 = mi.mirrored ? 1 : 2;
        o0 = make_float4(vx[0], vy[0], vz[0], vx[k1] - vx[0]);
        o1 = make_float4(vy[k1] - vy[0], vz[k1] - vz[0], vx[k2] - vx[0], vy[k2] - vy[0]);
        o2 = make_float4(vz[k2] - vz[0], c.w, __uint_as_float(flags | (mi.mirrored ? kTriMirrored : 0u)), 0.f);
    }
    float4* dp = (float4*)(tris + i);
    dp[0] = o0; dp[1] = o1; dp[2] = o2;
    shade[__float_as_uint(c.w)].instance = mi.instance;          // shading records are indexed by flattened triangle id
}

// Smallest biased exponent e with lo + 255 * 2^(e - 127) >= hi in float arithmetic (the builder's rule, scene_build.cpp).
__device__ __forceinline__ uint32_t quant_exponent(float lo, float hi, float* scale) {
    const float ext = fmaxf(hi - lo, 0.f);
    uint32_t e = 1u;
    if (ext > 0.f) {
        const uint32_t bits = __float_as_uint(ext / 255.f);
        e = ((bits >> 23) & 0xffu) + ((bits & 0x7fffffu) ? 1u : 0u);
        e = min(max(e, 1u), 254u);
    }
    for (;;) {
        const float sc = __uint_as_float(e << 23);
        if (lo + 255.f * sc >= hi || e >= 254u) { *scale = sc; return e; }
        ++e;
    }
}

__global__ __launch_bounds__(64) void bvh_refit_level_kernel(const uint32_t* __restrict__ levelNodes, uint32_t count, BvhNode8* __restrict__ nodes,
                                                             const TriIsect* __restrict__ tris, NodeBox* __restrict__ boxes) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
    const uint32_t ni = levelNodes[i];
    BvhNode8 nd = nodes[ni];
    float clo[8][3], chi[8][3];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float bl[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bh[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        if ((nd.imask >> k) & 1u) {
            const NodeBox b = boxes[nd.childBase + k];
#pragma unroll
            for (int a = 0; a < 3; ++a) { bl[a] = b.lo[a]; bh[a] = b.hi[a]; }
        } else if ((nd.leafTris >> (4 * k)) & 15u) {
            const uint32_t first = nd.triBase + 4u * (uint32_t)k, cnt = (uint32_t)__popc((nd.leafTris >> (4 * k)) & 15u);
            for (uint32_t t = first; t < first + cnt; ++t) {
                const TriIsect T = tris[t];
                const float vx[3] = {T.v0x, T.v0x + T.e1x, T.v0x + T.e2x}, vy[3] = {T.v0y, T.v0y + T.e1y, T.v0y + T.e2y}, vz[3] = {T.v0z, T.v0z + T.e1z, T.v0z + T.e2z};
                bool finite = true;
#pragma unroll
                for (int v = 0; v < 3; ++v) finite = finite && fabsf(vx[v]) <= 3.0e38f && fabsf(vy[v]) <= 3.0e38f && fabsf(vz[v]) <= 3.0e38f;   // NaN / inf: unhittable, no box
                if (!finite) continue;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    bl[0] = fminf(bl[0], vx[v]); bh[0] = fmaxf(bh[0], vx[v]);
                    bl[1] = fminf(bl[1], vy[v]); bh[1] = fmaxf(bh[1], vy[v]);
                    bl[2] = fminf(bl[2], vz[v]); bh[2] = fmaxf(bh[2], vz[v]);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { clo[k][a] = bl[a]; chi[k][a] = bh[a]; if (bl[a] <= bh[a]) { lo[a] = fminf(lo[a], bl[a]); hi[a] = fmaxf(hi[a], bh[a]); } }
    }
    const bool any = lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2];
    NodeBox nb;
#pragma unroll
    for (int a = 0; a < 3; ++a) { nb.lo[a] = any ? lo[a] : 3.0e38f; nb.hi[a] = any ? hi[a] : -3.0e38f; }
    nb.pad0 = nb.pad1 = 0.f;
    boxes[ni] = nb;
    if (!any) { lo[0] = lo[1] = lo[2] = 0.f; hi[0] = hi[1] = hi[2] = 0.f; }
    nd.ox = lo[0]; nd.oy = lo[1]; nd.oz = lo[2];
    float scale[3];
    nd.ex = (uint8_t)quant_exponent(lo[0], hi[0], &scale[0]);
    nd.ey = (uint8_t)quant_exponent(lo[1], hi[1], &scale[1]);
    nd.ez = (uint8_t)quant_exponent(lo[2], hi[2], &scale[2]);
    uint8_t* qlo[3] = {nd.qlox, nd.qloy, nd.qloz}; uint8_t* qhi[3] = {nd.qhix, nd.qhiy, nd.qhiz};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool valid = clo[k][0] <= chi[k][0] && clo[k][1] <= chi[k][1] && clo[k][2] <= chi[k][2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!valid) { qlo[a][k] = 255; qhi[a][k] = 0; continue; }       // nothing to hit below: inverted box
            int l = (int)floorf((clo[k][a] - lo[a]) / scale[a]), h = (int)ceilf((chi[k][a] - lo[a]) / scale[a]);
            l = min(max(l, 0), 255); h = min(max(h, 0), 255);
            while (l > 0 && lo[a] + (float)l * scale[a] > clo[k][a]) --l;          // conservative under the traversal's float decode
            while (h < 255 && lo[a] + (float)h * scale[a] < chi[k][a]) ++h;
            qlo[a][k] = (uint8_t)l; qhi[a][k] = (uint8_t)h;
        }
    }
    nodes[ni] = nd;
}
} // namespace incr

// Stream-ordered: re-transform every triangle, then refit range by range (scene_build.h refitRanges: children before parents --
// the meshes' subtrees deepest local level first, the mesh roots, the top tree deepest level first).  `ranges` is the host copy.
extern "C" hipError_t incr_launch_bvh_update(uint32_t numTris, const incr::TriObject* src, const float* meshLocalToWorld, const incr::MeshInstanceInfo* meshInfo,
                                             incr::TriIsect* tris, incr::TriShade* shade, incr::BvhNode8* nodes, const uint32_t* dRefitNodes,
                                             const uint32_t* ranges, uint32_t numRanges, void* nodeBoxes, hipStream_t stream) {
    using namespace incr;
    if (numTris) hipLaunchKernelGGL(bvh_retransform_kernel, dim3((numTris + 255u) / 256u), dim3(256), 0, stream, numTris, src, meshLocalToWorld, meshInfo, tris, shade);
    for (uint32_t r = 0; r < numRanges; ++r) {
        const uint32_t first = ranges[2 * r], count = ranges[2 * r + 1];
        if (count) hipLaunchKernelGGL(bvh_refit_level_kernel, dim3((count + 63u) / 64u), dim3(64), 0, stream, dRefitNodes + first, count, nodes, tris, (NodeBox*)nodeBoxes);
    }
    return hipGetLastError();
}
