// raster.hip -- the raster half of the hot path on MI355X (gfx950): batch culling, light culling, the visibility pass as
// a software rasteriser, the G-buffer resolve, the three deferred lighting passes and the TAA resolve.
// Reference: source/demo-dll/shaders/{culling,geo-raster,lighting,postprocess}/ as cited at each kernel; the arithmetic
// (operation order, no FMA contraction) is the same as oracle/oracle_raster.h so results are bit-identical.
//
// MI355X notes (DESIGN.md "Raster path"): every pass is HBM-bound byte/id work, one lane per pixel / meshlet / cluster,
// surfaces are linear arrays so a wave's 64 lanes touch 64 consecutive texels (256 B per 32-bit surface).  The
// visibility pass has no fixed-function rasteriser to lean on: one wave owns one meshlet (<= 64 vertices fit the 64
// lanes: one transform per lane, staged through LDS), triangles are rasterised lane-serially when small and by the
// whole wave in 8x8 pixel tiles when large, and depth test + id write are a single 64-bit atomicMax on depth|order-key.
#include <hip/hip_runtime.h>

#include "pt_shade.h"
#include "ibl_device.h"
#include "raster_params.h"

namespace incr {

// ---------------------------------------------------------------------------------------------- small helpers
DI void mat_mul16(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ r) {      // host_math.h m44_mul order
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = a[i * 4 + 0] * b[j] + a[i * 4 + 1] * b[4 + j];
            s = s + a[i * 4 + 2] * b[8 + j];
            s = s + a[i * 4 + 3] * b[12 + j];
            r[i * 4 + j] = s;
        }
}
DI float dot4p(f3 c, const float* p) { return ((c.x * p[0] + c.y * p[1]) + c.z * p[2]) + 1.f * p[3]; }
DI float len3(const float* p) { return sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]); }
DI uint32_t ftou(float f) { if (!(f > 0.f)) return 0u; if (f >= 4294967296.f) return 0xffffffffu; return (uint32_t)f; }   // D3D ftou
DI uint32_t unorm8(float c) { if (!(c > 0.f)) return 0u; if (c > 1.f) c = 1.f; return (uint32_t)floorf(c * 255.f + 0.5f); }
DI uint32_t pack_unorm8x4(f4 c) { return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24); }
DI f4 unpack_unorm8x4(uint32_t v) { return {(float)(v & 255u) / 255.f, (float)((v >> 8) & 255u) / 255.f, (float)((v >> 16) & 255u) / 255.f, (float)(v >> 24) / 255.f}; }
DI f2 OctEncode(f3 n) {                                                      // geo-raster/encoding.hlsli:37-43
    const float s = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    n = {n.x / s, n.y / s, n.z / s};
    f2 xy = {n.x, n.y};
    if (!(n.z >= 0.f)) xy = {(1.f - fabsf(n.y)) * (n.x >= 0.f ? 1.f : -1.f), (1.f - fabsf(n.x)) * (n.y >= 0.f ? 1.f : -1.f)};
    return {xy.x * 0.5f + 0.5f, xy.y * 0.5f + 0.5f};
}
DI f3 OctDecode(f2 f) {                                                      // :45-53
    f = {f.x * 2.f - 1.f, f.y * 2.f - 1.f};
    f3 n = {f.x, f.y, (1.f - fabsf(f.x)) - fabsf(f.y)};
    const float t = saturate(-n.z);
    n.x += n.x >= 0.f ? -t : t; n.y += n.y >= 0.f ? -t : t;
    return normalize(n);
}
// gbuffer-compute.hlsl:19-46 (rcp = 1/x), split into the per-triangle part and the per-pixel part: the same operations in the same
// order, but a caller that evaluates several pixels of one triangle (the four of a derivative quad, the fragments of a scanned
// triangle) pays the four divisions of the set-up once.
struct BarySetup { float invW0; f2 ndc0; f3 ddx, ddy; float ddxSum, ddySum; };
DI BarySetup bary_setup(f4 p0, f4 p1, f4 p2) {
    const f3 invW = {1.f / p0.w, 1.f / p1.w, 1.f / p2.w};
    const f2 ndc0 = {p0.x * invW.x, p0.y * invW.x}, ndc1 = {p1.x * invW.y, p1.y * invW.y}, ndc2 = {p2.x * invW.z, p2.y * invW.z};
    const f2 a = ndc2 - ndc1, b = ndc0 - ndc1;
    const float invDet = 1.f / (a.x * b.y - a.y * b.x);
    BarySetup s;
    s.invW0 = invW.x; s.ndc0 = ndc0;
    s.ddx = {((ndc1.y - ndc2.y) * invDet) * invW.x, ((ndc2.y - ndc0.y) * invDet) * invW.y, ((ndc0.y - ndc1.y) * invDet) * invW.z};
    s.ddy = {((ndc2.x - ndc1.x) * invDet) * invW.x, ((ndc0.x - ndc2.x) * invDet) * invW.y, ((ndc1.x - ndc0.x) * invDet) * invW.z};
    s.ddxSum = (s.ddx.x * 1.f + s.ddx.y * 1.f) + s.ddx.z * 1.f; s.ddySum = (s.ddy.x * 1.f + s.ddy.y * 1.f) + s.ddy.z * 1.f;
    return s;
}
DI f3 bary_eval(const BarySetup& s, f2 pixelNdc) {
    const f2 dv = pixelNdc - s.ndc0;
    const float interpInvW = (s.invW0 + dv.x * s.ddxSum) + dv.y * s.ddySum;
    const float interpW = 1.f / interpInvW;
    f3 l;
    l.x = interpW * ((s.invW0 + dv.x * s.ddx.x) + dv.y * s.ddy.x);
    l.y = interpW * ((0.0f + dv.x * s.ddx.y) + dv.y * s.ddy.y);
    l.z = interpW * ((0.0f + dv.x * s.ddx.z) + dv.y * s.ddy.z);
    return l;
}
DI f3 CalcBarycentrics(f4 p0, f4 p1, f4 p2, f2 pixelNdc) { return bary_eval(bary_setup(p0, p1, p2), pixelNdc); }
DI f2 pixel_ndc(uint32_t px, uint32_t py, float W, float H) {
    f2 n = {((float)px + 0.5f) / W, ((float)py + 0.5f) / H};
    n.x = 2.f * n.x - 1.f; n.y = -2.f * n.y + 1.f;
    return n;
}
DI f2 interp2(f2 a, f2 b, f2 c, f3 w) { return {(w.x * a.x + w.y * b.x) + w.z * c.x, (w.x * a.y + w.y * b.y) + w.z * c.y}; }
// Screen-space UV derivatives for Sample(): the differences a 2x2 quad of threads would form (ddx_coarse / ddy_coarse: taken at the
// quad's top-left pixel), with all three UVs evaluated on THIS pixel's triangle -- the pixel's own plane extrapolated to its quad
// neighbours, so the value does not depend on what the neighbouring lanes hold (texture unit).
struct UvGrad { f2 ddx, ddy; };
DI UvGrad quad_uv_gradient(const BarySetup& bs, f2 uv0, f2 uv1, f2 uv2, uint32_t x, uint32_t y, float W, float H) {
    const uint32_t x0 = x & ~1u, y0 = y & ~1u;
    const f2 a = interp2(uv0, uv1, uv2, bary_eval(bs, pixel_ndc(x0, y0, W, H)));
    const f2 b = interp2(uv0, uv1, uv2, bary_eval(bs, pixel_ndc(x0 + 1u, y0, W, H)));
    const f2 c = interp2(uv0, uv1, uv2, bary_eval(bs, pixel_ndc(x0, y0 + 1u, W, H)));
    return {b - a, c - a};
}
DI UvGrad quad_uv_gradient(f4 p0, f4 p1, f4 p2, f2 uv0, f2 uv1, f2 uv2, uint32_t x, uint32_t y, float W, float H) {
    return quad_uv_gradient(bary_setup(p0, p1, p2), uv0, uv1, uv2, x, y, W, H);
}
DI f3 interp3(f3 a, f3 b, f3 c, f3 w) { return {(w.x * a.x + w.y * b.x) + w.z * c.x, (w.x * a.y + w.y * b.y) + w.z * c.y, (w.x * a.z + w.y * b.z) + w.z * c.z}; }
DI f3 mul_vec3_m3(f3 v, f3 r0, f3 r1, f3 r2) { return {(v.x * r0.x + v.y * r1.x) + v.z * r2.x, (v.x * r0.y + v.y * r1.y) + v.z * r2.y, (v.x * r0.z + v.y * r1.z) + v.z * r2.z}; }

// ---------------------------------------------------------------------------------------------- per-mesh frame constants
__global__ __launch_bounds__(64) void mesh_frame_kernel(const RasterScene S, const ViewFrame vf, MeshFrame* __restrict__ out) {
    const uint32_t mesh = blockIdx.x * 64 + threadIdx.x;
    if (mesh >= S.numMeshes) return;
    float l2w[16], Mt[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) l2w[i] = S.meshLocalToWorld[(size_t)mesh * 16 + i];
    MeshFrame f;
    mat_mul16(l2w, vf.viewProj, f.localToClip);
    mat_mul16(l2w, vf.cullViewProj, Mt);           // batch-culling.hlsl:27: M = transpose(Mt), M[r] = column r of Mt
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float c0 = Mt[k * 4 + 0], c1 = Mt[k * 4 + 1], c2 = Mt[k * 4 + 2], c3 = Mt[k * 4 + 3];
        f.planes[0][k] = c3 - c2; f.planes[1][k] = c3 + c0; f.planes[2][k] = c3 - c0; f.planes[3][k] = c3 + c1; f.planes[4][k] = c3 - c1;
    }
#pragma unroll
    for (int p = 0; p < 5; ++p) f.planeLen[p] = len3(f.planes[p]);
    f.pad[0] = f.pad[1] = f.pad[2] = 0.f;
    out[mesh] = f;
}

// ---------------------------------------------------------------------------------------------- batch culling
// cs_meshlet_cull_main / cs_primitive_cull_main (batch-culling.hlsl:47-193).  One 1024-lane workgroup walks the meshlets
// (or primitives) in id order and compacts with ballots + an LDS scan, so both lists come out in ascending id without atomics.
__global__ __launch_bounds__(1024) void batch_cull_kernel(const RasterScene S, const MeshFrame* __restrict__ frames, int frustumCulling,
                                                           incr_draw_cmd* __restrict__ defList, incr_draw_cmd* __restrict__ dsList,
                                                           uint32_t* __restrict__ counts, incr_render_stats* __restrict__ stats,
                                                           const PrimDraw* __restrict__ prims, uint32_t numPrims) {
    __shared__ uint32_t s_cnt[2][16];
    __shared__ uint32_t s_base[2];
    __shared__ uint32_t s_culled;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid == 0) { s_base[0] = s_base[1] = 0; s_culled = 0; }
    __syncthreads();
    const uint32_t numItems = prims ? numPrims : S.numMeshlets;
    for (uint32_t start = 0; start < numItems; start += 1024) {
        const uint32_t id = start + tid;
        int cls = 0;                                   // 1 default list, 2 double-sided list, 3 culled
        uint32_t vcount = 0;
        if (id < numItems) {
            int32_t meshIndex, materialIndex; float sphere[4];
            if (prims) {
                const PrimDraw d = prims[id];
                meshIndex = d.meshIndex; materialIndex = d.materialIndex; vcount = d.pad[0];
                sphere[0] = d.sphere[0]; sphere[1] = d.sphere[1]; sphere[2] = d.sphere[2]; sphere[3] = d.sphere[3];
            } else {
                const incr_gpu_meshlet m = S.meshlets[id];
                meshIndex = m.meshIndex; materialIndex = m.materialIndex; vcount = m.triangleCount * 3u;
                sphere[0] = m.boundingSphere[0]; sphere[1] = m.boundingSphere[1]; sphere[2] = m.boundingSphere[2]; sphere[3] = m.boundingSphere[3];
            }
            if (S.meshVisibility[meshIndex] != 0) {
                bool inside = true;
                if (frustumCulling) {
                    const MeshFrame& f = frames[meshIndex];
                    const f3 c = {sphere[0], sphere[1], sphere[2]};
                    const float r = sphere[3];
#pragma unroll
                    for (int p = 0; p < 5; ++p) inside = inside && (dot4p(c, f.planes[p]) + r * f.planeLen[p] >= 0);
                }
                if (inside) cls = S.materials[materialIndex].doubleSided ? 2 : 1;
                else cls = 3;
            }
        }
        const unsigned long long m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (lane == 0) { s_cnt[0][wave] = __popcll(m1); s_cnt[1][wave] = __popcll(m2); if (m3) atomicAdd(&s_culled, (uint32_t)__popcll(m3)); }
        __syncthreads();
        if (cls == 1 || cls == 2) {
            const int l = cls - 1;
            uint32_t off = s_base[l];
            for (uint32_t w = 0; w < wave; ++w) off += s_cnt[l][w];
            off += __popcll((l ? m2 : m1) & lt);
            (l ? dsList : defList)[off] = incr_draw_cmd{id, vcount};
        }
        __syncthreads();
        if (tid == 0) { for (int w = 0; w < 16; ++w) { s_base[0] += s_cnt[0][w]; s_base[1] += s_cnt[1][w]; } }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = s_base[0]; counts[1] = s_base[1]; if (stats) stats->culledObjects = (int32_t)s_culled; }
}

// ---------------------------------------------------------------------------------------------- light culling
DI void plane_from(float* out, f3 a, f3 b, f3 P) { const f3 n = cross(a, b); out[0] = n.x; out[1] = n.y; out[2] = n.z; out[3] = -dot(n, P); }
// light-culling.hlsl:45-131 + cluster-culling.hlsli:44-129; one lane per cluster
__global__ __launch_bounds__(64) void light_cull_kernel(const LightCullParams P) {
    const uint32_t clusterId = blockIdx.x * 64 + threadIdx.x;
    const uint32_t numClusters = P.dim[0] * P.dim[1] * P.dim[2];
    uint32_t count = 0, culled = 0;
    if (clusterId < numClusters) {
        const uint32_t cxI = clusterId % P.dim[0], cyI = (clusterId / P.dim[0]) % P.dim[1], czI = clusterId / (P.dim[0] * P.dim[1]);
        float cx = (float)cxI / (float)P.dim[0], cy = (float)cyI / (float)P.dim[1];
        cx = 2.f * cx - 1.f; cy = 2.f * cy - 1.f; cy = -cy;
        const float sx = 2.f / (float)P.dim[0], sy = 2.f / (float)P.dim[1];
        const f4 np = mul_vec4({0.f, 0.f, P.slice[czI], 1.f}, P.proj), fp = mul_vec4({0.f, 0.f, P.slice[czI + 1], 1.f}, P.proj);
        const float zn = np.z / np.w, zf = fp.z / fp.w;
        const f4 pts[8] = {{cx, cy - sy, zn, 1.f}, {cx + sx, cy - sy, zn, 1.f}, {cx + sx, cy, zn, 1.f}, {cx, cy, zn, 1.f},
                           {cx, cy - sy, zf, 1.f}, {cx + sx, cy - sy, zf, 1.f}, {cx + sx, cy, zf, 1.f}, {cx, cy, zf, 1.f}};
        f3 Q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const f4 w = mul_vec4(pts[i], P.invViewProj); Q[i] = {w.x / w.w, w.y / w.w, w.z / w.w}; }
        float pl[6][4];                                // near, far, left, right, bottom, top (FrustumCull's test order)
        plane_from(pl[0], Q[1] - Q[0], Q[3] - Q[0], Q[0]);
        plane_from(pl[1], Q[7] - Q[4], Q[5] - Q[4], Q[4]);
        plane_from(pl[2], Q[3] - Q[0], Q[4] - Q[0], Q[0]);
        plane_from(pl[3], Q[5] - Q[1], Q[2] - Q[1], Q[1]);
        plane_from(pl[4], Q[4] - Q[0], Q[1] - Q[0], Q[0]);
        plane_from(pl[5], Q[6] - Q[2], Q[3] - Q[2], Q[2]);
        float plen[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) plen[p] = len3(pl[p]);
        uint32_t* __restrict__ list = P.lists + (size_t)clusterId * P.maxLights;
        for (uint32_t i = 0; i < P.numSceneLights && count < P.maxLights; ++i) {
            const uint32_t gi = (uint32_t)P.lightIndices[i];
            incr_light light = {};
            if (gi < P.numGlobalLights) light = P.globalLights[gi];
            if (light.type == INCR_LIGHT_DIRECTIONAL) continue;
            const float range = light.range == 0 ? 4.f : light.range;          // MAX_LIGHT_RANGE
            const float* __restrict__ m = P.lightTransforms + (size_t)i * 16;   // Q10: indexed with the loop index
            const f3 c = {m[12], m[13], m[14]};
            bool inside = true;
#pragma unroll
            for (int p = 0; p < 6; ++p) inside = inside && (dot4p(c, pl[p]) + range * plen[p] >= 0);
            if (inside) list[count++] = gi; else ++culled;
        }
        P.grid[clusterId] = incr_light_grid_data{clusterId * P.maxLights, count};
    }
    // wave reduction of the two counters, one atomic each
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { count += __shfl_down(count, off, 64); culled += __shfl_down(culled, off, 64); }
    if (threadIdx.x == 0) {
        if (count) atomicAdd(P.totalCount, count);
        if (P.stats && culled) atomicAdd((uint32_t*)&P.stats->culledLights, culled);
    }
}

// ---------------------------------------------------------------------------------------------- visibility pass
static constexpr int kVisWaves = 4;
static constexpr int kMaxMeshletVerts = 256;         // the shaders mask local vertex indices with 0xff
static constexpr int kBigBoxPixels = 256;            // larger bounding boxes are rasterised by the whole wave
// ... unless the triangle is alpha-tested: its fragments cost two dependent memory round trips each (surface word, texels), which a
// lane scanning its triangle serially pays one after the other and 64 lanes on an 8 x 8 tile pay once (raster frame 2.49 -> 2.26 ms;
// 0 / 16 / 64 pixels measure the same)
#ifndef INCR_VIS_ALPHA_BOX
#define INCR_VIS_ALPHA_BOX 16
#endif

struct ClipVert { float x, y, z, w; };
struct ScreenVert { int32_t X, Y; float z; };
DI bool to_screen(const ClipVert& c, float W, float H, ScreenVert* o) {
    if (!(c.w > 0.f)) return false;
    const float nx = c.x / c.w, ny = c.y / c.w;
    float sx = (nx * 0.5f + 0.5f) * W, sy = (0.5f - ny * 0.5f) * H;
    if (!(sx == sx) || !(sy == sy)) return false;
    sx = fminf(fmaxf(sx, -1048576.f), 1048576.f); sy = fminf(fmaxf(sy, -1048576.f), 1048576.f);
    o->X = (int32_t)floorf(sx * 256.f + 0.5f); o->Y = (int32_t)floorf(sy * 256.f + 0.5f);
    o->z = c.z / c.w;
    return true;
}
DI ClipVert clip_lerp(const ClipVert& a, const ClipVert& b) {        // a inside, b outside the near plane (z <= w)
    const float da = a.w - a.z, db = b.w - b.z;
    const float t = da / (da - db);
    return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)};
}

// One screen triangle ready to scan: clockwise vertex order, unbiased edge functions at pixel (px0, py0).
struct SubTri {
    int32_t px0, px1, py0, py1;
    long long e[3], sx[3], sy[3];        // E01, E12, E20: value at (px0, py0), step per pixel in x and in y
    int32_t bias[3];                     // 0 for top/left edges, -1 otherwise (D3D fill rule)
    double area;                         // 1 / (twice the signed screen area), in double
    float z0, dz1, dz2;
    uint32_t key, list;                  // visibility id (Encode{Meshlet,Primitive}Visibility) and the draw list that drew it (0 default, 1 double-sided)
    int32_t alphaTex;                    // base-colour texture to alpha-test against, -1 = none
    BarySetup bs;                        // interpolant set-up of the ORIGINAL triangle's clip-space vertices (alpha-tested triangles only)
    DeviceTexture tex;                   // descriptor of alphaTex, fetched once per triangle
    f2 uv0, uv1, uv2;
};
DI bool top_left(int32_t dx, int32_t dy) { return (dy == 0 && dx > 0) || dy < 0; }
// Facing + culling: returns false for degenerate and culled triangles, otherwise leaves the vertices in CLOCKWISE order.
DI bool orient_subtri(ScreenVert& v0, ScreenVert& v1, ScreenVert& v2, bool cullBack) {
    const long long area = (long long)(v1.X - v0.X) * (v2.Y - v0.Y) - (long long)(v2.X - v0.X) * (v1.Y - v0.Y);
    if (area == 0) return false;
    if (area > 0) return !cullBack;                               // clockwise on screen = back face (FrontCounterClockwise)
    const ScreenVert t = v1; v1 = v2; v2 = t;
    return true;
}
// Scan set-up of a clockwise screen triangle; false when its bounding box holds no pixel centre.
// The scan rectangle is columns [x0, x1) x rows [y0, y1): the frame's width and scissor rows, or a screen bin.
DI bool setup_subtri(const ScreenVert& v0, const ScreenVert& v1, const ScreenVert& v2, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, SubTri* st) {
    const long long area = (long long)(v1.X - v0.X) * (v2.Y - v0.Y) - (long long)(v2.X - v0.X) * (v1.Y - v0.Y);
    const int32_t minX = min(v0.X, min(v1.X, v2.X)), maxX = max(v0.X, max(v1.X, v2.X));
    const int32_t minY = min(v0.Y, min(v1.Y, v2.Y)), maxY = max(v0.Y, max(v1.Y, v2.Y));
    st->px0 = max((int32_t)x0, (minX + 127) >> 8); st->px1 = min((int32_t)x1 - 1, (maxX - 128) >> 8);
    st->py0 = max((int32_t)y0, (minY + 127) >> 8); st->py1 = min((int32_t)y1 - 1, (maxY - 128) >> 8);      // scissor rows [y0, y1)
    if (st->px0 > st->px1 || st->py0 > st->py1) return false;
    const long long Sx = (long long)st->px0 * 256 + 128, Sy = (long long)st->py0 * 256 + 128;
    const int32_t dx01 = v1.X - v0.X, dy01 = v1.Y - v0.Y, dx12 = v2.X - v1.X, dy12 = v2.Y - v1.Y, dx20 = v0.X - v2.X, dy20 = v0.Y - v2.Y;
    st->e[0] = (long long)dx01 * (Sy - v0.Y) - (long long)dy01 * (Sx - v0.X);
    st->e[1] = (long long)dx12 * (Sy - v1.Y) - (long long)dy12 * (Sx - v1.X);
    st->e[2] = (long long)dx20 * (Sy - v2.Y) - (long long)dy20 * (Sx - v2.X);
    st->sx[0] = -(long long)dy01 * 256; st->sx[1] = -(long long)dy12 * 256; st->sx[2] = -(long long)dy20 * 256;
    st->sy[0] = (long long)dx01 * 256; st->sy[1] = (long long)dx12 * 256; st->sy[2] = (long long)dx20 * 256;
    st->bias[0] = top_left(dx01, dy01) ? 0 : -1; st->bias[1] = top_left(dx12, dy12) ? 0 : -1; st->bias[2] = top_left(dx20, dy20) ? 0 : -1;
    st->area = 1.0 / (double)area;           // reciprocal once per triangle: the per-fragment weights are products
    st->z0 = v0.z; st->dz1 = v1.z - v0.z; st->dz2 = v2.z - v0.z;
    return true;
}
// Coverage + depth of one fragment -> the surface word it competes with (false: not covered / clipped)
DI bool fragment_word(const SubTri& st, long long E01, long long E12, long long E20, unsigned long long* word) {
    if (E01 + st.bias[0] < 0 || E12 + st.bias[1] < 0 || E20 + st.bias[2] < 0) return false;
    const float l1 = (float)((double)E20 * st.area), l2 = (float)((double)E01 * st.area);
    float depth = (st.z0 + l1 * st.dz1) + l2 * st.dz2;
    if (!(depth >= 0.f)) return false;
    if (depth > 1.f) depth = 1.f;
    *word = ((unsigned long long)__float_as_uint(depth) << 34) | ((unsigned long long)st.list << 33) | ((unsigned long long)st.key + 1ull);
    return true;
}
DI void fragment(const VisParams& P, const SubTri& st, int32_t px, int32_t py, long long E01, long long E12, long long E20) {
    if (E01 + st.bias[0] < 0 || E12 + st.bias[1] < 0 || E20 + st.bias[2] < 0) return;
#ifdef INCR_DIAG_F32_DEPTH       // diagnostic only (not bit-exact): the weights in single precision
    const float l1 = (float)E20 * (float)st.area, l2 = (float)E01 * (float)st.area;
#else
    const float l1 = (float)((double)E20 * st.area), l2 = (float)((double)E01 * st.area);
#endif
    float depth = (st.z0 + l1 * st.dz1) + l2 * st.dz2;
    if (!(depth >= 0.f)) return;
    if (depth > 1.f) depth = 1.f;
    // depth || draw list || id + 1: a depth in [0, 1] is a bit pattern below 2^30 (1.0f = 0x3F800000), the list bit decides equal depths
    // between the two draw lists, and the id keeps all of its 32 bits (EncodePrimitiveVisibility uses every one: 12 + 20)
    const unsigned long long v = ((unsigned long long)__float_as_uint(depth) << 34) | ((unsigned long long)st.list << 33) | ((unsigned long long)st.key + 1ull);
    unsigned long long* dst = P.packed + (size_t)py * P.W + px;
#ifndef INCR_DIAG_NO_ALPHA       // diagnostic only: every triangle scanned as opaque
    if (st.alphaTex >= 0)
#else
    if (false)
#endif
    {                                                              // ps_meshlet_main: clip(opacity - 0.5)
        // values only grow: a fragment that loses now loses for good -- worth a load before the texture sample (fetching the surface
        // word and the texels together was measured: +0.16 ms per frame, most alpha fragments lose here).  Opaque fragments skip
        // the check: the atomic is fire-and-forget, a load would put a memory round trip into every step.
        if (v <= __atomic_load_n(dst, __ATOMIC_RELAXED)) return;
        // per-lane serial loop: its cost is its dependent round trips -- the descriptor came with the triangle, the texels of both
        // trilinear levels are fetched together
        const f3 lam = bary_eval(st.bs, pixel_ndc((uint32_t)px, (uint32_t)py, (float)P.W, (float)P.H));
        const UvGrad g = quad_uv_gradient(st.bs, st.uv0, st.uv1, st.uv2, (uint32_t)px, (uint32_t)py, (float)P.W, (float)P.H);
        if (!AlphaTestGrad(st.tex, interp2(st.uv0, st.uv1, st.uv2, lam), g.ddx, g.ddy, P.S.maxAnisotropy)) return;
    }
#if defined(INCR_DIAG_NO_ATOMIC)     // diagnostic only: the fragment's arithmetic without its atomic
    if (v == 0x123456789abcull) *dst = v;
#elif defined(INCR_DIAG_STORE)       // diagnostic only: a plain store in the atomic's place
    *dst = v;
#else
    atomicMax(dst, v);
#endif
}
DI void raster_serial(const VisParams& P, const SubTri& st) {
    long long r0 = st.e[0], r1 = st.e[1], r2 = st.e[2];
    for (int32_t py = st.py0; py <= st.py1; ++py) {
        long long a = r0, b = r1, c = r2;
        for (int32_t px = st.px0; px <= st.px1; ++px) {
            fragment(P, st, px, py, a, b, c);
            a += st.sx[0]; b += st.sx[1]; c += st.sx[2];
        }
        r0 += st.sy[0]; r1 += st.sy[1]; r2 += st.sy[2];
    }
}
// whole wave, 8x8 pixel tiles; `st` is wave-uniform
DI void raster_coop(const VisParams& P, const SubTri& st, uint32_t lane) {
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    for (int32_t oy = st.py0; oy <= st.py1; oy += 8)
        for (int32_t ox = st.px0; ox <= st.px1; ox += 8) {
            const long long ix = ox - st.px0, iy = oy - st.py0;
            bool reject = false;
            long long base[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                base[k] = st.e[k] + ix * st.sx[k] + iy * st.sy[k];
                const long long hi = base[k] + (st.sx[k] > 0 ? 7 * st.sx[k] : 0) + (st.sy[k] > 0 ? 7 * st.sy[k] : 0) + st.bias[k];
                reject = reject || hi < 0;
            }
#if defined(INCR_DIAG_VIS_COUNT) && INCR_DIAG_VIS_COUNT >= 2       // (1.7 M same-address atomics: distorts the cycle counts of level 1)
            if (lane == 0) atomicAdd(P.vqCounters + (reject ? 17 : 16), 1u);
#endif
            if (reject) continue;
            const int32_t px = ox + lx, py = oy + ly;
            if (px <= st.px1 && py <= st.py1)
                fragment(P, st, px, py, base[0] + lx * st.sx[0] + ly * st.sy[0], base[1] + lx * st.sx[1] + ly * st.sy[1], base[2] + lx * st.sx[2] + ly * st.sy[2]);
        }
}
// The alpha-tested large triangle by the whole wave, its fragments COMPACTED before they cost anything: the tile loop only computes coverage and the surface word
// and appends the covered fragments to a queue of the wave in LDS (A); 64 of them at a time look at the stored word (one round trip per 64 fragments instead of
// one per tile) and the survivors move to a second queue (B); 64 of those at a time run the sampler's tap loop and the atomic.  Scanned tile by tile, a foliage
// triangle ran its taps with the lanes of one tile that were covered AND not yet hidden -- a third of the wave on the 4K Sponza-stat frame, whose alpha-tested
// items were 0.52 of the scan's 0.63 ms.  The queues hold 128 entries (fewer than 64 left + at most 64 new); `st` is wave-uniform.
struct FragQueues { uint32_t* pixA; unsigned long long* wordA; uint32_t* pixB; unsigned long long* wordB; };
static constexpr uint32_t kFragQueue = 128;
DI void alpha_look(const VisParams& P, const FragQueues& q, uint32_t lane, uint32_t first, uint32_t count, uint32_t& bn) {
    uint32_t pix = 0; unsigned long long v = 0ull; bool live = false;
    if (lane < count) {
        pix = q.pixA[first + lane]; v = q.wordA[first + lane];
        live = v > __atomic_load_n(P.packed + (size_t)(pix >> 16) * P.W + (pix & 0xffffu), __ATOMIC_RELAXED);      // values only grow: a fragment that loses now loses for good
    }
    const unsigned long long m = __ballot(live);
#if defined(INCR_DIAG_VIS_COUNT) && INCR_DIAG_VIS_COUNT >= 2
    if (lane == 0) { atomicAdd(P.vqCounters + 34, count); atomicAdd(P.vqCounters + 35, (uint32_t)__popcll(m)); atomicAdd(P.vqCounters + 37, 1u); }
#endif
    if (live) { const uint32_t at = bn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); q.pixB[at] = pix; q.wordB[at] = v; }
    bn += (uint32_t)__popcll(m);
}
DI void alpha_sample(const VisParams& P, const SubTri& st, const FragQueues& q, uint32_t lane, uint32_t first, uint32_t count) {
    if (lane >= count) return;
    const uint32_t pix = q.pixB[first + lane], px = pix & 0xffffu, py = pix >> 16;
    const unsigned long long v = q.wordB[first + lane];
    const f3 lam = bary_eval(st.bs, pixel_ndc(px, py, (float)P.W, (float)P.H));
    const UvGrad g = quad_uv_gradient(st.bs, st.uv0, st.uv1, st.uv2, px, py, (float)P.W, (float)P.H);
    const bool pass = AlphaTestGrad(st.tex, interp2(st.uv0, st.uv1, st.uv2, lam), g.ddx, g.ddy, P.S.maxAnisotropy);
#if defined(INCR_DIAG_VIS_COUNT) && INCR_DIAG_VIS_COUNT >= 2
    { const unsigned long long pm = __ballot(pass); if (lane == 0) { atomicAdd(P.vqCounters + 36, (uint32_t)__popcll(pm)); atomicAdd(P.vqCounters + 38, 1u); } }
#endif
    if (pass) atomicMax(P.packed + (size_t)py * P.W + px, v);
}
DI void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DI void raster_coop_alpha(const VisParams& P, const SubTri& st, uint32_t lane, const FragQueues& q) {
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    uint32_t an = 0, bn = 0;                                       // wave-uniform fill of the two queues
    auto drain = [&](bool all) {
        while (an >= 64u || (all && an != 0u)) {
            const uint32_t n = min(an, 64u);
            an -= n;
            wave_lds_fence();
            alpha_look(P, q, lane, an, n, bn);
            wave_lds_fence();
            if (bn >= 64u) { bn -= 64u; alpha_sample(P, st, q, lane, bn, 64u); }
        }
        if (all && bn != 0u) { wave_lds_fence(); alpha_sample(P, st, q, lane, 0u, bn); bn = 0u; }
    };
    for (int32_t oy = st.py0; oy <= st.py1; oy += 8)
        for (int32_t ox = st.px0; ox <= st.px1; ox += 8) {
            const long long ix = ox - st.px0, iy = oy - st.py0;
            bool reject = false;
            long long base[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                base[k] = st.e[k] + ix * st.sx[k] + iy * st.sy[k];
                const long long hi = base[k] + (st.sx[k] > 0 ? 7 * st.sx[k] : 0) + (st.sy[k] > 0 ? 7 * st.sy[k] : 0) + st.bias[k];
                reject = reject || hi < 0;
            }
            if (reject) continue;
            const int32_t px = ox + lx, py = oy + ly;
            unsigned long long v = 0ull;
            const bool covered = px <= st.px1 && py <= st.py1 &&
                fragment_word(st, base[0] + lx * st.sx[0] + ly * st.sy[0], base[1] + lx * st.sx[1] + ly * st.sy[1], base[2] + lx * st.sx[2] + ly * st.sy[2], &v);
            const unsigned long long m = __ballot(covered);
            if (m == 0ull) continue;
            if (covered) { const uint32_t at = an + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); q.pixA[at] = ((uint32_t)py << 16) | (uint32_t)px; q.wordA[at] = v; }
            an += (uint32_t)__popcll(m);
            drain(false);
        }
    drain(true);
}
// Work distribution (DESIGN.md 5.2 `vis_scan_kernel`): the set-up kernel turns every surviving screen triangle into a 48-byte record and files it
//   * per SCREEN BIN (round 5): opaque large triangles and small ones -- the scan kernel's workgroups first take bins and scan a bin's lists into an LDS copy of it (scan_bin);
//   * as band items  -- <= 32 8x8 tiles of an alpha-tested large triangle (or of any large triangle that found a bin list full), scanned by the whole wave, fragments compacted
//     before they sample (raster_coop_alpha);
//   * as whole items -- large triangles that did not fit the band queue (same scan, all bands by one wave);
//   * as small items -- small triangles that found a bin list full: 64 per wave, one per lane, scanned serially.
// The result does not depend on the order or on which path a triangle took: depth test and id write are one atomicMax on depth | order key, in LDS or on the surface.
DI void vq_load_alpha(const VisParams& P, uint32_t idx, SubTri* st) {
    const SubTriAlpha a = P.alphas[idx];
    st->bs = bary_setup({a.p[0], a.p[1], a.p[2], a.p[3]}, {a.p[4], a.p[5], a.p[6], a.p[7]}, {a.p[8], a.p[9], a.p[10], a.p[11]});
    st->uv0 = {a.uv[0], a.uv[1]}; st->uv1 = {a.uv[2], a.uv[3]}; st->uv2 = {a.uv[4], a.uv[5]};
    st->tex = P.S.textures[st->alphaTex];
    if (P.S.alphaQuads) { const uint32_t* q = P.S.alphaQuads[st->alphaTex]; if (q) { st->tex.texels = (const uint8_t*)q; st->tex.pad[2] = 1u; } }      // the alpha test reads the quad chain
}
DI bool subtri_from_rec(const VisParams& P, uint32_t idx, SubTri* st) {
    const SubTriRec r = P.recs[idx];
    const ScreenVert v0 = {r.X0, r.Y0, r.z0}, v1 = {r.X1, r.Y1, r.z1}, v2 = {r.X2, r.Y2, r.z2};
    if (!setup_subtri(v0, v1, v2, 0u, P.W, P.y0, P.y1, st)) return false;
    st->key = r.key; st->list = r.pad; st->alphaTex = r.alphaTex;
    if (r.alphaTex >= 0) vq_load_alpha(P, idx, st);
    return true;
}

// Can a pixel centre of columns [rx0, rx1] x rows [ry0, ry1] lie inside the clockwise triangle?  false when one edge function (with its fill-rule bias) is negative at
// the rectangle's corner where it is largest -- the scan's tile reject on a rectangle of any size.
DI bool rect_may_touch(const ScreenVert& v0, const ScreenVert& v1, const ScreenVert& v2, int32_t rx0, int32_t rx1, int32_t ry0, int32_t ry1) {
    const long long Sx = (long long)rx0 * 256 + 128, Sy = (long long)ry0 * 256 + 128, w = (long long)(rx1 - rx0) * 256, h = (long long)(ry1 - ry0) * 256;
    const ScreenVert* v[4] = {&v0, &v1, &v2, &v0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t dx = v[k + 1]->X - v[k]->X, dy = v[k + 1]->Y - v[k]->Y;
        const long long e = (long long)dx * (Sy - v[k]->Y) - (long long)dy * (Sx - v[k]->X);
        const long long hi = e + (dy < 0 ? -(long long)dy * w : 0) + (dx > 0 ? (long long)dx * h : 0) + (top_left(dx, dy) ? 0 : -1);
        if (hi < 0) return false;
    }
    return true;
}
// Primitive assembly for one triangle per lane: near-plane clip, viewport snap, facing / culling, 48-byte record, filing by
// bounding-box size.  MUST be called by all 64 lanes (queue slots are allocated with ballots: one atomic per wave and
// queue instead of one per triangle -- 930 K same-address atomics cost 9 ms on the Bistro-scale scene).
DI void emit_triangle(const VisParams& P, uint32_t lane, bool valid, const float4 c0, const float4 c1, const float4 c2,
                      const float2 uvA, const float2 uvB, const float2 uvC, uint32_t key, uint32_t list, int32_t alphaTex, bool cullBack) {
    const float Wf = (float)P.W, Hf = (float)P.H;
    const ClipVert cv[3] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}, {c2.x, c2.y, c2.z, c2.w}};
    ClipVert poly[4]; int np = 0;
    bool in[3]; int nin = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { in[k] = (cv[k].w - cv[k].z) >= 0.f; nin += in[k] ? 1 : 0; }
    if (nin == 3) { poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2]; np = 3; }
    else if (nin > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = (k + 1) % 3;
            if (in[k]) poly[np++] = cv[k];
            if (in[k] != in[j]) poly[np++] = in[k] ? clip_lerp(cv[k], cv[j]) : clip_lerp(cv[j], cv[k]);
        }
    }
    ScreenVert sv[4] = {}; bool ok = valid && np >= 3;
    for (int k = 0; k < np; ++k) ok = ok && to_screen(poly[k], Wf, Hf, &sv[k]);
    for (int s = 0; s < 2; ++s) {                                // a clipped triangle is a fan of two
        ScreenVert v0 = sv[0], v1 = sv[s + 1], v2 = sv[s + 2];
        SubTri st;
        bool have = ok && (s == 0 || np == 4);
        if (have) have = orient_subtri(v0, v1, v2, cullBack);
        if (have) have = setup_subtri(v0, v1, v2, 0u, P.W, P.y0, P.y1, &st);
        const unsigned long long hm = __ballot(have);
        if (hm == 0ull) continue;
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t rbase = 0;
        const int leader = __ffsll((long long)hm) - 1;
        if ((int)lane == leader) rbase = atomicAdd(P.vqCounters + 0, (uint32_t)__popcll(hm));
        rbase = (uint32_t)__shfl((int)rbase, leader, 64);
        const uint32_t idx = rbase + (uint32_t)__popcll(hm & below);
        have = have && idx < P.recCap;                           // cannot fail: recCap = 2 x scene triangles
        const uint32_t bw = have ? (uint32_t)(st.px1 - st.px0 + 1) : 0u, bh = have ? (uint32_t)(st.py1 - st.py0 + 1) : 0u;
        const bool small = have && bw * bh <= (uint32_t)(alphaTex >= 0 ? INCR_VIS_ALPHA_BOX : kBigBoxPixels);
        // Every triangle is filed per SCREEN BIN (scan_bin, the first phase of vis_scan_kernel, scans a bin's triangles into an LDS copy of the bin: a fragment's 64-bit atomic -- 184 cycles of a CU per
        // 8 x 8 tile on the device surface, 0.25 of the scan's 0.63 ms on the 4K Sponza-stat frame -- and the alpha-tested fragment's look at the stored word become LDS
        // operations), in one of three lists per bin: opaque large, alpha-tested large, small.  The (triangle, bin) pairs of the wave's triangles are flattened over the
        // lanes 64 at a time (a triangle at a time waits for every triangle's returning atomic in turn: set-up kernel 0.050 -> 0.128 ms; a lane per triangle waits for the
        // widest one).  A bin is skipped when one edge has it outside; a triangle that finds a list full takes the device-surface queues below instead -- fragments
        // scanned twice lose against themselves, the surface keeps the maximum.
#ifndef INCR_VIS_BIN_KINDS
#define INCR_VIS_BIN_KINDS 5     // bit per list: 1 opaque large, 2 alpha-tested large, 4 small
#endif
        const uint32_t kind = small ? (uint32_t)kBinSmall : (alphaTex >= 0 ? (uint32_t)kBinAlpha : (uint32_t)kBinOpaque);
        bool binned = have && P.binCap != 0u && (((uint32_t)INCR_VIS_BIN_KINDS >> kind) & 1u) != 0u;
        if (P.binCap != 0u && __ballot(binned) != 0ull) {
            const uint32_t sh = P.binShift;
            const uint32_t cx0 = binned ? (uint32_t)st.px0 >> sh : 0u, cy0 = binned ? ((uint32_t)st.py0 - P.y0) >> sh : 0u;
            const uint32_t ncx = binned ? ((uint32_t)st.px1 >> sh) - cx0 + 1u : 0u, nb = binned ? ncx * ((((uint32_t)st.py1 - P.y0) >> sh) - cy0 + 1u) : 0u;
            uint32_t incl = nb;                                   // inclusive prefix sum of the pair counts over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64); if ((int)lane >= off) incl += t; }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            bool full = false;                                    // per SOURCE lane: one of its lists was full
            for (uint32_t p0 = 0; p0 < total; p0 += 64u) {
                const uint32_t pair = p0 + lane;
                uint32_t src = 0;                                 // first lane whose inclusive sum exceeds `pair`
#pragma unroll
                for (uint32_t half = 32u; half != 0u; half >>= 1) { const uint32_t t = (uint32_t)__shfl((int)incl, (int)(src + half - 1u), 64); if (t <= pair) src += half; }
                src = min(src, 63u);
                ScreenVert a, b, c;
                a.X = __shfl(v0.X, (int)src, 64); a.Y = __shfl(v0.Y, (int)src, 64); b.X = __shfl(v1.X, (int)src, 64); b.Y = __shfl(v1.Y, (int)src, 64);
                c.X = __shfl(v2.X, (int)src, 64); c.Y = __shfl(v2.Y, (int)src, 64);
                const int32_t bx0p = __shfl(st.px0, (int)src, 64), bx1p = __shfl(st.px1, (int)src, 64), by0p = __shfl(st.py0, (int)src, 64), by1p = __shfl(st.py1, (int)src, 64);
                const uint32_t sidx = (uint32_t)__shfl((int)idx, (int)src, 64), sncx = (uint32_t)__shfl((int)ncx, (int)src, 64), skind = (uint32_t)__shfl((int)kind, (int)src, 64);
                const uint32_t j = pair - ((uint32_t)__shfl((int)incl, (int)src, 64) - (uint32_t)__shfl((int)nb, (int)src, 64));
                bool mine = false;
                if (pair < total) {
                    const uint32_t cx = ((uint32_t)bx0p >> sh) + j % sncx, cy = (((uint32_t)by0p - P.y0) >> sh) + j / sncx;
                    const int32_t rx0 = max(bx0p, (int32_t)(cx << sh)), rx1 = min(bx1p, (int32_t)(((cx + 1u) << sh) - 1u));
                    const int32_t ry0 = max(by0p, (int32_t)(P.y0 + (cy << sh))), ry1 = min(by1p, (int32_t)(P.y0 + ((cy + 1u) << sh) - 1u));
                    if (rect_may_touch(a, b, c, rx0, rx1, ry0, ry1)) {
                        const uint32_t list = (cy * P.binsX + cx) * (uint32_t)kVisBinLists + skind, slot = atomicAdd(P.binCount + list, 1u);
                        if (slot < P.binCap) P.binList[(size_t)list * P.binCap + slot] = sidx; else mine = true;
                    }
                }
                for (unsigned long long m = __ballot(mine); m != 0ull; m &= m - 1ull)          // rare: tell the triangle's own lane
                    if (lane == (uint32_t)__shfl((int)src, __ffsll((long long)m) - 1, 64)) full = true;
            }
            if (full) binned = false;
        }
        const unsigned long long sm = __ballot(small && !binned);
        if (sm != 0ull) {
            uint32_t sbase = 0;
            const int sleader = __ffsll((long long)sm) - 1;
            if ((int)lane == sleader) sbase = atomicAdd(P.vqCounters + 2, (uint32_t)__popcll(sm));
            sbase = (uint32_t)__shfl((int)sbase, sleader, 64);
            if (small && !binned) P.smallQueue[sbase + (uint32_t)__popcll(sm & below)] = idx;
        }
        if (!have) continue;
        SubTriRec r;
        r.X0 = v0.X; r.Y0 = v0.Y; r.X1 = v1.X; r.Y1 = v1.Y; r.X2 = v2.X; r.Y2 = v2.Y; r.z0 = v0.z; r.z1 = v1.z; r.z2 = v2.z;
        r.key = key; r.alphaTex = alphaTex; r.pad = list;
        P.recs[idx] = r;
        if (alphaTex >= 0) {
            SubTriAlpha al;
            al.p[0] = c0.x; al.p[1] = c0.y; al.p[2] = c0.z; al.p[3] = c0.w; al.p[4] = c1.x; al.p[5] = c1.y; al.p[6] = c1.z; al.p[7] = c1.w;
            al.p[8] = c2.x; al.p[9] = c2.y; al.p[10] = c2.z; al.p[11] = c2.w;
            al.uv[0] = uvA.x; al.uv[1] = uvA.y; al.uv[2] = uvB.x; al.uv[3] = uvB.y; al.uv[4] = uvC.x; al.uv[5] = uvC.y; al.pad[0] = al.pad[1] = 0.f;
            P.alphas[idx] = al;
        }
        if (small || binned) continue;
        const uint32_t tw = (bw + 7u) >> 3, th = (bh + 7u) >> 3;
        // tiles per band item: 64 for an opaque triangle, 32 for an alpha-tested one -- its fragments run the sampler's serial tap loop, a 64-tile item of foliage kept ONE
        // wave busy for most of the launch while the others had finished (PMC: the average wave lived a quarter of the kernel's duration).  Measured, raster frame at
        // 64 / 32 / 24 / 16 / 12 / 8 / 4 tiles: 2.68 / 2.54 / 2.54 / 2.56 / 2.60 / 2.64 / 2.70 ms; the opaque items at 32 or 16: no change (profiles/r04_ab_raster_waves.txt)
#ifndef INCR_VIS_ALPHA_ITEM_TILES
#define INCR_VIS_ALPHA_ITEM_TILES 32
#endif
#ifndef INCR_VIS_ITEM_TILES
#define INCR_VIS_ITEM_TILES 64
#endif
        const uint32_t tileBudget = alphaTex >= 0 ? (uint32_t)INCR_VIS_ALPHA_ITEM_TILES : (uint32_t)INCR_VIS_ITEM_TILES;
        const uint32_t rowsPerItem = max(1u, tileBudget / tw), nItems = (th + rowsPerItem - 1u) / rowsPerItem;
        const uint32_t base = atomicAdd(P.vqCounters + 1, nItems);
        if (base + nItems <= P.bigCap) {
            for (uint32_t i = 0; i < nItems; ++i) P.bigQueue[base + i] = make_uint2(idx, ((i * rowsPerItem) << 16) | min(rowsPerItem, th - i * rowsPerItem));
        } else {                                                 // band queue full: neutralise the reserved slots, scan the triangle whole
            for (uint32_t i = base; i < min(base + nItems, P.bigCap); ++i) P.bigQueue[i] = make_uint2(idx, 0u);
            P.wholeQueue[atomicAdd(P.vqCounters + 3, 1u)] = idx;
        }
    }
}

// vs_meshlet_main + primitive assembly (visibility-pass.hlsl:57-97): one wave per draw command
__global__ __launch_bounds__(kVisWaves * 64) void vis_setup_kernel(const VisParams P) {
    __shared__ float4 s_clip[kVisWaves][kMaxMeshletVerts];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n0 = P.counts[0], n1 = P.counts[1];
    for (uint32_t job = blockIdx.x * kVisWaves + wave; job < n0 + n1; job += gridDim.x * kVisWaves) {
        const uint32_t list = job >= n0 ? 1u : 0u;
        const incr_draw_cmd cmd = P.lists[list][list ? job - n0 : job];
        const incr_gpu_meshlet m = P.S.meshlets[cmd.id];
        const float* __restrict__ l2w = P.S.meshLocalToWorld + (size_t)m.meshIndex * 16;
        const uint32_t nv = min(m.vertexCount, (uint32_t)kMaxMeshletVerts);
        for (uint32_t v = lane; v < nv; v += 64) {                 // vs_meshlet_main:86-88
            const float4 pos = P.S.mvPos[m.vertexBegin + v];
            const f4 worldPos = mul_vec4({pos.x, pos.y, pos.z, 1.f}, l2w);
            const f4 c = mul_vec4(worldPos, P.viewProj);
            s_clip[wave][v] = make_float4(c.x, c.y, c.z, c.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const incr_material& mat = P.S.materials[m.materialIndex];
        int32_t alphaTex = mat.baseColorTextureIndex;
        if (alphaTex >= 0 && P.S.textures[alphaTex].pad[0]) alphaTex = -1;      // alpha is 255 everywhere: the clip can never fire
        const uint32_t ntri = cmd.vertex_count / 3u;
        for (uint32_t t0 = 0; t0 < ntri; t0 += 64) {             // wave-uniform trip count
            const uint32_t tri = t0 + lane;
            const bool valid = tri < ntri;
            const uint32_t packed = valid ? P.S.packedTris[m.triangleBegin + tri] : 0u;
            uint32_t li[3];
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) li[k] = min(0xffu & (packed >> ((2u - k) * 10u)), nv - 1u);
            const float4 c0 = s_clip[wave][li[0]], c1 = s_clip[wave][li[1]], c2 = s_clip[wave][li[2]];
            float2 uvA = make_float2(0.f, 0.f), uvB = uvA, uvC = uvA;
            if (alphaTex >= 0) { uvA = P.S.mvUv[m.vertexBegin + li[0]]; uvB = P.S.mvUv[m.vertexBegin + li[1]]; uvC = P.S.mvUv[m.vertexBegin + li[2]]; }
            emit_triangle(P, lane, valid, c0, c1, c2, uvA, uvB, uvC, (cmd.id << 7) | tri, list, alphaTex, list == 0);
        }
        __builtin_amdgcn_wave_barrier();                         // s_clip[wave] is rewritten by the next job
    }
}

// vs_primitive_main + primitive assembly (visibility-pass.hlsl:32-55; FConfig::UseMeshlets == false): the draw commands
// name whole primitives (up to 2^20 triangles), so the work is cut into static 64-triangle chunks, one per wave; a
// chunk looks up whether its primitive survived culling (primFlags: 0 = not drawn, 1 = default list, 2 = double-sided).
__global__ __launch_bounds__(256) void vis_setup_prim_kernel(const VisParams P) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const RasterPrims& R = P.prims;
    for (uint32_t chunk = blockIdx.x * 4 + wave; chunk < R.numChunks; chunk += gridDim.x * 4) {
        const uint32_t prim = R.chunkPrim[chunk];
        const uint32_t flag = R.primFlags[prim];
        if (flag == 0u) continue;                                 // wave-uniform
        const uint32_t list = flag - 1u;
        const PrimDraw d = R.draws[prim];
        const float* __restrict__ l2w = P.S.meshLocalToWorld + (size_t)d.meshIndex * 16;
        int32_t alphaTex = P.S.materials[d.materialIndex].baseColorTextureIndex;
        if (alphaTex >= 0 && P.S.textures[alphaTex].pad[0]) alphaTex = -1;
        const uint32_t tri = R.chunkFirstTri[chunk] + lane;
        const bool valid = tri < d.indexCount / 3u;
        float4 c[3]; float2 uv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                             // vs_primitive_main:44-50
            const uint32_t vi = valid ? R.indices[d.indexBegin + tri * 3u + (uint32_t)k] : 0u;
            const float4 pos = R.pvPos[d.vertexBase + vi];
            const f4 worldPos = mul_vec4({pos.x, pos.y, pos.z, 1.f}, l2w);
            const f4 cc = mul_vec4(worldPos, P.viewProj);
            c[k] = make_float4(cc.x, cc.y, cc.z, cc.w);
            uv[k] = alphaTex >= 0 ? R.pvUv[d.vertexBase + vi] : make_float2(0.f, 0.f);
        }
        // the forward pass draws primitive by primitive in scene order: no list bit in the depth-tie order
        emit_triangle(P, lane, valid, c[0], c[1], c[2], uv[0], uv[1], uv[2], (prim << 20) | tri, P.forward ? 0u : list, alphaTex, list == 0);
    }
}
// forward-pass.inl:68-179: every primitive of every visible mesh is drawn; CullMode NONE for double-sided materials
__global__ __launch_bounds__(256) void forward_flags_kernel(const RasterScene S, const PrimDraw* __restrict__ draws, uint32_t numPrims, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numPrims) return;
    const PrimDraw d = draws[i];
    flags[i] = S.meshVisibility[d.meshIndex] == 0u ? 0u : (S.materials[d.materialIndex].doubleSided ? 2u : 1u);
}
// draw lists -> per-primitive flags for the chunked set-up above
__global__ __launch_bounds__(256) void prim_flags_kernel(const incr_draw_cmd* __restrict__ defList, const incr_draw_cmd* __restrict__ dsList,
                                                         const uint32_t* __restrict__ counts, uint32_t* __restrict__ flags) {
    const uint32_t n0 = counts[0], n1 = counts[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += gridDim.x * blockDim.x)
        flags[i < n0 ? defList[i].id : dsList[i - n0].id] = i < n0 ? 1u : 2u;
}

// rasteriser + ps_meshlet_main (visibility-pass.hlsl:99-118): persistent waves over the three queues
// A wave's walk over one queue.  Its first `step` entries are static (entry = wave index x step), the cursor hands out what
// lies behind the grid's first round, `grab` steps per atomic: the three cursors share a cache line, and 8192 waves asking
// each of them at least once -- plus one atomic per band item -- were ~80 k serialised same-line atomics per frame, most
// of the kernel's time.  A wave whose static entry is already past the end leaves the queue without touching the cursor.
struct QueueWalk { uint32_t cur, end; bool done; };
DI void qw_init(QueueWalk& q, uint32_t n, uint32_t step) {
    q.cur = __builtin_amdgcn_readfirstlane((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * step);
    q.end = q.cur + step; q.done = q.cur >= n;
}
DI bool qw_next(QueueWalk& q, uint32_t* __restrict__ cursor, uint32_t n, uint32_t step, uint32_t grab, uint32_t* out) {
    if (q.done) return false;
    if (q.cur >= q.end) {
        uint32_t b = 0;
        if ((threadIdx.x & 63u) == 0) b = atomicAdd(cursor, step * grab);
        b = __builtin_amdgcn_readfirstlane(b) + gridDim.x * (blockDim.x >> 6) * step;
        q.cur = b; q.end = b + step * grab;
    }
    if (q.cur >= n) { q.done = true; return false; }
    *out = q.cur; q.cur += step;
    return true;
}
// The visibility scan, bin by bin (first phase of vis_scan_kernel): a workgroup takes a screen bin (2^SHIFT pixels square, origin column 0 / row y0) and scans the
// triangles filed for it into an LDS copy of the bin's surface words -- the same edge functions, depth and word as on the device surface, atomicMax on LDS --, then
// merges the copy into the surface with one atomicMax per covered pixel.  Large triangles: a wave each, 8 x 8 tiles aligned to the bin.  Small
// ones: a lane each.  The alpha-tested large triangles come last, behind a barrier: their fragments look at the stored word before they sample (ps_meshlet_main's
// clip), and by then the bin's opaque geometry is in place.  The wave-level queues of vis_scan_kernel follow: the alpha-tested large triangles (not binned by default: foliage piles up in few bins) and whatever found its list full.  LDS holds the bin tile by tile (a tile = 512
// contiguous bytes: the 64 atomics of a wave fall into different bank pairs).
template <uint32_t SHIFT> struct BinTile {
    static constexpr uint32_t kBin = 1u << SHIFT, kTilesX = kBin / 8u, kWords = kBin * kBin;
    static DI uint32_t word_index(uint32_t x, uint32_t y) { return ((y >> 3) * kTilesX + (x >> 3)) * 64u + (y & 7u) * 8u + (x & 7u); }
};
// One fragment against the LDS copy (fragment() on the device surface)
DI void fragment_lds(const VisParams& P, const SubTri& st, int32_t px, int32_t py, long long E01, long long E12, long long E20, unsigned long long* __restrict__ w) {
    unsigned long long v;
    if (!fragment_word(st, E01, E12, E20, &v)) return;
    if (st.alphaTex >= 0) {
        if (v <= *(volatile unsigned long long*)w) return;
        const f3 lam = bary_eval(st.bs, pixel_ndc((uint32_t)px, (uint32_t)py, (float)P.W, (float)P.H));
        const UvGrad g = quad_uv_gradient(st.bs, st.uv0, st.uv1, st.uv2, (uint32_t)px, (uint32_t)py, (float)P.W, (float)P.H);
        if (!AlphaTestGrad(st.tex, interp2(st.uv0, st.uv1, st.uv2, lam), g.ddx, g.ddy, P.S.maxAnisotropy)) return;
    }
    atomicMax(w, v);
}
// Record -> scan set-up inside the rectangle [X0, X1) x [Y0, Y1)
DI bool subtri_in_rect(const VisParams& P, uint32_t idx, uint32_t X0, uint32_t X1, uint32_t Y0, uint32_t Y1, SubTri* st) {
    const SubTriRec r = P.recs[idx];
    const ScreenVert v0 = {r.X0, r.Y0, r.z0}, v1 = {r.X1, r.Y1, r.z1}, v2 = {r.X2, r.Y2, r.z2};
    if (!setup_subtri(v0, v1, v2, X0, X1, Y0, Y1, st)) return false;
    st->key = r.key; st->list = r.pad; st->alphaTex = r.alphaTex;
    if (r.alphaTex >= 0) vq_load_alpha(P, idx, st);
    return true;
}
template <uint32_t SHIFT>
DI void scan_bin(const VisParams& P, uint32_t bin, unsigned long long* __restrict__ s_words, uint32_t* __restrict__ s_next) {
    using T = BinTile<SHIFT>;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cx = bin % P.binsX, cy = bin / P.binsX;
    const uint32_t X0 = cx << SHIFT, Y0 = P.y0 + (cy << SHIFT), X1 = min(X0 + T::kBin, P.W), Y1 = min(Y0 + T::kBin, P.y1);
    for (uint32_t i = threadIdx.x; i < T::kWords; i += 256u) s_words[i] = 0ull;
    if (threadIdx.x < (uint32_t)kVisBinLists) s_next[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t count[kVisBinLists];
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)kVisBinLists; ++k) count[k] = min(P.binCount[bin * (uint32_t)kVisBinLists + k], P.binCap);
    const int32_t lx = (int32_t)(lane & 7u), ly = (int32_t)(lane >> 3);
    // a large triangle by the whole wave
    auto scan_large = [&](uint32_t list) {
        const uint32_t* __restrict__ entries = P.binList + (size_t)(bin * (uint32_t)kVisBinLists + list) * P.binCap;
        for (;;) {
            uint32_t e = 0;
            if (lane == 0) e = atomicAdd(&s_next[list], 1u);
            e = __builtin_amdgcn_readfirstlane(e);
            if (e >= count[list]) break;
            SubTri st;
            if (!subtri_in_rect(P, __builtin_amdgcn_readfirstlane(entries[e]), X0, X1, Y0, Y1, &st)) continue;
            const int32_t tx0 = (st.px0 - (int32_t)X0) >> 3, tx1 = (st.px1 - (int32_t)X0) >> 3, ty0 = (st.py0 - (int32_t)Y0) >> 3, ty1 = (st.py1 - (int32_t)Y0) >> 3;
            for (int32_t ty = ty0; ty <= ty1; ++ty)
                for (int32_t tx = tx0; tx <= tx1; ++tx) {
                    const long long ix = (int32_t)X0 + tx * 8 - st.px0, iy = (int32_t)Y0 + ty * 8 - st.py0;       // >= -7: the tile may begin left of / above the rectangle
                    bool reject = false;
                    long long base[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        base[k] = st.e[k] + ix * st.sx[k] + iy * st.sy[k];
                        const long long hi = base[k] + (st.sx[k] > 0 ? 7 * st.sx[k] : 0) + (st.sy[k] > 0 ? 7 * st.sy[k] : 0) + st.bias[k];
                        reject = reject || hi < 0;
                    }
                    if (reject) continue;
                    const int32_t px = (int32_t)X0 + tx * 8 + lx, py = (int32_t)Y0 + ty * 8 + ly;
                    if (px >= st.px0 && px <= st.px1 && py >= st.py0 && py <= st.py1)
                        fragment_lds(P, st, px, py, base[0] + lx * st.sx[0] + ly * st.sy[0], base[1] + lx * st.sx[1] + ly * st.sy[1], base[2] + lx * st.sx[2] + ly * st.sy[2],
                                     &s_words[((uint32_t)ty * T::kTilesX + (uint32_t)tx) * 64u + lane]);
                }
        }
    };
    scan_large(kBinOpaque);
    for (;;) {                                                   // small triangles: one per lane, its part of the bin pixel by pixel
        uint32_t e = 0;
        if (lane == 0) e = atomicAdd(&s_next[kBinSmall], 64u);
        e = __builtin_amdgcn_readfirstlane(e);
        if (e >= count[kBinSmall]) break;
        SubTri st;
        if (e + lane < count[kBinSmall] && subtri_in_rect(P, P.binList[(size_t)(bin * (uint32_t)kVisBinLists + (uint32_t)kBinSmall) * P.binCap + e + lane], X0, X1, Y0, Y1, &st)) {
            long long r0 = st.e[0], r1 = st.e[1], r2 = st.e[2];
            for (int32_t py = st.py0; py <= st.py1; ++py) {
                long long a = r0, b = r1, c = r2;
                for (int32_t px = st.px0; px <= st.px1; ++px) {
                    fragment_lds(P, st, px, py, a, b, c, &s_words[T::word_index((uint32_t)px - X0, (uint32_t)py - Y0)]);
                    a += st.sx[0]; b += st.sx[1]; c += st.sx[2];
                }
                r0 += st.sy[0]; r1 += st.sy[1]; r2 += st.sy[2];
            }
        }
    }
    __syncthreads();
    scan_large(kBinAlpha);
    __syncthreads();
    // into the device surface: other workgroups' alpha-tested and overflowed triangles write there at the same time
    for (uint32_t i = threadIdx.x; i < T::kWords; i += 256u) {
        const uint32_t x = i & (T::kBin - 1u), y = i >> SHIFT;
        const unsigned long long w = s_words[T::word_index(x, y)];
        if (w != 0ull && X0 + x < X1 && Y0 + y < Y1) atomicMax(P.packed + (size_t)(Y0 + y) * P.W + X0 + x, w);
    }
    __syncthreads();                                              // the LDS copy is reused (next bin, the fragment queues)
}
// The band queue's walk: static first entry like the others, then STRIPED cursors -- one cursor for the whole queue was the scan's largest single cost (72 K band items
// on the 4K Sponza-stat frame: an atomic per item 1.05 ms for the pass, per two items 0.69, per eight 0.54 with the alpha test compiled out, ~11 ns per same-address
// atomic; profiles/r05_ab_vis_cursor.txt), and large grabs leave the foliage's expensive items to a few waves.  A wave draws from its home segment (workgroup index
// modulo the stripe count) and moves on to the next when that one is used up; a segment found empty is never asked again.
struct BandWalk { uint32_t cur, end, stripe, tried, dynBase, segLen; bool done; };
DI void bw_init(BandWalk& q, uint32_t n) {
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    q.cur = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    q.end = q.cur + 1u; q.done = q.cur >= n;
    q.stripe = blockIdx.x % (uint32_t)INCR_VIS_STRIPES; q.tried = 0u;
    q.dynBase = min(waves, n);
    q.segLen = (n - q.dynBase + (uint32_t)INCR_VIS_STRIPES - 1u) / (uint32_t)INCR_VIS_STRIPES;
}
DI bool bw_next(BandWalk& q, uint32_t* __restrict__ counters, uint32_t n, uint32_t grab, uint32_t* out) {
    if (q.done) return false;
    if (q.cur >= q.end) {
        for (;;) {
            if (q.tried >= (uint32_t)INCR_VIS_STRIPES || q.segLen == 0u) { q.done = true; return false; }
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(counters + 64u + q.stripe * 32u, grab);
            b = __builtin_amdgcn_readfirstlane(b);
            const uint32_t segBegin = q.dynBase + q.stripe * q.segLen;
            if (b < q.segLen && segBegin + b < n) { q.cur = segBegin + b; q.end = min(min(q.cur + grab, segBegin + q.segLen), n); break; }
            ++q.tried; q.stripe = q.stripe + 1u == (uint32_t)INCR_VIS_STRIPES ? 0u : q.stripe + 1u;
        }
    }
    *out = q.cur++;
    return true;
}
template <uint32_t SHIFT>
__global__ __launch_bounds__(256, 4) void vis_scan_kernel(const VisParams P) {
    // one LDS block: a bin's surface words while the workgroup scans bins, the four waves' fragment queues afterwards
    constexpr uint32_t kQueueWords = 4u * 2u * kFragQueue, kLdsWords = BinTile<SHIFT>::kWords > kQueueWords + kQueueWords / 2u ? BinTile<SHIFT>::kWords : kQueueWords + kQueueWords / 2u;
    __shared__ unsigned long long s_lds[kLdsWords];
    __shared__ uint32_t s_next[kVisBinLists];
    const uint32_t lane = threadIdx.x & 63u, waveInBlock = threadIdx.x >> 6;
    if (P.binCap != 0u)
        for (uint32_t bin = blockIdx.x; bin < P.binsX * P.binsY; bin += gridDim.x) scan_bin<SHIFT>(P, bin, s_lds, s_next);
    unsigned long long* const fragWord = s_lds + waveInBlock * 2u * kFragQueue;
    uint32_t* const fragPix = (uint32_t*)(s_lds + kQueueWords) + waveInBlock * 2u * kFragQueue;
    const FragQueues fq = {fragPix, fragWord, fragPix + kFragQueue, fragWord + kFragQueue};
    const uint32_t nBig = min(P.vqCounters[1], P.bigCap), nSmall = P.vqCounters[2], nWhole = P.vqCounters[3];
    QueueWalk q;
    BandWalk bq;
    bw_init(bq, nBig);
#ifndef INCR_VIS_GRAB
#define INCR_VIS_GRAB 1      // band items per atomic on a segment's cursor: with the striped cursors the atomic is cheap and the items that are left (alpha-tested: their cost is their
                             // surviving fragments, unknown beforehand) want the finest balance -- scan kernel 0.504 / 0.556 / 0.686 ms at 1 / 2 / 4
#endif
#ifdef INCR_DIAG_VIS_COUNT
    const long long diagT0 = clock64(), diagW0 = wall_clock64(); long long diagItemMax = 0, diagPrev = diagT0, diagSetup = 0, diagQueue = 0, diagCursor = 0, diagTop = diagT0; uint32_t diagItems = 0;
#endif
#ifdef INCR_DIAG_NO_BAND         // diagnostic only: the scan without its band items
    bq.done = true;
#endif
    for (uint32_t i; bw_next(bq, P.vqCounters, nBig, (uint32_t)INCR_VIS_GRAB, &i);) {          // band items
#ifdef INCR_DIAG_VIS_COUNT
        { const long long now = clock64(); if (diagItems) diagItemMax = max(diagItemMax, now - diagPrev); diagPrev = now; ++diagItems; diagCursor += now - diagTop; }
#endif
        const uint2 item = P.bigQueue[i];
#ifdef INCR_DIAG_VIS_COUNT
        { const uint32_t use = __builtin_amdgcn_readfirstlane(item.x); asm volatile("" :: "s"(use)); diagQueue += clock64() - diagPrev; }
#endif
        const uint32_t rowStart = item.y >> 16, rowCount = item.y & 0xffffu;
        if (!rowCount) continue;
        SubTri st;
        if (!subtri_from_rec(P, item.x, &st)) continue;
#ifdef INCR_DIAG_VIS_COUNT
        { const long long now = clock64(); diagSetup += now - diagPrev; }       // cursor + queue entry + record (+ alpha block) since the item began
#endif
        const int32_t skip = (int32_t)rowStart * 8;
#pragma unroll
        for (int k = 0; k < 3; ++k) st.e[k] += (long long)skip * st.sy[k];
        st.py0 += skip; st.py1 = min(st.py1, st.py0 + (int32_t)rowCount * 8 - 1);
        if (st.alphaTex >= 0) raster_coop_alpha(P, st, lane, fq); else raster_coop(P, st, lane);
#ifdef INCR_DIAG_VIS_COUNT
        diagTop = clock64();
#endif
    }
#ifdef INCR_DIAG_VIS_COUNT
    if (lane == 0) {
        const long long t = clock64() - diagT0;
        atomicAdd((unsigned long long*)(P.vqCounters + 28), (unsigned long long)(wall_clock64() - diagW0));
        atomicAdd((unsigned long long*)(P.vqCounters + 30), (unsigned long long)diagQueue);
        atomicAdd((unsigned long long*)(P.vqCounters + 32), (unsigned long long)diagCursor);
        atomicAdd((unsigned long long*)(P.vqCounters + 20), (unsigned long long)t);
        atomicAdd((unsigned long long*)(P.vqCounters + 26), (unsigned long long)diagSetup);
        atomicMax(P.vqCounters + 22, (uint32_t)(t >> 4));
        atomicMax(P.vqCounters + 23, (uint32_t)(diagItemMax >> 4));
        atomicMax(P.vqCounters + 24, diagItems);
        atomicAdd(P.vqCounters + 18, diagItems ? 1u : 0u);
    }
#endif
    qw_init(q, nWhole, 1u);
    for (uint32_t i; qw_next(q, P.vqCounters + 5, nWhole, 1u, 1u, &i);) {        // whole large triangles (band-queue overflow)
        SubTri st;
        if (subtri_from_rec(P, P.wholeQueue[i], &st)) { if (st.alphaTex >= 0) raster_coop_alpha(P, st, lane, fq); else raster_coop(P, st, lane); }
    }
    qw_init(q, nSmall, 64u);
#ifdef INCR_DIAG_NO_SMALL        // diagnostic only: the scan without its small triangles
    q.done = true;
#endif
    for (uint32_t base; qw_next(q, P.vqCounters + 6, nSmall, 64u, 1u, &base);) { // small triangles: one per lane
        const uint32_t i = base + lane;
        SubTri st;
        if (i < nSmall && subtri_from_rec(P, P.smallQueue[i], &st)) raster_serial(P, st);
    }
}
// packed depth|key -> R32_UINT visibility + D32 depth (clears: 0xFFFFF000 and 0, visibility-pass.inl:68-74)
// 16 B in (two depth|key words), 8 + 8 B out per lane; `first` is even and n is rounded up by the launcher's padding rule
__global__ __launch_bounds__(256) void vis_resolve_kernel(const unsigned long long* __restrict__ packed, uint32_t* __restrict__ vis, float* __restrict__ depth, size_t n) {
    auto id_of = [](unsigned long long v) { return v == 0 ? 0xFFFFF000u : (uint32_t)((v & 0x1ffffffffull) - 1ull); };
    auto z_of = [](unsigned long long v) { return v == 0 ? 0.f : __uint_as_float((uint32_t)(v >> 34)); };
    const bool wide = ((((uintptr_t)packed) & 15u) | (((uintptr_t)vis) & 7u) | (((uintptr_t)depth) & 7u)) == 0;
    const size_t pairs = wide ? n / 2 : 0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < pairs; q += (size_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = ((const ulonglong2*)packed)[q];
        ((uint2*)vis)[q] = make_uint2(id_of(v.x), id_of(v.y));
        ((float2*)depth)[q] = make_float2(z_of(v.x), z_of(v.y));
    }
    for (size_t i = pairs * 2 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long v = packed[i];
        vis[i] = id_of(v); depth[i] = z_of(v);
    }
}

// ---------------------------------------------------------------------------------------------- G-buffer
// gbuffer-compute.hlsl:157-253 (USING_MESHLETS); 16x16 thread groups there, 64x4 pixel blocks here (one wave = one row run)
#ifndef INCR_GBUFFER_MIN_BLOCKS
#define INCR_GBUFFER_MIN_BLOCKS 4
#endif
__global__ __launch_bounds__(256, INCR_GBUFFER_MIN_BLOCKS) void gbuffer_kernel(const GBufferParams P) {
    // the sRGB decode table in LDS: a trilinear tap of a base-colour map looks 24 values up, as many instructions on the vector memory
    // pipeline (which bounds this kernel) as its texel loads
    __shared__ float s_srgbLut[256];
    s_srgbLut[threadIdx.x] = P.S.srgbLut[threadIdx.x];
    __syncthreads();
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.y1) return;
    const size_t i = (size_t)y * P.W + x;
    const uint32_t v = P.vis[i];
    if (v == 0xFFFFF000u) { P.g0[i] = 0; P.g1[i] = 0; P.g2[i] = 0; P.color[i] = 0; return; }
    // DecodeMeshletVisibility / DecodePrimitiveVisibility + Get{Meshlet,Primitive}TriangleData (gbuffer-compute.hlsl:96-156)
    int32_t meshIndex, materialIndex;
    f4 p[3]; f2 uv[3]; f3 nrm[3]; f3 tan[3]; float tanW0 = 0.f;
    if (P.usePrims) {
        const uint32_t primId = v >> 20, tri = v & 0xfffffu;
        if (primId >= P.prims.numPrims) { P.g0[i] = 0; P.g1[i] = 0; P.g2[i] = 0; P.color[i] = 0; return; }   // not a primitive-path id
        const PrimDraw d = P.prims.draws[primId];
        if (tri * 3u + 2u >= d.indexCount) { P.g0[i] = 0; P.g1[i] = 0; P.g2[i] = 0; P.color[i] = 0; return; }
        meshIndex = d.meshIndex; materialIndex = d.materialIndex;
        const MeshFrame& mf = P.frames[meshIndex];
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
            const uint32_t vi = d.vertexBase + P.prims.indices[d.indexBegin + tri * 3u + k];
            const float4 pos = P.prims.pvPos[vi];
            p[k] = mul_vec4({pos.x, pos.y, pos.z, 1.f}, mf.localToClip);
            const float2 t = P.prims.pvUv[vi]; uv[k] = {t.x, t.y};
            const float4 n = P.prims.pvNormal[vi]; nrm[k] = {n.x, n.y, n.z};
            const float4 tg = P.prims.pvTangent[vi]; tan[k] = {tg.x, tg.y, tg.z};
            if (k == 0) tanW0 = tg.w;
        }
    } else {
        const uint32_t meshletId = v >> 7, tri = v & 127u;
        const incr_gpu_meshlet m = P.S.meshlets[meshletId];
        meshIndex = m.meshIndex; materialIndex = m.materialIndex;
        const uint32_t packed = P.S.packedTris[m.triangleBegin + tri];
        const MeshFrame& mf = P.frames[meshIndex];
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
            const uint32_t vi = m.vertexBegin + (0xffu & (packed >> ((2u - k) * 10u)));
            const float4 pos = P.S.mvPos[vi];
            p[k] = mul_vec4({pos.x, pos.y, pos.z, 1.f}, mf.localToClip);
            const float2 t = P.S.mvUv[vi]; uv[k] = {t.x, t.y};
            const float4 n = P.S.mvNormal[vi]; nrm[k] = {n.x, n.y, n.z};
            const float4 tg = P.S.mvTangent[vi]; tan[k] = {tg.x, tg.y, tg.z};
            if (k == 0) tanW0 = tg.w;
        }
    }
    const BarySetup bs = bary_setup(p[0], p[1], p[2]);
    const f3 lam = bary_eval(bs, pixel_ndc(x, y, (float)P.W, (float)P.H));
    f3 N = normalize(interp3(nrm[0], nrm[1], nrm[2], lam));
    const f3 T = normalize(interp3(tan[0], tan[1], tan[2], lam));
    const f3 B = normalize(cross(N, T) * tanW0);
    const f2 UV = interp2(uv[0], uv[1], uv[2], lam);
    const UvGrad gr = quad_uv_gradient(bs, uv[0], uv[1], uv[2], x, y, (float)P.W, (float)P.H);
    const MatProps mi = EvaluateMaterialPropertiesGrad(P.S.materials[materialIndex], P.S.textures, s_srgbLut, UV, gr.ddx, gr.ddy, P.S.maxAnisotropy,
                                                       P.S.materialTexels ? P.S.materialTexels[materialIndex].chain : nullptr);      // Sample(g_anisoSampler, uv), gbuffer-compute.hlsl:240
    if (mi.bHasNormalmap) {
        const float* __restrict__ L = P.S.meshLocalToWorld + (size_t)meshIndex * 16;
        const f3 l0 = {L[0], L[1], L[2]}, l1 = {L[4], L[5], L[6]}, l2 = {L[8], L[9], L[10]};
        const f3 r0 = mul_vec3_m3(T, l0, l1, l2), r1 = mul_vec3_m3(B, l0, l1, l2), r2 = mul_vec3_m3(N, l0, l1, l2);
        N = normalize(mul_vec3_m3(mi.normalmap, r0, r1, r2));
    }
    P.color[i] = pack_r11g11b10(mi.emissive * 20000.f, P.rounding);
    P.g0[i] = pack_unorm8x4({mi.basecolor.x, mi.basecolor.y, mi.basecolor.z, 0.f});
    const f2 oct = OctEncode(N);
    P.g1[i] = f32_to_f16_bits(oct.x, P.rounding) | (f32_to_f16_bits(oct.y, P.rounding) << 16);
    P.g2[i] = pack_unorm8x4({mi.metallic, mi.roughness, mi.ao, mi.aoblend});
}

// ---------------------------------------------------------------------------------------------- deferred lighting
struct GBufferPixel { f3 basecolor, N; float metallic, roughness, ao, aoBlend; };
DI GBufferPixel read_gbuffer(const LightingParams& P, size_t i) {
    GBufferPixel g;
    const f4 b = unpack_unorm8x4(P.g0[i]);
    g.basecolor = P.lightingOnly ? mk3(0.5f) : mk3(b.x, b.y, b.z);
    const uint32_t n = P.g1[i];
    g.N = OctDecode({f16_bits_to_f32(n & 0xffffu), f16_bits_to_f32(n >> 16)});
    const f4 m = unpack_unorm8x4(P.g2[i]);
    g.metallic = m.x; g.roughness = m.y; g.ao = m.z; g.aoBlend = m.w;
    return g;
}
DI f2 screen_pos(uint32_t x, uint32_t y, uint32_t W, uint32_t H) { return {2.f * ((float)x / (float)W) - 1.f, 2.f * ((float)y / (float)H) - 1.f}; }
DI void add_color(const LightingParams& P, size_t i, f3 radiance) { P.color[i] = pack_r11g11b10(unpack_r11g11b10(P.color[i]) + radiance, P.rounding); }
DI void load_global_light(const PtParams& S, uint32_t index, incr_light& light, const float*& m, bool& haveT) {     // D3D robust access: OOB reads 0
    light = {};
    if (index < S.numGlobalLights) light = S.globalLights[index];
    haveT = index < S.numSceneLights;
    m = S.lightTransforms + (haveT ? (size_t)index * 16 : 0);
}

// sky-lighting.hlsl:36-119 with g_envmapIndex == -1
DI f3 ShIrradiance(f3 n, const float4* __restrict__ sh) {                   // spherical-harmonics/common.hlsli:56-126
    const float K[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
    const float A[3] = {3.14159265f, 2.094395f, 0.785398f};
    float v[9];
    v[0] = K[0];
    v[1] = K[1] * n.y; v[2] = K[2] * n.z; v[3] = K[3] * n.x;
    v[4] = (K[4] * n.x) * n.y; v[5] = (K[5] * n.y) * n.z; v[6] = K[6] * ((3.0f * n.z) * n.z - 1.0f); v[7] = (K[7] * n.x) * n.z; v[8] = K[8] * (n.x * n.x - n.y * n.y);
    float c[9];
    c[0] = v[0] * A[0]; c[1] = v[1] * A[1]; c[2] = v[2] * A[1]; c[3] = v[3] * A[1];
#pragma unroll
    for (int i = 4; i < 9; ++i) c[i] = v[i] * A[2];
    f3 irr = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) { const float4 s = sh[i]; irr = irr + mk3(s.x, s.y, s.z) * c[i]; }
    return irr;
}
DI f2 sample_rg16f_bilinear_clamp(const uint32_t* __restrict__ img, uint32_t W, uint32_t H, f2 uv) {      // R16G16_FLOAT LUT, CLAMP, level 0
    float u = uv.x, v = uv.y;
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    // CLAMP: the fixed-point coordinate (pt_device.h linear_address) held inside [0, W - 1] texel centres; a NaN or out-of-range float saturates in the conversion
    u = fminf(fmaxf(u, 0.f), 1.f); v = fminf(fmaxf(v, 0.f), 1.f);
    int xi = linear_fixed(u * (float)W), yi = linear_fixed(v * (float)H), x0i, y0i; float fx, fy;
    xi = min(max(xi, 0), (int)(W - 1) * 256); yi = min(max(yi, 0), (int)(H - 1) * 256);
    linear_split(xi, &x0i, &fx); linear_split(yi, &y0i, &fy);
    const uint32_t x0 = (uint32_t)x0i, y0 = (uint32_t)y0i, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const uint32_t a = img[(size_t)y0 * W + x0], b = img[(size_t)y0 * W + x1], c = img[(size_t)y1 * W + x0], d = img[(size_t)y1 * W + x1];
    const f2 t00 = {f16_bits_to_f32(a & 0xffffu), f16_bits_to_f32(a >> 16)}, t10 = {f16_bits_to_f32(b & 0xffffu), f16_bits_to_f32(b >> 16)};
    const f2 t01 = {f16_bits_to_f32(c & 0xffffu), f16_bits_to_f32(c >> 16)}, t11 = {f16_bits_to_f32(d & 0xffffu), f16_bits_to_f32(d >> 16)};
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return {(t00.x * wx0 + t10.x * fx) * wy0 + (t01.x * wx0 + t11.x * fx) * fy, (t00.y * wx0 + t10.y * fx) * wy0 + (t01.y * wx0 + t11.y * fx) * fy};
}
// radiance the sky pass adds to pixel (x, y): SH9 diffuse + split-sum specular, scaled by the HBAO factor
DI f3 sky_radiance(const SkyParams& P, const GBufferPixel& g, uint32_t x, uint32_t y, size_t i) {
    f3 radiance = {0.f, 0.f, 0.f};
    if (P.diffuseIbl && P.sh9) {
        const f3 albedo = (1.f - g.metallic) * g.basecolor;
        f3 samplingNormal = g.N;
        if (P.useBent) {                                                         // USE_BENT_NORMALS, sky-lighting.hlsl:88-92
            const uint32_t b = P.bent ? P.bent[i] : 0x3c003c00u;                 // HBAO off: cleared to (1, 1)
            samplingNormal = OctDecode({f16_bits_to_f32(b & 0xffffu), f16_bits_to_f32(b >> 16)});
        }
        const f3 shDiffuse = ((albedo * K_INV_PI) * ShIrradiance(samplingNormal, P.sh9)) * 5.f;
        const f3 occluded = g.ao * shDiffuse;
        radiance = radiance + P.skyBrightness * (shDiffuse + g.aoBlend * (occluded - shDiffuse));
    }
    if (P.envmap) {                                                              // sky-lighting.hlsl:98-116: split-sum specular IBL
        f2 sp = screen_pos(x, y, P.L.W, P.L.H); sp.y = -sp.y;
        const f4 wp = mul_vec4({sp.x, sp.y, P.L.depth[i], 1.f}, P.skyInvViewProj);
        const f3 worldPos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
        const f3 eye = {P.L.eye[0], P.L.eye[1], P.L.eye[2]};
        const f3 V = normalize(eye - worldPos);
        const float NoV = hmax(dot(g.N, V), 0.01f);
        const f3 F0 = g.metallic * g.basecolor + (1.f - g.metallic) * 0.04f;
        const f3 R = normalize(reflect(-V, g.N));
        const CubeChain env = {P.envmapRes, P.envmapMips, P.envmap};
        const f3 prefilteredColor = cube_sample_trilinear(env, R, g.roughness * ((float)P.envmapMips - 1.f));
        const f2 envBrdf = sample_rg16f_bilinear_clamp(P.envBrdf, P.envBrdfW, P.envBrdfH, {NoV, g.roughness});
        const f3 specularIBL = prefilteredColor * (F0 * envBrdf.x + envBrdf.y);
        const f3 occludedSpec = g.ao * specularIBL;
        radiance = radiance + P.skyBrightness * (specularIBL + g.aoBlend * (occludedSpec - specularIBL));
    }
    const float hbao = P.ao ? (float)P.ao[i] / 255.f : 1.f;
    return hbao * radiance;
}
__global__ __launch_bounds__(256) void sky_lighting_kernel(const SkyParams P) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.L.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.L.W || y >= P.L.y1) return;
    const size_t i = (size_t)y * P.L.W + x;
    add_color(P.L, i, sky_radiance(P, read_gbuffer(P.L, i), x, y, i));
}

// Ray statistics of a lane-per-pixel pass (every thread of the block calls this once, outside divergent code).  One atomic
// per WAVE on the caller's counter was the critical path of the sun pass: 130 k same-address atomics at ~12 ns each are
// 1.5 ms at 4K.  The block's count is reduced through LDS and added to one of 64 striped counters (own cache lines);
// fold_ray_stats_kernel moves the stripes into incr_ray_stats::light_shadow_rays after the pass.
DI void count_rays(const PtParams& S, uint32_t rays) {
    __shared__ uint32_t s_rays;
    if (!S.stats) return;                                   // uniform: a kernel argument
    if (threadIdx.x == 0) s_rays = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63u) == 0 && rays) atomicAdd(&s_rays, rays);
    __syncthreads();
    if (threadIdx.x == 0 && s_rays) {
        if (S.statStripes) atomicAdd(S.statStripes + ((blockIdx.y * gridDim.x + blockIdx.x) & 63u) * 16u + 3u, (unsigned long long)s_rays);   // word 3 = light_shadow_rays
        else atomicAdd((unsigned long long*)&S.stats->light_shadow_rays, (unsigned long long)s_rays);
    }
}
// a stripe holds the five counters of incr_ray_stats in its first five words
__global__ __launch_bounds__(64) void fold_ray_stats_kernel(unsigned long long* __restrict__ stripes, incr_ray_stats* __restrict__ stats) {
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) {
        unsigned long long v = atomicExch(stripes + threadIdx.x * 16u + k, 0ull);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0 && v) atomicAdd((unsigned long long*)stats + k, v);
    }
}
extern "C" hipError_t incr_launch_fold_ray_stats(unsigned long long* stripes, incr_ray_stats* stats, hipStream_t stream) {
    hipLaunchKernelGGL(fold_ray_stats_kernel, dim3(1), dim3(64), 0, stream, stripes, stats);
    return hipGetLastError();
}

// ---- the sun's shadow rays as a PACKET ---------------------------------------------------------------------------------------
// The 64 rays of a wave -- an 8 x 8 pixel tile, one direction (the sun), origins on neighbouring surface points -- walk the tree
// TOGETHER: the wave visits a node when any of its rays overlaps it, so the node index, the stack, the leaf / triangle loop and
// the 80 + 48-byte fetches are wave-uniform (scalar registers, scalar loads), and every lane tests ITS ray against the eight
// boxes / the triangle with the whole wave active.  The lane-per-ray loop (pt_traverse.h trace_any) spends most of its vector
// instructions on lanes that are idle (no ray for a sky or back-facing pixel, a shorter walk than the neighbour's) or on the other
// kind of step.  Any-hit is order-independent, and a ray tested against a triangle of a leaf only its neighbours overlap gives the
// same answer as never testing it, so the result equals trace_any's for every ray.  All 64 lanes must call; `wants` marks the
// lanes that have a ray.  `stack`: kPacketStack words of LDS per wave.
static constexpr int kPacketStack = 32;      // one entry per tree level (kMaxBvhDepth = 25)
DI bool trace_any_packet(const PtParams& P, uint32_t* __restrict__ stack, f3 o, f3 d, float tmin, float tmax, bool wants) {
    const RaySetup rs = ray_setup(o, d);
    bool alive = wants, occluded = false;
    int sp = 0;
    uint32_t cur = 0;
    bool haveCur = true;
    for (;;) {
        if (__ballot(alive) == 0ull) break;
        if (!haveCur) {
            if (sp == 0) break;
            uint32_t e = __builtin_amdgcn_readfirstlane(stack[sp - 1]);
            cur = (e >> 9) + (uint32_t)__builtin_ctz(e & 0xffu);
            e &= e - 1u;
            if (e & 0xffu) stack[sp - 1] = e; else --sp;
        }
        haveCur = false;
        const float4* __restrict__ np = (const float4*)P.nodes + (size_t)cur * 5;        // wave-uniform address: scalar loads
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        Node8 N;
        node8_test(n0, n1, n2, n3, n4, rs, tmin, tmax, N);
        const uint32_t mine = alive ? N.hitMask : 0u;
        uint32_t any = 0u;                                                                 // children some live ray overlaps
#pragma unroll
        for (int k = 0; k < 8; ++k) any |= (__ballot((mine >> k) & 1u) != 0ull) ? (1u << k) : 0u;
        const uint32_t imask = __builtin_amdgcn_readfirstlane(N.imask), childBase = __builtin_amdgcn_readfirstlane(N.childBase);
        Node8 U = N; U.hitMask = any; U.imask = imask;
        U.leafTris = __builtin_amdgcn_readfirstlane(N.leafTris); U.sortAxis = __builtin_amdgcn_readfirstlane(N.sortAxis);
        uint32_t triMask = node8_tri_mask_uniform(U.hitMask, U.leafTris);
        const uint32_t triBase = __builtin_amdgcn_readfirstlane(N.triBase);
        while (triMask) {
            const uint32_t ti = triBase + (uint32_t)__builtin_ctz(triMask);
            triMask &= triMask - 1u;
            const float4* __restrict__ tp = (const float4*)(P.tris + ti);
            const float4 a = tp[0], b = tp[1], c = tp[2];
            float det, U, V, Tn;
            if (alive && tri_hit(a, b, c, rs.o, d, tmin, tmax, &det, &U, &V, &Tn)) { occluded = true; alive = false; }
            if (__ballot(alive) == 0ull) break;
        }
        const uint32_t inner = any & imask;
        if (inner) {
            const uint32_t rest = inner & (inner - 1u);
            if (rest) { if (sp < kPacketStack) stack[sp++] = stack_entry(childBase, false, rest); }
            cur = childBase + (uint32_t)__builtin_ctz(inner);
            haveCur = true;
        }
    }
    return occluded;
}

// ---- sun cover map (raster_params.h SunCover) ----
// One thread: the frame of the map from the sun's direction (light_setup's, from the light table the lighting pass reads) and the root node's box.
__global__ void sun_cover_header_kernel(const PtParams S, uint32_t sunIndex, SunCoverHeader* __restrict__ out) {
    SunCoverHeader h = {};
    incr_light light; const float* m; bool haveT;
    load_global_light(S, sunIndex, light, m, haveT);
    const LightSample ls = light_setup(light, m, haveT, {0.f, 0.f, 0.f}, {0.f, 1.f, 0.f});
    const f3 L = ls.L;
    const bool unit = fabsf(dot(L, L) - 1.f) < 1e-3f;             // also false for NaN
    if (light.type == INCR_LIGHT_DIRECTIONAL && unit && S.numNodes != 0u) {
        const f3 a = fabsf(L.x) < 0.577f ? mk3(1.f, 0.f, 0.f) : (fabsf(L.y) < 0.577f ? mk3(0.f, 1.f, 0.f) : mk3(0.f, 0.f, 1.f));
        const f3 U = normalize(cross(a, L)), V = cross(L, U);
        const BvhNode8 root = S.nodes[0];
        const f3 lo = {root.ox, root.oy, root.oz};
        const f3 hi = {root.ox + 256.f * __uint_as_float((uint32_t)root.ex << 23), root.oy + 256.f * __uint_as_float((uint32_t)root.ey << 23), root.oz + 256.f * __uint_as_float((uint32_t)root.ez << 23)};
        float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f, ext = 0.f;
        for (int c = 0; c < 8; ++c) {
            const f3 p = {c & 1 ? hi.x : lo.x, c & 2 ? hi.y : lo.y, c & 4 ? hi.z : lo.z};
            const float u = dot(p, U), v = dot(p, V);
            umin = fminf(umin, u); umax = fmaxf(umax, u); vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            ext = fmaxf(ext, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
        }
        const float side = fmaxf(umax - umin, vmax - vmin);
        if (side > 0.f && side < 3.0e37f && ext < 3.0e37f) {
            h.U[0] = U.x; h.U[1] = U.y; h.U[2] = U.z; h.V[0] = V.x; h.V[1] = V.y; h.V[2] = V.z; h.L[0] = L.x; h.L[1] = L.y; h.L[2] = L.z;
            // eps: far above the rounding of a projection (|p| 2^-22) and of the traversal's triangle test, far below a cell
            h.eps = fmaxf(ext, side) * (1.f / 16384.f);
            h.u0 = umin - 2.f * h.eps; h.v0 = vmin - 2.f * h.eps;
            h.invCell = (float)kSunCoverG / (side + 4.f * h.eps);
            h.valid = 1u;
        }
    }
#ifdef INCR_DIAG_SUN_COUNT
    printf("sun cover header: valid %u type %u L %g %g %g u0 %g v0 %g invCell %g eps %g nodes %u root %g %g %g e %u %u %u\n", h.valid, (unsigned)light.type, L.x, L.y, L.z, h.u0, h.v0, h.invCell, h.eps,
           S.numNodes, S.nodes[0].ox, S.nodes[0].oy, S.nodes[0].oz, (unsigned)S.nodes[0].ex, (unsigned)S.nodes[0].ey, (unsigned)S.nodes[0].ez);
#endif
    *out = h;
}
// One lane per triangle record, conservative rasterisation onto the map: every cell that the triangle's projection (grown by eps) can touch takes the greatest height of
// the triangle's PLANE over the cell, capped by the triangle's highest vertex (+ eps) -- a sloped triangle must not lift its whole bounding box to its top vertex, or the
// surface a shadow ray starts from would hide its own lit neighbourhood.  Triangles seen edge-on or nearly so (plane gradient beyond 1000) fall back to the vertex cap.
// Footprints of more than 16 cells are spread over the wave.
struct SunTri { float u[3], v[3]; float h0, gu, gv, hcap, orient; int32_t cu0, cu1, cv0, cv1; };
DI void sun_cover_cell(const SunCoverHeader& H, const SunTri& t, int32_t cu, int32_t cv, uint32_t* __restrict__ grid) {
    const float cell = 1.f / H.invCell;
    const float x0 = H.u0 + (float)cu * cell - H.eps, x1 = H.u0 + (float)(cu + 1) * cell + H.eps, y0 = H.v0 + (float)cv * cell - H.eps, y1 = H.v0 + (float)(cv + 1) * cell + H.eps;
    if (t.orient != 0.f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {                            // the cell's corner that lies deepest inside edge k must not be outside it
            const int j = k == 2 ? 0 : k + 1;
            const float A = -t.orient * (t.v[j] - t.v[k]), B = t.orient * (t.u[j] - t.u[k]);
            const float e = A * ((A > 0.f ? x1 : x0) - t.u[k]) + B * ((B > 0.f ? y1 : y0) - t.v[k]);
            if (e < -(fabsf(A) + fabsf(B)) * H.eps) return;
        }
    }
    float h = t.hcap;
    if (t.gu == t.gu) h = fminf(h, t.h0 + t.gu * ((t.gu > 0.f ? x1 : x0) - t.u[0]) + t.gv * ((t.gv > 0.f ? y1 : y0) - t.v[0]) + 2.f * H.eps);
    atomicMax(grid + (size_t)cv * kSunCoverG + (uint32_t)cu, height_key(h));
}
// Record -> what the rasterisation needs (false: a hole, a back face, a record with NaNs: nothing a shadow ray toward the sun can hit)
DI bool sun_tri_setup(const SunCoverHeader& H, const TriIsect& r, SunTri& t) {
    if (r.flatId == kTriHole) return false;
    const f3 p0 = {r.v0x, r.v0y, r.v0z}, p1 = {r.v0x + r.e1x, r.v0y + r.e1y, r.v0z + r.e1z}, p2 = {r.v0x + r.e2x, r.v0y + r.e2y, r.v0z + r.e2z};
    const f3 U = {H.U[0], H.U[1], H.U[2]}, V = {H.V[0], H.V[1], H.V[2]}, L = {H.L[0], H.L[1], H.L[2]};
    t.u[0] = dot(p0, U); t.u[1] = dot(p1, U); t.u[2] = dot(p2, U); t.v[0] = dot(p0, V); t.v[1] = dot(p1, V); t.v[2] = dot(p2, V);
    const float h0 = dot(p0, L), h1 = dot(p1, L), h2 = dot(p2, L);
    t.h0 = h0; t.hcap = fmaxf(h0, fmaxf(h1, h2)) + H.eps;
    const float ulo = fminf(t.u[0], fminf(t.u[1], t.u[2])) - H.eps, uhi = fmaxf(t.u[0], fmaxf(t.u[1], t.u[2])) + H.eps;
    const float vlo = fminf(t.v[0], fminf(t.v[1], t.v[2])) - H.eps, vhi = fmaxf(t.v[0], fmaxf(t.v[1], t.v[2])) + H.eps;
    // the shadow query culls back faces (pt_traverse.h tri_hit: det = e1 . (d x e2) > 0 with d = L): a triangle that fails this for the sun's direction
    // -- the courtyard's roof seen from below -- hides nothing.  The same expression; kept when within rounding of zero.
    const f3 e1 = {r.e1x, r.e1y, r.e1z}, e2 = {r.e2x, r.e2y, r.e2z};
    const float det = dot(e1, cross(L, e2));
    const bool faces = !(det < -1e-5f * sqrtf(dot(e1, e1) * dot(e2, e2)));
    if (!(faces && ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi && t.hcap == t.hcap)) return false;          // (a record with NaNs no ray can hit)
    const float G1 = (float)(kSunCoverG - 1u);
    t.cu0 = (int32_t)fminf(fmaxf(floorf((ulo - H.u0) * H.invCell), 0.f), G1); t.cu1 = (int32_t)fminf(fmaxf(floorf((uhi - H.u0) * H.invCell), 0.f), G1);
    t.cv0 = (int32_t)fminf(fmaxf(floorf((vlo - H.v0) * H.invCell), 0.f), G1); t.cv1 = (int32_t)fminf(fmaxf(floorf((vhi - H.v0) * H.invCell), 0.f), G1);
    const float du1 = t.u[1] - t.u[0], dv1 = t.v[1] - t.v[0], du2 = t.u[2] - t.u[0], dv2 = t.v[2] - t.v[0];
    const float area2 = du1 * dv2 - du2 * dv1;
    // the projection's area against its edges' rounding: below that the triangle is a sliver on the map -- every cell of its box, the vertex cap
    const bool solid = fabsf(area2) > 1e-4f * ((fabsf(du1) + fabsf(dv1)) * (fabsf(du2) + fabsf(dv2)));
    t.orient = solid ? (area2 > 0.f ? 1.f : -1.f) : 0.f;
    t.gu = t.gv = __uint_as_float(0x7fc00000u);
    if (solid) {
        const float gu = ((h1 - h0) * dv2 - (h2 - h0) * dv1) / area2, gv = ((h2 - h0) * du1 - (h1 - h0) * du2) / area2;
        if (fabsf(gu) < 1000.f && fabsf(gv) < 1000.f) { t.gu = gu; t.gv = gv; }
    }
    return true;
}
// Footprints by size: <= 16 cells by the triangle's own lane; <= 1024 spread over its wave; larger ones go to a list that sun_cover_large_kernel works off with a
// workgroup per 65 536 cells -- a ground plane of two triangles covers the whole map, and one wave would be at it for 50 ms.
static constexpr uint32_t kSunLargeCells = 1024, kSunLargeChunk = 65536, kSunLargeCap = 65536;
__global__ __launch_bounds__(256) void sun_cover_raster_kernel(const TriIsect* __restrict__ tris, uint32_t numTris, SunCoverHeader* __restrict__ header, uint32_t* __restrict__ grid, uint2* __restrict__ large) {
    const SunCoverHeader H = *header;
    if (!H.valid) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    SunTri t = {};
    bool have = i < numTris && sun_tri_setup(H, tris[i], t);
    const uint32_t ncu = have ? (uint32_t)(t.cu1 - t.cu0 + 1) : 0u, cells = have ? ncu * (uint32_t)(t.cv1 - t.cv0 + 1) : 0u;
    if (have && cells > kSunLargeCells) {
        const uint32_t chunks = (cells + kSunLargeChunk - 1u) / kSunLargeChunk, at = atomicAdd(&header->pad[0], chunks);
        if (at + chunks <= kSunLargeCap) { for (uint32_t c = 0; c < chunks; ++c) large[at + c] = make_uint2(i, c); have = false; }       // (a full list: the wave does it after all)
    }
    if (have && cells <= 16u)
        for (int32_t cv = t.cv0; cv <= t.cv1; ++cv)
            for (int32_t cu = t.cu0; cu <= t.cu1; ++cu) sun_cover_cell(H, t, cu, cv, grid);
    for (unsigned long long big = __ballot(have && cells > 16u); big != 0ull; big &= big - 1ull) {
        const int src = __ffsll((long long)big) - 1;
        SunTri s;
#pragma unroll
        for (int k = 0; k < 3; ++k) { s.u[k] = __shfl(t.u[k], src, 64); s.v[k] = __shfl(t.v[k], src, 64); }
        s.h0 = __shfl(t.h0, src, 64); s.gu = __shfl(t.gu, src, 64); s.gv = __shfl(t.gv, src, 64); s.hcap = __shfl(t.hcap, src, 64); s.orient = __shfl(t.orient, src, 64);
        s.cu0 = __shfl(t.cu0, src, 64); s.cv0 = __shfl(t.cv0, src, 64);
        const uint32_t sn = (uint32_t)__shfl((int)ncu, src, 64), sc = (uint32_t)__shfl((int)cells, src, 64);
        for (uint32_t j = lane; j < sc; j += 64u) sun_cover_cell(H, s, s.cu0 + (int32_t)(j % sn), s.cv0 + (int32_t)(j / sn), grid);
    }
}
__global__ __launch_bounds__(256) void sun_cover_large_kernel(const TriIsect* __restrict__ tris, const SunCoverHeader* __restrict__ header, uint32_t* __restrict__ grid, const uint2* __restrict__ large) {
    const SunCoverHeader H = *header;
    if (!H.valid) return;
    const uint32_t n = min(H.pad[0], kSunLargeCap);
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const uint2 item = large[e];
        SunTri t = {};
        if (!sun_tri_setup(H, tris[item.x], t)) continue;          // (cannot fail: it was filed)
        const uint32_t ncu = (uint32_t)(t.cu1 - t.cu0 + 1), cells = ncu * (uint32_t)(t.cv1 - t.cv0 + 1);
        const uint32_t lo = item.y * kSunLargeChunk, hi = min(cells, lo + kSunLargeChunk);
        for (uint32_t j = lo + threadIdx.x; j < hi; j += 256u) sun_cover_cell(H, t, t.cu0 + (int32_t)(j % ncu), t.cv0 + (int32_t)(j / ncu), grid);
    }
}
// direct-lighting.hlsl:31-85: the sun with a ray-traced shadow (lighting/common.hlsli:102-115), in two halves around the shadow
// query so that the query can be made by the whole wave at once
struct DirectSetup { LightSample s; f3 worldPos, V; bool wantsRay; };
DI DirectSetup direct_setup(const DirectParams& P, const GBufferPixel& g, uint32_t x, uint32_t y, size_t i, uint32_t& rays) {
    DirectSetup ds;
    const f2 sp = screen_pos(x, y, P.L.W, P.L.H);
    const f4 w = mul_vec4({sp.x, -sp.y, P.L.depth[i], 1.f}, P.L.invViewProj);
    ds.worldPos = {w.x / w.w, w.y / w.w, w.z / w.w};
    const f3 eye = {P.L.eye[0], P.L.eye[1], P.L.eye[2]};
    ds.V = normalize(eye - ds.worldPos);
    incr_light light; const float* m; bool haveT;
    load_global_light(P.S, P.sunIndex, light, m, haveT);
    ds.s = light_setup(light, m, haveT, ds.worldPos, g.N);
    ds.wantsRay = false;
    if (ds.s.needsShadow && light.type == INCR_LIGHT_DIRECTIONAL) {
        ++rays;
        // background pixels unproject to non-finite positions: such a ray hits nothing (every triangle test fails
        // on NaN/inf), but the box tests cannot reject it and the walk would visit the whole tree
        ds.wantsRay = isfinite(ds.worldPos.x) && isfinite(ds.worldPos.y) && isfinite(ds.worldPos.z) && isfinite(ds.s.L.x) && isfinite(ds.s.L.y) && isfinite(ds.s.L.z);
    }
    return ds;
}
DI f3 direct_finish(const DirectSetup& ds, const GBufferPixel& g, bool occluded) {
    if (!ds.s.needsShadow || occluded) return {0.f, 0.f, 0.f};
    return light_shade(ds.s, g.basecolor, g.metallic, g.roughness, g.N, ds.V);
}
// Pixel of a lane: one 8 x 8 tile per wave (the packet above wants neighbouring origins), four tiles side by side per workgroup
DI void tile_pixel(uint32_t y0, uint32_t* x, uint32_t* y) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    *x = blockIdx.x * 32u + (tid >> 6) * 8u + (lane & 7u);
    *y = y0 + blockIdx.y * 8u + (lane >> 3);
}
__global__ __launch_bounds__(kBlock) void direct_lighting_kernel(const DirectParams P) {
    __shared__ uint32_t s_stack[(kBlock / 64) * kPacketStack];
    uint32_t x, y; tile_pixel(P.L.y0, &x, &y);
    const bool valid = x < P.L.W && y < P.L.y1;
    const size_t i = valid ? (size_t)y * P.L.W + x : 0;
    uint32_t rays = 0;
    GBufferPixel g = {};
    DirectSetup ds = {};
    if (valid) { g = read_gbuffer(P.L, i); ds = direct_setup(P, g, x, y, i, rays); }
#ifdef INCR_DIAG_SUN_COUNT       // diagnostic only: light_shadow_rays = rays + 65536 x rays the cover map answers (=2) / occluded rays (otherwise)
    const bool diagWanted = ds.wantsRay;
    if (P.S.cover.header && valid && (x % 640u) == 320u && (y % 540u) == 270u) {
        const SunCoverHeader& H = *P.S.cover.header; const f3 p = ds.worldPos;
        const float u = ((p.x * H.U[0] + p.y * H.U[1]) + p.z * H.U[2] - H.u0) * H.invCell, v = ((p.x * H.V[0] + p.y * H.V[1]) + p.z * H.V[2] - H.v0) * H.invCell;
        const float h = (p.x * H.L[0] + p.y * H.L[1]) + p.z * H.L[2];
        const uint32_t k = (u >= 0.f && v >= 0.f && u < (float)kSunCoverG && v < (float)kSunCoverG) ? P.S.cover.grid[(size_t)(uint32_t)v * kSunCoverG + (uint32_t)u] : 0u;
        const float stored = __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
        printf("pixel %u %u: world %g %g %g wants %d  cell %g %g  height %g  map %g (key %08x)  N %g %g %g\n", x, y, p.x, p.y, p.z, (int)ds.wantsRay, u, v, h, stored, k, g.N.x, g.N.y, g.N.z);
    }
#endif
    if (P.S.cover.header && ds.wantsRay && sun_cover_clear(P.S.cover, ds.worldPos, 0.1f)) ds.wantsRay = false;     // provably unoccluded: `occluded` stays false
    const bool occluded = trace_any_packet(P.S, s_stack + (threadIdx.x >> 6) * kPacketStack, ds.worldPos, ds.s.L, 0.1f, 1000.f, valid && ds.wantsRay);
    if (valid) add_color(P.L, i, direct_finish(ds, g, occluded));
#ifdef INCR_DIAG_SUN_COUNT
    if (INCR_DIAG_SUN_COUNT == 2 ? (valid && diagWanted && !ds.wantsRay) : (valid && ds.wantsRay && occluded)) rays += 65536u;
#endif
    count_rays(P.S, rays);
}

// ---------------------------------------------------------------------------------------------- HBAO
// shaders/postprocess/hbao.hlsl.  cos / sin / acos are fixed polynomial evaluations (HLSL leaves their
// precision open): Cody-Waite reduction by pi/2 + Taylor polynomials on [-pi/4, pi/4], the arcsine series with the
// half-angle identity, rsqrt(x) = 1 / sqrt(x) -- operation for operation what the CPU oracle evaluates.
DI void hb_reduce(float x, float& r, int& q) {
    const float kf = floorf(x * 0.636619747f + 0.5f);
    r = (x - kf * 1.5707963705062866f) - kf * -4.3711388286737929e-08f;
    q = (kf == kf) ? ((int)kf & 3) : 0;
}
DI float hb_sin_poly(float r) {
    const float z = r * r;
    return r + (r * z) * (-0.166666672f + z * (0.00833333377f + z * (-0.000198412701f + z * 2.75573188e-06f)));
}
DI float hb_cos_poly(float r) {
    const float z = r * r;
    return 1.f + z * (-0.5f + z * (0.0416666679f + z * (-0.00138888892f + z * (2.48015876e-05f + z * -2.755732e-07f))));
}
DI void hb_sincos(float x, float& s, float& c) {
    float r; int q; hb_reduce(x, r, q);
    const float sp = hb_sin_poly(r), cp = hb_cos_poly(r);
    s = (q & 1) ? cp : sp; if (q & 2) s = -s;
    c = (q & 1) ? sp : cp; if ((q + 1) & 2) c = -c;
}
DI float hb_asin_half(float x) {
    const float z = x * x;
    const float p = 0.166666672f + z * (0.075000003f + z * (0.0446428582f + z * (0.030381944f + z * (0.0223721582f + z * (0.0173527636f
                  + z * (0.0139648439f + z * (0.0115518011f + z * (0.00976160914f + z * (0.00839033537f + z * 0.007312526f)))))))));
    return x + (x * z) * p;
}
DI float hb_acos(float x) {
    if (!(fabsf(x) > 0.5f)) return 1.5707963705062866f - (hb_asin_half(x) - -4.3711388286737929e-08f);
    if (x > 0.f) return 2.f * hb_asin_half(sqrtf((1.f - x) * 0.5f));
    return 3.1415927410125732f - (2.f * hb_asin_half(sqrtf((1.f + x) * 0.5f)) - -8.7422776573475858e-08f);
}
DI uint32_t hb_f16(float v, int rounding) { return v != v ? 0x7fffu : f32_to_f16_bits(v, rounding); }

static constexpr float kHbaoPi = 3.14159265358979323846f;
// ComputeHorizonAngle (hbao.hlsl:66-93): binary search of the horizon, at most 4 ray queries (TMin 0.1, TMax 1).  WAVE-SYNCHRONOUS (round 4):
// the wave is an 8 x 8 pixel tile and its 64 queries of one search step -- neighbouring origins, directions in neighbouring slice planes a
// few degrees apart -- go through the tree as one packet (trace_any_packet: wave-uniform node walk on scalar loads, every lane tests its
// own ray; an any-hit answer does not depend on the walk).  Rounds 1-3 ran one lane-per-ray query after the other with 24-KB LDS stacks:
// 25.6 ms per 4K frame.  All 64 lanes must call; `active` = the lane has a pixel.
DI float hbao_horizon(const PtParams& S, uint32_t* __restrict__ stack, float start, f3 pos, f3 sliceX, f3 sliceY, bool finitePos, bool active, uint32_t& rays) {
    const float terminate = kHbaoPi / 36.f;
    float lo = 0.f, hi = start;
    for (int k = 0; k < 4; ++k) {
        const bool searching = active && (hi - lo) > terminate;
        if (__ballot(searching) == 0ull) break;
        const float a = lo + 0.5f * (hi - lo);
        float sa, ca; hb_sincos(a, sa, ca);
        const f3 dir = ca * sliceY + sa * sliceX;
        if (searching) ++rays;
        const bool finite = finitePos && isfinite(dir.x) && isfinite(dir.y) && isfinite(dir.z);
        const bool occluded = trace_any_packet(S, stack, pos, dir, 0.1f, 1.f, searching && finite);
        if (searching) { if (finite && occluded) hi = a; else lo = a; }
    }
    return hi;
}
#ifndef INCR_HBAO_MIN_BLOCKS
#define INCR_HBAO_MIN_BLOCKS 8          // 64 VGPRs (36 B of scratch) instead of 72: the pass 14.25 -> 13.64 ms
#endif
__global__ __launch_bounds__(kBlock, INCR_HBAO_MIN_BLOCKS) void hbao_kernel(const HbaoParams P) {            // cs_main, hbao.hlsl:109-175
    __shared__ uint32_t s_stack[(kBlock / 64) * kPacketStack];
    uint32_t* stack = s_stack + (threadIdx.x >> 6) * kPacketStack;
    uint32_t x, y; tile_pixel(P.y0, &x, &y);
    const bool valid = x < P.W && y < P.y1;
    uint32_t rays = 0;
    const size_t i = valid ? (size_t)y * P.W + x : 0;
    f3 pos = {0.f, 0.f, 0.f}, localAt = {0.f, 0.f, 1.f}, localRight = {1.f, 0.f, 0.f}, localUp = {0.f, 1.f, 0.f}, N = {0.f, 0.f, 1.f};
    bool finitePos = false;
    if (valid) {
        const f2 ndc = pixel_ndc(x, y, (float)P.W, (float)P.H);
        const f4 wp = mul_vec4({ndc.x, ndc.y, P.depth[i], 1.f}, P.invViewProj);
        pos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
        finitePos = isfinite(pos.x) && isfinite(pos.y) && isfinite(pos.z);
        const f3 eye = {P.eye[0], P.eye[1], P.eye[2]}, camUp = {P.camUp[0], P.camUp[1], P.camUp[2]};
        localAt = normalize(eye - pos);
        localRight = normalize(cross(localAt, camUp));
        localUp = cross(localRight, localAt);
        const uint32_t n = P.normals[i];
        N = OctDecode({f16_bits_to_f32(n & 0xffffu), f16_bits_to_f32(n >> 16)});
    }
    float azimuth = 0.f, sumAO = 0.f;
    f3 avgBent = {0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) {
        float sa, ca; hb_sincos(azimuth, sa, ca);
        const f3 sliceT = ca * localRight + sa * localUp;
        const float pnx = dot(N, sliceT), pny = dot(N, localAt);
        const float t = -dot(sliceT, N) / dot(localAt, N);
        const float horizon = hb_acos(t * (1.f / sqrtf(1.f + t * t)));
        const float theta1 = hbao_horizon(P.S, stack, horizon, pos, sliceT, localAt, finitePos, valid, rays);
        const float theta0 = -hbao_horizon(P.S, stack, horizon, pos, -sliceT, localAt, finitePos, valid, rays);
        float s0, c0, s1, c1; hb_sincos(theta0, s0, c0); hb_sincos(theta1, s1, c1);
        const float nx = 0.5f * (((theta1 - theta0) + s0 * c0) - s1 * c1);
        const float ny = 0.5f * ((2.f - c0 * c0) - c1 * c1);
        sumAO += pnx * nx + pny * ny;
        avgBent = avgBent + normalize(sliceT * nx + localAt * ny);
        azimuth += kHbaoPi / 4.f;
    }
    if (valid) {
        const float aoRaw = sumAO / 4.f;
        const float ao = !(aoRaw > 0.f) ? 0.f : (aoRaw > 1.f ? 1.f : aoRaw);
        P.ao[i] = (uint8_t)unorm8(ao);
        const f2 oct = OctEncode(normalize(avgBent));
        P.bent[i] = hb_f16(oct.x, P.rounding) | (hb_f16(oct.y, P.rounding) << 16);
    }
    count_rays(P.S, rays);
}

// ---------------------------------------------------------------------------------------------- forward lighting
// forward-pass.hlsl vs_main + ps_main over the nearest fragment the forward-order visibility pass found: vertex-shader
// outputs re-evaluated per pixel and interpolated with the G-buffer pass's barycentric formula.
__global__ __launch_bounds__(kBlock) void forward_shade_kernel(const ForwardParams P) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    const uint32_t tid = threadIdx.x;
    const uint32_t x = blockIdx.x * 64 + (tid & 63u), y = P.y0 + blockIdx.y * 4 + (tid >> 6);
    uint32_t rays = 0;
    if (x < P.W && y < P.y1) {
        const size_t i = (size_t)y * P.W + x;
        const uint32_t v = P.vis[i];
        const uint32_t primId = v >> 20, tri = v & 0xfffffu;
        f3 out = {0.8f, 0.8f, 1.f};                                                  // clearColor, forward-pass.inl:62
        bool covered = v != 0xFFFFF000u && primId < P.prims.numPrims;
        PrimDraw d = {};
        if (covered) { d = P.prims.draws[primId]; covered = tri * 3u + 2u < d.indexCount; }
        if (covered) {
            LaneStack st; st.bind(s_stack + tid, nullptr); st.sp = 0;
            const float* __restrict__ l2w = P.S.meshLocalToWorld + (size_t)d.meshIndex * 16;
            f4 clip[3], wpos[3]; f2 uv[3]; f3 nW[3], tW[3], bW[3];
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {                                       // vs_main :57-90
                const uint32_t vi = d.vertexBase + P.prims.indices[d.indexBegin + tri * 3u + k];
                const float4 pos = P.prims.pvPos[vi];
                wpos[k] = mul_vec4({pos.x, pos.y, pos.z, 1.f}, l2w);
                clip[k] = mul_vec4(wpos[k], P.viewProj);
                const float2 t = P.prims.pvUv[vi]; uv[k] = {t.x, t.y};
                const float4 n4 = P.prims.pvNormal[vi], t4 = P.prims.pvTangent[vi];
                const f3 normal = {n4.x, n4.y, n4.z}, tangent = {t4.x, t4.y, t4.z};
                const f3 bitangent = cross(normal, tangent) * t4.w;
                const f4 a = mul_vec4({normal.x, normal.y, normal.z, 0.f}, l2w), b = mul_vec4({tangent.x, tangent.y, tangent.z, 0.f}, l2w);
                const f4 c = mul_vec4({bitangent.x, bitangent.y, bitangent.z, 0.f}, l2w);
                nW[k] = {a.x, a.y, a.z}; tW[k] = {b.x, b.y, b.z}; bW[k] = {c.x, c.y, c.z};
            }
            const BarySetup bs = bary_setup(clip[0], clip[1], clip[2]);
            const f3 lam = bary_eval(bs, pixel_ndc(x, y, (float)P.W, (float)P.H));
            const f2 UV = interp2(uv[0], uv[1], uv[2], lam);
            const incr_material mat = P.S.materials[d.materialIndex];
            const UvGrad gr = quad_uv_gradient(bs, uv[0], uv[1], uv[2], x, y, (float)P.W, (float)P.H);
            const MatProps p = EvaluateMaterialPropertiesGrad(mat, P.S.textures, P.S.srgbLut, UV, gr.ddx, gr.ddy, P.S.maxAnisotropy, P.S.materialTexels ? P.S.materialTexels[d.materialIndex].chain : nullptr);
            bool done = true;
            if (P.viewmode == 2) out = mk3(p.roughness);                             // ps_main :95-105 (`#if VIWEMODE == 1` never fires)
            else if (P.viewmode == 3) out = mk3(p.metallic);
            else if (P.viewmode == 4) out = p.basecolor;
            else if (P.viewmode == 5) out = p.emissive;
            else done = false;
            if (!done) {
                const f3 T = normalize(interp3(tW[0], tW[1], tW[2], lam)), B = normalize(interp3(bW[0], bW[1], bW[2], lam));
                f3 N = normalize(interp3(nW[0], nW[1], nW[2], lam));
                if (p.bHasNormalmap) {
                    const f3 nm = p.normalmap;
                    N = normalize(mk3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
                }
                const f3 wp = interp3({wpos[0].x, wpos[0].y, wpos[0].z}, {wpos[1].x, wpos[1].y, wpos[1].z}, {wpos[2].x, wpos[2].y, wpos[2].z}, lam);
                const float ww = (lam.x * wpos[0].w + lam.y * wpos[1].w) + lam.z * wpos[2].w;
                const f3 worldPos = {wp.x / ww, wp.y / ww, wp.z / ww};
                const f3 eye = {P.eye[0], P.eye[1], P.eye[2]};
                const f3 V = normalize(eye - worldPos);
                f3 radiance = p.emissive * 20000;
                const CubeChain env = {P.envmapRes, P.envmapMips, P.envmap};
                if (P.viewmode == 7 && P.envmap) {                                   // Reflections :123-131
                    out = cube_sample_trilinear(env, normalize(reflect(-V, N)), 0.f);
                    done = true;
                }
                if (!done) {
                    if (P.directLighting) {                                          // :133-147
                        for (uint32_t lightIndex = 0; lightIndex < P.Pt.numSceneLights; ++lightIndex) {
                            incr_light light; const float* m; bool haveT;
                            load_light(P.Pt, (int)lightIndex, light, m, haveT);
                            const LightSample s = light_setup(light, m, haveT, wp, N);
                            if (s.needsShadow) {
                                bool lit = true;
                                if (light.type == INCR_LIGHT_DIRECTIONAL) {          // !PATH_TRACING: only the sun is shadowed
                                    ++rays;
                                    const bool finite = isfinite(wp.x) && isfinite(wp.y) && isfinite(wp.z) && isfinite(s.L.x) && isfinite(s.L.y) && isfinite(s.L.z);
                                    lit = !finite || !trace_any(P.Pt, st, wp, s.L, 0.1f, 1000.f);
                                }
                                if (lit) radiance = radiance + light_shade(s, p.basecolor, p.metallic, p.roughness, N, V);
                            }
                        }
                    }
                    float transmission = mat.transmissionFactor;                     // material/common.hlsli:83-88
                    if (mat.transmissionTextureIndex != -1) transmission *= SampleGrad(P.S.textures, P.S.srgbLut, mat.transmissionTextureIndex, UV, gr.ddx, gr.ddy, P.S.maxAnisotropy).x;
                    if (P.diffuseIbl && P.sh9) {                                     // :149-165
                        const f3 albedo = ((1.f - p.metallic) * (1.f - transmission)) * p.basecolor;
                        const f3 shDiffuse = (albedo * K_INV_PI) * ShIrradiance(N, P.sh9);
                        radiance = radiance + (shDiffuse + p.aoblend * (p.ao * shDiffuse - shDiffuse));
                    }
                    if (P.specularIbl && P.envmap && P.envBrdf) {                    // :167-185
                        const float NoV = hmax(dot(N, V), 0.01f);
                        const f3 F0 = p.metallic * p.basecolor + (1.f - p.metallic) * 0.04f;
                        const f3 R = normalize(reflect(-V, N));
                        const f3 prefilteredColor = cube_sample_trilinear(env, R, p.roughness * ((float)P.envmapMips - 1.f));
                        const f2 envBrdf = sample_rg16f_bilinear_clamp(P.envBrdf, P.envBrdfW, P.envBrdfH, {NoV, p.roughness});
                        const f3 specularIBL = prefilteredColor * (F0 * envBrdf.x + envBrdf.y);
                        radiance = radiance + (specularIBL + p.aoblend * (p.ao * specularIBL - specularIBL));
                    }
                    out = radiance;
                }
            }
        }
        P.color[i] = pack_r11g11b10(out, P.rounding);
    }
    count_rays(P.Pt, rays);
}

// ---------------------------------------------------------------------------------------------- decals
// gbuffer-raster.hlsl vs_main + primitive assembly: one wave per 64-triangle chunk of the decal draws; every triangle
// writes its (at most two) clockwise screen triangles into its own two slots, so slot order = draw order.
__global__ __launch_bounds__(256) void decal_setup_kernel(const DecalParams P) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const RasterPrims& R = P.prims;
    const float fW = (float)P.W, fH = (float)P.H;
    for (uint32_t chunk = blockIdx.x * 4 + wave; chunk < R.numChunks; chunk += gridDim.x * 4) {
        const uint32_t prim = R.chunkPrim[chunk];
        const PrimDraw d = R.draws[prim];
        const float* __restrict__ l2w = P.decalLocalToWorld + (size_t)d.meshIndex * 16;
        const bool cullBack = P.S.materials[d.materialIndex].doubleSided == 0;
        const uint32_t tri = R.chunkFirstTri[chunk] + lane;
        SubTriRec out[2];
        out[0].key = out[1].key = 0xffffffffu;
        if (tri < d.indexCount / 3u) {
            ClipVert cv[3];
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const float4 pos = R.pvPos[d.vertexBase + R.indices[d.indexBegin + tri * 3u + k]];
                const f4 c = mul_vec4(mul_vec4({pos.x, pos.y, pos.z, 1.f}, l2w), P.viewProj);
                cv[k] = {c.x, c.y, c.z, c.w};
            }
            bool in[3]; int nin = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { in[k] = (cv[k].w - cv[k].z) >= 0.f; nin += in[k]; }
            ClipVert poly[4]; int np = 0;
            if (nin == 3) { poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2]; np = 3; }
            else if (nin > 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int j = (k + 1) % 3;
                    if (in[k]) poly[np++] = cv[k];
                    if (in[k] != in[j]) poly[np++] = in[k] ? clip_lerp(cv[k], cv[j]) : clip_lerp(cv[j], cv[k]);
                }
            }
            ScreenVert sv[4]; bool ok = np >= 3;
            for (int k = 0; k < np; ++k) ok = ok && to_screen(poly[k], fW, fH, &sv[k]);
            if (ok) {
                for (int j = 0; j < (np == 4 ? 2 : 1); ++j) {
                    ScreenVert a = sv[0], b = sv[1 + j], c = sv[2 + j];
                    if (!orient_subtri(a, b, c, cullBack)) continue;
                    SubTriRec& r = out[j];
                    r.X0 = a.X; r.Y0 = a.Y; r.X1 = b.X; r.Y1 = b.Y; r.X2 = c.X; r.Y2 = c.Y; r.z0 = a.z; r.z1 = b.z; r.z2 = c.z;
                    r.key = (prim << 20) | tri; r.alphaTex = -1; r.pad = 0;
                }
            }
        }
        const size_t slot = ((size_t)chunk * 64 + lane) * 2;
        P.recs[slot] = out[0]; P.recs[slot + 1] = out[1];
    }
}
static constexpr uint32_t kDecalListCap = 1024;
// Per fragment: gbuffer-raster.hlsl ps_main + the blend states of gbuffer-raster-pass.inl:126-183
DI void decal_fragment(const DecalParams& P, const SubTriRec& r, uint32_t px, uint32_t py, float sceneDepth, uint32_t& g0, uint32_t& g1, uint32_t& g2, bool& dirty) {
    const long long Sx = (long long)px * 256 + 128, Sy = (long long)py * 256 + 128;
    const int32_t dx01 = r.X1 - r.X0, dy01 = r.Y1 - r.Y0, dx12 = r.X2 - r.X1, dy12 = r.Y2 - r.Y1, dx20 = r.X0 - r.X2, dy20 = r.Y0 - r.Y2;
    const long long E01 = (long long)dx01 * (Sy - r.Y0) - (long long)dy01 * (Sx - r.X0);
    const long long E12 = (long long)dx12 * (Sy - r.Y1) - (long long)dy12 * (Sx - r.X1);
    const long long E20 = (long long)dx20 * (Sy - r.Y2) - (long long)dy20 * (Sx - r.X2);
    if (E01 + (top_left(dx01, dy01) ? 0 : -1) < 0 || E12 + (top_left(dx12, dy12) ? 0 : -1) < 0 || E20 + (top_left(dx20, dy20) ? 0 : -1) < 0) return;
    const long long area = (long long)(r.X1 - r.X0) * (r.Y2 - r.Y0) - (long long)(r.X2 - r.X0) * (r.Y1 - r.Y0);
    const double invArea = 1.0 / (double)area;
    const float l1 = (float)((double)E20 * invArea), l2 = (float)((double)E01 * invArea);
    float depth = (r.z0 + l1 * (r.z1 - r.z0)) + l2 * (r.z2 - r.z0);
    if (!(depth >= 0.f)) return;
    if (depth > 1.f) depth = 1.f;
    if (!(depth >= sceneDepth)) return;                                              // GREATER_EQUAL, no depth write
    const PrimDraw d = P.prims.draws[r.key >> 20];
    const uint32_t tri = r.key & 0xfffffu;
    const float* __restrict__ l2w = P.decalLocalToWorld + (size_t)d.meshIndex * 16;
    f4 clip[3]; f2 uv[3]; f3 nW[3], tW[3], bW[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {                                               // vs_main :46-78
        const uint32_t vi = d.vertexBase + P.prims.indices[d.indexBegin + tri * 3u + k];
        const float4 pos = P.prims.pvPos[vi];
        clip[k] = mul_vec4(mul_vec4({pos.x, pos.y, pos.z, 1.f}, l2w), P.viewProj);
        const float2 t = P.prims.pvUv[vi]; uv[k] = {t.x, t.y};
        const float4 n4 = P.prims.pvNormal[vi], t4 = P.prims.pvTangent[vi];
        const f3 normal = {n4.x, n4.y, n4.z}, tangent = {t4.x, t4.y, t4.z};
        const f3 bitangent = cross(normal, tangent) * t4.w;
        const f4 a = mul_vec4({normal.x, normal.y, normal.z, 0.f}, l2w), b = mul_vec4({tangent.x, tangent.y, tangent.z, 0.f}, l2w);
        const f4 c = mul_vec4({bitangent.x, bitangent.y, bitangent.z, 0.f}, l2w);
        nW[k] = {a.x, a.y, a.z}; tW[k] = {b.x, b.y, b.z}; bW[k] = {c.x, c.y, c.z};
    }
    const BarySetup bs = bary_setup(clip[0], clip[1], clip[2]);
    const f3 lam = bary_eval(bs, pixel_ndc(px, py, (float)P.W, (float)P.H));
    const incr_material mat = P.S.materials[d.materialIndex];
    const UvGrad gr = quad_uv_gradient(bs, uv[0], uv[1], uv[2], px, py, (float)P.W, (float)P.H);
    const MatProps p = EvaluateMaterialPropertiesGrad(mat, P.S.textures, P.S.srgbLut, interp2(uv[0], uv[1], uv[2], lam), gr.ddx, gr.ddy, P.S.maxAnisotropy, P.S.materialTexels ? P.S.materialTexels[d.materialIndex].chain : nullptr);
    const f3 T = normalize(interp3(tW[0], tW[1], tW[2], lam)), B = normalize(interp3(bW[0], bW[1], bW[2], lam));
    f3 N = normalize(interp3(nW[0], nW[1], nW[2], lam));
    if (p.bHasNormalmap) {
        const f3 nm = p.normalmap;
        N = normalize(mk3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
    }
    const f4 dst = unpack_unorm8x4(g0);
    const float a = p.opacity, ia = 1.f - a;
    g0 = pack_unorm8x4({p.basecolor.x * a + dst.x * ia, p.basecolor.y * a + dst.y * ia, p.basecolor.z * a + dst.z * ia, a * a + dst.w * ia});
    if (mat.normalTextureIndex != -1) {
        const f2 oct = OctEncode(N);
        g1 = f32_to_f16_bits(oct.x, P.rounding) | (f32_to_f16_bits(oct.y, P.rounding) << 16);
    }
    if (mat.metallicRoughnessTextureIndex != -1) g2 = pack_unorm8x4({p.metallic, p.roughness, p.ao, p.aoblend});
    dirty = true;
}
// One 64x4 pixel tile per workgroup: the slots are culled against the tile in order (ballot prefix + LDS), then every pixel
// walks the surviving list front to back -- sequential per pixel, so the blend order is the slot order.
__global__ __launch_bounds__(256) void decal_blend_kernel(const DecalParams P) {
    __shared__ uint32_t s_list[kDecalListCap];
    __shared__ uint32_t s_waveCnt[4];
    __shared__ uint32_t s_count;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t tx0 = blockIdx.x * 64, ty0 = P.y0 + blockIdx.y * 4;
    const uint32_t px = tx0 + lane, py = ty0 + wave;
    const bool inside = px < P.W && py < P.y1;
    const size_t i = (size_t)py * P.W + px;
    uint32_t g0 = 0, g1 = 0, g2 = 0; float sceneDepth = 0.f; bool dirty = false;
    if (inside) { g0 = P.g0[i]; g1 = P.g1[i]; g2 = P.g2[i]; sceneDepth = P.depth[i]; }
    const int32_t tileX0 = (int32_t)tx0, tileX1 = (int32_t)min(tx0 + 63u, P.W - 1u), tileY0 = (int32_t)ty0, tileY1 = (int32_t)min(ty0 + 3u, P.y1 - 1u);
    if (tid == 0) s_count = 0;
    __syncthreads();
    auto flush = [&]() {
        const uint32_t n = s_count;
        if (inside) {
            for (uint32_t k = 0; k < n; ++k) {
                const SubTriRec r = P.recs[s_list[k]];
                const int32_t minX = min(r.X0, min(r.X1, r.X2)), maxX = max(r.X0, max(r.X1, r.X2));
                const int32_t minY = min(r.Y0, min(r.Y1, r.Y2)), maxY = max(r.Y0, max(r.Y1, r.Y2));
                if ((int32_t)px < ((minX + 127) >> 8) || (int32_t)px > ((maxX - 128) >> 8) || (int32_t)py < ((minY + 127) >> 8) || (int32_t)py > ((maxY - 128) >> 8)) continue;
                decal_fragment(P, r, px, py, sceneDepth, g0, g1, g2, dirty);
            }
        }
        __syncthreads();
        if (tid == 0) s_count = 0;
        __syncthreads();
    };
    for (uint32_t base = 0; base < P.numSlots; base += 256) {
        if (s_count + 256u > P.listCap) flush();
        const uint32_t slot = base + tid;
        bool hit = false;
        if (slot < P.numSlots) {
            const SubTriRec r = P.recs[slot];
            if (r.key != 0xffffffffu) {
                const int32_t minX = min(r.X0, min(r.X1, r.X2)), maxX = max(r.X0, max(r.X1, r.X2));
                const int32_t minY = min(r.Y0, min(r.Y1, r.Y2)), maxY = max(r.Y0, max(r.Y1, r.Y2));
                const int32_t bx0 = (minX + 127) >> 8, bx1 = (maxX - 128) >> 8, by0 = (minY + 127) >> 8, by1 = (maxY - 128) >> 8;
                hit = bx0 <= tileX1 && bx1 >= tileX0 && by0 <= tileY1 && by1 >= tileY0 && bx0 <= bx1 && by0 <= by1;
            }
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_waveCnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        if (hit) {
            uint32_t off = s_count;
            for (uint32_t w = 0; w < wave; ++w) off += s_waveCnt[w];
            off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            s_list[off] = slot;
        }
        __syncthreads();
        if (tid == 0) s_count += s_waveCnt[0] + s_waveCnt[1] + s_waveCnt[2] + s_waveCnt[3];
        __syncthreads();
    }
    flush();
    if (inside && dirty) { P.g0[i] = g0; P.g1[i] = g1; P.g2[i] = g2; }
}

// clustered-lighting.hlsl:35-115 + cluster-culling.hlsli:26-41
DI f3 clustered_radiance(const ClusteredParams& P, const GBufferPixel& g, uint32_t x, uint32_t y, size_t i) {
    f2 sp = screen_pos(x, y, P.L.W, P.L.H); sp.y = -sp.y;
    const float depth = P.L.depth[i];
    const f4 wp = mul_vec4({sp.x, sp.y, depth, 1.f}, P.L.invViewProj);
    const f3 worldPos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
    const f4 vp = mul_vec4({sp.x, sp.y, depth, 1.f}, P.L.invProj);
    const float viewZ = vp.z / vp.w;
    const float gridResX = (float)P.L.W / (float)P.dim[0], gridResY = (float)P.L.H / (float)P.dim[1];
    const uint32_t cx = ftou((float)x / gridResX), cy = ftou((float)y / gridResY);
    const uint32_t cz = ftou(floorf(logf(viewZ) * P.scale + P.bias));
    const uint32_t clusterId = ftou(((float)(P.dim[0] * P.dim[1]) * (float)cz + (float)P.dim[0] * (float)cy) + (float)cx);
    const uint32_t numClusters = P.dim[0] * P.dim[1] * P.dim[2];
    const f3 eye = {P.L.eye[0], P.L.eye[1], P.L.eye[2]};
    const f3 V = normalize(eye - worldPos);
    f3 radiance = {0.f, 0.f, 0.f};
    incr_light_grid_data ci = {0u, 0u};
    if (clusterId < numClusters) ci = P.grid[clusterId];
    for (uint32_t k = 0; k < ci.count; ++k) {
        const uint32_t lightIndex = P.lists[ci.offset + k];
        incr_light light; const float* m; bool haveT;
        load_global_light(P.S, lightIndex, light, m, haveT);
        if (light.range == 0) light.range = 4.f;                            // MAX_LIGHT_RANGE
        const LightSample s = light_setup(light, m, haveT, worldPos, g.N);
        // clustered lights are punctual: GetDirectRadiance traces shadows for directional lights only outside PATH_TRACING
        if (s.needsShadow) radiance = radiance + light_shade(s, g.basecolor, g.metallic, g.roughness, g.N, V);
        else radiance = radiance + mk3(0.f);
    }
    return radiance;
}
__global__ __launch_bounds__(256) void clustered_lighting_kernel(const ClusteredParams P) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.L.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.L.W || y >= P.L.y1) return;
    const size_t i = (size_t)y * P.L.W + x;
    if (P.requiresClear) P.L.color[i] = 0;
    add_color(P.L, i, clustered_radiance(P, read_gbuffer(P.L, i), x, y, i));
}

// Fused deferred lighting: sky + sun + clustered lights in ONE pass over the pixel -- the three G-buffer words and depth are
// read once and the colour target is read and written once, where the unfused chain (the reference's three blended draws /
// dispatches, sky-lighting.inl, direct-lighting.inl, clustered-lighting.inl) moves 24 B per pixel per pass.  The colour is
// re-quantised to R11G11B10F between the parts exactly where the chain stores it, so the result is bit-identical to running
// the three kernels one after the other (tests/test_gpu_raster.py::test_fused_deferred_lighting_equals_the_chain).
#ifndef INCR_LIGHTING_MIN_BLOCKS
#define INCR_LIGHTING_MIN_BLOCKS 8      // 64 VGPRs (28 B of scratch) for 8 waves per SIMD instead of 78 for 6: fused lighting 0.667-0.694 -> 0.638-0.649 ms (tools/ab_occupancy.sh)
#endif
__global__ __launch_bounds__(kBlock, INCR_LIGHTING_MIN_BLOCKS) void deferred_lighting_kernel(const FusedLightingParams P) {
    __shared__ uint32_t s_stack[(kBlock / 64) * kPacketStack];
    const LightingParams& L = P.haveSky ? P.sky.L : (P.haveDirect ? P.direct.L : P.clustered.L);
    uint32_t x, y; tile_pixel(L.y0, &x, &y);
    const bool valid = x < L.W && y < L.y1;
    const size_t i = valid ? (size_t)y * L.W + x : 0;
    uint32_t rays = 0;
    GBufferPixel g = {};
    uint32_t c = 0;
    if (valid) {
        g = read_gbuffer(L, i);
        c = L.color[i];
        if (P.haveSky) c = pack_r11g11b10(unpack_r11g11b10(c) + sky_radiance(P.sky, g, x, y, i), L.rounding);
    }
    if (P.haveDirect) {          // uniform: the shadow query is made by the whole wave (trace_any_packet)
        DirectSetup ds = {};
        if (valid) ds = direct_setup(P.direct, g, x, y, i, rays);
        if (P.direct.S.cover.header && ds.wantsRay && sun_cover_clear(P.direct.S.cover, ds.worldPos, 0.1f)) ds.wantsRay = false;     // provably unoccluded: `occluded` stays false
#ifdef INCR_DIAG_NO_SUN_WALK     // diagnostic only: the pass without the sun's shadow walk (everything lit)
        const bool occluded = false;
#else
        const bool occluded = trace_any_packet(P.direct.S, s_stack + (threadIdx.x >> 6) * kPacketStack, ds.worldPos, ds.s.L, 0.1f, 1000.f, valid && ds.wantsRay);
#endif
        if (valid) c = pack_r11g11b10(unpack_r11g11b10(c) + direct_finish(ds, g, occluded), L.rounding);
    }
    if (valid) {
        if (P.haveClustered) {
            if (P.clustered.requiresClear) c = 0;
            c = pack_r11g11b10(unpack_r11g11b10(c) + clustered_radiance(P.clustered, g, x, y, i), L.rounding);
        }
        L.color[i] = c;
    }
    if (P.haveDirect) count_rays(P.direct.S, rays);
}

// ---------------------------------------------------------------------------------------------- dynamic sky background
// dynamic-sky.hlsl: a full-screen triangle at z = 0.0001 under GREATER_EQUAL (dynamic-sky.inl:100-102) overwrites every
// pixel whose depth is <= 0.0001 with 1000 x Preetham along the direction the SCREEN uv maps to in lat-long
// parameterisation (the reference shader uses the interpolated uv, not the view ray).
__global__ __launch_bounds__(256) void dynamic_sky_kernel(const DynamicSkyParams P) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.y1) return;
    const size_t i = (size_t)y * P.W + x;
    if (!(0.0001f >= P.depth[i])) return;
    const f2 uv = {((float)x + 0.5f) / (float)P.W, ((float)y + 0.5f) / (float)P.H};
    const float theta = K_PI * uv.y, phi = K_PI * (uv.x * 2.f - 1.f);                 // LatlongUV2Polar (common/math.hlsli:46-52)
    const float sint = sinf(theta), cost = cosf(theta), sinp = sinf(phi), cosp = cosf(phi);
    const f3 dir = {sint * sinp, cost, sint * cosp};                                  // Polar2Cartesian(..., World) = p.xzy
    const f3 radiance = 1000.f * CalculateSkyRadianceRGB(P.sun, normalize(dir), P.perez);
    P.color[i] = pack_r11g11b10(radiance, P.rounding);
}

// envmap.hlsl:22-54 (EnvSkyMode::HDRI background): sky_brightness * cube along the translation-free view ray
__global__ __launch_bounds__(256) void environmentmap_kernel(const EnvmapParams P) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.y1) return;
    const size_t i = (size_t)y * P.W + x;
    if (!(0.0001f >= P.depth[i])) return;
    const f2 ndc = pixel_ndc(x, y, (float)P.W, (float)P.H);
    const f4 wp = mul_vec4({ndc.x, ndc.y, 0.0001f, 1.f}, P.invParallaxViewProj);
    const f3 dir = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
    const f3 luminance = P.skyBrightness * cube_sample_level(CubeChain{P.envmapRes, P.envmapMips, P.envmap}, dir, 0u);
    P.color[i] = pack_r11g11b10(luminance, P.rounding);
}

// ---------------------------------------------------------------------------------------------- debug views
// debug-visualization.hlsl:52-203: overwrites the HDR colour target, no depth test, no blending
DI f3 hsv2rgb(f3 hsv) {                                                       // common/color-space.hlsli:2-48
    f3 sat;
    while (hsv.x < 0) hsv.x += 360;
    while (hsv.x > 360) hsv.x -= 360;
    if (hsv.x < 120) { sat.x = (120 - hsv.x) / 60.0f; sat.y = hsv.x / 60.0f; sat.z = 0; }
    else if (hsv.x < 240) { sat.x = 0; sat.y = (240 - hsv.x) / 60.0f; sat.z = (hsv.x - 120) / 60.0f; }
    else { sat.x = (hsv.x - 240) / 60.0f; sat.y = 0; sat.z = (360 - hsv.x) / 60.0f; }
    sat.x = hmin(sat.x, 1.f); sat.y = hmin(sat.y, 1.f); sat.z = hmin(sat.z, 1.f);
    return {((1 - hsv.y) + hsv.y * sat.x) * hsv.z, ((1 - hsv.y) + hsv.y * sat.y) * hsv.z, ((1 - hsv.y) + hsv.y * sat.z) * hsv.z};
}
DI float debug_halton(uint32_t sampleIndex, uint32_t base) {                  // common/uniform-sampling.hlsli:123-136
    float result = 0.f, f = 1.f;
    while (sampleIndex > 0) { f = f / (float)base; result += f * (float)(sampleIndex % base); sampleIndex = sampleIndex / base; }
    return result;
}
__global__ __launch_bounds__(256) void debug_viz_kernel(const DebugVizParams P) {
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.y1) return;
    const size_t i = (size_t)y * P.W + x;
    f3 c = {0.f, 0.f, 0.f};
    switch (P.viewmode) {
    case 2: { const float g = unpack_unorm8x4(P.g2[i]).y; c = {g, g, g}; break; }
    case 3: { const float r = unpack_unorm8x4(P.g2[i]).x; c = {r, r, r}; break; }
    case 4: { const f4 b = unpack_unorm8x4(P.g0[i]); c = {b.x, b.y, b.z}; break; }
    case 8: case 9: {
        const uint32_t v = P.vis[i];
        const uint32_t objectId = P.usePrims ? v >> 20 : v >> 7, triangleId = P.usePrims ? v & 0xfffffu : v & 127u;
        c = P.viewmode == 8 ? hsv2rgb({debug_halton(objectId, 3) * 360.f, 0.85f, 0.95f}) : hsv2rgb({debug_halton(triangleId, 5) * 360.f, 0.85f, 0.95f});
        break;
    }
    case 10: case 13: {
        const uint32_t n = P.viewmode == 10 ? P.g1[i] : (P.bent ? P.bent[i] : 0x3c003c00u);
        const f3 N = OctDecode({f16_bits_to_f32(n & 0xffffu), f16_bits_to_f32(n >> 16)});
        c = {N.x * 0.5f + 0.5f, N.y * 0.5f + 0.5f, N.z * 0.5f + 0.5f};
        break;
    }
    case 11: {
        const f2 uv = {((float)x + 0.5f) / (float)P.W, ((float)y + 0.5f) / (float)P.H};
        const f4 pv = mul_vec4({2.f * uv.x - 1.f, -2.f * uv.y + 1.f, hmax(0.001f, P.depth[i]), 1.f}, P.invProj);
        const float z = pv.z / pv.w;
        const uint32_t sliceIndex = ftou(floorf(P.sliceScale * logf(z) + P.sliceBias));
        c = hsv2rgb({(float)sliceIndex / P.numSlices * 360.f, 0.85f, 0.95f});
        break;
    }
    case 12: { const float a = P.ao ? (float)P.ao[i] / 255.f : 1.f; c = {a, a, a}; break; }
    default: break;
    }
    P.color[i] = pack_r11g11b10(c, P.rounding);
}

// ---------------------------------------------------------------------------------------------- TAA resolve
DI f3 load_hdr(const uint32_t* __restrict__ img, uint32_t W, uint32_t H, uint32_t x, uint32_t y) {
    if (x >= W || y >= H) return {0.f, 0.f, 0.f};
    return unpack_r11g11b10(img[(size_t)y * W + x]);
}
DI int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }
DI f3 sample_bilinear_clamp(const uint32_t* __restrict__ img, uint32_t W, uint32_t H, f2 uv) {
    float u = uv.x, v = uv.y;
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    // the coordinate clamped to one texel beyond the frame on either side, in fixed point (pt_device.h linear_address), the texel indices clamped to the frame
    u = fminf(fmaxf(u, -2.f), 3.f); v = fminf(fmaxf(v, -2.f), 3.f);
    int xi = linear_fixed(u * (float)W), yi = linear_fixed(v * (float)H), xf, yf; float fx, fy;
    xi = min(max(xi, -256), (int)W * 256); yi = min(max(yi, -256), (int)H * 256);
    linear_split(xi, &xf, &fx); linear_split(yi, &yf, &fy);
    const int x0 = clampi(xf, (int)W), x1 = clampi(xf + 1, (int)W), y0 = clampi(yf, (int)H), y1 = clampi(yf + 1, (int)H);
    const f3 t00 = load_hdr(img, W, H, x0, y0), t10 = load_hdr(img, W, H, x1, y0), t01 = load_hdr(img, W, H, x0, y1), t11 = load_hdr(img, W, H, x1, y1);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}
DI float lum_weight(f3 c) { return 1.f / (1.f + ((c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f)); }
// taa-resolve.hlsl:34-117
__global__ __launch_bounds__(256) void taa_resolve_kernel(const TaaParams P) {
    __shared__ float s_steps[kSrgbSteps];
    if (P.ldr) {                                                  // kernel-uniform
        for (uint32_t i = threadIdx.x; i < kSrgbSteps; i += 256u) s_steps[i] = P.srgbSteps[i];
        __syncthreads();
    }
    const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63u), y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.W || y >= P.y1) return;
    const size_t i = (size_t)y * P.W + x;
    const f3 cur = unpack_r11g11b10(P.hdr[i]);
    if (P.historyIndex == 0) {
        const uint32_t o = pack_r11g11b10(cur, P.rounding);
        P.histOut[i] = o;
        if (P.ldr) P.ldr[i] = tonemap_texel(s_steps, o, Exposure(P.ev100), P.viewmode);
        return;
    }
    const f2 uv = {((float)x + 0.5f) / (float)P.W, ((float)y + 0.5f) / (float)P.H};
    const f4 clip = {2.f * uv.x - 1.f, -2.f * uv.y + 1.f, P.depth[i], 1.f};
    const f4 world = mul_vec4(clip, P.invViewProj);
    const f4 pc = mul_vec4(world, P.prevViewProj);
    const f2 ruv = {0.5f * (pc.x / pc.w) + 0.5f, -0.5f * (pc.y / pc.w) + 0.5f};
    const f3 previous = sample_bilinear_clamp(P.histIn, P.W, P.H, ruv);
    f3 mn = mk3(10000.f), mx = mk3(-10000.f);
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const f3 c = load_hdr(P.hdr, P.W, P.H, ftou((float)x + (float)dx), ftou((float)y + (float)dy));
            mn = {hmin(mn.x, c.x), hmin(mn.y, c.y), hmin(mn.z, c.z)};
            mx = {hmax(mx.x, c.x), hmax(mx.y, c.y), hmax(mx.z, c.z)};
        }
    const f3 clamped = {hmin(hmax(previous.x, mn.x), mx.x), hmin(hmax(previous.y, mn.y), mx.y), hmin(hmax(previous.z, mn.z), mx.z)};
    const float cw = 0.1f * lum_weight(cur), pw = 0.9f * lum_weight(clamped);
    f3 out = cur * cw + clamped * pw;
    out = out / (cw + pw);
    const uint32_t o = pack_r11g11b10(out, P.rounding);
    P.histOut[i] = o;
    // TonemapPass fused into the resolve when the caller presents this surface: the tonemap reads exactly the stored texel
    if (P.ldr) P.ldr[i] = tonemap_texel(s_steps, o, Exposure(P.ev100), P.viewmode);
}

} // namespace incr

// ---- launchers (called from raster_api.hip) ------------------------------------------------------------
using namespace incr;
static inline dim3 pixel_grid(uint32_t W, uint32_t y0, uint32_t y1) { return dim3((W + 63) / 64, (y1 - y0 + 3) / 4); }
static inline dim3 tile_grid(uint32_t W, uint32_t y0, uint32_t y1) { return dim3((W + 31) / 32, (y1 - y0 + 7) / 8); }      // tile_pixel: 4 tiles of 8 x 8 per workgroup

extern "C" hipError_t incr_launch_mesh_frames(const RasterScene* S, const ViewFrame* vf, MeshFrame* out, hipStream_t stream) {
    if (!S->numMeshes) return hipSuccess;
    hipLaunchKernelGGL(mesh_frame_kernel, dim3((S->numMeshes + 63) / 64), dim3(64), 0, stream, *S, *vf, out);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_batch_cull(const RasterScene* S, const MeshFrame* frames, int frustumCulling, incr_draw_cmd* defList, incr_draw_cmd* dsList,
                                             uint32_t* counts, incr_render_stats* stats, const PrimDraw* prims, uint32_t numPrims, hipStream_t stream) {
    hipLaunchKernelGGL(batch_cull_kernel, dim3(1), dim3(1024), 0, stream, *S, frames, frustumCulling, defList, dsList, counts, stats, prims, numPrims);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_light_cull(const LightCullParams* p, hipStream_t stream) {
    const uint32_t n = p->dim[0] * p->dim[1] * p->dim[2];
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(light_cull_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_visibility(const VisParams* p, uint32_t maxJobs, int numCUs, uint32_t* vis, float* depth, hipStream_t stream) {
    if (p->y1 <= p->y0 || !p->W) return hipSuccess;
    const size_t n = (size_t)p->W * (p->y1 - p->y0), first = (size_t)p->W * p->y0;     // only the scissor rows are cleared, drawn and resolved
    hipError_t e = hipMemsetAsync(p->packed + first, 0, n * 8, stream);
    if (e != hipSuccess) return e;
    const bool bins = maxJobs && p->binCap;
    // the queue counters and, right behind them, the bins' triangle counts
    if ((e = hipMemsetAsync(p->vqCounters, 0, bins ? (size_t)((const uint8_t*)(p->binCount + (size_t)p->binsX * p->binsY * kVisBinLists) - (const uint8_t*)p->vqCounters) : kVisCounterBytes, stream)) != hipSuccess) return e;
    if (maxJobs) {
        const uint32_t blocks = std::min<uint32_t>((maxJobs + kVisWaves - 1) / kVisWaves, (uint32_t)numCUs * 8u);
        if (p->prims.draws) {            // primitive path: draw lists -> flags, then one wave per 64-triangle chunk
            if (p->forward) {
                hipLaunchKernelGGL(forward_flags_kernel, dim3((p->prims.numPrims + 255u) / 256u), dim3(256), 0, stream, p->S, p->prims.draws, p->prims.numPrims, p->prims.primFlags);
            } else {
                if ((e = hipMemsetAsync(p->prims.primFlags, 0, (size_t)p->prims.numPrims * 4, stream)) != hipSuccess) return e;
                hipLaunchKernelGGL(prim_flags_kernel, dim3(std::max(1u, std::min((p->prims.numPrims + 255u) / 256u, 256u))), dim3(256), 0, stream,
                                   p->lists[0], p->lists[1], p->counts, p->prims.primFlags);
            }
            hipLaunchKernelGGL(vis_setup_prim_kernel, dim3(blocks), dim3(256), 0, stream, *p);
        } else {
            hipLaunchKernelGGL(vis_setup_kernel, dim3(blocks), dim3(kVisWaves * 64), 0, stream, *p);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // every workgroup resident (4 per CU: registers and LDS): the bins are the grid's first round, the wave-level queues what follows
        if (p->binShift == 6u) hipLaunchKernelGGL(vis_scan_kernel<6u>, dim3((uint32_t)numCUs * 4u), dim3(256), 0, stream, *p);
        else hipLaunchKernelGGL(vis_scan_kernel<5u>, dim3((uint32_t)numCUs * 4u), dim3(256), 0, stream, *p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(vis_resolve_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, p->packed + first, vis + first, depth + first, n);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_gbuffer(const GBufferParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(gbuffer_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_sky_lighting(const SkyParams* p, hipStream_t stream) {
    if (!p->L.W || p->L.y1 <= p->L.y0) return hipSuccess;
    hipLaunchKernelGGL(sky_lighting_kernel, pixel_grid(p->L.W, p->L.y0, p->L.y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" size_t incr_sun_cover_bytes() { return 256 + (size_t)kSunCoverG * kSunCoverG * 4 + (size_t)kSunLargeCap * sizeof(uint2); }      // header, map, the list of large footprints
extern "C" hipError_t incr_launch_sun_cover(const PtParams* S, uint32_t sunIndex, SunCoverHeader* header, uint32_t* grid, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(grid, 0, (size_t)kSunCoverG * kSunCoverG * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sun_cover_header_kernel, dim3(1), dim3(1), 0, stream, *S, sunIndex, header);
    if (S->numTris) {
        uint2* large = (uint2*)(grid + (size_t)kSunCoverG * kSunCoverG);          // behind the map (incr_sun_cover_bytes)
        hipLaunchKernelGGL(sun_cover_raster_kernel, dim3((S->numTris + 255u) / 256u), dim3(256), 0, stream, S->tris, S->numTris, header, grid, large);
        hipLaunchKernelGGL(sun_cover_large_kernel, dim3(1024), dim3(256), 0, stream, S->tris, header, grid, large);
    }
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_direct_lighting(const DirectParams* p, hipStream_t stream) {
    if (!p->L.W || p->L.y1 <= p->L.y0) return hipSuccess;
    hipLaunchKernelGGL(direct_lighting_kernel, tile_grid(p->L.W, p->L.y0, p->L.y1), dim3(kBlock), 0, stream, *p);
    if (p->S.stats && p->S.statStripes) hipLaunchKernelGGL(fold_ray_stats_kernel, dim3(1), dim3(64), 0, stream, p->S.statStripes, p->S.stats);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_hbao(const HbaoParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(hbao_kernel, tile_grid(p->W, p->y0, p->y1), dim3(kBlock), 0, stream, *p);
    if (p->S.stats && p->S.statStripes) hipLaunchKernelGGL(fold_ray_stats_kernel, dim3(1), dim3(64), 0, stream, p->S.statStripes, p->S.stats);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_decals(const DecalParams* p, int numCUs, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0 || !p->prims.numChunks) return hipSuccess;
    hipLaunchKernelGGL(decal_setup_kernel, dim3(std::min<uint32_t>((p->prims.numChunks + 3u) / 4u, (uint32_t)numCUs * 8u)), dim3(256), 0, stream, *p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(decal_blend_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_forward_shade(const ForwardParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(forward_shade_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(kBlock), 0, stream, *p);
    if (p->Pt.stats && p->Pt.statStripes) hipLaunchKernelGGL(fold_ray_stats_kernel, dim3(1), dim3(64), 0, stream, p->Pt.statStripes, p->Pt.stats);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_clustered_lighting(const ClusteredParams* p, hipStream_t stream) {
    if (!p->L.W || p->L.y1 <= p->L.y0) return hipSuccess;
    hipLaunchKernelGGL(clustered_lighting_kernel, pixel_grid(p->L.W, p->L.y0, p->L.y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_deferred_lighting(const FusedLightingParams* p, hipStream_t stream) {
    const LightingParams& L = p->haveSky ? p->sky.L : (p->haveDirect ? p->direct.L : p->clustered.L);
    if (!L.W || L.y1 <= L.y0) return hipSuccess;
    hipLaunchKernelGGL(deferred_lighting_kernel, tile_grid(L.W, L.y0, L.y1), dim3(kBlock), 0, stream, *p);
    if (p->haveDirect && p->direct.S.stats && p->direct.S.statStripes) (void)incr_launch_fold_ray_stats(p->direct.S.statStripes, p->direct.S.stats, stream);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_dynamic_sky(const DynamicSkyParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(dynamic_sky_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_environmentmap(const EnvmapParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(environmentmap_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_debug_viz(const DebugVizParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(debug_viz_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_taa_resolve(const TaaParams* p, hipStream_t stream) {
    if (!p->W || p->y1 <= p->y0) return hipSuccess;
    hipLaunchKernelGGL(taa_resolve_kernel, pixel_grid(p->W, p->y0, p->y1), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
