// TEST INFRASTRUCTURE ONLY -- CPU oracle for the raster half of the Incarnation hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
// PARITY UNPINNED (see oracle_math.h): restatement of the HLSL / host code cited at each function; the reference ships
// no golden vectors.  Included at the end of oracle_pt.cpp (shares its scene access, texture unit and BRDF code).
//
// Fixed-function stages the reference leaves to D3D12 hardware are PINNED here (DESIGN.md "Rasterisation rules"):
//   * clip only against the near plane (z <= w; the reference's reverse-Z infinite projection has no far plane),
//     new vertices a + t (b - a) with a the inside vertex, t = d_a / (d_a - d_b), d = w - z;
//   * viewport: sx = (x/w * 0.5 + 0.5) * W, sy = (0.5 - y/w * 0.5) * H, snapped to 24.8 fixed point (round half up),
//     clamped to +-2^20 pixels; sample point = pixel centre; D3D top-left fill rule on exact integer edge functions;
//   * front face = counter-clockwise on screen (visibility-pass.inl FrontCounterClockwise = TRUE);
//   * depth = z0 + l1 (z1 - z0) + l2 (z2 - z0) with l = (float)((double)E * (1.0 / (double)area)), clamped to <= 1;
//   * depth test GREATER_EQUAL in draw order: default list, then double-sided list, each in ascending meshlet id,
//     triangles in ascending index -> among equal depths the LAST drawn fragment wins;
//   * pixel-shader interpolants (uv) use the same screen-space barycentric formula as the G-buffer pass;
//   * textures: mip 0 bilinear (the raster path's anisotropic/trilinear LOD selection is not reproduced);
//   * rcp(x) = 1 / x, UNORM stores round to nearest, float stores as incr_*_desc::storage_rounding says.
#pragma once

// ------------------------------------------------------------------------------------------------ view constants
static float RasterHalton(uint64_t sampleIndex, uint32_t base) {             // renderer.cpp:53-66
    float result = 0.f, f = 1.f;
    while (sampleIndex > 0) { f = f / (float)base; result += f * (float)(sampleIndex % base); sampleIndex = sampleIndex / base; }
    return result;
}
static Mat4 mat_translation(float x, float y) { Mat4 t = mat_identity(); t.m[12] = x; t.m[13] = y; return t; }
static void put(incr_matrix* d, const Mat4& m) { memcpy(d->m, m.m, 64); }

static void make_view_constants(const incr_view* view, const incr_view* cull, const float jitter[2], const incr_matrix* prev, float nearPlane,
                                uint32_t resX, uint32_t resY, incr_view_constants* o) {    // renderer.cpp:1638-1665
    if (!cull) cull = view;
    const float jx = jitter ? jitter[0] : 0.f, jy = jitter ? jitter[1] : 0.f;
    const Mat4 V = to_mat4(view->view_transform), P = to_mat4(view->projection_transform);
    const Mat4 Vc = to_mat4(cull->view_transform), Pc = to_mat4(cull->projection_transform);
    const Mat4 T = mat_translation(jx, jy);
    const Mat4 PT = mat_mul(P, T), VPT = mat_mul(V, PT), VP = mat_mul(V, P);
    memset(o, 0, sizeof *o);
    put(&o->view_transform, V); put(&o->proj_transform, PT); put(&o->view_proj_transform, VPT);
    put(&o->inv_view_proj_transform, mat_inverse(VPT)); put(&o->inv_proj_transform, mat_inverse(PT));
    put(&o->cull_view_proj_transform, mat_mul(Vc, PT));
    put(&o->prev_view_proj_transform, prev ? to_mat4(*prev) : mat_identity());
    put(&o->lightcull_proj_transform, mat_mul(Pc, T));                                     // light-culling.inl:111
    put(&o->lightcull_inv_view_proj, mat_inverse(mat_mul(mat_mul(Vc, Pc), T)));            // light-culling.inl:112
    put(&o->sky_inv_view_proj, mat_inverse(mat_mul(VP, T)));                               // sky-lighting.inl:124
    put(&o->unjittered_view_proj, VP); put(&o->taa_inv_view_proj, mat_inverse(VP));       // renderer.cpp:2007-2017
    for (int i = 0; i < 3; ++i) { o->eye_pos[i] = view->position[i]; o->camera_up[i] = view->up[i]; }      // renderer.cpp:1659-1660
    Mat4 parallaxView = V; parallaxView.m[12] = parallaxView.m[13] = parallaxView.m[14] = 0.f;             // environmentmap.inl:127-133
    put(&o->envmap_inv_parallax_view_proj, mat_inverse(mat_mul(parallaxView, P)));
    o->near_plane = nearPlane; o->res_x = resX; o->res_y = resY; o->jitter[0] = jx; o->jitter[1] = jy;
}

// ------------------------------------------------------------------------------------------------ helpers
static inline float dot4p(float3 c, const float p[4]) { return ((c.x * p[0] + c.y * p[1]) + c.z * p[2]) + 1.f * p[3]; }   // dot(float4(c,1), plane)
static inline float len3(const float p[4]) { return sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]); }
static inline Mat4 mesh_local_to_world(const Scene& s, int mesh) {           // mul(meshTransform, sceneRotation)
    return mat_mul(to_mat4(s.d.mesh_transforms[mesh]), to_mat4(s.d.root_transform));
}
// rows [y0, y1) a pass touches (incr_view_constants::scissor_y0/y1: sort-first multi-GPU row bands; 0,0 = whole frame)
static inline void scissor_rows(const incr_view_constants& v, uint32_t* y0, uint32_t* y1) {
    const bool whole = v.scissor_y0 == 0 && v.scissor_y1 == 0;
    *y0 = whole ? 0u : std::min(v.scissor_y0, v.res_y);
    *y1 = whole ? v.res_y : std::min(v.scissor_y1, v.res_y);
    if (*y1 < *y0) *y1 = *y0;
}
// D3D ftou: NaN -> 0, negative -> 0, saturating
static inline uint32_t ftou(float f) { if (!(f > 0.f)) return 0u; if (f >= 4294967296.f) return 0xffffffffu; return (uint32_t)f; }

// ------------------------------------------------------------------------------------------------ batch culling
// shaders/culling/batch-culling.hlsl:22-46 FrustumCull
static bool BatchFrustumCull(const float sphere[4], const Mat4& localToWorld, const Mat4& cullViewProj) {
    const Mat4 Mt = mat_mul(localToWorld, cullViewProj);      // M = transpose(Mt): M[r] is column r of Mt
    float n[4], l[4], r[4], b[4], t[4];
    for (int k = 0; k < 4; ++k) {
        const float c0 = Mt.m[k * 4 + 0], c1 = Mt.m[k * 4 + 1], c2 = Mt.m[k * 4 + 2], c3 = Mt.m[k * 4 + 3];
        n[k] = c3 - c2; l[k] = c3 + c0; r[k] = c3 - c0; b[k] = c3 + c1; t[k] = c3 - c1;
    }
    const float3 c = {sphere[0], sphere[1], sphere[2]}; const float rad = sphere[3];
    return (dot4p(c, n) + rad * len3(n) >= 0) && (dot4p(c, l) + rad * len3(l) >= 0) && (dot4p(c, r) + rad * len3(r) >= 0)
        && (dot4p(c, b) + rad * len3(b) >= 0) && (dot4p(c, t) + rad * len3(t) >= 0);
}
// cs_meshlet_cull_main :119-193 (lists in ascending meshlet id: one of the orders the atomic append can produce)
// and cs_primitive_cull_main :53-117 (usePrims)
static void batch_cull(const Scene& s, const incr_view_constants& vc, int frustumCulling, incr_draw_cmd* defList, incr_draw_cmd* dsList,
                       uint32_t counts[2], int32_t* culled, bool usePrims = false) {
    counts[0] = counts[1] = 0; *culled = 0;
    const Mat4 cvp = to_mat4(vc.cull_view_proj_transform);
    if (usePrims) {
        for (uint32_t id = 0; id < s.d.num_primitives; ++id) {
            const incr_gpu_primitive& pr = s.d.primitives[id];
            const uint32_t visible = s.d.mesh_visibility ? s.d.mesh_visibility[pr.meshIndex] : 1u;
            if (!visible) continue;
            if (!frustumCulling || BatchFrustumCull(pr.boundingSphere, mesh_local_to_world(s, pr.meshIndex), cvp)) {
                const incr_draw_cmd cmd = {id, pr.indexCount > 0 ? (uint32_t)pr.indexCount : 0u};
                if (s.d.materials[pr.materialIndex].doubleSided) dsList[counts[1]++] = cmd; else defList[counts[0]++] = cmd;
            } else ++*culled;
        }
        return;
    }
    for (uint32_t id = 0; id < s.d.num_meshlets; ++id) {
        const incr_gpu_meshlet& m = s.d.meshlets[id];
        const uint32_t visible = s.d.mesh_visibility ? s.d.mesh_visibility[m.meshIndex] : 1u;
        if (!visible) continue;
        if (!frustumCulling || BatchFrustumCull(m.boundingSphere, mesh_local_to_world(s, m.meshIndex), cvp)) {
            const incr_draw_cmd cmd = {id, m.triangleCount * 3u};
            if (s.d.materials[m.materialIndex].doubleSided) dsList[counts[1]++] = cmd; else defList[counts[0]++] = cmd;
        } else ++*culled;
    }
}

// ------------------------------------------------------------------------------------------------ light culling
struct Frustum6 { float nearP[4], farP[4], leftP[4], rightP[4], bottomP[4], topP[4]; };
static void plane_from(float out[4], float3 a, float3 b, float3 P) { float3 n = cross(a, b); out[0] = n.x; out[1] = n.y; out[2] = n.z; out[3] = -dot(n, P); }
// shaders/common/cluster-culling.hlsli:44-113 GetClusterFrustum
static Frustum6 GetClusterFrustum(const uint32_t ci[3], const uint32_t grid[3], float zNear, float zFar, const Mat4& proj, const Mat4& invViewProj) {
    float cx = (float)ci[0] / (float)grid[0], cy = (float)ci[1] / (float)grid[1];
    cx = 2.f * cx - 1.f; cy = 2.f * cy - 1.f; cy = -cy;
    const float sx = 2.f / (float)grid[0], sy = 2.f / (float)grid[1];
    const float d0 = zNear * powf(zFar / zNear, (float)ci[2] / (float)grid[2]);
    const float d1 = zNear * powf(zFar / zNear, ((float)ci[2] + 1.f) / (float)grid[2]);
    const float4 np = mul_vec4({0.f, 0.f, d0, 1.f}, proj), fp = mul_vec4({0.f, 0.f, d1, 1.f}, proj);
    const float zn = np.z / np.w, zf = fp.z / fp.w;
    const float4 pts[8] = {{cx, cy - sy, zn, 1.f}, {cx + sx, cy - sy, zn, 1.f}, {cx + sx, cy, zn, 1.f}, {cx, cy, zn, 1.f},
                           {cx, cy - sy, zf, 1.f}, {cx + sx, cy - sy, zf, 1.f}, {cx + sx, cy, zf, 1.f}, {cx, cy, zf, 1.f}};
    float3 P[8];
    for (int i = 0; i < 8; ++i) { const float4 w = mul_vec4(pts[i], invViewProj); P[i] = {w.x / w.w, w.y / w.w, w.z / w.w}; }
    Frustum6 f;
    plane_from(f.nearP, P[1] - P[0], P[3] - P[0], P[0]);
    plane_from(f.farP, P[7] - P[4], P[5] - P[4], P[4]);
    plane_from(f.leftP, P[3] - P[0], P[4] - P[0], P[0]);
    plane_from(f.rightP, P[5] - P[1], P[2] - P[1], P[1]);
    plane_from(f.topP, P[6] - P[2], P[3] - P[2], P[2]);
    plane_from(f.bottomP, P[4] - P[0], P[1] - P[0], P[0]);
    return f;
}
static bool ClusterFrustumCull(const Frustum6& f, float3 c, float r) {       // cluster-culling.hlsli:116-129
    return (dot4p(c, f.nearP) + r * len3(f.nearP) >= 0) && (dot4p(c, f.farP) + r * len3(f.farP) >= 0) && (dot4p(c, f.leftP) + r * len3(f.leftP) >= 0)
        && (dot4p(c, f.rightP) + r * len3(f.rightP) >= 0) && (dot4p(c, f.bottomP) + r * len3(f.bottomP) >= 0) && (dot4p(c, f.topP) + r * len3(f.topP) >= 0);
}
static const float kMaxLightRange = 4.f;
// shaders/culling/light-culling.hlsl:45-131; offsets are c * maxLights (the reference allocates them with one atomic per cluster)
static void light_cull(const Scene& s, const incr_view_constants& vc, const uint32_t grid[3], uint32_t maxLights, float depthExtent,
                       uint32_t* lists, incr_light_grid_data* lightGrid, uint32_t* totalCount, int32_t* culled) {
    const Mat4 proj = to_mat4(vc.lightcull_proj_transform), ivp = to_mat4(vc.lightcull_inv_view_proj);
    *totalCount = 0; *culled = 0;
    for (uint32_t z = 0; z < grid[2]; ++z) for (uint32_t y = 0; y < grid[1]; ++y) for (uint32_t x = 0; x < grid[0]; ++x) {
        const uint32_t ci[3] = {x, y, z};
        const uint32_t clusterId = (grid[0] * grid[1]) * z + grid[0] * y + x;
        const Frustum6 f = GetClusterFrustum(ci, grid, vc.near_plane, depthExtent, proj, ivp);
        uint32_t count = 0;
        for (uint32_t i = 0; i < s.d.num_scene_lights && count < maxLights; ++i) {
            const uint32_t gi = (uint32_t)s.d.light_indices[i];
            const incr_light& light = s.d.global_lights[gi];
            if (light.type == INCR_LIGHT_DIRECTIONAL) continue;
            const float range = light.range == 0 ? kMaxLightRange : light.range;
            const float* m = s.d.light_transforms[i].m;           // Q10: the transform is indexed with the loop index
            if (ClusterFrustumCull(f, {m[12], m[13], m[14]}, range)) lists[(size_t)clusterId * maxLights + count++] = gi;
            else ++*culled;
        }
        lightGrid[clusterId].offset = clusterId * maxLights; lightGrid[clusterId].count = count;
        *totalCount += count;
    }
}

// ------------------------------------------------------------------------------------------------ visibility pass
struct ClipVert { float x, y, z, w; };
struct ScreenVert { int32_t X, Y; float z; };
static inline bool to_screen(const ClipVert& c, float W, float H, ScreenVert* o) {
    if (!(c.w > 0.f)) return false;
    const float nx = c.x / c.w, ny = c.y / c.w;
    float sx = (nx * 0.5f + 0.5f) * W, sy = (0.5f - ny * 0.5f) * H;
    if (!(sx == sx) || !(sy == sy)) return false;
    sx = fminf(fmaxf(sx, -1048576.f), 1048576.f); sy = fminf(fmaxf(sy, -1048576.f), 1048576.f);
    o->X = (int32_t)floorf(sx * 256.f + 0.5f); o->Y = (int32_t)floorf(sy * 256.f + 0.5f);
    o->z = c.z / c.w;
    return true;
}
static inline ClipVert clip_lerp(const ClipVert& a, const ClipVert& b) {      // a inside, b outside the near plane
    const float da = a.w - a.z, db = b.w - b.z;
    const float t = da / (da - db);
    return {a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z), a.w + t * (b.w - a.w)};
}
// gbuffer-compute.hlsl:19-46 CalcBarycentrics (rcp pinned to 1/x)
static float3 CalcBarycentrics(float4 p0, float4 p1, float4 p2, float2 pixelNdc) {
    const float3 invW = {1.f / p0.w, 1.f / p1.w, 1.f / p2.w};
    const float2 ndc0 = {p0.x * invW.x, p0.y * invW.x}, ndc1 = {p1.x * invW.y, p1.y * invW.y}, ndc2 = {p2.x * invW.z, p2.y * invW.z};
    const float2 a = ndc2 - ndc1, b = ndc0 - ndc1;
    const float invDet = 1.f / (a.x * b.y - a.y * b.x);
    const float3 ddx = {((ndc1.y - ndc2.y) * invDet) * invW.x, ((ndc2.y - ndc0.y) * invDet) * invW.y, ((ndc0.y - ndc1.y) * invDet) * invW.z};
    const float3 ddy = {((ndc2.x - ndc1.x) * invDet) * invW.x, ((ndc0.x - ndc2.x) * invDet) * invW.y, ((ndc1.x - ndc0.x) * invDet) * invW.z};
    const float ddxSum = (ddx.x * 1.f + ddx.y * 1.f) + ddx.z * 1.f, ddySum = (ddy.x * 1.f + ddy.y * 1.f) + ddy.z * 1.f;
    const float2 dv = pixelNdc - ndc0;
    const float interpInvW = (invW.x + dv.x * ddxSum) + dv.y * ddySum;
    const float interpW = 1.f / interpInvW;
    float3 l;
    l.x = interpW * ((invW.x + dv.x * ddx.x) + dv.y * ddy.x);
    l.y = interpW * ((0.0f + dv.x * ddx.y) + dv.y * ddy.y);
    l.z = interpW * ((0.0f + dv.x * ddx.z) + dv.y * ddy.z);
    return l;
}
static inline float2 pixel_ndc(uint32_t px, uint32_t py, float W, float H) {   // gbuffer-compute.hlsl:207-210
    float2 n = {((float)px + 0.5f) / W, ((float)py + 0.5f) / H};
    n.x = 2.f * n.x - 1.f; n.y = -2.f * n.y + 1.f;
    return n;
}
static inline float2 interp2(float2 a, float2 b, float2 c, float3 w) { return {(w.x * a.x + w.y * b.x) + w.z * c.x, (w.x * a.y + w.y * b.y) + w.z * c.y}; }
// UV derivatives of Sample(): ddx_coarse / ddy_coarse of the pixel's 2x2 quad, the three UVs taken on the pixel's OWN triangle
// (its plane extrapolated to the quad's top-left pixel and that pixel's right / lower neighbour) -- pinned, DESIGN.md section 3
struct UvGrad { float2 ddx, ddy; };
static inline UvGrad quad_uv_gradient(float4 p0, float4 p1, float4 p2, float2 uv0, float2 uv1, float2 uv2, uint32_t x, uint32_t y, float W, float H) {
    const uint32_t x0 = x & ~1u, y0 = y & ~1u;
    const float2 a = interp2(uv0, uv1, uv2, CalcBarycentrics(p0, p1, p2, pixel_ndc(x0, y0, W, H)));
    const float2 b = interp2(uv0, uv1, uv2, CalcBarycentrics(p0, p1, p2, pixel_ndc(x0 + 1u, y0, W, H)));
    const float2 c = interp2(uv0, uv1, uv2, CalcBarycentrics(p0, p1, p2, pixel_ndc(x0, y0 + 1u, W, H)));
    return {{b.x - a.x, b.y - a.y}, {c.x - a.x, c.y - a.y}};
}
static inline float3 interp3(float3 a, float3 b, float3 c, float3 w) {
    return {(w.x * a.x + w.y * b.x) + w.z * c.x, (w.x * a.y + w.y * b.y) + w.z * c.y, (w.x * a.z + w.y * b.z) + w.z * c.z};
}
static inline void meshlet_tri_verts(const Scene& s, const incr_gpu_meshlet& m, uint32_t tri, uint32_t out[3]) {   // visibility-pass.hlsl:72-82
    const uint32_t packed = s.d.meshlet_packed_triangles[m.triangleBegin + tri];
    for (uint32_t k = 0; k < 3; ++k) out[k] = s.d.meshlet_vertex_indices[m.vertexBegin + (0xffu & (packed >> ((2u - k) * 10u)))];
}

struct RasterTarget { uint32_t W, H, y0, y1; std::vector<uint64_t> packed; };

static void raster_subtri(const Scene& s, RasterTarget& rt, ScreenVert v0, ScreenVert v1, ScreenVert v2, bool cullBack, uint32_t key, uint32_t list,
                          const incr_material& mat, const float4 p[3], const float2 uv[3]) {
    int64_t area = (int64_t)(v1.X - v0.X) * (v2.Y - v0.Y) - (int64_t)(v2.X - v0.X) * (v1.Y - v0.Y);
    if (area == 0) return;
    if (area > 0) { if (cullBack) return; }                    // clockwise on screen = back face
    else { ScreenVert t = v1; v1 = v2; v2 = t; area = -area; }  // make the vertex order clockwise
    const int32_t minX = std::min(v0.X, std::min(v1.X, v2.X)), maxX = std::max(v0.X, std::max(v1.X, v2.X));
    const int32_t minY = std::min(v0.Y, std::min(v1.Y, v2.Y)), maxY = std::max(v0.Y, std::max(v1.Y, v2.Y));
    const int32_t px0 = std::max(0, (minX + 127) >> 8), px1 = std::min((int32_t)rt.W - 1, (maxX - 128) >> 8);
    const int32_t py0 = std::max((int32_t)rt.y0, (minY + 127) >> 8), py1 = std::min((int32_t)rt.y1 - 1, (maxY - 128) >> 8);
    auto topLeft = [](int32_t dx, int32_t dy) { return (dy == 0 && dx > 0) || dy < 0; };
    const double invArea = 1.0 / (double)area;
    const int64_t b01 = topLeft(v1.X - v0.X, v1.Y - v0.Y) ? 0 : -1, b12 = topLeft(v2.X - v1.X, v2.Y - v1.Y) ? 0 : -1, b20 = topLeft(v0.X - v2.X, v0.Y - v2.Y) ? 0 : -1;
    const float W = (float)rt.W, H = (float)rt.H;
    for (int32_t py = py0; py <= py1; ++py)
        for (int32_t px = px0; px <= px1; ++px) {
            const int64_t Sx = (int64_t)px * 256 + 128, Sy = (int64_t)py * 256 + 128;
            const int64_t E01 = (int64_t)(v1.X - v0.X) * (Sy - v0.Y) - (int64_t)(v1.Y - v0.Y) * (Sx - v0.X);
            const int64_t E12 = (int64_t)(v2.X - v1.X) * (Sy - v1.Y) - (int64_t)(v2.Y - v1.Y) * (Sx - v1.X);
            const int64_t E20 = (int64_t)(v0.X - v2.X) * (Sy - v2.Y) - (int64_t)(v0.Y - v2.Y) * (Sx - v2.X);
            if (E01 + b01 < 0 || E12 + b12 < 0 || E20 + b20 < 0) continue;
            const float l1 = (float)((double)E20 * invArea), l2 = (float)((double)E01 * invArea);
            float depth = (v0.z + l1 * (v1.z - v0.z)) + l2 * (v2.z - v0.z);
            if (!(depth >= 0.f)) continue;
            if (depth > 1.f) depth = 1.f;
            if (mat.baseColorTextureIndex != -1) {             // ps_meshlet_main: clip(opacity - 0.5)
                const float3 lam = CalcBarycentrics(p[0], p[1], p[2], pixel_ndc((uint32_t)px, (uint32_t)py, W, H));
                const UvGrad g = quad_uv_gradient(p[0], p[1], p[2], uv[0], uv[1], uv[2], (uint32_t)px, (uint32_t)py, W, H);
                const float opacity = SampleGrad(s, mat.baseColorTextureIndex, interp2(uv[0], uv[1], uv[2], lam), g.ddx, g.ddy).w;
                if (opacity - 0.5f < 0.f) continue;
            }
            // GREATER_EQUAL in draw order (default list, then double-sided list; ascending id inside a list): the survivor of a pixel is the
            // maximum of (depth, list, id).  One 64-bit word: depth in [0, 1] has a bit pattern below 2^30, one bit of list, 33 bits of id + 1
            // (the id keeps all 32 bits: EncodePrimitiveVisibility is 12 + 20)
            const uint64_t v = ((uint64_t)f2u(depth) << 34) | ((uint64_t)list << 33) | ((uint64_t)key + 1u);
            uint64_t& dst = rt.packed[(size_t)py * rt.W + px];
            if (v > dst) dst = v;
        }
}

// visibility-pass.hlsl vs_meshlet_main + ps_meshlet_main over both draw lists
// (usePrims: vs_primitive_main :32-55 + ps_primitive_main -- a draw command names a whole primitive, SV_PrimitiveID is the triangle)
static void visibility_pass(const Scene& s, const incr_view_constants& vc, const incr_draw_cmd* defList, const incr_draw_cmd* dsList,
                            const uint32_t counts[2], uint32_t* vis, float* depthOut, bool usePrims = false, bool forward = false) {
    RasterTarget rt; rt.W = vc.res_x; rt.H = vc.res_y; rt.packed.assign((size_t)rt.W * rt.H, 0);
    scissor_rows(vc, &rt.y0, &rt.y1);
    const Mat4 viewProj = to_mat4(vc.view_proj_transform);
    const float W = (float)rt.W, H = (float)rt.H;
    for (int list = 0; list < 2; ++list) {
        const incr_draw_cmd* cmds = list ? dsList : defList;
        for (uint32_t c = 0; c < counts[list]; ++c) {
            const uint32_t objectId = cmds[c].id;                 // meshlet id, or primitive id
            const incr_gpu_meshlet* m = usePrims ? nullptr : &s.d.meshlets[objectId];
            const incr_gpu_primitive* pr = usePrims ? &s.d.primitives[objectId] : nullptr;
            const Mat4 localToWorld = mesh_local_to_world(s, usePrims ? pr->meshIndex : m->meshIndex);
            const incr_material& mat = s.d.materials[usePrims ? pr->materialIndex : m->materialIndex];
            const int posAcc = usePrims ? pr->positionAccessor : m->positionAccessor, uvAcc = usePrims ? pr->uvAccessor : m->uvAccessor;
            for (uint32_t tri = 0; tri < cmds[c].vertex_count / 3u; ++tri) {
                uint32_t vi[3];
                if (usePrims) GetUint3(s, (int)(tri * 3u), pr->indexAccessor, vi); else meshlet_tri_verts(s, *m, tri, vi);
                ClipVert cv[3]; float4 p[3]; float2 uv[3];
                for (int k = 0; k < 3; ++k) {
                    const float3 pos = GetFloat3(s, (int)vi[k], posAcc);
                    const float4 worldPos = mul_vec4({pos.x, pos.y, pos.z, 1.f}, localToWorld);
                    p[k] = mul_vec4(worldPos, viewProj);
                    cv[k] = {p[k].x, p[k].y, p[k].z, p[k].w};
                    uv[k] = GetFloat2(s, (int)vi[k], uvAcc);
                }
                // forward pass (forward-pass.inl:68-179): one draw per primitive in scene order, cull mode from the material
                const bool cullBack = forward ? !mat.doubleSided : list == 0;
                const uint32_t key = usePrims ? ((objectId << 20) | tri) : ((objectId << 7) | tri);      // Encode{Primitive,Meshlet}Visibility, encoding.hlsli:4-28
                const uint32_t listBit = forward ? 0u : (uint32_t)list;
                // near-plane clip (z <= w)
                bool in[3]; int nin = 0;
                for (int k = 0; k < 3; ++k) { in[k] = (cv[k].w - cv[k].z) >= 0.f; nin += in[k]; }
                if (nin == 0) continue;
                ClipVert poly[4]; int np = 0;
                if (nin == 3) { poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2]; np = 3; }
                else for (int k = 0; k < 3; ++k) {
                    const int j = (k + 1) % 3;
                    if (in[k]) poly[np++] = cv[k];
                    if (in[k] != in[j]) poly[np++] = in[k] ? clip_lerp(cv[k], cv[j]) : clip_lerp(cv[j], cv[k]);
                }
                ScreenVert sv[4]; bool ok = true;
                for (int k = 0; k < np; ++k) ok = ok && to_screen(poly[k], W, H, &sv[k]);
                if (!ok) continue;
                raster_subtri(s, rt, sv[0], sv[1], sv[2], cullBack, key, listBit, mat, p, uv);
                if (np == 4) raster_subtri(s, rt, sv[0], sv[2], sv[3], cullBack, key, listBit, mat, p, uv);
            }
        }
    }
    for (size_t i = (size_t)rt.y0 * rt.W; i < (size_t)rt.y1 * rt.W; ++i) {       // only the scissor rows are cleared and written
        const uint64_t v = rt.packed[i];
        if (v == 0) { vis[i] = 0xFFFFF000u; depthOut[i] = 0.f; }
        else { vis[i] = (uint32_t)((v & 0x1ffffffffull) - 1u); depthOut[i] = u2f((uint32_t)(v >> 34)); }
    }
}

// ------------------------------------------------------------------------------------------------ G-buffer
static inline uint32_t unorm8(float c) { if (!(c > 0.f)) return 0; if (c > 1.f) c = 1.f; return (uint32_t)floorf(c * 255.f + 0.5f); }
static inline uint32_t pack_unorm8x4(float4 c) { return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24); }
static inline float4 unpack_unorm8x4(uint32_t v) { return {(float)(v & 255u) / 255.f, (float)((v >> 8) & 255u) / 255.f, (float)((v >> 16) & 255u) / 255.f, (float)(v >> 24) / 255.f}; }
static float2 OctEncode(float3 n) {                                          // geo-raster/encoding.hlsli:37-43
    const float s = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    n = {n.x / s, n.y / s, n.z / s};
    float2 xy = {n.x, n.y};
    if (!(n.z >= 0.f)) xy = {(1.f - fabsf(n.y)) * (n.x >= 0.f ? 1.f : -1.f), (1.f - fabsf(n.x)) * (n.y >= 0.f ? 1.f : -1.f)};
    return {xy.x * 0.5f + 0.5f, xy.y * 0.5f + 0.5f};
}
static float3 OctDecode(float2 f) {                                          // :45-53
    f = {f.x * 2.f - 1.f, f.y * 2.f - 1.f};
    float3 n = {f.x, f.y, (1.f - fabsf(f.x)) - fabsf(f.y)};
    const float t = saturate(-n.z);
    n.x += n.x >= 0.f ? -t : t; n.y += n.y >= 0.f ? -t : t;
    return normalize(n);
}
static inline float3 mul_vec3_m3(float3 v, const float r0[3], const float r1[3], const float r2[3]) {     // mul(v, float3x3(rows))
    return {(v.x * r0[0] + v.y * r1[0]) + v.z * r2[0], (v.x * r0[1] + v.y * r1[1]) + v.z * r2[1], (v.x * r0[2] + v.y * r1[2]) + v.z * r2[2]};
}
// gbuffer-compute.hlsl:157-253 cs_main (USING_MESHLETS = 1)
// usePrims: USING_MESHLETS = 0 (DecodePrimitiveVisibility + GetPrimitiveTriangleData :96-121)
static void gbuffer_pass(const Scene& s, const incr_view_constants& vc, const uint32_t* vis, uint32_t* g0, uint32_t* g1, uint32_t* g2,
                         uint32_t* color, int rounding, bool usePrims = false) {
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 viewProj = to_mat4(vc.view_proj_transform);
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        color[i] = 0;                                              // gbuffer-compute-pass.inl:94-100 clear
        const uint32_t v = vis[i];
        if (v == 0xFFFFF000u) { g0[i] = 0; g1[i] = 0; g2[i] = 0; continue; }
        uint32_t vi[3]; int meshIndex, materialIndex, posAcc, uvAcc, nrmAcc, tanAcc;
        if (usePrims) {
            const uint32_t primId = v >> 20, tri = v & 0xfffffu;
            const incr_gpu_primitive& pr = s.d.primitives[primId];
            GetUint3(s, (int)(tri * 3u), pr.indexAccessor, vi);
            meshIndex = pr.meshIndex; materialIndex = pr.materialIndex;
            posAcc = pr.positionAccessor; uvAcc = pr.uvAccessor; nrmAcc = pr.normalAccessor; tanAcc = pr.tangentAccessor;
        } else {
            const uint32_t meshletId = v >> 7, tri = v & 127u;
            const incr_gpu_meshlet& m = s.d.meshlets[meshletId];
            meshlet_tri_verts(s, m, tri, vi);
            meshIndex = m.meshIndex; materialIndex = m.materialIndex;
            posAcc = m.positionAccessor; uvAcc = m.uvAccessor; nrmAcc = m.normalAccessor; tanAcc = m.tangentAccessor;
        }
        const Mat4 localToWorld = mesh_local_to_world(s, meshIndex);
        const Mat4 localToClip = mat_mul(localToWorld, viewProj);
        float4 p[3]; float2 uv[3]; float3 nrm[3]; float4 tan[3];
        for (int k = 0; k < 3; ++k) {
            const float3 pos = GetFloat3(s, (int)vi[k], posAcc);
            p[k] = mul_vec4({pos.x, pos.y, pos.z, 1.f}, localToClip);
            uv[k] = GetFloat2(s, (int)vi[k], uvAcc);
            nrm[k] = GetFloat3(s, (int)vi[k], nrmAcc);
            tan[k] = GetFloat4(s, (int)vi[k], tanAcc);
        }
        const float3 lam = CalcBarycentrics(p[0], p[1], p[2], pixel_ndc(x, y, (float)W, (float)H));
        float3 N = normalize(interp3(nrm[0], nrm[1], nrm[2], lam));
        const float3 T = normalize(interp3({tan[0].x, tan[0].y, tan[0].z}, {tan[1].x, tan[1].y, tan[1].z}, {tan[2].x, tan[2].y, tan[2].z}, lam));
        const float3 B = normalize(cross(N, T) * tan[0].w);
        const float2 UV = interp2(uv[0], uv[1], uv[2], lam);
        const UvGrad gr = quad_uv_gradient(p[0], p[1], p[2], uv[0], uv[1], uv[2], x, y, (float)W, (float)H);
        const MatProps mi = EvaluateMaterialPropertiesGrad(s, s.d.materials[materialIndex], UV, gr.ddx, gr.ddy);      // Sample(g_anisoSampler, uv)
        if (mi.bHasNormalmap) {
            // tangentToWorld = mul(float3x3(T, B, N), (float3x3)localToWorld); N = normalize(mul(normalmap, tangentToWorld))
            const float* L = localToWorld.m;
            const float l0[3] = {L[0], L[1], L[2]}, l1[3] = {L[4], L[5], L[6]}, l2[3] = {L[8], L[9], L[10]};
            const float3 r0 = mul_vec3_m3(T, l0, l1, l2), r1 = mul_vec3_m3(B, l0, l1, l2), r2 = mul_vec3_m3(N, l0, l1, l2);
            const float a0[3] = {r0.x, r0.y, r0.z}, a1[3] = {r1.x, r1.y, r1.z}, a2[3] = {r2.x, r2.y, r2.z};
            N = normalize(mul_vec3_m3(mi.normalmap, a0, a1, a2));
        }
        color[i] = pack_r11g11b10(mi.emissive * 20000.f, rounding);
        g0[i] = pack_unorm8x4({mi.basecolor.x, mi.basecolor.y, mi.basecolor.z, 0.f});
        const float2 oct = OctEncode(N);
        g1[i] = (uint32_t)f32_to_f16(oct.x, rounding) | ((uint32_t)f32_to_f16(oct.y, rounding) << 16);
        g2[i] = pack_unorm8x4({mi.metallic, mi.roughness, mi.ao, mi.aoblend});
    }
}

// ------------------------------------------------------------------------------------------------ decals
// render-jobs/gbuffer-raster-pass.inl:17-233 + shaders/geo-raster/gbuffer-raster.hlsl: every primitive of
// FScene::m_sceneMeshDecals is drawn in order over the G-buffers -- depth test GREATER_EQUAL against the scene depth without
// writes, back faces culled unless the material is double-sided, RT0 (base colour) blended SRC_ALPHA / INV_SRC_ALPHA in
// colour AND alpha, RT1 (normals) overwritten only when the decal has a normal map, RT2 only when it has a
// metallic-roughness map.  Rasterisation rules as the visibility pass; UNORM8 blending in float, stored per fragment.
template <class F> static void raster_cover(uint32_t W, uint32_t y0, uint32_t y1, ScreenVert v0, ScreenVert v1, ScreenVert v2, bool cullBack, F frag) {
    int64_t area = (int64_t)(v1.X - v0.X) * (v2.Y - v0.Y) - (int64_t)(v2.X - v0.X) * (v1.Y - v0.Y);
    if (area == 0) return;
    if (area > 0) { if (cullBack) return; }
    else { ScreenVert t = v1; v1 = v2; v2 = t; area = -area; }
    const int32_t minX = std::min(v0.X, std::min(v1.X, v2.X)), maxX = std::max(v0.X, std::max(v1.X, v2.X));
    const int32_t minY = std::min(v0.Y, std::min(v1.Y, v2.Y)), maxY = std::max(v0.Y, std::max(v1.Y, v2.Y));
    const int32_t px0 = std::max(0, (minX + 127) >> 8), px1 = std::min((int32_t)W - 1, (maxX - 128) >> 8);
    const int32_t py0 = std::max((int32_t)y0, (minY + 127) >> 8), py1 = std::min((int32_t)y1 - 1, (maxY - 128) >> 8);
    auto topLeft = [](int32_t dx, int32_t dy) { return (dy == 0 && dx > 0) || dy < 0; };
    const double invArea = 1.0 / (double)area;
    const int64_t b01 = topLeft(v1.X - v0.X, v1.Y - v0.Y) ? 0 : -1, b12 = topLeft(v2.X - v1.X, v2.Y - v1.Y) ? 0 : -1, b20 = topLeft(v0.X - v2.X, v0.Y - v2.Y) ? 0 : -1;
    for (int32_t py = py0; py <= py1; ++py)
        for (int32_t px = px0; px <= px1; ++px) {
            const int64_t Sx = (int64_t)px * 256 + 128, Sy = (int64_t)py * 256 + 128;
            const int64_t E01 = (int64_t)(v1.X - v0.X) * (Sy - v0.Y) - (int64_t)(v1.Y - v0.Y) * (Sx - v0.X);
            const int64_t E12 = (int64_t)(v2.X - v1.X) * (Sy - v1.Y) - (int64_t)(v2.Y - v1.Y) * (Sx - v1.X);
            const int64_t E20 = (int64_t)(v0.X - v2.X) * (Sy - v2.Y) - (int64_t)(v0.Y - v2.Y) * (Sx - v2.X);
            if (E01 + b01 < 0 || E12 + b12 < 0 || E20 + b20 < 0) continue;
            const float l1 = (float)((double)E20 * invArea), l2 = (float)((double)E01 * invArea);
            float depth = (v0.z + l1 * (v1.z - v0.z)) + l2 * (v2.z - v0.z);
            if (!(depth >= 0.f)) continue;
            if (depth > 1.f) depth = 1.f;
            frag((uint32_t)px, (uint32_t)py, depth);
        }
}
static void gbuffer_decals(const Scene& s, const incr_gbuffer_decals_desc& d) {
    const incr_view_constants& vc = *d.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    const Mat4 viewProj = to_mat4(vc.view_proj_transform), root = to_mat4(s.d.root_transform);
    const float Wf = (float)W, Hf = (float)H;
    for (uint32_t di = 0; di < s.d.num_decal_primitives; ++di) {
        const incr_gpu_primitive& pr = s.d.decal_primitives[di];
        const incr_material& mat = s.d.materials[pr.materialIndex];
        const Mat4 localToWorld = mat_mul(to_mat4(s.d.decal_transforms[di]), root);          // gbuffer-raster.hlsl:51
        const uint32_t ntri = pr.indexCount > 0 ? (uint32_t)pr.indexCount / 3u : 0u;
        for (uint32_t tri = 0; tri < ntri; ++tri) {
            uint32_t vi[3]; GetUint3(s, (int)(tri * 3u), pr.indexAccessor, vi);
            ClipVert cv[3]; float4 clip[3]; float2 uv[3]; float3 nW[3], tW[3], bW[3];
            for (int k = 0; k < 3; ++k) {                                                     // vs_main :46-78
                const float3 pos = GetFloat3(s, (int)vi[k], pr.positionAccessor);
                const float4 worldPos = mul_vec4({pos.x, pos.y, pos.z, 1.f}, localToWorld);
                clip[k] = mul_vec4(worldPos, viewProj);
                cv[k] = {clip[k].x, clip[k].y, clip[k].z, clip[k].w};
                uv[k] = GetFloat2(s, (int)vi[k], pr.uvAccessor);
                const float3 normal = GetFloat3(s, (int)vi[k], pr.normalAccessor);
                const float4 pt = GetFloat4(s, (int)vi[k], pr.tangentAccessor);
                const float3 tangent = {pt.x, pt.y, pt.z};
                const float3 bitangent = cross(normal, tangent) * pt.w;
                const float4 n4 = mul_vec4({normal.x, normal.y, normal.z, 0.f}, localToWorld), t4 = mul_vec4({tangent.x, tangent.y, tangent.z, 0.f}, localToWorld);
                const float4 b4 = mul_vec4({bitangent.x, bitangent.y, bitangent.z, 0.f}, localToWorld);
                nW[k] = {n4.x, n4.y, n4.z}; tW[k] = {t4.x, t4.y, t4.z}; bW[k] = {b4.x, b4.y, b4.z};
            }
            bool in[3]; int nin = 0;
            for (int k = 0; k < 3; ++k) { in[k] = (cv[k].w - cv[k].z) >= 0.f; nin += in[k]; }
            if (nin == 0) continue;
            ClipVert poly[4]; int np = 0;
            if (nin == 3) { poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2]; np = 3; }
            else for (int k = 0; k < 3; ++k) {
                const int j = (k + 1) % 3;
                if (in[k]) poly[np++] = cv[k];
                if (in[k] != in[j]) poly[np++] = in[k] ? clip_lerp(cv[k], cv[j]) : clip_lerp(cv[j], cv[k]);
            }
            ScreenVert sv[4]; bool ok = true;
            for (int k = 0; k < np; ++k) ok = ok && to_screen(poly[k], Wf, Hf, &sv[k]);
            if (!ok) continue;
            auto frag = [&](uint32_t px, uint32_t py, float depth) {
                const size_t i = (size_t)py * W + px;
                if (!(depth >= d.depth_buffer[i])) return;                                     // GREATER_EQUAL, DepthWriteMask ZERO
                const float3 lam = CalcBarycentrics(clip[0], clip[1], clip[2], pixel_ndc(px, py, Wf, Hf));
                const UvGrad gr = quad_uv_gradient(clip[0], clip[1], clip[2], uv[0], uv[1], uv[2], px, py, Wf, Hf);
                const MatProps p = EvaluateMaterialPropertiesGrad(s, mat, interp2(uv[0], uv[1], uv[2], lam), gr.ddx, gr.ddy);
                const float3 T = normalize(interp3(tW[0], tW[1], tW[2], lam)), B = normalize(interp3(bW[0], bW[1], bW[2], lam));
                float3 N = normalize(interp3(nW[0], nW[1], nW[2], lam));
                if (p.bHasNormalmap) {
                    const float3 nm = p.normalmap;
                    N = normalize(f3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
                }
                const float4 dst = unpack_unorm8x4(d.gbuffer_basecolor[i]);
                const float a = p.opacity, ia = 1.f - a;
                d.gbuffer_basecolor[i] = pack_unorm8x4({p.basecolor.x * a + dst.x * ia, p.basecolor.y * a + dst.y * ia, p.basecolor.z * a + dst.z * ia, a * a + dst.w * ia});
                if (mat.normalTextureIndex != -1) {
                    const float2 oct = OctEncode(N);
                    d.gbuffer_normals[i] = (uint32_t)f32_to_f16(oct.x, d.storage_rounding) | ((uint32_t)f32_to_f16(oct.y, d.storage_rounding) << 16);
                }
                if (mat.metallicRoughnessTextureIndex != -1) d.gbuffer_metallic_roughness_ao[i] = pack_unorm8x4({p.metallic, p.roughness, p.ao, p.aoblend});
            };
            const bool cullBack = !mat.doubleSided;
            raster_cover(W, y0, y1, sv[0], sv[1], sv[2], cullBack, frag);
            if (np == 4) raster_cover(W, y0, y1, sv[0], sv[2], sv[3], cullBack, frag);
        }
    }
}

// ------------------------------------------------------------------------------------------------ HBAO
// shaders/postprocess/hbao.hlsl:1-175.  HLSL leaves the precision of cos / sin / acos / rsqrt to the implementation; the
// pinned choice is a fixed evaluation both sides restate: Cody-Waite reduction by pi/2 + Taylor polynomials
// in Horner form for sin / cos on [-pi, pi], the arcsine series through x^23 with the half-angle identity for acos,
// rsqrt(x) = 1 / sqrt(x), saturate(NaN) = 0, NaN stored to R16_FLOAT as 0x7FFF.
static inline void hb_reduce(float x, float* r, int* q) {
    const float kf = floorf(x * 0.636619747f + 0.5f);
    *r = (x - kf * 1.5707963705062866f) - kf * -4.3711388286737929e-08f;
    *q = (kf == kf) ? ((int)kf & 3) : 0;
}
static inline float hb_sin_poly(float r) {
    const float z = r * r;
    return r + (r * z) * (-0.166666672f + z * (0.00833333377f + z * (-0.000198412701f + z * 2.75573188e-06f)));
}
static inline float hb_cos_poly(float r) {
    const float z = r * r;
    return 1.f + z * (-0.5f + z * (0.0416666679f + z * (-0.00138888892f + z * (2.48015876e-05f + z * -2.755732e-07f))));
}
static inline float hb_sin(float x) {
    float r; int q; hb_reduce(x, &r, &q);
    const float s = (q & 1) ? hb_cos_poly(r) : hb_sin_poly(r);
    return (q & 2) ? -s : s;
}
static inline float hb_cos(float x) {
    float r; int q; hb_reduce(x, &r, &q);
    const float c = (q & 1) ? hb_sin_poly(r) : hb_cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}
static inline float hb_asin_half(float x) {            // |x| <= 0.5 (plus rounding)
    const float z = x * x;
    const float p = 0.166666672f + z * (0.075000003f + z * (0.0446428582f + z * (0.030381944f + z * (0.0223721582f + z * (0.0173527636f
                  + z * (0.0139648439f + z * (0.0115518011f + z * (0.00976160914f + z * (0.00839033537f + z * 0.007312526f)))))))));
    return x + (x * z) * p;
}
static inline float hb_acos(float x) {
    if (!(fabsf(x) > 0.5f)) return 1.5707963705062866f - (hb_asin_half(x) - -4.3711388286737929e-08f);
    if (x > 0.f) return 2.f * hb_asin_half(sqrtf((1.f - x) * 0.5f));
    return 3.1415927410125732f - (2.f * hb_asin_half(sqrtf((1.f + x) * 0.5f)) - -8.7422776573475858e-08f);
}
static inline float hb_saturate(float v) { return !(v > 0.f) ? 0.f : (v > 1.f ? 1.f : v); }
static inline uint32_t hb_f16(float v, int rounding) { return v != v ? 0x7fffu : (uint32_t)f32_to_f16(v, rounding); }

static const float kHbaoPi = 3.14159265358979323846f;
static const uint32_t kHbaoNumSlices = 4, kHbaoMaxTracesPerSlice = 4;

static bool hbao_ray_hit(const Scene& s, float3 origin, float3 dir, uint64_t* rays) {            // RayHit :41-64
    ++*rays;
    if (!finite_ray(origin, dir)) return false;
    return trace_any(s, origin, dir, 0.1f, 1.f);
}
static float ComputeHorizonAngle(const Scene& s, float horizonAngleStart, float3 pos, float3 sliceX, float3 sliceY, uint64_t* rays) {   // :66-93
    const float terminate = kHbaoPi / 36.f;
    float minHorizonAngle = 0.f, maxHorizonAngle = horizonAngleStart;
    uint32_t traceIdx = 0;
    while (traceIdx < kHbaoMaxTracesPerSlice && (maxHorizonAngle - minHorizonAngle) > terminate) {
        const float a = minHorizonAngle + 0.5f * (maxHorizonAngle - minHorizonAngle);
        const float3 dir = hb_cos(a) * sliceY + hb_sin(a) * sliceX;
        if (hbao_ray_hit(s, pos, dir, rays)) maxHorizonAngle = a; else minHorizonAngle = a;
        ++traceIdx;
    }
    return maxHorizonAngle;
}
static void hbao_pass(const Scene& s, const incr_hbao_desc& d, uint64_t* rayCount) {           // cs_main :109-175
    const incr_view_constants& vc = *d.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 ivp = to_mat4(vc.inv_view_proj_transform);
    const float3 eye = {vc.eye_pos[0], vc.eye_pos[1], vc.eye_pos[2]}, camUp = {vc.camera_up[0], vc.camera_up[1], vc.camera_up[2]};
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    uint64_t rays = 0;
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const float2 ndc = pixel_ndc(x, y, (float)W, (float)H);                      // (xy + 0.5) / res, y flipped
        const float4 wp = mul_vec4({ndc.x, ndc.y, d.depth_buffer[i], 1.f}, ivp);
        const float3 pos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
        const float3 localAt = normalize(eye - pos);
        const float3 localRight = normalize(cross(localAt, camUp));
        const float3 localUp = cross(localRight, localAt);
        const uint32_t n = d.gbuffer_normals[i];
        const float3 N = OctDecode({f16_to_f32((uint16_t)(n & 0xffffu)), f16_to_f32((uint16_t)(n >> 16))});
        float azimuth = 0.f, sumAO = 0.f;
        float3 avgBent = {0.f, 0.f, 0.f};
        for (uint32_t k = 0; k < kHbaoNumSlices; ++k) {
            const float3 sliceT = hb_cos(azimuth) * localRight + hb_sin(azimuth) * localUp;
            const float pnx = dot(N, sliceT), pny = dot(N, localAt);
            const float t = -dot(sliceT, N) / dot(localAt, N);
            const float horizon = hb_acos(t * (1.f / sqrtf(1.f + t * t)));
            // GatherAO :95-106
            const float theta1 = ComputeHorizonAngle(s, horizon, pos, sliceT, localAt, &rays);
            const float theta0 = -ComputeHorizonAngle(s, horizon, pos, -sliceT, localAt, &rays);
            const float s0 = hb_sin(theta0), c0 = hb_cos(theta0), s1 = hb_sin(theta1), c1 = hb_cos(theta1);
            const float nx = 0.5f * (((theta1 - theta0) + s0 * c0) - s1 * c1);
            const float ny = 0.5f * ((2.f - c0 * c0) - c1 * c1);
            sumAO += pnx * nx + pny * ny;
            avgBent = avgBent + normalize(sliceT * nx + localAt * ny);
            azimuth += kHbaoPi / (float)kHbaoNumSlices;
        }
        const float ao = hb_saturate(sumAO / (float)kHbaoNumSlices);
        d.ao_target[i] = (uint8_t)unorm8(ao);
        const float2 oct = OctEncode(normalize(avgBent));
        d.bent_normal_target[i] = hb_f16(oct.x, d.storage_rounding) | (hb_f16(oct.y, d.storage_rounding) << 16);
    }
    if (rayCount) *rayCount += rays;
}

// ------------------------------------------------------------------------------------------------ deferred lighting
struct GBufferPixel { float3 basecolor, N; float metallic, roughness, ao, aoBlend; };
static inline GBufferPixel read_gbuffer(const incr_lighting_inputs& in, size_t i) {
    GBufferPixel g;
    const float4 b = unpack_unorm8x4(in.gbuffer_basecolor[i]);
    g.basecolor = in.lighting_only ? f3(0.5f) : f3(b.x, b.y, b.z);
    const uint32_t n = in.gbuffer_normals[i];
    g.N = OctDecode({f16_to_f32((uint16_t)(n & 0xffffu)), f16_to_f32((uint16_t)(n >> 16))});
    const float4 m = unpack_unorm8x4(in.gbuffer_metallic_roughness_ao[i]);
    g.metallic = m.x; g.roughness = m.y; g.ao = m.z; g.aoBlend = m.w;
    return g;
}
static inline float2 screen_pos(uint32_t x, uint32_t y, uint32_t W, uint32_t H) {       // dispatchThreadId.xy / res; 2 s - 1 (no half-pixel offset)
    return {2.f * ((float)x / (float)W) - 1.f, 2.f * ((float)y / (float)H) - 1.f};
}
static inline float3 unproject(float2 sp, float depth, const Mat4& inv) {
    const float4 w = mul_vec4({sp.x, -sp.y, depth, 1.f}, inv);
    return {w.x / w.w, w.y / w.w, w.z / w.w};
}
// lightPropertiesBuffer / lightTransformsBuffer are both indexed with the GLOBAL light index (direct-lighting.hlsl:73-74,
// clustered-lighting.hlsl:100-108); an index past the end of a buffer reads zeros (D3D robust buffer access)
static inline void load_global_light(const Scene& s, uint32_t index, incr_light* light, Mat4* transform) {
    memset(light, 0, sizeof *light); memset(transform, 0, sizeof *transform);
    if (index < s.d.num_global_lights) *light = s.d.global_lights[index];
    if (index < s.d.num_scene_lights) *transform = to_mat4(s.d.light_transforms[index]);
}
static inline void add_color(uint32_t* color, size_t i, float3 radiance, int rounding) {   // colorTarget[xy] += radiance on R11G11B10_FLOAT
    color[i] = pack_r11g11b10(unpack_r11g11b10(color[i]) + radiance, rounding);
}

// shaders/lighting/direct-lighting.hlsl:31-85
static void direct_lighting(const Scene& s, const incr_lighting_inputs& in, int sunIndex, uint64_t* shadowRays) {
    const incr_view_constants& vc = *in.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 ivp = to_mat4(vc.inv_view_proj_transform);
    const float3 eye = {vc.eye_pos[0], vc.eye_pos[1], vc.eye_pos[2]};
    incr_light light; Mat4 lightTransform;
    load_global_light(s, (uint32_t)sunIndex, &light, &lightTransform);
    RayStats rs;
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const GBufferPixel g = read_gbuffer(in, i);
        const float3 worldPos = unproject(screen_pos(x, y, W, H), in.depth_buffer[i], ivp);
        const float3 V = normalize(eye - worldPos);
        const float3 radiance = GetDirectRadiance(s, light, lightTransform, worldPos, g.basecolor, g.metallic, g.roughness, g.N, V, false, &rs);
        add_color(in.color_target, i, radiance, in.storage_rounding);
    }
    if (shadowRays) *shadowRays += rs.lightShadow;
}

// shaders/lighting/clustered-lighting.hlsl:35-115 + common/cluster-culling.hlsli:26-41
static void clustered_lighting(const Scene& s, const incr_lighting_inputs& in, const uint32_t* lists, const incr_light_grid_data* lightGrid,
                               const uint32_t grid[3], float depthExtent, int requiresClear) {
    const incr_view_constants& vc = *in.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 ivp = to_mat4(vc.inv_view_proj_transform), ip = to_mat4(vc.inv_proj_transform);
    const float3 eye = {vc.eye_pos[0], vc.eye_pos[1], vc.eye_pos[2]};
    const float scale = (float)grid[2] / logf(depthExtent / vc.near_plane);          // clustered-lighting.inl:88,102
    const float bias = -scale * logf(vc.near_plane);
    const float gridResX = (float)W / (float)grid[0], gridResY = (float)H / (float)grid[1];
    const uint32_t numClusters = grid[0] * grid[1] * grid[2];
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        if (requiresClear) in.color_target[i] = 0;                                    // clustered-lighting.inl:108-117
        const GBufferPixel g = read_gbuffer(in, i);
        float2 sp = screen_pos(x, y, W, H); sp.y = -sp.y;
        const float depth = in.depth_buffer[i];
        const float4 wp = mul_vec4({sp.x, sp.y, depth, 1.f}, ivp);
        const float3 worldPos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
        const float4 vp = mul_vec4({sp.x, sp.y, depth, 1.f}, ip);
        const float viewZ = vp.z / vp.w;
        const uint32_t cx = ftou((float)x / gridResX), cy = ftou((float)y / gridResY);
        const uint32_t cz = ftou(floorf(logf(viewZ) * scale + bias));
        const uint32_t clusterId = ftou(((float)(grid[0] * grid[1]) * (float)cz + (float)grid[0] * (float)cy) + (float)cx);
        const float3 V = normalize(eye - worldPos);
        float3 radiance = {0, 0, 0};
        const uint32_t count = clusterId < numClusters ? lightGrid[clusterId].count : 0u;   // out-of-bounds buffer loads return 0
        const uint32_t offset = clusterId < numClusters ? lightGrid[clusterId].offset : 0u;
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t lightIndex = lists[offset + k];
            incr_light light; Mat4 lightTransform;
            load_global_light(s, lightIndex, &light, &lightTransform);
            if (light.range == 0) light.range = kMaxLightRange;
            radiance += GetDirectRadiance(s, light, lightTransform, worldPos, g.basecolor, g.metallic, g.roughness, g.N, V, false, nullptr);
        }
        add_color(in.color_target, i, radiance, in.storage_rounding);
    }
}

// shaders/image-based-lighting/spherical-harmonics/common.hlsli:56-126
static float3 ShIrradiance(float3 n, const float3 sh[9]) {
    static const float K[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
    static const float A[3] = {3.14159265f, 2.094395f, 0.785398f};
    float v[9];
    v[0] = K[0];
    v[1] = K[1] * n.y; v[2] = K[2] * n.z; v[3] = K[3] * n.x;
    v[4] = (K[4] * n.x) * n.y; v[5] = (K[5] * n.y) * n.z; v[6] = K[6] * ((3.0f * n.z) * n.z - 1.0f); v[7] = (K[7] * n.x) * n.z; v[8] = K[8] * (n.x * n.x - n.y * n.y);
    float c[9];
    c[0] = v[0] * A[0]; c[1] = v[1] * A[1]; c[2] = v[2] * A[1]; c[3] = v[3] * A[1];
    for (int i = 4; i < 9; ++i) c[i] = v[i] * A[2];
    float3 irr = {0, 0, 0};
    for (int i = 0; i < 9; ++i) irr += sh[i] * c[i];
    return irr;
}
// shaders/lighting/sky-lighting.hlsl:36-119 with g_envmapIndex == -1 (no specular IBL inputs this round)
// R16G16_FLOAT LUT, bilinear, CLAMP addressing, level 0
static float2 sample_rg16f_bilinear_clamp(const uint32_t* img, uint32_t W, uint32_t H, float2 uv) {
    float u = uv.x, v = uv.y;
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    u = fminf(fmaxf(u, 0.f), 1.f); v = fminf(fmaxf(v, 0.f), 1.f);
    int xi = linear_fixed(u * (float)W), yi = linear_fixed(v * (float)H), x0i, y0i; float fx, fy;          // CLAMP in fixed point: [0, W - 1] texel centres
    xi = std::min(std::max(xi, 0), (int)(W - 1) * 256); yi = std::min(std::max(yi, 0), (int)(H - 1) * 256);
    linear_split(xi, &x0i, &fx); linear_split(yi, &y0i, &fy);
    const uint32_t x0 = (uint32_t)x0i, y0 = (uint32_t)y0i, x1 = std::min(x0 + 1, W - 1), y1 = std::min(y0 + 1, H - 1);
    auto ld = [&](uint32_t xx, uint32_t yy) -> float2 { const uint32_t t = img[(size_t)yy * W + xx]; return {f16_to_f32((uint16_t)(t & 0xffffu)), f16_to_f32((uint16_t)(t >> 16))}; };
    const float2 t00 = ld(x0, y0), t10 = ld(x1, y0), t01 = ld(x0, y1), t11 = ld(x1, y1);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return {(t00.x * wx0 + t10.x * fx) * wy0 + (t01.x * wx0 + t11.x * fx) * fy, (t00.y * wx0 + t10.y * fx) * wy0 + (t01.y * wx0 + t11.y * fx) * fy};
}
static void sky_lighting(const Scene& s, const incr_sky_lighting_desc& d) {
    (void)s;
    const incr_lighting_inputs& in = d.in;
    const float* sh9rgba = d.sh9_rgba; const uint8_t* aoTex = d.ambient_occlusion;
    const float skyBrightness = d.sky_brightness; const int diffuseIbl = d.diffuse_ibl;
    const bool specular = d.specular_ibl && d.prefiltered_envmap && d.env_brdf;
    const CubeChain env = {d.envmap_res, d.envmap_mips, d.prefiltered_envmap};
    const incr_view_constants& vc = *in.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 ivp = to_mat4(vc.sky_inv_view_proj);
    const float3 eye = {vc.eye_pos[0], vc.eye_pos[1], vc.eye_pos[2]};
    float3 sh[9];
    if (sh9rgba) for (int k = 0; k < 9; ++k) sh[k] = {sh9rgba[k * 4], sh9rgba[k * 4 + 1], sh9rgba[k * 4 + 2]};
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const GBufferPixel g = read_gbuffer(in, i);
        float3 radiance = {0, 0, 0};
        if (diffuseIbl && sh9rgba) {
            const float3 albedo = (1.f - g.metallic) * g.basecolor;
            float3 samplingNormal = g.N;
            if (d.use_bent_normals) {                                                          // USE_BENT_NORMALS, sky-lighting.hlsl:88-92
                const uint32_t b = d.bent_normals ? d.bent_normals[i] : 0x3c003c00u;          // HBAO off: cleared to (1, 1)
                samplingNormal = OctDecode({f16_to_f32((uint16_t)(b & 0xffffu)), f16_to_f32((uint16_t)(b >> 16))});
            }
            const float3 shDiffuse = ((albedo * Fd_Lambert()) * ShIrradiance(samplingNormal, sh)) * 5.f;
            const float3 occluded = g.ao * shDiffuse;
            radiance += skyBrightness * (shDiffuse + g.aoBlend * (occluded - shDiffuse));      // lerp(a, b, t) = a + t (b - a)
        }
        if (specular) {                                                                        // sky-lighting.hlsl:98-116
            float2 sp = screen_pos(x, y, W, H); sp.y = -sp.y;
            const float4 wp = mul_vec4({sp.x, sp.y, in.depth_buffer[i], 1.f}, ivp);
            const float3 worldPos = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
            const float3 V = normalize(eye - worldPos);
            const float NoV = hmax(dot(g.N, V), 0.01f);
            const float3 F0 = g.metallic * g.basecolor + (1.f - g.metallic) * 0.04f;
            const float3 R = normalize(reflect(-V, g.N));
            const float3 prefilteredColor = cube_sample_trilinear(env, R, g.roughness * ((float)d.envmap_mips - 1.f));
            const float2 envBrdf = sample_rg16f_bilinear_clamp(d.env_brdf, d.env_brdf_width, d.env_brdf_height, {NoV, g.roughness});
            const float3 specularIBL = prefilteredColor * (F0 * envBrdf.x + envBrdf.y);
            const float3 occludedSpec = g.ao * specularIBL;
            radiance += skyBrightness * (specularIBL + g.aoBlend * (occludedSpec - specularIBL));
        }
        const float hbao = aoTex ? (float)aoTex[i] / 255.f : 1.f;
        add_color(in.color_target, i, hbao * radiance, in.storage_rounding);
    }
}

// ------------------------------------------------------------------------------------------------ dynamic sky background
// shaders/environment-sky/dynamic-sky.hlsl:21-49 + render-jobs/dynamic-sky.inl:100-102,119-153: full-screen triangle at
// z = 0.0001, depth GREATER_EQUAL, blend off; the sky direction comes from the SCREEN uv in lat-long parameterisation
// (common/math.hlsli:46-79), not from the view ray.
static void dynamic_sky(const Scene& s, const incr_view_constants& vc, uint32_t* color, const float* depth, float turbidity, int rounding) {
    const uint32_t W = vc.res_x, H = vc.res_y;
    incr_perez_distribution pc; perez_from_turbidity(turbidity, &pc);
    const float3 L = normalize(f3(s.d.sun_dir[0], s.d.sun_dir[1], s.d.sun_dir[2]));
    const SkySunTerms sun = sky_sun_terms(L, turbidity, pc);
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        if (!(0.0001f >= depth[i])) continue;
        const float2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
        const float theta = k_Pi * uv.y, phi = k_Pi * (uv.x * 2.f - 1.f);
        const float sint = sinf(theta), cost = cosf(theta), sinp = sinf(phi), cosp = cosf(phi);
        const float3 dir = {sint * sinp, cost, sint * cosp};
        color[i] = pack_r11g11b10(1000.f * CalculateSkyRadianceRGB(sun, normalize(dir), pc), rounding);
    }
}

// ------------------------------------------------------------------------------------------------ forward lighting
// render-jobs/forward-pass.inl:17-190 + shaders/geo-raster/forward-pass.hlsl (FConfig::ForwardLighting): clears colour to
// (0.8, 0.8, 1) and depth to 0, draws every primitive of every visible mesh (no frustum culling; cull mode from the
// material; depth GREATER_EQUAL with writes; clip(opacity - 0.5)) and shades in the pixel shader.  The nearest fragment per
// pixel is found with the primitive visibility pass in forward order; vertex-shader outputs are interpolated with the
// G-buffer pass's screen-space barycentric formula (the pinned attribute interpolation).
static void forward_lighting(const Scene& s, const incr_forward_lighting_desc& d, uint64_t* shadowRays) {
    const incr_view_constants& vc = *d.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    std::vector<incr_draw_cmd> cmds;
    for (uint32_t id = 0; id < s.d.num_primitives; ++id) {
        const incr_gpu_primitive& pr = s.d.primitives[id];
        if (s.d.mesh_visibility && !s.d.mesh_visibility[pr.meshIndex]) continue;               // forward-pass.inl:70-74 bHidden
        cmds.push_back({id, pr.indexCount > 0 ? (uint32_t)pr.indexCount : 0u});
    }
    const uint32_t counts[2] = {(uint32_t)cmds.size(), 0u};
    std::vector<uint32_t> ids((size_t)W * H, 0xFFFFF000u);
    visibility_pass(s, vc, cmds.data(), nullptr, counts, ids.data(), d.depth_buffer, true, true);
    const Mat4 viewProj = to_mat4(vc.view_proj_transform);
    const float3 eye = {vc.eye_pos[0], vc.eye_pos[1], vc.eye_pos[2]};
    const bool specular = d.specular_ibl && d.prefiltered_envmap && d.env_brdf;
    const CubeChain env = {d.envmap_res, d.envmap_mips, d.prefiltered_envmap};
    float3 sh[9];
    if (d.sh9_rgba) for (int k = 0; k < 9; ++k) sh[k] = {d.sh9_rgba[k * 4], d.sh9_rgba[k * 4 + 1], d.sh9_rgba[k * 4 + 2]};
    RayStats rs;
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const uint32_t v = ids[i];
        if (v == 0xFFFFF000u) { d.color_target[i] = pack_r11g11b10({0.8f, 0.8f, 1.f}, d.storage_rounding); continue; }
        const incr_gpu_primitive& pr = s.d.primitives[v >> 20];
        uint32_t vi[3]; GetUint3(s, (int)((v & 0xfffffu) * 3u), pr.indexAccessor, vi);
        const Mat4 localToWorld = mesh_local_to_world(s, pr.meshIndex);
        float4 clip[3], wpos[3]; float2 uv[3]; float3 nW[3], tW[3], bW[3];
        for (int k = 0; k < 3; ++k) {                                                          // vs_main :57-90
            const float3 pos = GetFloat3(s, (int)vi[k], pr.positionAccessor);
            wpos[k] = mul_vec4({pos.x, pos.y, pos.z, 1.f}, localToWorld);
            clip[k] = mul_vec4(wpos[k], viewProj);
            uv[k] = GetFloat2(s, (int)vi[k], pr.uvAccessor);
            const float3 normal = GetFloat3(s, (int)vi[k], pr.normalAccessor);
            const float4 pt = GetFloat4(s, (int)vi[k], pr.tangentAccessor);
            const float3 tangent = {pt.x, pt.y, pt.z};
            const float3 bitangent = cross(normal, tangent) * pt.w;
            const float4 n4 = mul_vec4({normal.x, normal.y, normal.z, 0.f}, localToWorld), t4 = mul_vec4({tangent.x, tangent.y, tangent.z, 0.f}, localToWorld);
            const float4 b4 = mul_vec4({bitangent.x, bitangent.y, bitangent.z, 0.f}, localToWorld);
            nW[k] = {n4.x, n4.y, n4.z}; tW[k] = {t4.x, t4.y, t4.z}; bW[k] = {b4.x, b4.y, b4.z};
        }
        const float3 lam = CalcBarycentrics(clip[0], clip[1], clip[2], pixel_ndc(x, y, (float)W, (float)H));
        const float2 UV = interp2(uv[0], uv[1], uv[2], lam);
        const incr_material& mat = s.d.materials[pr.materialIndex];
        const UvGrad gr = quad_uv_gradient(clip[0], clip[1], clip[2], uv[0], uv[1], uv[2], x, y, (float)W, (float)H);
        MatProps p = EvaluateMaterialPropertiesGrad(s, mat, UV, gr.ddx, gr.ddy);
        float3 out = {0.f, 0.f, 0.f};
        bool done = true;
        if (d.viewmode == 2) out = f3(p.roughness);                                            // ps_main :95-105 (VIEWMODE 1 is a typo there: no effect)
        else if (d.viewmode == 3) out = f3(p.metallic);
        else if (d.viewmode == 4) out = p.basecolor;
        else if (d.viewmode == 5) out = p.emissive;
        else done = false;
        if (!done) {
            const float3 T = normalize(interp3(tW[0], tW[1], tW[2], lam)), B = normalize(interp3(bW[0], bW[1], bW[2], lam));
            float3 N = normalize(interp3(nW[0], nW[1], nW[2], lam));
            if (p.bHasNormalmap) {
                const float3 nm = p.normalmap;
                N = normalize(f3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
            }
            const float3 wp = interp3({wpos[0].x, wpos[0].y, wpos[0].z}, {wpos[1].x, wpos[1].y, wpos[1].z}, {wpos[2].x, wpos[2].y, wpos[2].z}, lam);
            const float ww = (lam.x * wpos[0].w + lam.y * wpos[1].w) + lam.z * wpos[2].w;
            const float3 worldPos = {wp.x / ww, wp.y / ww, wp.z / ww};
            const float3 V = normalize(eye - worldPos);
            float3 radiance = p.emissive * 20000;
            if (d.viewmode == 7 && d.prefiltered_envmap) {                                     // Reflections :123-131
                const float3 R = normalize(reflect(-V, N));
                out = cube_sample_trilinear(env, R, 0.f);
                done = true;
            }
            if (!done) {
                if (d.direct_lighting) {                                                       // :133-147, input.worldPos.xyz (not divided)
                    for (uint32_t lightIndex = 0; lightIndex < s.d.num_scene_lights; ++lightIndex) {
                        const int lightId = s.d.light_indices[lightIndex];
                        incr_light light{}; Mat4 lightTransform{};
                        if ((uint32_t)lightId < s.d.num_global_lights) light = s.d.global_lights[lightId];
                        if ((uint32_t)lightId < s.d.num_scene_lights) lightTransform = to_mat4(s.d.light_transforms[lightId]);
                        radiance += GetDirectRadiance(s, light, lightTransform, wp, p.basecolor, p.metallic, p.roughness, N, V, false, &rs);
                    }
                }
                float transmission = mat.transmissionFactor;                                   // material/common.hlsli:83-88
                if (mat.transmissionTextureIndex != -1) transmission *= SampleGrad(s, mat.transmissionTextureIndex, UV, gr.ddx, gr.ddy).x;
                if (d.diffuse_ibl && d.sh9_rgba) {                                             // :149-165
                    const float3 albedo = ((1.f - p.metallic) * (1.f - transmission)) * p.basecolor;
                    const float3 shDiffuse = (albedo * Fd_Lambert()) * ShIrradiance(N, sh);
                    radiance += shDiffuse + p.aoblend * (p.ao * shDiffuse - shDiffuse);
                }
                if (specular) {                                                                // :167-185
                    const float NoV = hmax(dot(N, V), 0.01f);
                    const float3 F0 = p.metallic * p.basecolor + (1.f - p.metallic) * 0.04f;
                    const float3 R = normalize(reflect(-V, N));
                    const float3 prefilteredColor = cube_sample_trilinear(env, R, p.roughness * ((float)d.envmap_mips - 1.f));
                    const float2 envBrdf = sample_rg16f_bilinear_clamp(d.env_brdf, d.env_brdf_width, d.env_brdf_height, {NoV, p.roughness});
                    const float3 specularIBL = prefilteredColor * (F0 * envBrdf.x + envBrdf.y);
                    radiance += specularIBL + p.aoblend * (p.ao * specularIBL - specularIBL);
                }
                out = radiance;
            }
        }
        d.color_target[i] = pack_r11g11b10(out, d.storage_rounding);
    }
    if (shadowRays) *shadowRays += rs.lightShadow;
}

// ------------------------------------------------------------------------------------------------ HDRI background
// shaders/environment-sky/envmap.hlsl:22-54 + render-jobs/environmentmap.inl:100-146: full-screen triangle at z = 0.0001,
// depth GREATER_EQUAL, blend off; the interpolated clip position of the pixel centre goes through
// ((V without translation) * P)^-1 and the cube is sampled along it (pinned: level 0, DESIGN.md 3 texture unit).
static void environmentmap_pass(const incr_environmentmap_desc& d) {
    const incr_view_constants& vc = *d.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 inv = to_mat4(vc.envmap_inv_parallax_view_proj);
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        if (!(0.0001f >= d.depth_buffer[i])) continue;
        const float2 ndc = pixel_ndc(x, y, (float)W, (float)H);
        float4 wp = mul_vec4({ndc.x, ndc.y, 0.0001f, 1.f}, inv);
        const float3 dir = {wp.x / wp.w, wp.y / wp.w, wp.z / wp.w};
        const float3 luminance = d.sky_brightness * env_cube_level0(d.envmap_r11g11b10, d.envmap_res, d.envmap_mips, dir);
        d.color_target[i] = pack_r11g11b10(luminance, d.storage_rounding);
    }
}

// ------------------------------------------------------------------------------------------------ debug views
// shaders/postprocess/debug-visualization.hlsl:52-203 + render-jobs/debug-visualization.inl: a full-screen pass without
// depth test or blending that overwrites the HDR colour target when Viewmode is neither Normal nor LightingOnly.
// (The mouse-pick store into the highlight pass's indirect args, :97-131, is UI and not restated.)
static float3 hsv2rgb(float3 hsv) {                                           // common/color-space.hlsli:2-48
    float3 sat;
    while (hsv.x < 0) hsv.x += 360;
    while (hsv.x > 360) hsv.x -= 360;
    if (hsv.x < 120) { sat.x = (120 - hsv.x) / 60.0f; sat.y = hsv.x / 60.0f; sat.z = 0; }
    else if (hsv.x < 240) { sat.x = 0; sat.y = (240 - hsv.x) / 60.0f; sat.z = (hsv.x - 120) / 60.0f; }
    else { sat.x = (hsv.x - 240) / 60.0f; sat.y = 0; sat.z = (360 - hsv.x) / 60.0f; }
    sat.x = hmin(sat.x, 1.f); sat.y = hmin(sat.y, 1.f); sat.z = hmin(sat.z, 1.f);
    return {((1 - hsv.y) + hsv.y * sat.x) * hsv.z, ((1 - hsv.y) + hsv.y * sat.y) * hsv.z, ((1 - hsv.y) + hsv.y * sat.z) * hsv.z};
}
static float DebugHalton(uint32_t sampleIndex, uint32_t base) {               // common/uniform-sampling.hlsli:123-136
    float result = 0.f, f = 1.f;
    while (sampleIndex > 0) { f = f / (float)base; result += f * (float)(sampleIndex % base); sampleIndex = sampleIndex / base; }
    return result;
}
static void debug_visualization(const incr_debug_viz_desc& d) {
    const incr_view_constants& vc = *d.view;
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 ip = to_mat4(vc.inv_proj_transform);
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        float3 c = {0.f, 0.f, 0.f};
        switch (d.viewmode) {
        case 2: { const float g = unpack_unorm8x4(d.gbuffer_metallic_roughness_ao[i]).y; c = {g, g, g}; break; }
        case 3: { const float r = unpack_unorm8x4(d.gbuffer_metallic_roughness_ao[i]).x; c = {r, r, r}; break; }
        case 4: { const float4 b = unpack_unorm8x4(d.gbuffer_basecolor[i]); c = {b.x, b.y, b.z}; break; }
        case 8: case 9: {
            const uint32_t v = d.vis_buffer[i];
            const uint32_t objectId = d.use_primitives ? v >> 20 : v >> 7, triangleId = d.use_primitives ? v & 0xfffffu : v & 127u;
            c = d.viewmode == 8 ? hsv2rgb({DebugHalton(objectId, 3) * 360.f, 0.85f, 0.95f}) : hsv2rgb({DebugHalton(triangleId, 5) * 360.f, 0.85f, 0.95f});
            break;
        }
        case 10: case 13: {
            const uint32_t n = d.viewmode == 10 ? d.gbuffer_normals[i] : (d.bent_normals ? d.bent_normals[i] : 0x3c003c00u);
            const float3 N = OctDecode({f16_to_f32((uint16_t)(n & 0xffffu)), f16_to_f32((uint16_t)(n >> 16))});
            c = {N.x * 0.5f + 0.5f, N.y * 0.5f + 0.5f, N.z * 0.5f + 0.5f};
            break;
        }
        case 11: {
            const float2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
            const float4 ndc = {2.f * uv.x - 1.f, -2.f * uv.y + 1.f, hmax(0.001f, d.depth_buffer[i]), 1.f};
            const float4 pv = mul_vec4(ndc, ip);
            const float z = pv.z / pv.w;
            const float numSlices = (float)d.light_cluster_slices;
            const float scale = numSlices / logf(1000.f / 5.f), bias = -logf(5.f) * scale;
            const uint32_t sliceIndex = ftou(floorf(scale * logf(z) + bias));
            c = hsv2rgb({(float)sliceIndex / numSlices * 360.f, 0.85f, 0.95f});
            break;
        }
        case 12: { const float a = d.ambient_occlusion ? (float)d.ambient_occlusion[i] / 255.f : 1.f; c = {a, a, a}; break; }
        default: break;                                                       // Emissive / NanCheck / Reflections: return 0.xxxx
        }
        d.color_target[i] = pack_r11g11b10(c, d.storage_rounding);
    }
}

// ------------------------------------------------------------------------------------------------ TAA resolve
static inline float3 load_hdr(const uint32_t* img, uint32_t W, uint32_t H, uint32_t x, uint32_t y) {       // Texture2D.Load: out of bounds -> 0
    if (x >= W || y >= H) return {0, 0, 0};
    return unpack_r11g11b10(img[(size_t)y * W + x]);
}
static float3 sample_bilinear_clamp(const uint32_t* img, uint32_t W, uint32_t H, float2 uv) {
    float u = uv.x, v = uv.y;
    if (!(u == u)) u = 0.f;
    if (!(v == v)) v = 0.f;
    u = fminf(fmaxf(u, -2.f), 3.f); v = fminf(fmaxf(v, -2.f), 3.f);
    int xi = linear_fixed(u * (float)W), yi = linear_fixed(v * (float)H), xf, yf; float fx, fy;          // one texel beyond the frame on either side, in fixed point
    xi = std::min(std::max(xi, -256), (int)W * 256); yi = std::min(std::max(yi, -256), (int)H * 256);
    linear_split(xi, &xf, &fx); linear_split(yi, &yf, &fy);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); };
    const int x0 = cl(xf, (int)W), x1 = cl(xf + 1, (int)W), y0 = cl(yf, (int)H), y1 = cl(yf + 1, (int)H);
    const float3 t00 = load_hdr(img, W, H, x0, y0), t10 = load_hdr(img, W, H, x1, y0), t01 = load_hdr(img, W, H, x0, y1), t11 = load_hdr(img, W, H, x1, y1);
    const float wx0 = 1.f - fx, wy0 = 1.f - fy;
    return (t00 * wx0 + t10 * fx) * wy0 + (t01 * wx0 + t11 * fx) * fy;
}
static inline float lum_weight(float3 c) { return 1.f / (1.f + ((c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f)); }
// shaders/postprocess/taa-resolve.hlsl:34-117 (history read from `in`, result written to `out`)
static void taa_resolve(const incr_view_constants& vc, const uint32_t* hdr, const uint32_t* histIn, uint32_t* histOut, const float* depthTex,
                        uint32_t historyIndex, int rounding) {
    const uint32_t W = vc.res_x, H = vc.res_y;
    const Mat4 inv = to_mat4(vc.taa_inv_view_proj), prev = to_mat4(vc.prev_view_proj_transform);
    uint32_t y0, y1; scissor_rows(vc, &y0, &y1);
    for (uint32_t y = y0; y < y1; ++y) for (uint32_t x = 0; x < W; ++x) {
        const size_t i = (size_t)y * W + x;
        const float3 cur = unpack_r11g11b10(hdr[i]);
        if (historyIndex == 0) { histOut[i] = pack_r11g11b10(cur, rounding); continue; }
        const float2 uv = {((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H};
        const float4 clip = {2.f * uv.x - 1.f, -2.f * uv.y + 1.f, depthTex[i], 1.f};
        const float4 world = mul_vec4(clip, inv);
        float4 pc = mul_vec4(world, prev);
        pc = {pc.x / pc.w, pc.y / pc.w, pc.z / pc.w, pc.w / pc.w};
        const float2 ruv = {0.5f * pc.x + 0.5f, -0.5f * pc.y + 0.5f};
        const float3 previous = sample_bilinear_clamp(histIn, W, H, ruv);
        float3 mn = f3(10000.f), mx = f3(-10000.f);
        for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) {
            const float3 c = load_hdr(hdr, W, H, ftou((float)x + (float)dx), ftou((float)y + (float)dy));
            mn = {hmin(mn.x, c.x), hmin(mn.y, c.y), hmin(mn.z, c.z)};
            mx = {hmax(mx.x, c.x), hmax(mx.y, c.y), hmax(mx.z, c.z)};
        }
        const float3 clamped = {hmin(hmax(previous.x, mn.x), mx.x), hmin(hmax(previous.y, mn.y), mx.y), hmin(hmax(previous.z, mn.z), mx.z)};
        const float cw = 0.1f * lum_weight(cur), pw = 0.9f * lum_weight(clamped);
        float3 out = cur * cw + clamped * pw;
        out = out / (cw + pw);
        histOut[i] = pack_r11g11b10(out, rounding);
    }
}

// ------------------------------------------------------------------------------------------------ C exports
extern "C" {
void orc_pixel_jitter(uint32_t frameIndex, uint32_t resX, uint32_t resY, float out[2]) {       // renderer.cpp:126-131
    const uint32_t i = frameIndex % 16u;
    out[0] = 2.f * (RasterHalton(i, 2) - 0.5f) / (float)resX;
    out[1] = 2.f * (RasterHalton(i, 3) - 0.5f) / (float)resY;
}
void orc_view_constants(const incr_view* view, const incr_view* cull, const float jitter[2], const incr_matrix* prev, float nearPlane,
                        uint32_t resX, uint32_t resY, incr_view_constants* out) { make_view_constants(view, cull, jitter, prev, nearPlane, resX, resY, out); }
void orc_batch_cull(void* scene, const incr_batch_cull_desc* d, int32_t* culled) {
    batch_cull(*(Scene*)scene, *d->view, d->frustum_culling, d->default_list, d->double_sided_list, d->counts, culled, d->use_primitives != 0);
}
void orc_light_cull(void* scene, const incr_light_cull_desc* d, int32_t* culled) {
    light_cull(*(Scene*)scene, *d->view, d->cluster_dim, d->max_lights_per_cluster, d->cluster_depth_extent, d->culled_light_lists, d->light_grid,
               d->culled_light_count, culled);
}
void orc_visibility(void* scene, const incr_visibility_desc* d) {
    visibility_pass(*(Scene*)scene, *d->view, d->default_list, d->double_sided_list, d->counts, d->vis_buffer, d->depth_buffer, d->use_primitives != 0);
}
void orc_gbuffer(void* scene, const incr_gbuffer_desc* d) {
    gbuffer_pass(*(Scene*)scene, *d->view, d->vis_buffer, d->gbuffer_basecolor, d->gbuffer_normals, d->gbuffer_metallic_roughness_ao, d->color_target, d->storage_rounding, d->use_primitives != 0);
}
// KAT hook: the six planes (near, far, left, right, bottom, top) of one light cluster, 24 floats
void orc_cluster_frustum(const uint32_t ci[3], const uint32_t grid[3], float zNear, float zFar, const incr_matrix* proj, const incr_matrix* invViewProj, float out[24]) {
    const Frustum6 f = GetClusterFrustum(ci, grid, zNear, zFar, to_mat4(*proj), to_mat4(*invViewProj));
    memcpy(out, &f, sizeof f);
}
void orc_environmentmap(const incr_environmentmap_desc* d) { environmentmap_pass(*d); }
void orc_debug_visualization(const incr_debug_viz_desc* d) { debug_visualization(*d); }
void orc_gbuffer_decals(void* scene, const incr_gbuffer_decals_desc* d) { gbuffer_decals(*(Scene*)scene, *d); }
void orc_forward_lighting(void* scene, const incr_forward_lighting_desc* d, uint64_t* shadowRays) { forward_lighting(*(Scene*)scene, *d, shadowRays); }
void orc_hbao(void* scene, const incr_hbao_desc* d, uint64_t* rays) { hbao_pass(*(Scene*)scene, *d, rays); }
void orc_hb_math(float x, float out[3]) { out[0] = hb_sin(x); out[1] = hb_cos(x); out[2] = hb_acos(x); }
void orc_sky_lighting(void* scene, const incr_sky_lighting_desc* d) {
    sky_lighting(*(Scene*)scene, *d);
}
void orc_direct_lighting(void* scene, const incr_direct_lighting_desc* d, uint64_t* shadowRays) {
    direct_lighting(*(Scene*)scene, d->in, d->directional_light_index, shadowRays);
}
void orc_clustered_lighting(void* scene, const incr_clustered_lighting_desc* d) {
    clustered_lighting(*(Scene*)scene, d->in, d->culled_light_lists, d->light_grid, d->cluster_dim, d->cluster_depth_extent, d->requires_clear);
}
void orc_dynamic_sky(void* scene, const incr_dynamic_sky_desc* d) {
    dynamic_sky(*(Scene*)scene, *d->view, d->color_target, d->depth_buffer, d->turbidity, d->storage_rounding);
}
void orc_taa_resolve(const incr_taa_resolve_desc* d) {
    taa_resolve(*d->view, d->hdr_scene_color, d->history_in, d->history_out, d->depth_buffer, d->history_index, d->storage_rounding);
}
void orc_oct_roundtrip(const float n[3], float enc[2], float dec[3]) {
    const float2 e = OctEncode({n[0], n[1], n[2]}); enc[0] = e.x; enc[1] = e.y;
    const float3 d = OctDecode(e); dec[0] = d.x; dec[1] = d.y; dec[2] = d.z;
}
} // extern "C"
