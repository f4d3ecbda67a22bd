// raster_api.hip -- C ABI entry points of the raster path (include/incr.h "Raster path"): argument checks, kernel
// argument blocks, launches.  Host code only; the kernels are in raster.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "api_internal.h"
#include "test_hooks.h"

using namespace incr;

extern "C" hipError_t incr_launch_mesh_frames(const RasterScene* S, const ViewFrame* vf, MeshFrame* out, hipStream_t stream);
extern "C" hipError_t incr_launch_batch_cull(const RasterScene* S, const MeshFrame* frames, int frustumCulling, incr_draw_cmd* defList, incr_draw_cmd* dsList,
                                             uint32_t* counts, incr_render_stats* stats, const PrimDraw* prims, uint32_t numPrims, hipStream_t stream);
extern "C" hipError_t incr_launch_light_cull(const LightCullParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_visibility(const VisParams* p, uint32_t maxJobs, int numCUs, uint32_t* vis, float* depth, hipStream_t stream);
extern "C" hipError_t incr_launch_gbuffer(const GBufferParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_sky_lighting(const SkyParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_direct_lighting(const DirectParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_clustered_lighting(const ClusteredParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_deferred_lighting(const FusedLightingParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_hbao(const HbaoParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_forward_shade(const ForwardParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_decals(const DecalParams* p, int numCUs, hipStream_t stream);
extern "C" hipError_t incr_launch_environmentmap(const EnvmapParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_debug_viz(const DebugVizParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_taa_resolve(const TaaParams* p, hipStream_t stream);
extern "C" hipError_t incr_launch_sun_cover(const PtParams* S, uint32_t sunIndex, SunCoverHeader* header, uint32_t* grid, hipStream_t stream);
extern "C" size_t incr_sun_cover_bytes();
extern "C" hipError_t incr_launch_dynamic_sky(const DynamicSkyParams* p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------- scene data
namespace {

// MeshMaterial::GetFloat{2,3,4} (shaders/common/mesh-material.hlsli:92-170): stride-checked typed fetch, 0 when absent
const uint8_t* accessor_base(const HostScene& s, int accessorIndex, uint32_t* stride, size_t* avail) {
    if (accessorIndex < 0 || (size_t)accessorIndex >= s.accessors.size()) return nullptr;
    const incr_mesh_accessor& a = s.accessors[accessorIndex];
    if (a.bufferViewIndex < 0 || (size_t)a.bufferViewIndex >= s.bufferViews.size()) return nullptr;
    const incr_mesh_buffer_view& v = s.bufferViews[a.bufferViewIndex];
    if (v.bufferSrvIndex < 0 || (size_t)v.bufferSrvIndex >= s.buffers.size()) return nullptr;
    const std::vector<uint8_t>& b = s.buffers[v.bufferSrvIndex];
    const size_t off = (size_t)a.byteOffset + v.byteOffset;
    if (off > b.size()) return nullptr;
    *stride = a.byteStride; *avail = b.size() - off;
    return b.data() + off;
}
void fetch(const HostScene& s, int accessorIndex, uint32_t index, int ncomp, float* out) {
    for (int i = 0; i < 4; ++i) out[i] = 0.f;
    uint32_t stride = 0; size_t avail = 0;
    const uint8_t* base = accessor_base(s, accessorIndex, &stride, &avail);
    if (!base) return;
    bool ok = false;
    if (ncomp == 4) ok = stride == 16;
    else if (ncomp == 3) ok = stride == 12 || stride == 16;
    else ok = stride == 8 || stride == 12 || stride == 16;
    if (!ok || (size_t)index * stride + (size_t)ncomp * 4 > avail) return;
    memcpy(out, base + (size_t)index * stride, (size_t)ncomp * 4);
}

} // namespace

struct BakedPrims {
    std::vector<PrimDraw> draws; std::vector<uint32_t> indices, chunkPrim, chunkFirstTri;
    std::vector<float4> pos, nrm, tan; std::vector<float2> uv;
};
static int bake_primitives(const HostScene& h, const std::vector<incr_gpu_primitive>& prims, const char* what, BakedPrims* out);

int incr_raster_scene_upload(incr_scene* sc) {
    const HostScene& h = sc->host;
    const size_t nm = h.meshlets.size(), nv = h.meshletVertexIndices.size(), nt = h.meshletPackedTriangles.size();
    for (size_t i = 0; i < nm; ++i) {
        const incr_gpu_meshlet& m = h.meshlets[i];
        if ((size_t)m.vertexBegin + m.vertexCount > nv || (size_t)m.triangleBegin + m.triangleCount > nt)
            return fail(INCR_ERR_INVALID_ARGUMENT, "meshlet %zu: vertex/triangle range outside the packed buffers", i);
        if (m.vertexCount > 256 || m.triangleCount > 128 || (m.triangleCount && !m.vertexCount))
            return fail(INCR_ERR_INVALID_ARGUMENT, "meshlet %zu: %u vertices / %u triangles exceed the 8-bit vertex and 7-bit triangle ids of the visibility buffer", i, m.vertexCount, m.triangleCount);
        if (m.meshIndex < 0 || (size_t)m.meshIndex >= h.meshTransforms.size()) return fail(INCR_ERR_INVALID_ARGUMENT, "meshlet %zu: meshIndex out of range", i);
        if (m.materialIndex < 0 || (size_t)m.materialIndex >= h.materials.size()) return fail(INCR_ERR_INVALID_ARGUMENT, "meshlet %zu: materialIndex out of range", i);
        for (uint32_t t = 0; t < m.triangleCount; ++t) {
            const uint32_t p = h.meshletPackedTriangles[m.triangleBegin + t];
            for (int k = 0; k < 3; ++k) if ((0xffu & (p >> (k * 10))) >= m.vertexCount) return fail(INCR_ERR_INVALID_ARGUMENT, "meshlet %zu triangle %u: local vertex index out of range", i, t);
        }
    }
    if (nm >= (1u << 24)) return fail(INCR_ERR_INVALID_ARGUMENT, "more than 2^24 meshlets do not fit the visibility pass's order key");
    // resolve the accessor indirection once: per packed meshlet vertex, the attributes the shaders fetch
    std::vector<float4> pos(nv), nrm(nv), tan(nv); std::vector<float2> uv(nv);
    std::vector<uint8_t> seen(nv, 0);
    for (size_t i = 0; i < nm; ++i) {
        const incr_gpu_meshlet& m = h.meshlets[i];
        for (uint32_t v = 0; v < m.vertexCount; ++v) {
            const size_t slot = (size_t)m.vertexBegin + v;
            const uint32_t index = h.meshletVertexIndices[slot];
            float f[4];
            fetch(h, m.positionAccessor, index, 3, f); pos[slot] = make_float4(f[0], f[1], f[2], 0.f);
            fetch(h, m.uvAccessor, index, 2, f); uv[slot] = make_float2(f[0], f[1]);
            fetch(h, m.normalAccessor, index, 3, f); nrm[slot] = make_float4(f[0], f[1], f[2], 0.f);
            fetch(h, m.tangentAccessor, index, 4, f); tan[slot] = make_float4(f[0], f[1], f[2], f[3]);
            seen[slot] = 1;
        }
    }
    int rc = upload_vec(&sc->dMeshlets, h.meshlets.data(), nm);
    if (!rc) rc = upload_vec(&sc->dPackedTris, h.meshletPackedTriangles.data(), nt);
    if (!rc) rc = upload_vec(&sc->dMeshVisibility, h.meshVisibility.data(), h.meshVisibility.size());
    if (!rc) rc = upload_vec(&sc->dMvPos, pos.data(), nv);
    if (!rc) rc = upload_vec(&sc->dMvUv, uv.data(), nv);
    if (!rc) rc = upload_vec(&sc->dMvNormal, nrm.data(), nv);
    if (!rc) rc = upload_vec(&sc->dMvTangent, tan.data(), nv);
    if (rc) return rc;

    BakedPrims bp;
    if ((rc = bake_primitives(h, h.primitives, "primitive", &bp))) return rc;
    sc->numPrimChunks = (uint32_t)bp.chunkPrim.size();
    rc = upload_vec(&sc->dPrimDraws, bp.draws.data(), bp.draws.size());
    if (!rc) rc = upload_vec(&sc->dPrimIndices, bp.indices.data(), bp.indices.size());
    if (!rc) rc = upload_vec(&sc->dPvPos, bp.pos.data(), bp.pos.size());
    if (!rc) rc = upload_vec(&sc->dPvUv, bp.uv.data(), bp.uv.size());
    if (!rc) rc = upload_vec(&sc->dPvNormal, bp.nrm.data(), bp.nrm.size());
    if (!rc) rc = upload_vec(&sc->dPvTangent, bp.tan.data(), bp.tan.size());
    if (!rc) rc = upload_vec(&sc->dChunkPrim, bp.chunkPrim.data(), bp.chunkPrim.size());
    if (!rc) rc = upload_vec(&sc->dChunkFirstTri, bp.chunkFirstTri.data(), bp.chunkFirstTri.size());
    if (rc) return rc;

    // ---- decals (FScene::m_sceneMeshDecals): the same bake; meshIndex of a decal draw indexes dDecalLocalToWorld ----
    BakedPrims bd;
    std::vector<incr_gpu_primitive> decals = h.decalPrimitives;
    for (size_t i = 0; i < decals.size(); ++i) {
        if (decals[i].materialIndex < 0 || (size_t)decals[i].materialIndex >= h.materials.size()) return fail(INCR_ERR_INVALID_ARGUMENT, "decal primitive %zu: materialIndex out of range", i);
        decals[i].meshIndex = (int32_t)i;
    }
    if ((rc = bake_primitives(h, decals, "decal primitive", &bd))) return rc;
    std::vector<float> l2w(decals.size() * 16);
    for (size_t i = 0; i < decals.size(); ++i) {       // mul(g_primCb.localToWorld, g_sceneCb.m_sceneRotation), gbuffer-raster.hlsl:51
        const M44 m = m44_mul(h.decalTransforms[i], h.rootTransform);
        memcpy(&l2w[i * 16], m.m, 64);
    }
    sc->numDecalChunks = (uint32_t)bd.chunkPrim.size(); sc->numDecals = (uint32_t)decals.size();
    rc = upload_vec(&sc->dDecalDraws, bd.draws.data(), bd.draws.size());
    if (!rc) rc = upload_vec(&sc->dDecalIndices, bd.indices.data(), bd.indices.size());
    if (!rc) rc = upload_vec(&sc->dDvPos, bd.pos.data(), bd.pos.size());
    if (!rc) rc = upload_vec(&sc->dDvUv, bd.uv.data(), bd.uv.size());
    if (!rc) rc = upload_vec(&sc->dDvNormal, bd.nrm.data(), bd.nrm.size());
    if (!rc) rc = upload_vec(&sc->dDvTangent, bd.tan.data(), bd.tan.size());
    if (!rc) rc = upload_vec(&sc->dDecalChunkPrim, bd.chunkPrim.data(), bd.chunkPrim.size());
    if (!rc) rc = upload_vec(&sc->dDecalChunkFirstTri, bd.chunkFirstTri.data(), bd.chunkFirstTri.size());
    if (!rc) rc = upload_vec(&sc->dDecalLocalToWorld, l2w.data(), l2w.size());
    return rc;
}

// Per primitive: the index list widened to 32 bit (MeshMaterial::GetUint, mesh-material.hlsli:5-43: 2- or 4-byte
// indices), the attributes of vertices 0 .. max index, and a table that cuts the triangles into 64-triangle chunks.
static int bake_primitives(const HostScene& h, const std::vector<incr_gpu_primitive>& prims, const char* what, BakedPrims* out) {
    std::vector<PrimDraw>& draws = out->draws; draws.assign(prims.size(), PrimDraw{});
    std::vector<uint32_t>& indices = out->indices; std::vector<uint32_t>& chunkPrim = out->chunkPrim; std::vector<uint32_t>& chunkFirstTri = out->chunkFirstTri;
    std::vector<float4>& ppos = out->pos; std::vector<float4>& pnrm = out->nrm; std::vector<float4>& ptan = out->tan; std::vector<float2>& puv = out->uv;
    for (size_t pi = 0; pi < prims.size(); ++pi) {
        const incr_gpu_primitive& pr = prims[pi];
        PrimDraw& d = draws[pi];
        memset(&d, 0, sizeof d);
        d.indexBegin = (uint32_t)indices.size(); d.indexCount = pr.indexCount > 0 ? (uint32_t)pr.indexCount / 3u * 3u : 0u;
        d.vertexBase = (uint32_t)ppos.size(); d.meshIndex = pr.meshIndex; d.materialIndex = pr.materialIndex;
        memcpy(d.sphere, pr.boundingSphere, 16);
        d.pad[0] = pr.indexCount > 0 ? (uint32_t)pr.indexCount : 0u;     // the draw's vertexCount, batch-culling.hlsl:87
        uint32_t stride = 0; size_t avail = 0;
        const uint8_t* ib = accessor_base(h, pr.indexAccessor, &stride, &avail);
        uint32_t maxIndex = 0;
        for (uint32_t i = 0; i < d.indexCount; ++i) {
            uint32_t v = 0;
            if (ib && stride == 2 && (size_t)i * 2 + 2 <= avail) { uint16_t t; memcpy(&t, ib + (size_t)i * 2, 2); v = t; }
            else if (ib && stride == 4 && (size_t)i * 4 + 4 <= avail) memcpy(&v, ib + (size_t)i * 4, 4);
            indices.push_back(v); maxIndex = std::max(maxIndex, v);
        }
        if (d.indexCount / 3u > (1u << 20)) return fail(INCR_ERR_INVALID_ARGUMENT, "%s %zu has more than 2^20 triangles: does not fit the visibility encoding", what, pi);
        const uint32_t nverts = d.indexCount ? maxIndex + 1u : 0u;
        for (uint32_t v = 0; v < nverts; ++v) {
            float f[4];
            fetch(h, pr.positionAccessor, v, 3, f); ppos.push_back(make_float4(f[0], f[1], f[2], 0.f));
            fetch(h, pr.uvAccessor, v, 2, f); puv.push_back(make_float2(f[0], f[1]));
            fetch(h, pr.normalAccessor, v, 3, f); pnrm.push_back(make_float4(f[0], f[1], f[2], 0.f));
            fetch(h, pr.tangentAccessor, v, 4, f); ptan.push_back(make_float4(f[0], f[1], f[2], f[3]));
        }
        for (uint32_t t = 0; t < d.indexCount / 3u; t += 64) { chunkPrim.push_back((uint32_t)pi); chunkFirstTri.push_back(t); }
    }
    return INCR_OK;
}
int incr_raster_scene_update(incr_scene* sc, hipStream_t stream) {
    const HostScene& h = sc->host;
    // batch culling and the forward-pass flags read dMeshVisibility; decals were baked with meshTransform * rootTransform
    // (called inside incr_update_bvh's staging bracket: the sources are copied into the pinned slot before the call returns)
    int rc = INCR_OK;
    if (!h.meshVisibility.empty()) rc = incr_staging_upload(sc, sc->dMeshVisibility, h.meshVisibility.data(), h.meshVisibility.size() * 4, stream);
    if (!rc && sc->numDecals) {
        std::vector<float> l2w((size_t)sc->numDecals * 16);
        for (size_t i = 0; i < sc->numDecals; ++i) { const M44 m = m44_mul(h.decalTransforms[i], h.rootTransform); memcpy(&l2w[i * 16], m.m, 64); }
        rc = incr_staging_upload(sc, sc->dDecalLocalToWorld, l2w.data(), l2w.size() * 4, stream);
    }
    return rc;
}

void incr_raster_scene_free(incr_scene* sc) {
    void* ptrs[] = {sc->dMeshlets, sc->dPackedTris, sc->dMeshVisibility, sc->dMvPos, sc->dMvUv, sc->dMvNormal, sc->dMvTangent,
                    sc->dPrimDraws, sc->dPrimIndices, sc->dPvPos, sc->dPvUv, sc->dPvNormal, sc->dPvTangent, sc->dChunkPrim, sc->dChunkFirstTri,
                    sc->dDecalDraws, sc->dDecalIndices, sc->dDvPos, sc->dDvUv, sc->dDvNormal, sc->dDvTangent, sc->dDecalChunkPrim, sc->dDecalChunkFirstTri,
                    sc->dDecalLocalToWorld};
    for (void* p : ptrs) if (p) (void)hipFree(p);
}

// ---------------------------------------------------------------------------------------------- helpers
namespace {

void fill_raster_scene(const incr_context* ctx, const incr_scene* sc, RasterScene& S) {
    memset(&S, 0, sizeof S);
    S.meshlets = sc->dMeshlets; S.packedTris = sc->dPackedTris; S.meshVisibility = sc->dMeshVisibility;
    S.mvPos = sc->dMvPos; S.mvUv = sc->dMvUv; S.mvNormal = sc->dMvNormal; S.mvTangent = sc->dMvTangent;
    S.meshLocalToWorld = sc->dMeshLocalToWorld; S.materials = sc->dMaterials; S.textures = sc->dTextures; S.materialTexels = sc->dMaterialTexels; S.alphaQuads = sc->dAlphaQuads; S.srgbLut = ctx->srgbLut;
    S.numMeshlets = (uint32_t)sc->host.meshlets.size(); S.numMeshes = (uint32_t)sc->host.meshTransforms.size();
    S.maxAnisotropy = sc->host.maxAnisotropy;
}
// FConfig::UseMeshlets == false.  EncodePrimitiveVisibility keeps 12 bits of primitive id (encoding.hlsli:4-11): 4,096 primitives, of which the
// last one shares its id range with the clear value 0xFFFFF000 (visibility-pass.inl:68-74) from triangle 0xFF000 on -- in the reference too.
int fill_raster_prims(const incr_scene* sc, uint32_t* primFlags, RasterPrims& R) {
    const size_t n = sc->host.primitives.size();
    if (n > 4096) return fail(INCR_ERR_INVALID_ARGUMENT, "%zu primitives: EncodePrimitiveVisibility holds at most 4096 (use meshlets)", n);
    R.draws = sc->dPrimDraws; R.indices = sc->dPrimIndices;
    R.pvPos = sc->dPvPos; R.pvUv = sc->dPvUv; R.pvNormal = sc->dPvNormal; R.pvTangent = sc->dPvTangent;
    R.chunkPrim = sc->dChunkPrim; R.chunkFirstTri = sc->dChunkFirstTri; R.primFlags = primFlags;
    R.numPrims = (uint32_t)n; R.numChunks = sc->numPrimChunks;
    return INCR_OK;
}
void fill_view_frame(const incr_view_constants& vc, ViewFrame& vf) {
    memcpy(vf.viewProj, vc.view_proj_transform.m, 64); memcpy(vf.cullViewProj, vc.cull_view_proj_transform.m, 64);
    memcpy(vf.invViewProj, vc.inv_view_proj_transform.m, 64); memcpy(vf.invProj, vc.inv_proj_transform.m, 64);
    memcpy(vf.eye, vc.eye_pos, 12); vf.nearPlane = vc.near_plane; vf.W = vc.res_x; vf.H = vc.res_y;
}
// One allocation, grown on demand: [per-mesh frame constants][queue counters][packed depth|key surface][triangle
// records + alpha blocks][small / whole / band queues].  The frame constants sit at offset 0, so passes that only need
// them (pixels == 0) never move anything.
struct RasterWorkspace {
    MeshFrame* frames; uint32_t* counters; unsigned long long* packed;
    SubTriRec* recs; SubTriAlpha* alphas; uint32_t* smallQueue; uint32_t* wholeQueue; uint2* bigQueue;
    uint32_t recCap, bigCap;
    uint32_t* primFlags;                   // primitive path: one word per primitive
    uint32_t* ids;                         // forward pass: its own visibility ids (the frame's vis buffer keeps the deferred ids)
    uint32_t* binCount; uint32_t* binList; uint32_t binCap;      // screen bins (binCount sits right behind `counters`: one clear)
};
// Screen bins of a W-wide surface with `rows` scissor rows
static inline uint32_t vis_bin_shift(uint32_t W, uint32_t H) {
    uint32_t shift = (size_t)W * H >= ((size_t)1 << 22) ? 6u : 5u;
    if (const char* v = test_hook_env("INCR_VIS_BIN_SHIFT")) shift = strtoul(v, nullptr, 10) == 6 ? 6u : 5u;
    return shift;
}
static inline uint32_t vis_bins(uint32_t pixels, uint32_t shift) { return (pixels + (1u << shift) - 1u) >> shift; }
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
int ensure_raster_workspace(incr_context* ctx, size_t pixels, size_t meshes, size_t tris, hipStream_t stream, RasterWorkspace* ws, size_t prims = 0, size_t bins = 0) {
    size_t recCap = pixels ? 2 * tris + 64 : 0, bigCap = pixels ? recCap + pixels / 16 + 64 : 0;
    // test hook: a tiny band queue forces the overflow path (large triangles scanned whole by one wave)
    if (const char* cap = pixels ? test_hook_env("INCR_VIS_BAND_QUEUE_CAP") : nullptr) bigCap = (size_t)strtoul(cap, nullptr, 10) + 1;
    // test hook: a tiny bin capacity forces the bins' overflow path (the triangle takes the band queue), 0 turns the bins off
    size_t binCap = bins ? kVisBinCapDefault : 0;
    if (const char* cap = bins ? test_hook_env("INCR_VIS_BIN_CAP") : nullptr) binCap = (size_t)strtoul(cap, nullptr, 10);
    if (!binCap) bins = 0;
    const size_t szFrames = align256((meshes + 1) * sizeof(MeshFrame)), szCounters = align256(kVisCounterBytes) + align256(bins * kVisBinLists * 4), szPacked = align256(pixels * 8);
    const size_t szBinList = align256(bins * kVisBinLists * binCap * 4);
    const size_t szRecs = align256(recCap * sizeof(SubTriRec)), szAlpha = align256(recCap * sizeof(SubTriAlpha));
    const size_t szSmall = align256(recCap * 4), szWhole = szSmall, szBig = align256(bigCap * 8);
    const size_t szFlags = align256((prims + 1) * 4), szIds = align256(pixels * 4);
    const size_t need = szFrames + szCounters + szPacked + szRecs + szAlpha + szSmall + szWhole + szBig + szFlags + szIds + szBinList;
    if (recCap > 0xffffffffull || bigCap > 0xffffffffull) return fail(INCR_ERR_INVALID_ARGUMENT, "scene too large for the visibility queues");
    if (ctx->rasterBytes < need || !ctx->rasterBlock) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (ctx->rasterBlock) { HIP_TRY(hipFree(ctx->rasterBlock)); ctx->rasterBlock = nullptr; ctx->rasterBytes = 0; ctx->rasterCounters = nullptr; }
        HIP_TRY(hipMalloc(&ctx->rasterBlock, need));
        ctx->rasterBytes = need;
    }
    uint8_t* p = (uint8_t*)ctx->rasterBlock;
    ws->frames = (MeshFrame*)p; p += szFrames;
    ws->counters = (uint32_t*)p; ws->binCount = (uint32_t*)(p + align256(kVisCounterBytes)); p += szCounters;
    ws->packed = (unsigned long long*)p; p += szPacked;
    ws->recs = (SubTriRec*)p; p += szRecs;
    ws->alphas = (SubTriAlpha*)p; p += szAlpha;
    ws->smallQueue = (uint32_t*)p; p += szSmall;
    ws->wholeQueue = (uint32_t*)p; p += szWhole;
    ws->bigQueue = (uint2*)p; p += szBig;
    ws->primFlags = (uint32_t*)p; p += szFlags;
    ws->ids = (uint32_t*)p; p += szIds;
    ws->binList = (uint32_t*)p; ws->binCap = (uint32_t)binCap;
    ws->recCap = (uint32_t)recCap; ws->bigCap = (uint32_t)bigCap;
    return INCR_OK;
}
// scissor rows of a pass: [y0, y1) clamped to the frame; 0,0 = whole frame
void scissor_rows(const incr_view_constants& v, uint32_t* y0, uint32_t* y1) {
    const bool whole = v.scissor_y0 == 0 && v.scissor_y1 == 0;
    *y0 = whole ? 0u : (v.scissor_y0 < v.res_y ? v.scissor_y0 : v.res_y);
    *y1 = whole ? v.res_y : (v.scissor_y1 < v.res_y ? v.scissor_y1 : v.res_y);
    if (*y1 < *y0) *y1 = *y0;
}
int check_view(const incr_view_constants* v) {
    if (!v) return fail(INCR_ERR_INVALID_ARGUMENT, "view constants are NULL");
    if (!v->res_x || !v->res_y) return fail(INCR_ERR_INVALID_ARGUMENT, "zero resolution");
    if ((uint64_t)v->res_x * v->res_y > 0x7fffffffull) return fail(INCR_ERR_INVALID_ARGUMENT, "resolution too large");
    return INCR_OK;
}
int fill_lighting(const incr_lighting_inputs& in, LightingParams& L) {
    int rc = check_view(in.view);
    if (rc) return rc;
    if (!in.scene) return fail(INCR_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!in.color_target || !in.depth_buffer || !in.gbuffer_basecolor || !in.gbuffer_normals || !in.gbuffer_metallic_roughness_ao)
        return fail(INCR_ERR_INVALID_ARGUMENT, "a lighting input surface is NULL");
    memset(&L, 0, sizeof L);
    L.W = in.view->res_x; L.H = in.view->res_y;
    scissor_rows(*in.view, &L.y0, &L.y1);
    L.color = in.color_target; L.depth = in.depth_buffer; L.g0 = in.gbuffer_basecolor; L.g1 = in.gbuffer_normals; L.g2 = in.gbuffer_metallic_roughness_ao;
    L.lightingOnly = in.lighting_only; L.rounding = in.storage_rounding;
    memcpy(L.invViewProj, in.view->inv_view_proj_transform.m, 64); memcpy(L.invProj, in.view->inv_proj_transform.m, 64);
    memcpy(L.eye, in.view->eye_pos, 12);
    return INCR_OK;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------- batch culling
int incr_batch_cull(incr_context* ctx, const incr_batch_cull_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->default_list || !d->double_sided_list || !d->counts) return fail(INCR_ERR_INVALID_ARGUMENT, "an output list is NULL");
    const incr_scene* sc = d->scene;
    RasterScene S; fill_raster_scene(ctx, sc, S);
    ViewFrame vf; fill_view_frame(*d->view, vf);
    RasterWorkspace ws;
    if ((rc = ensure_raster_workspace(ctx, 0, S.numMeshes, 0, (hipStream_t)stream, &ws))) return rc;
    HIP_TRY(incr_launch_mesh_frames(&S, &vf, ws.frames, (hipStream_t)stream));
    HIP_TRY(incr_launch_batch_cull(&S, ws.frames, d->frustum_culling, d->default_list, d->double_sided_list, d->counts, d->render_stats,
                                   d->use_primitives ? sc->dPrimDraws : nullptr, (uint32_t)sc->host.primitives.size(), (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- light culling
int incr_light_cull(incr_context* ctx, const incr_light_cull_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->culled_light_lists || !d->light_grid || !d->culled_light_count) return fail(INCR_ERR_INVALID_ARGUMENT, "an output buffer is NULL");
    if (!d->cluster_dim[0] || !d->cluster_dim[1] || !d->cluster_dim[2] || d->cluster_dim[2] > kMaxClusterSlices)
        return fail(INCR_ERR_INVALID_ARGUMENT, "cluster grid %ux%ux%u unsupported (each >= 1, Z <= %u)", d->cluster_dim[0], d->cluster_dim[1], d->cluster_dim[2], kMaxClusterSlices);
    if (!d->max_lights_per_cluster || d->max_lights_per_cluster > kMaxLightsPerClusterCap) return fail(INCR_ERR_INVALID_ARGUMENT, "max_lights_per_cluster must be 1..%u", kMaxLightsPerClusterCap);
    if (!(d->view->near_plane > 0.f) || !(d->cluster_depth_extent > d->view->near_plane)) return fail(INCR_ERR_INVALID_ARGUMENT, "cluster depth extent must exceed the near plane");
    const incr_scene* sc = d->scene;
    LightCullParams P; memset(&P, 0, sizeof P);
    P.globalLights = sc->dGlobalLights; P.lightIndices = sc->dLightIndices; P.lightTransforms = sc->dLightTransforms;
    P.numSceneLights = (uint32_t)sc->host.lightIndices.size(); P.numGlobalLights = (uint32_t)sc->host.globalLights.size();
    for (int a = 0; a < 3; ++a) P.dim[a] = d->cluster_dim[a];
    P.maxLights = d->max_lights_per_cluster;
    // cluster-culling.hlsli:57-60: zNear * pow(zFar / zNear, k / dimZ) -- powf once per slice on the host
    const float zNear = d->view->near_plane, zFar = d->cluster_depth_extent;
    for (uint32_t k = 0; k <= P.dim[2]; ++k) P.slice[k] = zNear * powf(zFar / zNear, (float)k / (float)P.dim[2]);
    memcpy(P.proj, d->view->lightcull_proj_transform.m, 64); memcpy(P.invViewProj, d->view->lightcull_inv_view_proj.m, 64);
    P.lists = d->culled_light_lists; P.grid = d->light_grid; P.totalCount = d->culled_light_count; P.stats = d->render_stats;
    // light-culling.inl:115-127 clears the count (and the lists); renderStatsBuffer.Store(sizeof(int), 0)
    HIP_TRY(hipMemsetAsync(d->culled_light_count, 0, 4, (hipStream_t)stream));
    if (d->render_stats) HIP_TRY(hipMemsetAsync(&d->render_stats->culledLights, 0, 4, (hipStream_t)stream));
    HIP_TRY(incr_launch_light_cull(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- visibility
// Introspection (bench / tuning): the queue counters the last incr_visibility call left behind --
// [0] screen triangles after clip + cull, [1] band items, [2] small triangles, [3] whole-triangle items.
int incr_visibility_get_stats(incr_context* ctx, uint32_t out_counters[4]) {
    if (!ctx || !out_counters) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!ctx->rasterBlock || !ctx->rasterCounters) return fail(INCR_ERR_INVALID_ARGUMENT, "incr_visibility has not run yet");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_counters, ctx->rasterCounters, 16, hipMemcpyDeviceToHost));
#ifdef INCR_DIAG_VIS_COUNT
    uint32_t c[64];
    HIP_TRY(hipMemcpy(c, ctx->rasterCounters, 256, hipMemcpyDeviceToHost));
    unsigned long long sum, setup; memcpy(&sum, c + 20, 8); memcpy(&setup, c + 26, 8);
    unsigned long long wall, queue, cursor; memcpy(&wall, c + 28, 8); memcpy(&queue, c + 30, 8); memcpy(&cursor, c + 32, 8);
    fprintf(stderr, "vis diag: of the band phase cycles: cursor %llu, queue entry %llu, queue entry + record + set-up %llu; band phase in 100 MHz ticks: %llu\n", cursor, queue, setup, wall);
    fprintf(stderr, "vis diag: alpha fragments covered %u (in %u batches), not hidden when looked at %u, passed the alpha test %u (%u sampler batches)\n", c[34], c[37], c[35], c[36], c[38]);
    fprintf(stderr, "vis diag: tiles scanned %u rejected %u; waves with band items %u; band phase cycles: sum %llu, max %llu; longest item %llu cycles; most items by one wave %u\n",
            c[16], c[17], c[18], sum, (unsigned long long)c[22] << 4, (unsigned long long)c[23] << 4, c[24]);
#endif
    return INCR_OK;
}

int incr_visibility(incr_context* ctx, const incr_visibility_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->default_list || !d->double_sided_list || !d->counts || !d->vis_buffer || !d->depth_buffer) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL buffer");
    const incr_scene* sc = d->scene;
    VisParams P; memset(&P, 0, sizeof P);
    fill_raster_scene(ctx, sc, P.S);
    memcpy(P.viewProj, d->view->view_proj_transform.m, 64);
    P.W = d->view->res_x; P.H = d->view->res_y;
    scissor_rows(*d->view, &P.y0, &P.y1);
    P.lists[0] = d->default_list; P.lists[1] = d->double_sided_list; P.counts = d->counts;
    RasterWorkspace ws;
    size_t tris = sc->host.meshletPackedTriangles.size();
    if (d->use_primitives) { tris = 0; for (const incr_gpu_primitive& pr : sc->host.primitives) tris += pr.indexCount > 0 ? (size_t)pr.indexCount / 3 : 0; }
    P.binShift = vis_bin_shift(P.W, P.H); P.binsX = vis_bins(P.W, P.binShift); P.binsY = vis_bins(P.y1 > P.y0 ? P.y1 - P.y0 : 0u, P.binShift);
    if ((rc = ensure_raster_workspace(ctx, (size_t)P.W * P.H, P.S.numMeshes, tris, (hipStream_t)stream, &ws, sc->host.primitives.size(), (size_t)P.binsX * vis_bins(P.H, P.binShift)))) return rc;
    if (d->use_primitives && (rc = fill_raster_prims(sc, ws.primFlags, P.prims))) return rc;
    P.packed = ws.packed; P.recs = ws.recs; P.alphas = ws.alphas; P.bigQueue = ws.bigQueue; P.smallQueue = ws.smallQueue; P.wholeQueue = ws.wholeQueue;
    P.vqCounters = ws.counters; P.recCap = ws.recCap; P.bigCap = ws.bigCap;
    P.binCount = ws.binCount; P.binList = ws.binList; P.binCap = ws.binCap;
    ctx->rasterCounters = ws.counters;
    HIP_TRY(incr_launch_visibility(&P, d->use_primitives ? P.prims.numChunks : P.S.numMeshlets, ctx->props.multiProcessorCount, d->vis_buffer, d->depth_buffer, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- G-buffer
int incr_gbuffer(incr_context* ctx, const incr_gbuffer_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->vis_buffer || !d->gbuffer_basecolor || !d->gbuffer_normals || !d->gbuffer_metallic_roughness_ao || !d->color_target) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    const incr_scene* sc = d->scene;
    GBufferParams P; memset(&P, 0, sizeof P);
    fill_raster_scene(ctx, sc, P.S);
    ViewFrame vf; fill_view_frame(*d->view, vf);
    RasterWorkspace ws;
    if ((rc = ensure_raster_workspace(ctx, 0, P.S.numMeshes, 0, (hipStream_t)stream, &ws))) return rc;
    HIP_TRY(incr_launch_mesh_frames(&P.S, &vf, ws.frames, (hipStream_t)stream));
    P.frames = ws.frames; P.W = d->view->res_x; P.H = d->view->res_y; P.vis = d->vis_buffer;
    scissor_rows(*d->view, &P.y0, &P.y1);
    P.g0 = d->gbuffer_basecolor; P.g1 = d->gbuffer_normals; P.g2 = d->gbuffer_metallic_roughness_ao; P.color = d->color_target;
    P.rounding = d->storage_rounding;
    if (d->use_primitives) { P.usePrims = 1; if ((rc = fill_raster_prims(sc, nullptr, P.prims))) return rc; }
    HIP_TRY(incr_launch_gbuffer(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- lighting
static int fill_sky(const incr_sky_lighting_desc* d, SkyParams& P) {
    memset(&P, 0, sizeof P);
    int rc = fill_lighting(d->in, P.L);
    if (rc) return rc;
    P.sh9 = (const float4*)d->sh9_rgba; P.ao = d->ambient_occlusion; P.skyBrightness = d->sky_brightness; P.diffuseIbl = d->diffuse_ibl;
    if (d->specular_ibl && d->prefiltered_envmap && d->env_brdf) {        // g_envmapIndex != -1 && g_envBrdfTextureIndex != -1
        if (!d->envmap_res || !d->envmap_mips || d->envmap_mips > 16 || !d->env_brdf_width || !d->env_brdf_height)
            return fail(INCR_ERR_INVALID_ARGUMENT, "specular IBL inputs: zero envmap / LUT dimensions");
        P.envmap = d->prefiltered_envmap; P.envmapRes = d->envmap_res; P.envmapMips = d->envmap_mips;
        P.envBrdf = d->env_brdf; P.envBrdfW = d->env_brdf_width; P.envBrdfH = d->env_brdf_height;
        memcpy(P.skyInvViewProj, d->in.view->sky_inv_view_proj.m, 64);
    }
    P.bent = d->bent_normals; P.useBent = d->use_bent_normals;
    return INCR_OK;
}
int incr_sky_lighting(incr_context* ctx, const incr_sky_lighting_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    SkyParams P;
    int rc = fill_sky(d, P);
    if (rc) return rc;
    HIP_TRY(incr_launch_sky_lighting(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- decals
int incr_gbuffer_decals(incr_context* ctx, const incr_gbuffer_decals_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->depth_buffer || !d->gbuffer_basecolor || !d->gbuffer_normals || !d->gbuffer_metallic_roughness_ao) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    const incr_scene* sc = d->scene;
    if (!sc->numDecals || !sc->numDecalChunks) return INCR_OK;                    // no decals in this scene
    if (sc->numDecals >= 4096) return fail(INCR_ERR_INVALID_ARGUMENT, "%u decal primitives: at most 4095 (12-bit id in the record key)", sc->numDecals);
    DecalParams P; memset(&P, 0, sizeof P);
    fill_raster_scene(ctx, sc, P.S);
    P.prims.draws = sc->dDecalDraws; P.prims.indices = sc->dDecalIndices;
    P.prims.pvPos = sc->dDvPos; P.prims.pvUv = sc->dDvUv; P.prims.pvNormal = sc->dDvNormal; P.prims.pvTangent = sc->dDvTangent;
    P.prims.chunkPrim = sc->dDecalChunkPrim; P.prims.chunkFirstTri = sc->dDecalChunkFirstTri;
    P.prims.numPrims = sc->numDecals; P.prims.numChunks = sc->numDecalChunks;
    P.decalLocalToWorld = sc->dDecalLocalToWorld;
    memcpy(P.viewProj, d->view->view_proj_transform.m, 64);
    P.W = d->view->res_x; P.H = d->view->res_y; scissor_rows(*d->view, &P.y0, &P.y1);
    P.rounding = d->storage_rounding;
    P.depth = d->depth_buffer; P.g0 = d->gbuffer_basecolor; P.g1 = d->gbuffer_normals; P.g2 = d->gbuffer_metallic_roughness_ao;
    // two record slots per decal triangle of every chunk: the visibility workspace's record array is reused
    const size_t slots = (size_t)sc->numDecalChunks * 64 * 2;
    RasterWorkspace ws;
    if ((rc = ensure_raster_workspace(ctx, 1, P.S.numMeshes, slots / 2 + 1, (hipStream_t)stream, &ws))) return rc;
    if (ws.recCap < slots) return fail(INCR_ERR_OUT_OF_MEMORY, "decal record workspace too small");
    P.recs = ws.recs; P.numSlots = (uint32_t)slots;
    P.listCap = 1024;
    // test hook: a small tile list forces flushes in the middle of the slot walk (the blend order must survive them)
    if (const char* cap = test_hook_env("INCR_DECAL_LIST_CAP")) { const unsigned long v = strtoul(cap, nullptr, 10); P.listCap = v < 256 ? 256u : (v > 1024 ? 1024u : (uint32_t)v); }
    HIP_TRY(incr_launch_decals(&P, ctx->props.multiProcessorCount, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- forward lighting
int incr_forward_lighting(incr_context* ctx, const incr_forward_lighting_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->color_target || !d->depth_buffer) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    if (d->viewmode < 0 || d->viewmode > 13) return fail(INCR_ERR_INVALID_ARGUMENT, "viewmode %d out of range (0..13)", d->viewmode);
    const incr_scene* sc = d->scene;
    // 1. nearest fragment per pixel: the primitive visibility pass in forward order (every primitive of every visible mesh)
    VisParams V; memset(&V, 0, sizeof V);
    fill_raster_scene(ctx, sc, V.S);
    memcpy(V.viewProj, d->view->view_proj_transform.m, 64);
    V.W = d->view->res_x; V.H = d->view->res_y;
    scissor_rows(*d->view, &V.y0, &V.y1);
    V.forward = 1;
    size_t tris = 0;
    for (const incr_gpu_primitive& pr : sc->host.primitives) tris += pr.indexCount > 0 ? (size_t)pr.indexCount / 3 : 0;
    RasterWorkspace ws;
    V.binShift = vis_bin_shift(V.W, V.H); V.binsX = vis_bins(V.W, V.binShift); V.binsY = vis_bins(V.y1 > V.y0 ? V.y1 - V.y0 : 0u, V.binShift);
    if ((rc = ensure_raster_workspace(ctx, (size_t)V.W * V.H, V.S.numMeshes, tris, (hipStream_t)stream, &ws, sc->host.primitives.size(), (size_t)V.binsX * vis_bins(V.H, V.binShift)))) return rc;
    if ((rc = fill_raster_prims(sc, ws.primFlags, V.prims))) return rc;
    V.packed = ws.packed; V.recs = ws.recs; V.alphas = ws.alphas; V.bigQueue = ws.bigQueue; V.smallQueue = ws.smallQueue; V.wholeQueue = ws.wholeQueue;
    V.vqCounters = ws.counters; V.recCap = ws.recCap; V.bigCap = ws.bigCap;
    V.binCount = ws.binCount; V.binList = ws.binList; V.binCap = ws.binCap;
    ctx->rasterCounters = ws.counters;
    HIP_TRY(incr_launch_visibility(&V, V.prims.numChunks, ctx->props.multiProcessorCount, ws.ids, d->depth_buffer, (hipStream_t)stream));
    // 2. pixel shader
    ForwardParams P; memset(&P, 0, sizeof P);
    P.S = V.S; P.prims = V.prims;
    incr_fill_scene_params(ctx, sc, P.Pt);
    P.Pt.stats = d->ray_stats;
    P.W = V.W; P.H = V.H; P.y0 = V.y0; P.y1 = V.y1; P.rounding = d->storage_rounding;
    P.vis = ws.ids; P.color = d->color_target;
    memcpy(P.viewProj, d->view->view_proj_transform.m, 64); memcpy(P.eye, d->view->eye_pos, 12);
    P.sh9 = (const float4*)d->sh9_rgba;
    if (d->prefiltered_envmap) {
        if (!d->envmap_res || !d->envmap_mips || d->envmap_mips > 16) return fail(INCR_ERR_INVALID_ARGUMENT, "envmap given with zero resolution / level count");
        P.envmap = d->prefiltered_envmap; P.envmapRes = d->envmap_res; P.envmapMips = d->envmap_mips;
    }
    if (d->env_brdf) {
        if (!d->env_brdf_width || !d->env_brdf_height) return fail(INCR_ERR_INVALID_ARGUMENT, "env BRDF LUT given with zero size");
        P.envBrdf = d->env_brdf; P.envBrdfW = d->env_brdf_width; P.envBrdfH = d->env_brdf_height;
    }
    P.viewmode = d->viewmode; P.directLighting = d->direct_lighting; P.diffuseIbl = d->diffuse_ibl; P.specularIbl = d->specular_ibl;
    HIP_TRY(incr_launch_forward_shade(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- HBAO
int incr_hbao(incr_context* ctx, const incr_hbao_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->depth_buffer || !d->gbuffer_normals || !d->ao_target || !d->bent_normal_target) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    HbaoParams P; memset(&P, 0, sizeof P);
    P.W = d->view->res_x; P.H = d->view->res_y;
    scissor_rows(*d->view, &P.y0, &P.y1);
    P.rounding = d->storage_rounding;
    P.depth = d->depth_buffer; P.normals = d->gbuffer_normals; P.ao = d->ao_target; P.bent = d->bent_normal_target;
    memcpy(P.invViewProj, d->view->inv_view_proj_transform.m, 64);
    memcpy(P.eye, d->view->eye_pos, 12); memcpy(P.camUp, d->view->camera_up, 12);
    incr_fill_scene_params(ctx, d->scene, P.S);
    P.S.stats = d->ray_stats;
    HIP_TRY(incr_launch_hbao(&P, (hipStream_t)stream));
    return INCR_OK;
}

static int fill_direct(incr_context* ctx, const incr_direct_lighting_desc* d, DirectParams& P) {
    memset(&P, 0, sizeof P);
    int rc = fill_lighting(d->in, P.L);
    if (rc) return rc;
    if (d->directional_light_index < 0) return fail(INCR_ERR_INVALID_ARGUMENT, "directional_light_index is negative: the reference skips this pass when the scene has no sun (renderer.cpp:1886)");
    incr_fill_scene_params(ctx, d->in.scene, P.S);
    P.S.stats = d->ray_stats;
    P.sunIndex = (uint32_t)d->directional_light_index;
    return INCR_OK;
}
// The scene's sun cover map (pt_params.h SunCover) for global light `sunIndex`, (re)built on `stream` when stale, handed to the pass in S.cover.  Also called by
// incr_pathtrace (api.hip).  Test hook INCR_SUN_COVER=0: no map.
int incr_ensure_sun_cover(incr_context* ctx, const incr_scene* sc, uint32_t sunIndex, PtParams& S, hipStream_t stream) {
    (void)ctx;
    if (const char* v = test_hook_env("INCR_SUN_COVER")) if (v[0] == '0') return INCR_OK;
    if (!S.numTris || !S.numNodes) return INCR_OK;
    std::lock_guard<std::mutex> guard(sc->sunCoverLock);
    if (!sc->sunCover) {
        HIP_TRY(hipMalloc((void**)&sc->sunCover, incr_sun_cover_bytes())); sc->sunCoverEpoch = 0;
        HIP_TRY(hipEventCreateWithFlags(&sc->sunCoverBuilt, hipEventDisableTiming));
    }
    SunCoverHeader* header = (SunCoverHeader*)sc->sunCover; uint32_t* grid = (uint32_t*)((uint8_t*)sc->sunCover + 256);
    // a stream that is being captured (the C++ renderer replays its frame as a HIP graph) neither waits for an event recorded outside the capture nor records one that
    // streams outside would wait for: the graph is launched on the stream the earlier frames -- and with them the build -- were issued on
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &capture);
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (sc->sunCoverEpoch != sc->geometryEpoch || sc->sunCoverIndex != sunIndex) {
        HIP_TRY(incr_launch_sun_cover(&S, sunIndex, header, grid, stream));
        if (!capturing) { HIP_TRY(hipEventRecord(sc->sunCoverBuilt, stream)); sc->sunCoverStream = stream; }
        sc->sunCoverEpoch = sc->geometryEpoch; sc->sunCoverIndex = sunIndex;
    } else if (!capturing && sc->sunCoverStream && stream != sc->sunCoverStream) HIP_TRY(hipStreamWaitEvent(stream, sc->sunCoverBuilt, 0));
    S.cover.header = header; S.cover.grid = grid;
    return INCR_OK;
}
int incr_direct_lighting(incr_context* ctx, const incr_direct_lighting_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    DirectParams P;
    int rc = fill_direct(ctx, d, P);
    if (rc) return rc;
    if ((rc = incr_ensure_sun_cover(ctx, d->in.scene, P.sunIndex, P.S, (hipStream_t)stream))) return rc;
    HIP_TRY(incr_launch_direct_lighting(&P, (hipStream_t)stream));
    return INCR_OK;
}

static int fill_clustered(incr_context* ctx, const incr_clustered_lighting_desc* d, ClusteredParams& P) {
    memset(&P, 0, sizeof P);
    int rc = fill_lighting(d->in, P.L);
    if (rc) return rc;
    if (!d->culled_light_lists || !d->light_grid) return fail(INCR_ERR_INVALID_ARGUMENT, "light lists / grid are NULL");
    if (!d->cluster_dim[0] || !d->cluster_dim[1] || !d->cluster_dim[2]) return fail(INCR_ERR_INVALID_ARGUMENT, "zero cluster dimension");
    const float nearPlane = d->in.view->near_plane;
    if (!(nearPlane > 0.f) || !(d->cluster_depth_extent > nearPlane)) return fail(INCR_ERR_INVALID_ARGUMENT, "cluster depth extent must exceed the near plane");
    incr_fill_scene_params(ctx, d->in.scene, P.S);
    P.lists = d->culled_light_lists; P.grid = d->light_grid;
    for (int a = 0; a < 3; ++a) P.dim[a] = d->cluster_dim[a];
    P.scale = (float)d->cluster_dim[2] / logf(d->cluster_depth_extent / nearPlane);          // clustered-lighting.inl:88
    P.bias = -P.scale * logf(nearPlane);                                                      // :102
    P.requiresClear = d->requires_clear;
    return INCR_OK;
}
int incr_clustered_lighting(incr_context* ctx, const incr_clustered_lighting_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    ClusteredParams P;
    int rc = fill_clustered(ctx, d, P);
    if (rc) return rc;
    HIP_TRY(incr_launch_clustered_lighting(&P, (hipStream_t)stream));
    return INCR_OK;
}

// sky + sun + clustered lights in one kernel (raster.hip deferred_lighting_kernel); bit-identical to the three calls in that order
int incr_deferred_lighting(incr_context* ctx, const incr_deferred_lighting_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    if (!d->sky && !d->direct && !d->clustered) return INCR_OK;
    static thread_local FusedLightingParams P;           // ~2 KB of kernel arguments: kept off the stack of the caller's thread
    int rc = INCR_OK;
    P.haveSky = d->sky != nullptr; P.haveDirect = d->direct != nullptr; P.haveClustered = d->clustered != nullptr;
    if (d->sky && (rc = fill_sky(d->sky, P.sky))) return rc;
    if (d->direct && ((rc = fill_direct(ctx, d->direct, P.direct)) || (rc = incr_ensure_sun_cover(ctx, d->direct->in.scene, P.direct.sunIndex, P.direct.S, (hipStream_t)stream)))) return rc;
    if (d->clustered && (rc = fill_clustered(ctx, d->clustered, P.clustered))) return rc;
    const LightingParams* parts[3] = {d->sky ? &P.sky.L : nullptr, d->direct ? &P.direct.L : nullptr, d->clustered ? &P.clustered.L : nullptr};
    const LightingParams* first = nullptr;
    for (const LightingParams* l : parts) {
        if (!l) continue;
        if (!first) { first = l; continue; }
        if (l->color != first->color || l->depth != first->depth || l->g0 != first->g0 || l->g1 != first->g1 || l->g2 != first->g2 || l->W != first->W || l->H != first->H ||
            l->y0 != first->y0 || l->y1 != first->y1 || l->rounding != first->rounding || l->lightingOnly != first->lightingOnly)
            return fail(INCR_ERR_INVALID_ARGUMENT, "the parts of a fused lighting pass must address the same surfaces, rows and storage rounding");
    }
    HIP_TRY(incr_launch_deferred_lighting(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- dynamic sky background
int incr_dynamic_sky(incr_context* ctx, const incr_dynamic_sky_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->scene) return fail(INCR_ERR_INVALID_ARGUMENT, "desc or scene is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->color_target || !d->depth_buffer) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    DynamicSkyParams P; memset(&P, 0, sizeof P);
    P.W = d->view->res_x; P.H = d->view->res_y; scissor_rows(*d->view, &P.y0, &P.y1);
    P.rounding = d->storage_rounding; P.color = d->color_target; P.depth = d->depth_buffer;
    incr_perez_from_turbidity(d->turbidity, &P.perez);                   // dynamic-sky.inl:119-126
    const float* s = d->scene->host.sunDir;                              // Vector3 L = scene->m_sunDir; L.Normalize() (:146-147)
    const V3 L = normalize3(v3(s[0], s[1], s[2]));
    const float Ln[3] = {L.x, L.y, L.z};
    incr_sky_sun_terms(Ln, d->turbidity, P.perez, &P.sun);
    HIP_TRY(incr_launch_dynamic_sky(&P, (hipStream_t)stream));
    return INCR_OK;
}

int incr_environmentmap(incr_context* ctx, const incr_environmentmap_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->color_target || !d->depth_buffer) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    if (!d->envmap_r11g11b10 || !d->envmap_res || !d->envmap_mips || d->envmap_mips > 16) return fail(INCR_ERR_INVALID_ARGUMENT, "environment cube missing or zero-sized");
    EnvmapParams P; memset(&P, 0, sizeof P);
    P.W = d->view->res_x; P.H = d->view->res_y; scissor_rows(*d->view, &P.y0, &P.y1);
    P.rounding = d->storage_rounding; P.color = d->color_target; P.depth = d->depth_buffer;
    P.envmap = d->envmap_r11g11b10; P.envmapRes = d->envmap_res; P.envmapMips = d->envmap_mips;
    P.skyBrightness = d->sky_brightness;
    memcpy(P.invParallaxViewProj, d->view->envmap_inv_parallax_view_proj.m, 64);
    HIP_TRY(incr_launch_environmentmap(&P, (hipStream_t)stream));
    return INCR_OK;
}

int incr_debug_visualization(incr_context* ctx, const incr_debug_viz_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->color_target) return fail(INCR_ERR_INVALID_ARGUMENT, "color_target is NULL");
    const int m = d->viewmode;
    if (m < 0 || m > 13) return fail(INCR_ERR_INVALID_ARGUMENT, "viewmode %d out of range (0..13)", m);
    if (((m == 2 || m == 3) && !d->gbuffer_metallic_roughness_ao) || (m == 4 && !d->gbuffer_basecolor) || ((m == 8 || m == 9) && !d->vis_buffer) ||
        (m == 10 && !d->gbuffer_normals) || (m == 11 && (!d->depth_buffer || !d->light_cluster_slices)))
        return fail(INCR_ERR_INVALID_ARGUMENT, "viewmode %d: its source surface is NULL", m);
    DebugVizParams P; memset(&P, 0, sizeof P);
    P.W = d->view->res_x; P.H = d->view->res_y; scissor_rows(*d->view, &P.y0, &P.y1);
    P.rounding = d->storage_rounding; P.viewmode = m; P.usePrims = d->use_primitives;
    P.vis = d->vis_buffer; P.g0 = d->gbuffer_basecolor; P.g1 = d->gbuffer_normals; P.g2 = d->gbuffer_metallic_roughness_ao;
    P.depth = d->depth_buffer; P.ao = d->ambient_occlusion; P.bent = d->bent_normals; P.color = d->color_target;
    // debug-visualization.hlsl:176-181: zFar 1000, zNear 5 -- logf of the constants once, on the host
    P.numSlices = (float)d->light_cluster_slices;
    P.sliceScale = P.numSlices / logf(1000.f / 5.f); P.sliceBias = -logf(5.f) * P.sliceScale;
    memcpy(P.invProj, d->view->inv_proj_transform.m, 64);
    HIP_TRY(incr_launch_debug_viz(&P, (hipStream_t)stream));
    return INCR_OK;
}

// ---------------------------------------------------------------------------------------------- TAA
int incr_taa_resolve(incr_context* ctx, const incr_taa_resolve_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d) return fail(INCR_ERR_INVALID_ARGUMENT, "desc is NULL");
    int rc = check_view(d->view);
    if (rc) return rc;
    if (!d->hdr_scene_color || !d->history_out || !d->depth_buffer) return fail(INCR_ERR_INVALID_ARGUMENT, "NULL surface");
    if (d->history_index != 0 && !d->history_in) return fail(INCR_ERR_INVALID_ARGUMENT, "history_in is NULL with history_index != 0");
    if (d->history_in == d->history_out) return fail(INCR_ERR_INVALID_ARGUMENT, "history_in and history_out must not alias");
    TaaParams P; memset(&P, 0, sizeof P);
    P.W = d->view->res_x; P.H = d->view->res_y; P.historyIndex = d->history_index; P.rounding = d->storage_rounding;
    scissor_rows(*d->view, &P.y0, &P.y1);
    P.hdr = d->hdr_scene_color; P.histIn = d->history_in; P.histOut = d->history_out; P.depth = d->depth_buffer;
    memcpy(P.invViewProj, d->view->taa_inv_view_proj.m, 64); memcpy(P.prevViewProj, d->view->prev_view_proj_transform.m, 64);
    P.ldr = (uint32_t*)d->ldr_rgba8; P.ev100 = (int)d->exposure; P.viewmode = d->viewmode; P.srgbSteps = ctx->srgbSteps;
    HIP_TRY(incr_launch_taa_resolve(&P, (hipStream_t)stream));
    return INCR_OK;
}

} // extern "C"
