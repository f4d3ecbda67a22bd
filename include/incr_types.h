/*
 * incr_types.h -- data contract of the MI355X-native Incarnation render hot path.
 *
 * Plain-C mirrors of the structs the reference shares between its C++ host and
 * its HLSL shaders (reference: source/demo-dll/inc/gpu-shared-types.h:59-269).
 * Field order, sizes and meaning are identical so that the packed scene arrays
 * `FScene::CreateGpuGeometryBuffers` builds (reference: source/demo-dll/src/scene.cpp:480-644)
 * can be handed to `incr_scene_create` verbatim.  Bindless descriptor indices
 * of the reference become plain table indices (buffers[], textures[]).
 *
 * Conventions (reference: shaders/common/math.hlsli:4-12, src/shadercompiler.cpp:12-14 `-Zpr`):
 *   left-handed, Y up, Z far; matrices are row-major and vectors are ROW vectors
 *   (v' = v * M), exactly as DirectX SimpleMath / HLSL `mul(v, M)`.
 */
#ifndef INCR_TYPES_H
#define INCR_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4x4 float matrix, row-major, row-vector convention.  m[r*4+c] == SimpleMath `_rc` (1-based). */
typedef struct incr_matrix { float m[16]; } incr_matrix;

/* gpu-shared-types.h:59-64  (glTF bufferView) -- 12 B */
typedef struct incr_mesh_buffer_view {
    int32_t  bufferSrvIndex;   /* index into incr_scene_desc::buffers */
    uint32_t byteOffset;
    uint32_t byteLength;
} incr_mesh_buffer_view;

/* gpu-shared-types.h:67-72  (glTF accessor) -- 12 B */
typedef struct incr_mesh_accessor {
    int32_t  bufferViewIndex;
    uint32_t byteOffset;
    uint32_t byteStride;
} incr_mesh_accessor;

/* gpu-shared-types.h:74-86 -- 52 B */
typedef struct incr_gpu_primitive {
    float   boundingSphere[4];
    int32_t meshIndex;
    int32_t indexAccessor;
    int32_t positionAccessor;
    int32_t uvAccessor;
    int32_t normalAccessor;
    int32_t tangentAccessor;
    int32_t materialIndex;
    int32_t indicesPerTriangle;
    int32_t indexCount;
} incr_gpu_primitive;

/* gpu-shared-types.h:88-102 -- 56 B */
typedef struct incr_gpu_meshlet {
    uint32_t vertexBegin;
    uint32_t vertexCount;
    uint32_t triangleBegin;
    uint32_t triangleCount;
    float    boundingSphere[4];
    int32_t  meshIndex;
    int32_t  positionAccessor;
    int32_t  uvAccessor;
    int32_t  normalAccessor;
    int32_t  tangentAccessor;
    int32_t  materialIndex;
} incr_gpu_meshlet;

/* gpu-shared-types.h:33-41 */
enum { INCR_ALPHA_OPAQUE = 0, INCR_ALPHA_MASKED = 1, INCR_ALPHA_BLEND = 2 };

/* gpu-shared-types.h:104-137 -- 128 B.  Texture indices index incr_scene_desc::textures (-1 = none). */
typedef struct incr_material {
    float   emissiveFactor[3];
    float   baseColorFactor[3];
    float   metallicFactor;
    float   roughnessFactor;
    float   aoStrength;
    float   transmissionFactor;
    float   clearcoatFactor;
    float   clearcoatRoughnessFactor;

    int32_t emissiveTextureIndex;
    int32_t baseColorTextureIndex;
    int32_t metallicRoughnessTextureIndex;
    int32_t normalTextureIndex;
    int32_t aoTextureIndex;
    int32_t transmissionTextureIndex;
    int32_t clearcoatTextureIndex;
    int32_t clearcoatRoughnessTextureIndex;
    int32_t clearcoatNormalTextureIndex;

    int32_t emissiveSamplerIndex;
    int32_t baseColorSamplerIndex;
    int32_t metallicRoughnessSamplerIndex;
    int32_t normalSamplerIndex;
    int32_t aoSamplerIndex;
    int32_t transmissionSamplerIndex;
    int32_t clearcoatSamplerIndex;
    int32_t clearcoatRoughnessSamplerIndex;
    int32_t clearcoatNormalSamplerIndex;

    int32_t alphaMode;
    int32_t doubleSided;       /* HLSL bool = 4 B */
} incr_material;

/* gpu-shared-types.h:195-215 -- 32 B */
enum { INCR_LIGHT_DIRECTIONAL = 0, INCR_LIGHT_POINT = 1, INCR_LIGHT_SPOT = 2 };
typedef struct incr_light {
    int32_t type;
    float   color[3];
    float   intensity;
    float   range;
    float   spotAngles[2];
} incr_light;

/* gpu-shared-types.h:251-255 */
typedef struct incr_render_stats {
    int32_t culledObjects;
    int32_t culledLights;
} incr_render_stats;

/* gpu-shared-types.h:266-269 (Preetham sky) */
typedef struct incr_perez_distribution { float A[4], B[4], C[4], D[4], E[4]; } incr_perez_distribution;

/* light-culling: FLightGridData{offset,count} (shaders/culling/light-culling.hlsl) -- 8 B per cluster */
typedef struct incr_light_grid_data { uint32_t offset; uint32_t count; } incr_light_grid_data;

/*
 * Texture table entry.  The reference block-compresses textures with DirectXTex
 * (scene.cpp:889-933, 1072-1100, 1157-1420: mip chain, then DirectX::Compress).  The ENCODER is a closed third-party step,
 * so the boundary takes either
 *   - the *uncompressed* RGBA8 texels the reference feeds to the compressor, plus the channel interpretation the chosen DXGI
 *     format implies (formats 0-3; the library builds the mip chain and the vMF-filtered normal / roughness levels itself), or
 *   - the COMPRESSED mip chain the reference uploads (formats 4-7: what `compressedScratch` / the content cache's .dds hold):
 *     `texels` = the blocks of level 0, 1, ... `mip_levels - 1` back to back (ceil(w/4) x ceil(h/4) blocks per level, row-major;
 *     16 bytes per block, 8 for BC4), decoded by the library as the texture unit does; nothing is
 *     generated or filtered for such a texture.
 */
enum {
    INCR_TEX_RGBA8_UNORM      = 0, /* R8G8B8A8_UNORM -> BC4/BC5_UNORM: value = c/255                          */
    INCR_TEX_RGBA8_SRGB       = 1, /* R8G8B8A8_UNORM_SRGB -> BC3_UNORM_SRGB: rgb sRGB-decoded per texel, a linear */
    INCR_TEX_RG_SNORM         = 2, /* R8G8B8A8_UNORM -> BC5_SNORM (normal maps): value = 2*c/255 - 1             */
    INCR_TEX_BGRA8_SWIZZLE_RG = 3, /* metal/rough: sampled .r = texel.b (metal), .g = texel.g (roughness)
                                      (scene.cpp:932 B8G8R8A8 swizzle; prefilter-normal-roughness.hlsl:41 `.bgra`) */
    INCR_TEX_BC3_SRGB         = 4, /* DXGI_FORMAT_BC3_UNORM_SRGB blocks: base colour, emissive (scene.cpp:902-903)        */
    INCR_TEX_BC4_UNORM        = 5, /* DXGI_FORMAT_BC4_UNORM blocks: AO, transmission, clearcoat (scene.cpp:904, 948, 971) */
    INCR_TEX_BC5_UNORM        = 6, /* DXGI_FORMAT_BC5_UNORM blocks: .r = metal, .g = roughness (scene.cpp:1163)           */
    INCR_TEX_BC5_SNORM        = 7, /* DXGI_FORMAT_BC5_SNORM blocks: normal maps (scene.cpp:1162)                          */
    /* what the samplers see after decode (never passed in): two 16-bit numerators per texel, value = n x float(1 / 8925) or n x float(1 / 4445) */
    INCR_TEX_RG16_OVER_8925   = 8,
    INCR_TEX_RG16S_OVER_4445  = 9
};
typedef struct incr_texture_desc {
    const uint8_t* texels;   /* formats 0-3: width*height*4 bytes, row-major, top row first, RGBA order as decoded from the image;
                                formats 4-7: the compressed mip chain (above) */
    uint32_t width;
    uint32_t height;
    uint32_t format;         /* INCR_TEX_* (0-7) */
    uint32_t mip_levels;     /* formats 4-7: levels in `texels` (>= 1); ignored for formats 0-3 */
} incr_texture_desc;

/*
 * The packed scene: every array of SURVEY.md section 8a row 2, taken verbatim.
 * All pointers are HOST pointers; the library copies what it needs to HBM.
 */
typedef struct incr_scene_desc {
    uint32_t num_buffers;                 const void* const* buffers; const uint64_t* buffer_sizes; /* raw glTF buffers */
    uint32_t num_buffer_views;            const incr_mesh_buffer_view* buffer_views;
    uint32_t num_accessors;               const incr_mesh_accessor*    accessors;
    uint32_t num_meshes;                  const uint32_t* primitive_counts;     /* per mesh (scene.cpp:596-616) */
                                          const incr_matrix* mesh_transforms;   /* per mesh localToWorld (scene.cpp:619-640) */
                                          const uint32_t* mesh_visibility;      /* per mesh 0/1 (renderer.cpp:1560-1589); NULL = all visible */
    uint32_t num_primitives;              const incr_gpu_primitive* primitives;
    uint32_t num_materials;               const incr_material* materials;
    uint32_t num_textures;                const incr_texture_desc* textures;
    uint32_t num_meshlets;                const incr_gpu_meshlet* meshlets;
    uint32_t num_meshlet_vertex_indices;  const uint32_t* meshlet_vertex_indices;
    uint32_t num_meshlet_triangles;       const uint32_t* meshlet_packed_triangles; /* i0:10 | i1:10 | i2:10 (mesh-utils.h:8-14) */
    uint32_t num_global_lights;           const incr_light* global_lights;      /* FScene::m_globalLightList */
    uint32_t num_scene_lights;            const int32_t* light_indices;         /* FScene::m_sceneLights.m_entityList */
                                          const incr_matrix* light_transforms;  /* per SCENE light (renderer.cpp:1540-1558) */
    incr_matrix root_transform;           /* FScene::m_rootTransform == shader `sceneRotation` */
    float    sun_dir[3];                  /* FScene::m_sunDir from time of day (scene.cpp:1594-1655); feeds the dynamic-sky environment */
    uint32_t max_anisotropy;              /* MaxAnisotropy of the raster passes' g_anisoSampler: 0 = 16, the default of the reference's static
                                           * samplers (gbuffer-compute.hlsl:20, d3dx12_root_signature.h:277); 1 = isotropic trilinear; at most 16 */
    /* FScene::m_sceneMeshDecals (scene.cpp:281: meshes whose name starts with "decal"): kept out of the packed primitives
     * and the BVH, drawn only by the decal G-buffer pass (incr_gbuffer_decals).  Accessors / materials index the arrays above. */
    uint32_t num_decal_primitives;        const incr_gpu_primitive* decal_primitives;   /* in draw order: mesh, then primitive */
                                          const incr_matrix* decal_transforms;         /* per decal PRIMITIVE: its mesh's m_transformList entry */
} incr_scene_desc;

/* FView (inc/renderer.h) fields the passes read. */
typedef struct incr_view {
    incr_matrix view_transform;        /* FView::m_viewTransform */
    incr_matrix projection_transform;  /* FView::m_projectionTransform (reverse-Z infinite, demo.cpp:94-102) */
    float position[3];
    float right[3];
    float up[3];
    float look[3];
    float fov;
} incr_view;

#ifdef __cplusplus
}
#endif
#endif /* INCR_TYPES_H */
