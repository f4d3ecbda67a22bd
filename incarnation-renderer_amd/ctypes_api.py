"""ctypes mirrors of include/incr_types.h and include/incr.h (the C-ABI boundary).

Field order and sizes must match the headers exactly; tests/test_abi.py checks sizeof() of every
struct against the values the reference's gpu-shared-types.h implies (52/56/128/32/12/12 bytes).
"""
import ctypes as C

c_float3 = C.c_float * 3
c_float4 = C.c_float * 4


class Matrix(C.Structure):
    _fields_ = [("m", C.c_float * 16)]


class MeshBufferView(C.Structure):
    _fields_ = [("bufferSrvIndex", C.c_int32), ("byteOffset", C.c_uint32), ("byteLength", C.c_uint32)]


class MeshAccessor(C.Structure):
    _fields_ = [("bufferViewIndex", C.c_int32), ("byteOffset", C.c_uint32), ("byteStride", C.c_uint32)]


class GpuPrimitive(C.Structure):
    _fields_ = [("boundingSphere", c_float4), ("meshIndex", C.c_int32), ("indexAccessor", C.c_int32),
                ("positionAccessor", C.c_int32), ("uvAccessor", C.c_int32), ("normalAccessor", C.c_int32),
                ("tangentAccessor", C.c_int32), ("materialIndex", C.c_int32), ("indicesPerTriangle", C.c_int32),
                ("indexCount", C.c_int32)]


class GpuMeshlet(C.Structure):
    _fields_ = [("vertexBegin", C.c_uint32), ("vertexCount", C.c_uint32), ("triangleBegin", C.c_uint32),
                ("triangleCount", C.c_uint32), ("boundingSphere", c_float4), ("meshIndex", C.c_int32),
                ("positionAccessor", C.c_int32), ("uvAccessor", C.c_int32), ("normalAccessor", C.c_int32),
                ("tangentAccessor", C.c_int32), ("materialIndex", C.c_int32)]


_MAT_TEX = ["emissive", "baseColor", "metallicRoughness", "normal", "ao", "transmission", "clearcoat",
            "clearcoatRoughness", "clearcoatNormal"]


class Material(C.Structure):
    _fields_ = ([("emissiveFactor", c_float3), ("baseColorFactor", c_float3), ("metallicFactor", C.c_float),
                 ("roughnessFactor", C.c_float), ("aoStrength", C.c_float), ("transmissionFactor", C.c_float),
                 ("clearcoatFactor", C.c_float), ("clearcoatRoughnessFactor", C.c_float)]
                + [(n + "TextureIndex", C.c_int32) for n in _MAT_TEX]
                + [(n + "SamplerIndex", C.c_int32) for n in _MAT_TEX]
                + [("alphaMode", C.c_int32), ("doubleSided", C.c_int32)])


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", c_float3), ("intensity", C.c_float), ("range", C.c_float),
                ("spotAngles", C.c_float * 2)]


class RenderStats(C.Structure):
    _fields_ = [("culledObjects", C.c_int32), ("culledLights", C.c_int32)]


class PerezDistribution(C.Structure):
    _fields_ = [("A", c_float4), ("B", c_float4), ("C", c_float4), ("D", c_float4), ("E", c_float4)]


class LightGridData(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("count", C.c_uint32)]


TEX_RGBA8_UNORM, TEX_RGBA8_SRGB, TEX_RG_SNORM, TEX_BGRA8_SWIZZLE_RG = 0, 1, 2, 3
TEX_BC3_SRGB, TEX_BC4_UNORM, TEX_BC5_UNORM, TEX_BC5_SNORM, TEX_RG16_OVER_8925, TEX_RG16S_OVER_4445 = 4, 5, 6, 7, 8, 9
ALPHA_OPAQUE, ALPHA_MASKED, ALPHA_BLEND = 0, 1, 2
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT = 0, 1, 2


class TextureDesc(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32),
                ("mip_levels", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("num_buffers", C.c_uint32), ("buffers", C.POINTER(C.c_void_p)), ("buffer_sizes", C.POINTER(C.c_uint64)),
        ("num_buffer_views", C.c_uint32), ("buffer_views", C.POINTER(MeshBufferView)),
        ("num_accessors", C.c_uint32), ("accessors", C.POINTER(MeshAccessor)),
        ("num_meshes", C.c_uint32), ("primitive_counts", C.POINTER(C.c_uint32)),
        ("mesh_transforms", C.POINTER(Matrix)), ("mesh_visibility", C.POINTER(C.c_uint32)),
        ("num_primitives", C.c_uint32), ("primitives", C.POINTER(GpuPrimitive)),
        ("num_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("num_textures", C.c_uint32), ("textures", C.POINTER(TextureDesc)),
        ("num_meshlets", C.c_uint32), ("meshlets", C.POINTER(GpuMeshlet)),
        ("num_meshlet_vertex_indices", C.c_uint32), ("meshlet_vertex_indices", C.POINTER(C.c_uint32)),
        ("num_meshlet_triangles", C.c_uint32), ("meshlet_packed_triangles", C.POINTER(C.c_uint32)),
        ("num_global_lights", C.c_uint32), ("global_lights", C.POINTER(Light)),
        ("num_scene_lights", C.c_uint32), ("light_indices", C.POINTER(C.c_int32)),
        ("light_transforms", C.POINTER(Matrix)),
        ("root_transform", Matrix),
        ("sun_dir", c_float3), ("max_anisotropy", C.c_uint32),
        ("num_decal_primitives", C.c_uint32), ("decal_primitives", C.POINTER(GpuPrimitive)), ("decal_transforms", C.POINTER(Matrix)),
    ]


class View(C.Structure):
    _fields_ = [("view_transform", Matrix), ("projection_transform", Matrix), ("position", c_float3),
                ("right", c_float3), ("up", c_float3), ("look", c_float3), ("fov", C.c_float)]


class Shard(C.Structure):
    _fields_ = [("row_block_height", C.c_uint32), ("row_block_first", C.c_uint32), ("row_block_stride", C.c_uint32)]


class RayStats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("sky_shadow_rays", C.c_uint64),
                ("light_shadow_rays", C.c_uint64), ("shaded_hits", C.c_uint64), ("reserved", C.c_uint64 * 3)]

    def total_rays(self):
        return self.camera_rays + self.bounce_rays + self.sky_shadow_rays + self.light_shadow_rays


class SceneInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint64), ("num_bvh_nodes", C.c_uint64), ("bvh_bytes", C.c_uint64),
                ("triangle_bytes", C.c_uint64), ("shading_bytes", C.c_uint64), ("texture_bytes", C.c_uint64),
                ("num_instances", C.c_uint32), ("bvh_max_depth", C.c_uint32)]


class PathtraceDesc(C.Structure):
    _fields_ = [
        ("scene", C.c_void_p), ("target_rgba16f", C.c_void_p), ("history_r11g11b10", C.c_void_p),
        ("res_x", C.c_uint32), ("res_y", C.c_uint32), ("shard", Shard),
        ("current_sample_index", C.c_uint32), ("max_sample_count", C.c_uint32),
        ("view", View),
        ("camera_aperture", C.c_float), ("camera_focal_length", C.c_float),
        ("env_sky_mode", C.c_int32), ("direct_lighting", C.c_int32), ("viewmode", C.c_int32),
        ("turbidity", C.c_float), ("sky_brightness", C.c_float), ("storage_rounding", C.c_int32),
        ("ray_stats", C.c_void_p),
        ("envmap_r11g11b10", C.c_void_p), ("envmap_res", C.c_uint32), ("envmap_mips", C.c_uint32),
        ("num_samples", C.c_uint32), ("reserved0", C.c_uint32),
        ("debug_hit_chain", C.c_void_p),
    ]


class PathtraceProfile(C.Structure):
    _fields_ = [("generate_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("shadow_gen_ms", C.c_double),
                ("shadow_ms", C.c_double), ("accumulate_ms", C.c_double), ("launches", C.c_uint32 * 5), ("frames", C.c_uint32), ("pipelines", C.c_uint32),
                ("overlap", C.c_uint32), ("shadow_queue_overflows", C.c_uint32), ("shadow_rays_inline", C.c_uint64), ("shadow_rays_asked", C.c_uint64), ("shadow_rays_answered", C.c_uint64)]


class TonemapDesc(C.Structure):
    _fields_ = [("hdr_r11g11b10", C.c_void_p), ("ldr_rgba8", C.c_void_p), ("res_x", C.c_uint32), ("rows", C.c_uint32),
                ("exposure", C.c_float), ("viewmode", C.c_int32)]


# ---- raster path (include/incr.h "Raster path") ----
class ViewConstants(C.Structure):
    _fields_ = [(n, Matrix) for n in (
        "view_transform", "proj_transform", "view_proj_transform", "inv_view_proj_transform", "inv_proj_transform",
        "cull_view_proj_transform", "prev_view_proj_transform", "lightcull_proj_transform", "lightcull_inv_view_proj",
        "sky_inv_view_proj", "unjittered_view_proj", "taa_inv_view_proj")] + [
        ("eye_pos", c_float3), ("near_plane", C.c_float), ("res_x", C.c_uint32), ("res_y", C.c_uint32), ("jitter", C.c_float * 2),
        ("scissor_y0", C.c_uint32), ("scissor_y1", C.c_uint32), ("camera_up", c_float3), ("pad0", C.c_float),
        ("envmap_inv_parallax_view_proj", Matrix)]


class DrawCmd(C.Structure):
    _fields_ = [("id", C.c_uint32), ("vertex_count", C.c_uint32)]


class LightGridData(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("count", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("culledObjects", C.c_int32), ("culledLights", C.c_int32)]


_VC = C.POINTER(ViewConstants)


class BatchCullDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("default_list", C.c_void_p), ("double_sided_list", C.c_void_p),
                ("counts", C.c_void_p), ("render_stats", C.c_void_p), ("frustum_culling", C.c_int32), ("use_primitives", C.c_int32)]


class LightCullDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("cluster_dim", C.c_uint32 * 3), ("max_lights_per_cluster", C.c_uint32),
                ("cluster_depth_extent", C.c_float), ("culled_light_lists", C.c_void_p), ("light_grid", C.c_void_p),
                ("culled_light_count", C.c_void_p), ("render_stats", C.c_void_p)]


class VisibilityDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("default_list", C.c_void_p), ("double_sided_list", C.c_void_p),
                ("counts", C.c_void_p), ("vis_buffer", C.c_void_p), ("depth_buffer", C.c_void_p), ("use_primitives", C.c_int32)]


class GBufferDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("vis_buffer", C.c_void_p), ("gbuffer_basecolor", C.c_void_p),
                ("gbuffer_normals", C.c_void_p), ("gbuffer_metallic_roughness_ao", C.c_void_p), ("color_target", C.c_void_p),
                ("storage_rounding", C.c_int32), ("use_primitives", C.c_int32)]


class LightingInputs(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("color_target", C.c_void_p), ("depth_buffer", C.c_void_p),
                ("gbuffer_basecolor", C.c_void_p), ("gbuffer_normals", C.c_void_p), ("gbuffer_metallic_roughness_ao", C.c_void_p),
                ("lighting_only", C.c_int32), ("storage_rounding", C.c_int32)]


class SkyLightingDesc(C.Structure):
    _fields_ = [("inp", LightingInputs), ("sh9_rgba", C.c_void_p), ("ambient_occlusion", C.c_void_p),
                ("sky_brightness", C.c_float), ("diffuse_ibl", C.c_int32),
                ("prefiltered_envmap", C.c_void_p), ("envmap_res", C.c_uint32), ("envmap_mips", C.c_uint32),
                ("env_brdf", C.c_void_p), ("env_brdf_width", C.c_uint32), ("env_brdf_height", C.c_uint32), ("specular_ibl", C.c_int32),
                ("bent_normals", C.c_void_p), ("use_bent_normals", C.c_int32)]


class EnvironmentmapDesc(C.Structure):
    _fields_ = [("view", _VC), ("color_target", C.c_void_p), ("depth_buffer", C.c_void_p), ("envmap_r11g11b10", C.c_void_p),
                ("envmap_res", C.c_uint32), ("envmap_mips", C.c_uint32), ("sky_brightness", C.c_float), ("storage_rounding", C.c_int32)]


class DebugVizDesc(C.Structure):
    _fields_ = [("view", _VC), ("viewmode", C.c_int32), ("vis_buffer", C.c_void_p), ("gbuffer_basecolor", C.c_void_p),
                ("gbuffer_normals", C.c_void_p), ("gbuffer_metallic_roughness_ao", C.c_void_p), ("depth_buffer", C.c_void_p),
                ("ambient_occlusion", C.c_void_p), ("bent_normals", C.c_void_p), ("color_target", C.c_void_p),
                ("light_cluster_slices", C.c_uint32), ("use_primitives", C.c_int32), ("storage_rounding", C.c_int32)]


class GBufferDecalsDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("depth_buffer", C.c_void_p), ("gbuffer_basecolor", C.c_void_p),
                ("gbuffer_normals", C.c_void_p), ("gbuffer_metallic_roughness_ao", C.c_void_p), ("storage_rounding", C.c_int32)]


class ForwardLightingDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("color_target", C.c_void_p), ("depth_buffer", C.c_void_p), ("sh9_rgba", C.c_void_p),
                ("prefiltered_envmap", C.c_void_p), ("envmap_res", C.c_uint32), ("envmap_mips", C.c_uint32),
                ("env_brdf", C.c_void_p), ("env_brdf_width", C.c_uint32), ("env_brdf_height", C.c_uint32),
                ("viewmode", C.c_int32), ("direct_lighting", C.c_int32), ("diffuse_ibl", C.c_int32), ("specular_ibl", C.c_int32),
                ("ray_stats", C.c_void_p), ("storage_rounding", C.c_int32)]


class HbaoDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("depth_buffer", C.c_void_p), ("gbuffer_normals", C.c_void_p),
                ("ao_target", C.c_void_p), ("bent_normal_target", C.c_void_p), ("ray_stats", C.c_void_p), ("storage_rounding", C.c_int32)]


class PrefilterNormalRoughnessDesc(C.Structure):
    _fields_ = [("normalmap", C.c_void_p), ("metallic_roughness", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("num_mips", C.c_uint32), ("reserved", C.c_uint32), ("out_normalmap", C.c_void_p), ("out_metallic_roughness", C.c_void_p)]


class DirectLightingDesc(C.Structure):
    _fields_ = [("inp", LightingInputs), ("directional_light_index", C.c_int32), ("ray_stats", C.c_void_p)]


class ClusteredLightingDesc(C.Structure):
    _fields_ = [("inp", LightingInputs), ("culled_light_lists", C.c_void_p), ("light_grid", C.c_void_p),
                ("cluster_dim", C.c_uint32 * 3), ("cluster_depth_extent", C.c_float), ("requires_clear", C.c_int32)]


class DynamicSkyDesc(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("view", _VC), ("color_target", C.c_void_p), ("depth_buffer", C.c_void_p),
                ("turbidity", C.c_float), ("storage_rounding", C.c_int32)]


class TaaResolveDesc(C.Structure):
    _fields_ = [("view", _VC), ("hdr_scene_color", C.c_void_p), ("history_in", C.c_void_p), ("history_out", C.c_void_p),
                ("depth_buffer", C.c_void_p), ("history_index", C.c_uint32), ("storage_rounding", C.c_int32),
                ("ldr_rgba8", C.c_void_p), ("exposure", C.c_float), ("viewmode", C.c_int32)]


class DeferredLightingDesc(C.Structure):
    _fields_ = [("sky", C.POINTER(SkyLightingDesc)), ("direct", C.POINTER(DirectLightingDesc)), ("clustered", C.POINTER(ClusteredLightingDesc))]


class SkyIblInfo(C.Structure):
    _fields_ = [("cubemap_res", C.c_uint32), ("num_mips", C.c_uint32), ("latlong_r11g11b10", C.c_void_p),
                ("source_cube_r11g11b10", C.c_void_p), ("envmap_r11g11b10", C.c_void_p), ("sh9_rgba", C.c_void_p),
                ("latlong_texels", C.c_uint64), ("cube_texels", C.c_uint64),
                ("latlong_rgba32f", C.c_void_p), ("latlong_width", C.c_uint32), ("latlong_height", C.c_uint32),
                ("source_cube_res", C.c_uint32), ("source_cube_mips", C.c_uint32)]


class RendererScene(C.Structure):
    _fields_ = [("scene", C.c_void_p), ("directional_light_index", C.c_int32), ("punctual_light_count", C.c_uint32),
                ("meshlet_count", C.c_uint32), ("skylight_sh9_rgba", C.c_void_p),
                ("prefiltered_envmap", C.c_void_p), ("envmap_res", C.c_uint32), ("envmap_mips", C.c_uint32),
                ("env_brdf", C.c_void_p), ("env_brdf_width", C.c_uint32), ("env_brdf_height", C.c_uint32),
                ("primitive_count", C.c_uint32)]


class RendererConfig(C.Structure):
    _fields_ = [("path_trace", C.c_int32), ("viewmode", C.c_int32), ("enable_taa", C.c_int32), ("frustum_culling", C.c_int32),
                ("enable_direct_lighting", C.c_int32), ("enable_sky_lighting", C.c_int32), ("enable_diffuse_ibl", C.c_int32),
                ("exposure", C.c_float), ("sky_brightness", C.c_float), ("camera_near_plane", C.c_float),
                ("light_cluster_dim", C.c_uint32 * 3), ("max_lights_per_cluster", C.c_uint32), ("cluster_depth_extent", C.c_float),
                ("max_sample_count", C.c_uint32), ("camera_aperture", C.c_float), ("camera_focal_length", C.c_float),
                ("turbidity", C.c_float), ("env_sky_mode", C.c_int32), ("storage_rounding", C.c_int32), ("enable_specular_ibl", C.c_int32),
                ("use_primitives", C.c_int32), ("enable_hbao", C.c_int32), ("use_bent_normals", C.c_int32),
                ("forward_lighting", C.c_int32), ("unfused_passes", C.c_int32), ("record_threads", C.c_int32), ("use_graph", C.c_int32)]


(SURFACE_BACK_BUFFER, SURFACE_HDR_RASTER, SURFACE_HDR_RAYTRACE, SURFACE_PATHTRACE_HISTORY, SURFACE_TAA_ACCUMULATION, SURFACE_DEPTH,
 SURFACE_VISIBILITY, SURFACE_GBUFFER_BASECOLOR, SURFACE_GBUFFER_NORMALS, SURFACE_GBUFFER_MRA, SURFACE_RENDER_STATS, SURFACE_RAY_STATS,
 SURFACE_BATCH_COUNTS, SURFACE_AMBIENT_OCCLUSION, SURFACE_BENT_NORMALS) = range(15)


# FConfig defaults the hot path reads (reference: source/demo-dll/inc/common.h:6-48)
class FConfig:
    Fov = 0.25 * 3.14159265358979323846
    Exposure = 13.0
    CameraNearPlane = 1.0
    Viewmode = 0
    EnvSkyMode = 1
    SkyBrightness = 25000.0
    EnableDirectLighting = True
    EnableSkyLighting = True
    EnableDiffuseIBL = True
    EnableSpecularIBL = True
    PathTrace = False
    ForwardLighting = False
    FrustumCulling = True
    EnableTAA = True
    EnableHBAO = False
    UseBentNormals = False           # inc/common.h:29 `UseBentNormals = EnableHBAO`
    UseMeshlets = True
    LightClusterDimX, LightClusterDimY, LightClusterDimZ = 16, 9, 24
    MaxLightsPerCluster = 64
    ClusterDepthExtent = 200.0
    MaxSampleCount = 256
    Pathtracing_CameraAperture = 0.01
    Pathtracing_CameraFocalLength = 7.0
    Turbidity = 2.0
    ToD_Enable = True
    ToD_DecimalHours = 11.0
    ToD_JulianDate = 200
    ToD_Latitude = 42.5
    EnvmapResolution = 256
    # not in the reference: how the C++ Renderer issues the frame (bit-identical surfaces for every combination)
    FusePasses = True
    RecordThreads = 0
    UseGraph = False
