"""Host mirror of the raster branch of Renderer::Render (reference: source/demo-dll/src/renderer.cpp:1711-2036):
meshlet generation at load time, per-frame view constants, and the pass sequence batch cull -> light cull ->
visibility -> g-buffer -> sky lighting -> direct lighting -> clustered lighting -> TAA resolve -> tonemap, each a
call through the C ABI (include/incr.h).  No CPU fallback: every pass needs the HIP library and a GPU.
"""
import ctypes as C

import numpy as np

from . import ctypes_api as A
from . import runtime as rt

MAX_MESHLET_VERTS = 64        # scene.cpp:1545-1546
MAX_MESHLET_PRIMS = 126


def generate_meshlets(scene, max_verts=MAX_MESHLET_VERTS, max_prims=MAX_MESHLET_PRIMS):
    """FScene::GenerateMeshlets + the meshlet packing of CreateGpuGeometryBuffers (scene.cpp:1488-1558, 503-560):
    meshletize every primitive with the host utility of the HIP library and pack mesh -> primitive -> meshlet.
    Returns a copy of `scene` that carries the meshlets (the arrays of `scene` are shared, never replaced)."""
    lib = rt.load()
    meshlets, vert_lists, tri_lists = [], [], []
    vbase = tbase = 0
    for mesh_index, prims in enumerate(scene.mesh_primitives):
        for p in prims:
            n = p["index_count"]
            stride = scene.accessors[p["index"]][2]
            idx = scene.accessor_array(p["index"], np.uint16 if stride == 2 else np.uint32, 1, n).astype(np.uint32).ravel()
            # positions.size() of the POSITION accessor (scene.cpp:1531-1542): the adjacency hash has vertexCount / 3 buckets
            nverts = max(int(p.get("vertex_count", 0)), int(idx.max()) + 1 if n else 0)
            pos = np.ascontiguousarray(scene.accessor_array(p["position"], np.float32, 3, nverts))
            idx = np.ascontiguousarray(idx)
            handle = C.c_void_p()
            rt.check(lib.incr_meshletize(idx.ctypes.data, n, pos.ctypes.data, nverts, 12, max_verts, max_prims, C.byref(handle)))
            try:
                nvi, ntri = C.c_uint32(), C.c_uint32()
                nm = lib.incr_meshlet_set_counts(handle, C.byref(nvi), C.byref(ntri))
                ml = (A.GpuMeshlet * max(nm, 1))()
                vi = np.zeros(max(nvi.value, 1), dtype=np.uint32)
                tri = np.zeros(max(ntri.value, 1), dtype=np.uint32)
                rt.check(lib.incr_meshlet_set_copy(handle, ml, vi.ctypes.data, tri.ctypes.data))
            finally:
                lib.incr_meshlet_set_free(handle)
            for k in range(nm):
                m = ml[k]
                m.vertexBegin += vbase
                m.triangleBegin += tbase
                m.meshIndex = mesh_index
                m.positionAccessor, m.uvAccessor = p["position"], p["uv"]
                m.normalAccessor, m.tangentAccessor = p["normal"], p["tangent"]
                m.materialIndex = p["material"]
                meshlets.append(m)
            vert_lists.append(vi[:nvi.value]); tri_lists.append(tri[:ntri.value])
            vbase += nvi.value; tbase += ntri.value
    arr = (A.GpuMeshlet * max(len(meshlets), 1))()
    for i, m in enumerate(meshlets):
        C.memmove(C.byref(arr, i * C.sizeof(A.GpuMeshlet)), C.byref(m), C.sizeof(A.GpuMeshlet))
    vi = np.concatenate(vert_lists) if vert_lists else np.zeros(0, np.uint32)
    tri = np.concatenate(tri_lists) if tri_lists else np.zeros(0, np.uint32)
    # A NEW scene object: the caller's scene and everything built from it (a GPU scene, an oracle scene that borrows its arrays by pointer) stay as they were
    import copy
    out = copy.copy(scene)
    out._keep, out._desc = [], None
    out.meshlets = (arr, np.ascontiguousarray(vi, dtype=np.uint32), np.ascontiguousarray(tri, dtype=np.uint32))
    out.num_meshlets = len(meshlets)
    return out


def pixel_jitter(frame_index, res_x, res_y):
    out = (C.c_float * 2)()
    rt.load().incr_pixel_jitter(frame_index, res_x, res_y, out)
    return np.array([out[0], out[1]], dtype=np.float32)


def make_view_constants(view, res_x, res_y, jitter=(0.0, 0.0), prev_view_proj=None, near_plane=A.FConfig.CameraNearPlane, culling_view=None):
    vc = A.ViewConstants()
    v = view.to_ctypes()
    cv = culling_view.to_ctypes() if culling_view is not None else None
    j = (C.c_float * 2)(float(jitter[0]), float(jitter[1]))
    prev = None
    if prev_view_proj is not None:
        prev = A.Matrix()
        flat = np.asarray(prev_view_proj, dtype=np.float32).ravel()
        for i in range(16):
            prev.m[i] = float(flat[i])
    rt.check(rt.load().incr_make_view_constants(C.byref(v), C.byref(cv) if cv is not None else None, j,
                                                C.byref(prev) if prev is not None else None, near_plane, res_x, res_y, C.byref(vc)))
    return vc


def matrix_to_numpy(m):
    return np.array(list(m.m), dtype=np.float32).reshape(4, 4)


def directional_light_index(scene):
    """FScene::GetDirectionalLight (scene.cpp:1581-1591): GLOBAL index of the first directional scene light, or -1."""
    for gi in scene.light_indices:
        if scene.global_lights[gi].type == A.LIGHT_DIRECTIONAL:
            return int(gi)
    return -1


def punctual_light_count(scene):
    return sum(1 for gi in scene.light_indices if scene.global_lights[gi].type != A.LIGHT_DIRECTIONAL)


class RasterSurfaces:
    """Device surfaces of one raster frame (renderer.cpp:1364-1520), linear res_x*res_y arrays."""

    SPECS = dict(vis=4, depth=4, gbuffer_basecolor=4, gbuffer_normals=4, gbuffer_mra=4, color=4, taa0=4, taa1=4, ldr=4)

    def __init__(self, ctx, res_x, res_y, num_meshlets, cfg=A.FConfig):
        self.ctx, self.res_x, self.res_y = ctx, res_x, res_y
        n = res_x * res_y
        self.ptr = {k: ctx.buffer(n * b) for k, b in self.SPECS.items()}
        self.clusters = cfg.LightClusterDimX * cfg.LightClusterDimY * cfg.LightClusterDimZ
        self.max_lights = cfg.MaxLightsPerCluster
        self.num_meshlets = num_meshlets
        self.ptr["default_list"] = ctx.buffer(max(num_meshlets, 1) * 8)
        self.ptr["double_sided_list"] = ctx.buffer(max(num_meshlets, 1) * 8)
        self.ptr["ao"] = ctx.buffer(n)                      # R8_UNORM hbao target
        self.ptr["bent_normals"] = ctx.buffer(n * 4)        # R16G16_FLOAT
        self.ptr["counts"] = ctx.buffer(8)
        self.ptr["render_stats"] = ctx.buffer(8)
        self.ptr["light_lists"] = ctx.buffer(self.clusters * self.max_lights * 4)
        self.ptr["light_grid"] = ctx.buffer(self.clusters * 8)
        self.ptr["light_count"] = ctx.buffer(4)
        self.ptr["ray_stats"] = ctx.buffer(C.sizeof(A.RayStats))
        self.ptr["sh9"] = ctx.buffer(9 * 16)
        for k in ("counts", "render_stats", "light_count"):
            ctx.clear(self.ptr[k], 8 if k != "light_count" else 4)
        ctx.clear(self.ptr["ray_stats"], C.sizeof(A.RayStats))
        ctx.clear(self.ptr["light_grid"], self.clusters * 8)
        ctx.clear(self.ptr["light_lists"], self.clusters * self.max_lights * 4)
        ctx.clear(self.ptr["taa0"], n * 4); ctx.clear(self.ptr["taa1"], n * 4)

    def download(self, name):
        ctx, n = self.ctx, self.res_x * self.res_y
        shape2 = (self.res_y, self.res_x)
        if name == "depth":
            return ctx.download(self.ptr[name], shape2, np.float32)
        if name in self.SPECS or name == "bent_normals":
            return ctx.download(self.ptr[name], shape2, np.uint32)
        if name == "ao":
            return ctx.download(self.ptr[name], shape2, np.uint8)
        if name in ("default_list", "double_sided_list"):
            return ctx.download(self.ptr[name], (max(self.num_meshlets, 1), 2), np.uint32)
        if name == "counts":
            return ctx.download(self.ptr[name], (2,), np.uint32)
        if name == "render_stats":
            return ctx.download(self.ptr[name], (2,), np.int32)
        if name == "light_lists":
            return ctx.download(self.ptr[name], (self.clusters, self.max_lights), np.uint32)
        if name == "light_grid":
            return ctx.download(self.ptr[name], (self.clusters, 2), np.uint32)
        if name == "light_count":
            return ctx.download(self.ptr[name], (1,), np.uint32)
        if name == "ray_stats":
            raw = ctx.download(self.ptr[name], (C.sizeof(A.RayStats),), np.uint8)
            return A.RayStats.from_buffer_copy(raw.tobytes())
        raise KeyError(name)

    def close(self):
        for p in self.ptr.values():
            self.ctx.free(p)
        self.ptr = {}


def lighting_inputs(scene_handle, vc, surf, cfg=A.FConfig, storage_rounding=0):
    i = A.LightingInputs()
    i.scene = scene_handle
    i.view = C.pointer(vc)
    i.color_target, i.depth_buffer = surf["color"], surf["depth"]
    i.gbuffer_basecolor, i.gbuffer_normals, i.gbuffer_metallic_roughness_ao = surf["gbuffer_basecolor"], surf["gbuffer_normals"], surf["gbuffer_mra"]
    i.lighting_only = 1 if cfg.Viewmode == 1 else 0
    i.storage_rounding = storage_rounding
    return i


def with_scissor(vc, row_begin, row_end):
    c = A.ViewConstants.from_buffer_copy(vc)
    c.scissor_y0, c.scissor_y1 = row_begin, row_end
    return c


def build_descs(scene_handle, packed_scene, vc, surf, frame_index, cfg=A.FConfig, storage_rounding=0, have_sh=True, band=None, ibl=None):
    """All pass descs of one raster frame over a dict of surface pointers (device pointers for the HIP library, host
    pointers for the oracle: both take the same structs).  Returns an ordered list of (pass name, desc).
    band = (row_begin, row_end): the sort-first multi-GPU variant -- draw and shade the band plus a one-row halo,
    resolve TAA for the band only (csrc/host/renderer.cpp does the same in C++)."""
    out = []
    vc_full = vc
    vc_taa = vc
    if band is not None:
        y0, y1 = band
        vc = with_scissor(vc_full, max(y0, 1) - 1, min(y1 + 1, vc_full.res_y))
        vc_taa = with_scissor(vc_full, y0, min(y1, vc_full.res_y))
    vcp = C.pointer(vc)
    d = A.BatchCullDesc(); d.scene, d.view = scene_handle, vcp
    d.default_list, d.double_sided_list, d.counts, d.render_stats = surf["default_list"], surf["double_sided_list"], surf["counts"], surf["render_stats"]
    d.frustum_culling = 1 if cfg.FrustumCulling else 0
    use_prims = 0 if cfg.UseMeshlets else 1          # renderer.cpp:1716-1761: the primitive variants of cull / visibility / gbuffer
    d.use_primitives = use_prims
    out.append(("batch_cull", d))
    punctual = punctual_light_count(packed_scene)
    sun = directional_light_index(packed_scene)
    dims = (cfg.LightClusterDimX, cfg.LightClusterDimY, cfg.LightClusterDimZ)
    if punctual > 0 and cfg.EnableDirectLighting:
        d = A.LightCullDesc(); d.scene, d.view = scene_handle, vcp
        d.cluster_dim[:] = dims
        d.max_lights_per_cluster, d.cluster_depth_extent = cfg.MaxLightsPerCluster, cfg.ClusterDepthExtent
        d.culled_light_lists, d.light_grid, d.culled_light_count, d.render_stats = surf["light_lists"], surf["light_grid"], surf["light_count"], surf["render_stats"]
        out.append(("light_cull", d))
    d = A.VisibilityDesc(); d.scene, d.view = scene_handle, vcp
    d.default_list, d.double_sided_list, d.counts = surf["default_list"], surf["double_sided_list"], surf["counts"]
    d.vis_buffer, d.depth_buffer = surf["vis"], surf["depth"]
    d.use_primitives = use_prims
    out.append(("visibility", d))
    d = A.GBufferDesc(); d.scene, d.view = scene_handle, vcp
    d.vis_buffer, d.gbuffer_basecolor, d.gbuffer_normals, d.gbuffer_metallic_roughness_ao = surf["vis"], surf["gbuffer_basecolor"], surf["gbuffer_normals"], surf["gbuffer_mra"]
    d.color_target, d.storage_rounding = surf["color"], storage_rounding
    d.use_primitives = use_prims
    out.append(("gbuffer", d))
    if getattr(packed_scene, "decal_primitives", None):      # renderer.cpp:1783-1798 GBufferRasterPass (decals), unconditional there
        d = A.GBufferDecalsDesc(); d.scene, d.view = scene_handle, vcp
        d.depth_buffer, d.gbuffer_basecolor, d.gbuffer_normals, d.gbuffer_metallic_roughness_ao = surf["depth"], surf["gbuffer_basecolor"], surf["gbuffer_normals"], surf["gbuffer_mra"]
        d.storage_rounding = storage_rounding
        out.append(("gbuffer_decals", d))
    if cfg.EnableHBAO:              # renderer.cpp:1801-1815; off = both targets cleared to 1 = NULL inputs of sky lighting
        d = A.HbaoDesc(); d.scene, d.view = scene_handle, vcp
        d.depth_buffer, d.gbuffer_normals = surf["depth"], surf["gbuffer_normals"]
        d.ao_target, d.bent_normal_target, d.ray_stats = surf["ao"], surf["bent_normals"], surf["ray_stats"]
        d.storage_rounding = storage_rounding
        out.append(("hbao", d))
    if cfg.EnableSkyLighting:
        d = A.SkyLightingDesc(); d.inp = lighting_inputs(scene_handle, vc, surf, cfg, storage_rounding)
        d.sh9_rgba = surf["sh9"] if have_sh else None
        d.ambient_occlusion = surf["ao"] if cfg.EnableHBAO else None
        d.bent_normals = surf["bent_normals"] if cfg.EnableHBAO else None
        d.use_bent_normals = 1 if cfg.UseBentNormals else 0
        d.sky_brightness, d.diffuse_ibl = cfg.SkyBrightness, 1 if cfg.EnableDiffuseIBL else 0
        d.specular_ibl = 1 if cfg.EnableSpecularIBL else 0
        if ibl is not None:         # IblInputs: pointers valid for whoever runs the pass (device for the library, host for the oracle)
            if ibl.sh9:
                d.sh9_rgba = ibl.sh9
            d.prefiltered_envmap, d.envmap_res, d.envmap_mips = ibl.envmap, ibl.envmap_res, ibl.envmap_mips
            d.env_brdf, d.env_brdf_width, d.env_brdf_height = ibl.env_brdf, ibl.env_brdf_size[0], ibl.env_brdf_size[1]
        out.append(("sky_lighting", d))
    if cfg.ForwardLighting:         # renderer.cpp:1865-1880: the forward pass replaces the deferred direct + clustered passes
        d = A.ForwardLightingDesc(); d.scene, d.view = scene_handle, vcp
        d.color_target, d.depth_buffer = surf["color"], surf["depth"]
        d.sh9_rgba = surf["sh9"] if have_sh else None
        if ibl is not None:
            if ibl.sh9:
                d.sh9_rgba = ibl.sh9
            d.prefiltered_envmap, d.envmap_res, d.envmap_mips = ibl.envmap, ibl.envmap_res, ibl.envmap_mips
            d.env_brdf, d.env_brdf_width, d.env_brdf_height = ibl.env_brdf, ibl.env_brdf_size[0], ibl.env_brdf_size[1]
        d.viewmode, d.direct_lighting = cfg.Viewmode, 1 if cfg.EnableDirectLighting else 0
        d.diffuse_ibl, d.specular_ibl = 1 if cfg.EnableDiffuseIBL else 0, 1 if cfg.EnableSpecularIBL else 0
        d.ray_stats, d.storage_rounding = surf["ray_stats"], storage_rounding
        out.append(("forward_lighting", d))
    if sun != -1 and cfg.EnableDirectLighting and not cfg.ForwardLighting:
        d = A.DirectLightingDesc(); d.inp = lighting_inputs(scene_handle, vc, surf, cfg, storage_rounding)
        d.directional_light_index, d.ray_stats = sun, surf["ray_stats"]
        out.append(("direct_lighting", d))
    if punctual > 0 and cfg.EnableDirectLighting and not cfg.ForwardLighting:
        d = A.ClusteredLightingDesc(); d.inp = lighting_inputs(scene_handle, vc, surf, cfg, storage_rounding)
        d.culled_light_lists, d.light_grid = surf["light_lists"], surf["light_grid"]
        d.cluster_dim[:] = dims
        d.cluster_depth_extent, d.requires_clear = cfg.ClusterDepthExtent, 1 if sun == -1 else 0
        out.append(("clustered_lighting", d))
    if cfg.EnvSkyMode == 0:         # renderer.cpp:1927-1943 EnvSkyMode::HDRI: the environment cube fills the background
        if ibl is None or not ibl.envmap:
            raise ValueError("EnvSkyMode 0 (HDRI) needs the environment cube (IblInputs.envmap)")
        d = A.EnvironmentmapDesc(); d.view = vcp
        d.color_target, d.depth_buffer = surf["color"], surf["depth"]
        d.envmap_r11g11b10, d.envmap_res, d.envmap_mips = ibl.envmap, ibl.envmap_res, ibl.envmap_mips
        d.sky_brightness, d.storage_rounding = cfg.SkyBrightness, storage_rounding
        out.append(("environmentmap", d))
    else:                           # :1944-1959: dynamic sky
        d = A.DynamicSkyDesc(); d.scene, d.view = scene_handle, vcp
        d.color_target, d.depth_buffer = surf["color"], surf["depth"]
        d.turbidity, d.storage_rounding = cfg.Turbidity, storage_rounding
        out.append(("dynamic_sky", d))
    if cfg.Viewmode not in (0, 1):  # renderer.cpp:1961-1984 bDebugView
        d = A.DebugVizDesc(); d.view, d.viewmode = vcp, cfg.Viewmode
        d.vis_buffer, d.gbuffer_basecolor, d.gbuffer_normals, d.gbuffer_metallic_roughness_ao = surf["vis"], surf["gbuffer_basecolor"], surf["gbuffer_normals"], surf["gbuffer_mra"]
        d.depth_buffer, d.color_target = surf["depth"], surf["color"]
        d.ambient_occlusion = surf["ao"] if cfg.EnableHBAO else None
        d.bent_normals = surf["bent_normals"] if cfg.EnableHBAO else None
        d.light_cluster_slices, d.use_primitives, d.storage_rounding = cfg.LightClusterDimZ, use_prims, storage_rounding
        out.append(("debug_visualization", d))
    if cfg.EnableTAA:
        d = A.TaaResolveDesc(); d.view = C.pointer(vc_taa)
        d.hdr_scene_color, d.depth_buffer = surf["color"], surf["depth"]
        d.history_in, d.history_out = surf["taa%d" % ((frame_index + 1) & 1)], surf["taa%d" % (frame_index & 1)]
        d.history_index, d.storage_rounding = frame_index, storage_rounding
        out.append(("taa_resolve", d))
    return out


def renderer_config(cfg=A.FConfig, path_trace=None, storage_rounding=0):
    c = A.RendererConfig()
    c.path_trace = int(cfg.PathTrace if path_trace is None else path_trace)
    c.viewmode, c.enable_taa, c.frustum_culling = cfg.Viewmode, int(cfg.EnableTAA), int(cfg.FrustumCulling)
    c.enable_direct_lighting, c.enable_sky_lighting, c.enable_diffuse_ibl = int(cfg.EnableDirectLighting), int(cfg.EnableSkyLighting), int(cfg.EnableDiffuseIBL)
    c.exposure, c.sky_brightness, c.camera_near_plane = cfg.Exposure, cfg.SkyBrightness, cfg.CameraNearPlane
    c.light_cluster_dim[:] = (cfg.LightClusterDimX, cfg.LightClusterDimY, cfg.LightClusterDimZ)
    c.max_lights_per_cluster, c.cluster_depth_extent = cfg.MaxLightsPerCluster, cfg.ClusterDepthExtent
    c.max_sample_count, c.camera_aperture, c.camera_focal_length = cfg.MaxSampleCount, cfg.Pathtracing_CameraAperture, cfg.Pathtracing_CameraFocalLength
    c.turbidity, c.env_sky_mode, c.storage_rounding = cfg.Turbidity, cfg.EnvSkyMode, storage_rounding
    c.enable_specular_ibl = int(cfg.EnableSpecularIBL)
    c.use_primitives = 0 if cfg.UseMeshlets else 1
    c.enable_hbao, c.use_bent_normals = int(cfg.EnableHBAO), int(cfg.UseBentNormals)
    c.forward_lighting = int(cfg.ForwardLighting)
    c.unfused_passes = 0 if getattr(cfg, "FusePasses", True) else 1
    c.record_threads = int(getattr(cfg, "RecordThreads", 0)); c.use_graph = int(getattr(cfg, "UseGraph", False))
    return c


class IblInputs:
    """Device pointers of the sky-lighting inputs that the precompute produces (ibl.SkyIbl + ibl.generate_env_brdf)."""

    def __init__(self, sh9=None, envmap=None, envmap_res=0, envmap_mips=0, env_brdf=None, env_brdf_size=(0, 0)):
        self.sh9, self.envmap, self.envmap_res, self.envmap_mips = sh9, envmap, envmap_res, envmap_mips
        self.env_brdf, self.env_brdf_size = env_brdf, env_brdf_size

    @staticmethod
    def from_sky(sky, env_brdf=None, env_brdf_size=(0, 0)):
        i = sky.info
        return IblInputs(i.sh9_rgba, i.envmap_r11g11b10, i.cubemap_res, i.num_mips, env_brdf, env_brdf_size)


class HostRenderer:
    """The C++ host mirror of Renderer::Render (csrc/host/renderer.cpp) through incr_renderer_*: one call per frame.
    This is the production frame loop; RasterRenderer below issues the same passes one by one from Python and exists so
    that tests can compare every intermediate surface with the oracle."""

    def __init__(self, ctx, scene, res_x, res_y, cfg=A.FConfig, path_trace=False, sh9_rgb=None, stream=None, storage_rounding=0, ibl=None):
        self.ctx, self.scene, self.res_x, self.res_y = ctx, scene, res_x, res_y
        self.sh9 = None
        if sh9_rgb is not None:
            a = np.zeros((9, 4), dtype=np.float32); a[:, :3] = np.asarray(sh9_rgb, dtype=np.float32).reshape(9, 3)
            self.sh9 = ctx.buffer(9 * 16)
            ctx.upload(self.sh9, a)
        s = A.RendererScene()
        s.scene = scene.handle
        s.directional_light_index = directional_light_index(scene.packed)
        s.punctual_light_count = punctual_light_count(scene.packed)
        s.meshlet_count = getattr(scene.packed, "num_meshlets", 0)
        s.primitive_count = sum(len(p) for p in scene.packed.mesh_primitives)
        s.skylight_sh9_rgba = self.sh9
        if ibl is not None:         # products of the sky / IBL precompute (IblInputs)
            if ibl.sh9:
                s.skylight_sh9_rgba = ibl.sh9
            s.prefiltered_envmap, s.envmap_res, s.envmap_mips = ibl.envmap, ibl.envmap_res, ibl.envmap_mips
            s.env_brdf, s.env_brdf_width, s.env_brdf_height = ibl.env_brdf, ibl.env_brdf_size[0], ibl.env_brdf_size[1]
        c = renderer_config(cfg, path_trace, storage_rounding)
        self.handle = C.c_void_p()
        rt.check(ctx.lib.incr_renderer_create(ctx.handle, stream, res_x, res_y, C.byref(s), C.byref(c), C.byref(self.handle)))

    def render(self, view, culling_view=None):
        v = view.to_ctypes()
        cv = culling_view.to_ctypes() if culling_view is not None else None
        rc = self.ctx.lib.incr_renderer_render(self.handle, C.byref(v), C.byref(cv) if cv is not None else None)
        if rc != 0:
            raise rt.IncrError(rc, self.ctx.lib.incr_renderer_last_error(self.handle).decode())

    def set_band(self, row_begin, row_end):
        """Sort-first multi-GPU: render only rows [row_begin, row_end) (+ halo); the caller gathers the TAA bands and calls tonemap()."""
        rt.check(self.ctx.lib.incr_renderer_set_band(self.handle, row_begin, row_end))

    def tonemap(self):
        rt.check(self.ctx.lib.incr_renderer_tonemap(self.handle))

    def surface_ptr(self, which):
        return self.ctx.lib.incr_renderer_surface(self.handle, which)

    def pass_times(self):
        """GPU ms per pass of the last frame rendered with profiling on (set_profiling(True) first): list of (pass, ms)."""
        names = ((C.c_char * 32) * 64)(); ms = (C.c_float * 64)(); n = C.c_uint32()
        rt.check(self.ctx.lib.incr_renderer_get_pass_times(self.handle, names, ms, 64, C.byref(n)))
        return [(names[i].value.decode(), float(ms[i])) for i in range(n.value)]

    def set_profiling(self, on):
        rt.check(self.ctx.lib.incr_renderer_set_profiling(self.handle, int(bool(on))))

    def frame_info(self):
        """(host seconds of the last render call, frames replayed from the HIP graph, in-place graph updates, graph rebuilds)."""
        t = C.c_double(); a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rt.check(self.ctx.lib.incr_renderer_get_frame_info(self.handle, C.byref(t), C.byref(a), C.byref(b), C.byref(c)))
        return t.value, a.value, b.value, c.value

    def stats(self):
        """Renderer::GetRenderStats: ((culled objects, culled lights) of the last frame, accumulated RayStats)."""
        rs, rays = (C.c_int32 * 2)(), A.RayStats()
        rt.check(self.ctx.lib.incr_renderer_get_stats(self.handle, C.byref(rs), C.byref(rays)))
        return (rs[0], rs[1]), rays

    def download(self, which, dtype=np.uint32, texel_words=1):
        shape = (self.res_y, self.res_x) if texel_words == 1 else (self.res_y, self.res_x, texel_words)
        return self.ctx.download(self.surface_ptr(which), shape, dtype)

    def close(self):
        if self.handle:
            self.ctx.lib.incr_renderer_destroy(self.handle)
            self.handle = C.c_void_p()
        if self.sh9:
            self.ctx.free(self.sh9); self.sh9 = None


class RasterRenderer:
    """One raster frame per render() call, all passes enqueued on one stream (job-sync.h:27-61 token order)."""

    PASS_FN = dict(batch_cull="incr_batch_cull", light_cull="incr_light_cull", visibility="incr_visibility", gbuffer="incr_gbuffer",
                   sky_lighting="incr_sky_lighting", direct_lighting="incr_direct_lighting", clustered_lighting="incr_clustered_lighting",
                   taa_resolve="incr_taa_resolve", dynamic_sky="incr_dynamic_sky", hbao="incr_hbao", environmentmap="incr_environmentmap", debug_visualization="incr_debug_visualization",
                   forward_lighting="incr_forward_lighting", gbuffer_decals="incr_gbuffer_decals")

    def __init__(self, ctx, scene, res_x, res_y, cfg=A.FConfig):
        self.ctx, self.scene, self.cfg = ctx, scene, cfg
        self.res_x, self.res_y = res_x, res_y
        draws = max(getattr(scene.packed, "num_meshlets", 0), sum(len(p) for p in scene.packed.mesh_primitives))   # renderer.cpp:1375-1377
        self.surf = RasterSurfaces(ctx, res_x, res_y, draws, cfg)
        self.prev_view_proj = None
        self.frame_index = 0
        self.have_sh = False

    def set_sh9(self, sh9_rgb):
        """Skylight probe: 9 RGB SH coefficients (FScene::m_skylight.m_shTextureIndex, a 9x1 RGBA32F texture)."""
        a = np.zeros((9, 4), dtype=np.float32); a[:, :3] = np.asarray(sh9_rgb, dtype=np.float32).reshape(9, 3)
        self.ctx.upload(self.surf.ptr["sh9"], a)
        self.have_sh = True

    def view_constants(self, view, culling_view=None):
        cfg = self.cfg
        jit = pixel_jitter(self.frame_index, self.res_x, self.res_y) if (cfg.EnableTAA and cfg.Viewmode == 0) else (0.0, 0.0)
        return make_view_constants(view, self.res_x, self.res_y, jit, self.prev_view_proj, cfg.CameraNearPlane, culling_view)

    def render(self, view, stream=None, storage_rounding=0, tonemap=True, vc=None):
        ctx, lib = self.ctx, self.ctx.lib
        vc = vc if vc is not None else self.view_constants(view)
        self.last_vc = vc
        fused = getattr(self, "fused", False)        # FConfig::FusePasses of the C++ renderer, issued from Python: one lighting kernel, tonemap inside TAA
        lighting = {}
        descs = build_descs(self.scene.handle, self.scene.packed, vc, self.surf.ptr, self.frame_index, self.cfg, storage_rounding, self.have_sh,
                            ibl=getattr(self, "ibl", None))      # self.ibl: IblInputs with device pointers (optional)
        last_lighting = max([k for k, (n, _) in enumerate(descs) if n in ("sky_lighting", "direct_lighting", "clustered_lighting")], default=-1)
        for k, (name, d) in enumerate(descs):
            if fused and not self.cfg.ForwardLighting and name in ("sky_lighting", "direct_lighting", "clustered_lighting"):
                lighting[name] = d
                if k == last_lighting:
                    f = A.DeferredLightingDesc()
                    f.sky = C.pointer(lighting["sky_lighting"]) if "sky_lighting" in lighting else None
                    f.direct = C.pointer(lighting["direct_lighting"]) if "direct_lighting" in lighting else None
                    f.clustered = C.pointer(lighting["clustered_lighting"]) if "clustered_lighting" in lighting else None
                    rt.check(lib.incr_deferred_lighting(ctx.handle, C.byref(f), stream))
                continue
            if fused and name == "taa_resolve" and tonemap:
                d.ldr_rgba8, d.exposure, d.viewmode = self.surf.ptr["ldr"], self.cfg.Exposure, self.cfg.Viewmode
                tonemap = False
            rt.check(getattr(lib, self.PASS_FN[name])(ctx.handle, C.byref(d), stream))
        hdr = self.surf.ptr["taa%d" % (self.frame_index & 1)] if self.cfg.EnableTAA else self.surf.ptr["color"]
        if tonemap:
            t = A.TonemapDesc(); t.hdr_r11g11b10, t.ldr_rgba8 = hdr, self.surf.ptr["ldr"]
            t.res_x, t.rows, t.exposure, t.viewmode = self.res_x, self.res_y, self.cfg.Exposure, self.cfg.Viewmode
            ctx.tonemap(t, stream)
        self.prev_view_proj = matrix_to_numpy(vc.unjittered_view_proj)       # renderer.cpp:2025
        self.frame_index += 1
        return vc

    def close(self):
        self.surf.close()
