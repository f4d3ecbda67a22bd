"""Loader and thin object wrappers for the C-ABI library csrc/libincr_hip.so (include/incr.h).

There is NO fallback path: `load()` raises if the HIP library is missing, and `Context()` raises
INCR_ERR_NO_DEVICE when no GPU is visible.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import ctypes_api as A

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libincr_hip.so")

# every symbol include/incr.h declares (tests/test_abi.py checks the library exports all of them)
EXPORTS = [
    "incr_initialize", "incr_teardown", "incr_last_error", "incr_version", "incr_device_count",
    "incr_buffer_create", "incr_buffer_destroy", "incr_buffer_upload", "incr_buffer_download", "incr_buffer_clear",
    "incr_stream_synchronize", "incr_scene_create", "incr_scene_destroy", "incr_update_bvh", "incr_scene_set_sun_dir", "incr_scene_set_max_anisotropy",
    "incr_scene_get_info", "incr_scene_texture_info", "incr_scene_texture_download", "incr_shard_local_rows", "incr_pathtrace", "incr_tonemap", "incr_unshard",
    "incr_trace_rays", "incr_selftest_small_float", "incr_selftest_reciprocal", "incr_scene_validate_host", "incr_profile_begin", "incr_profile_end", "incr_set_pathtrace_pipelines", "incr_set_pathtrace_overlap", "incr_set_pathtrace_accumulate_placement",
    # raster path
    "incr_generate_tangents", "incr_decode_jpeg", "incr_meshletize", "incr_meshlet_set_counts", "incr_meshlet_set_copy", "incr_meshlet_set_free",
    "incr_pixel_jitter", "incr_make_view_constants", "incr_batch_cull", "incr_light_cull", "incr_visibility", "incr_gbuffer",
    "incr_sky_lighting", "incr_direct_lighting", "incr_clustered_lighting", "incr_taa_resolve", "incr_dynamic_sky",
    "incr_hbao", "incr_environmentmap", "incr_debug_visualization", "incr_forward_lighting", "incr_gbuffer_decals",
    "incr_deferred_lighting",
    # frame level (C++ host mirror of Renderer::Render)
    "incr_renderer_create", "incr_renderer_destroy", "incr_renderer_set_config", "incr_renderer_render",
    "incr_renderer_reset_accumulation", "incr_renderer_last_error", "incr_renderer_surface",
    "incr_renderer_set_band", "incr_renderer_tonemap", "incr_visibility_get_stats", "incr_renderer_get_stats", "incr_renderer_get_frame_info", "incr_renderer_tick", "incr_renderer_set_profiling", "incr_renderer_get_pass_times",
    # sky / IBL precompute
    "incr_update_dynamic_sky", "incr_cache_hdri", "incr_decode_hdr", "incr_sky_ibl_get_info", "incr_sky_ibl_destroy", "incr_generate_env_brdf",
    # texture ingest, GPU step
    "incr_compressed_mip_count", "incr_mip_chain_texels", "incr_prefilter_normal_roughness",
]


class IncrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("incr error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j4"], check=True)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IncrError(-5, "HIP library %s is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the render path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.incr_initialize.restype = C.c_int; L.incr_initialize.argtypes = [C.c_int, C.POINTER(vp)]
    L.incr_teardown.restype = C.c_int; L.incr_teardown.argtypes = [vp]
    L.incr_last_error.restype = C.c_char_p; L.incr_last_error.argtypes = []
    L.incr_version.restype = C.c_char_p; L.incr_version.argtypes = []
    L.incr_device_count.restype = C.c_int; L.incr_device_count.argtypes = []
    L.incr_buffer_create.restype = C.c_int; L.incr_buffer_create.argtypes = [vp, u64, C.POINTER(vp)]
    L.incr_buffer_destroy.restype = C.c_int; L.incr_buffer_destroy.argtypes = [vp, vp]
    L.incr_buffer_upload.restype = C.c_int; L.incr_buffer_upload.argtypes = [vp, vp, vp, u64, vp]
    L.incr_buffer_download.restype = C.c_int; L.incr_buffer_download.argtypes = [vp, vp, vp, u64, vp]
    L.incr_buffer_clear.restype = C.c_int; L.incr_buffer_clear.argtypes = [vp, vp, u32, u64, vp]
    L.incr_stream_synchronize.restype = C.c_int; L.incr_stream_synchronize.argtypes = [vp, vp]
    L.incr_scene_create.restype = C.c_int; L.incr_scene_create.argtypes = [vp, C.POINTER(A.SceneDesc), C.POINTER(vp)]
    L.incr_scene_destroy.restype = C.c_int; L.incr_scene_destroy.argtypes = [vp, vp]
    L.incr_update_bvh.restype = C.c_int; L.incr_update_bvh.argtypes = [vp, vp, vp, vp, vp, vp]
    L.incr_scene_set_sun_dir.restype = C.c_int; L.incr_scene_set_sun_dir.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.incr_scene_set_max_anisotropy.restype = C.c_int; L.incr_scene_set_max_anisotropy.argtypes = [vp, vp, u32]
    L.incr_scene_texture_info.restype = C.c_int
    L.incr_scene_texture_info.argtypes = [vp, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
    L.incr_scene_texture_download.restype = C.c_int; L.incr_scene_texture_download.argtypes = [vp, vp, u32, vp, u64]
    L.incr_scene_get_info.restype = C.c_int; L.incr_scene_get_info.argtypes = [vp, vp, C.POINTER(A.SceneInfo)]
    L.incr_shard_local_rows.restype = u32; L.incr_shard_local_rows.argtypes = [u32, C.POINTER(A.Shard)]
    L.incr_pathtrace.restype = C.c_int; L.incr_pathtrace.argtypes = [vp, C.POINTER(A.PathtraceDesc), vp]
    L.incr_tonemap.restype = C.c_int; L.incr_tonemap.argtypes = [vp, C.POINTER(A.TonemapDesc), vp]
    L.incr_unshard.restype = C.c_int; L.incr_unshard.argtypes = [vp, vp, vp, u32, u32, u32, u32, u32, vp]
    L.incr_trace_rays.restype = C.c_int
    L.incr_trace_rays.argtypes = [vp, vp, u32, vp, vp, C.c_float, C.c_float, C.c_int, vp, vp, vp]
    L.incr_selftest_small_float.restype = C.c_int; L.incr_selftest_small_float.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.incr_profile_begin.restype = C.c_int; L.incr_profile_begin.argtypes = [vp]
    L.incr_set_pathtrace_pipelines.restype = C.c_int; L.incr_set_pathtrace_pipelines.argtypes = [vp, C.c_uint32]
    L.incr_set_pathtrace_overlap.restype = C.c_int; L.incr_set_pathtrace_overlap.argtypes = [vp, C.c_int32]
    L.incr_set_pathtrace_accumulate_placement.restype = C.c_int; L.incr_set_pathtrace_accumulate_placement.argtypes = [vp, C.c_int32]
    L.incr_profile_end.restype = C.c_int; L.incr_profile_end.argtypes = [vp, C.POINTER(A.PathtraceProfile)]
    L.incr_scene_validate_host.restype = C.c_int
    L.incr_scene_validate_host.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(A.SceneInfo)]
    # raster path
    L.incr_meshletize.restype = C.c_int
    L.incr_meshletize.argtypes = [vp, u32, vp, u32, u32, u32, u32, C.POINTER(vp)]
    L.incr_generate_tangents.restype = C.c_int; L.incr_generate_tangents.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    L.incr_decode_jpeg.restype = C.c_int; L.incr_decode_jpeg.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp, C.c_size_t]
    L.incr_meshlet_set_counts.restype = u32; L.incr_meshlet_set_counts.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.incr_meshlet_set_copy.restype = C.c_int; L.incr_meshlet_set_copy.argtypes = [vp, vp, vp, vp]
    L.incr_meshlet_set_free.restype = None; L.incr_meshlet_set_free.argtypes = [vp]
    L.incr_pixel_jitter.restype = None; L.incr_pixel_jitter.argtypes = [u32, u32, u32, C.POINTER(C.c_float)]
    L.incr_make_view_constants.restype = C.c_int
    L.incr_make_view_constants.argtypes = [C.POINTER(A.View), C.POINTER(A.View), C.POINTER(C.c_float), C.POINTER(A.Matrix), C.c_float, u32, u32,
                                           C.POINTER(A.ViewConstants)]
    for name, desc in (("incr_batch_cull", A.BatchCullDesc), ("incr_light_cull", A.LightCullDesc), ("incr_visibility", A.VisibilityDesc),
                       ("incr_gbuffer", A.GBufferDesc), ("incr_sky_lighting", A.SkyLightingDesc), ("incr_direct_lighting", A.DirectLightingDesc),
                       ("incr_clustered_lighting", A.ClusteredLightingDesc), ("incr_taa_resolve", A.TaaResolveDesc),
                       ("incr_dynamic_sky", A.DynamicSkyDesc), ("incr_hbao", A.HbaoDesc), ("incr_environmentmap", A.EnvironmentmapDesc),
                       ("incr_debug_visualization", A.DebugVizDesc), ("incr_forward_lighting", A.ForwardLightingDesc),
                       ("incr_gbuffer_decals", A.GBufferDecalsDesc), ("incr_deferred_lighting", A.DeferredLightingDesc)):
        fn = getattr(L, name)
        fn.restype = C.c_int; fn.argtypes = [vp, C.POINTER(desc), vp]
    L.incr_renderer_create.restype = C.c_int
    L.incr_renderer_create.argtypes = [vp, vp, u32, u32, C.POINTER(A.RendererScene), C.POINTER(A.RendererConfig), C.POINTER(vp)]
    L.incr_renderer_destroy.restype = None; L.incr_renderer_destroy.argtypes = [vp]
    L.incr_renderer_set_config.restype = C.c_int; L.incr_renderer_set_config.argtypes = [vp, C.POINTER(A.RendererConfig)]
    L.incr_renderer_render.restype = C.c_int; L.incr_renderer_render.argtypes = [vp, C.POINTER(A.View), C.POINTER(A.View)]
    L.incr_renderer_reset_accumulation.restype = None; L.incr_renderer_reset_accumulation.argtypes = [vp]
    L.incr_renderer_last_error.restype = C.c_char_p; L.incr_renderer_last_error.argtypes = [vp]
    L.incr_renderer_surface.restype = vp; L.incr_renderer_surface.argtypes = [vp, C.c_int]
    L.incr_renderer_set_band.restype = C.c_int; L.incr_renderer_set_band.argtypes = [vp, u32, u32]
    L.incr_renderer_tonemap.restype = C.c_int; L.incr_renderer_tonemap.argtypes = [vp]
    L.incr_renderer_get_stats.restype = C.c_int; L.incr_renderer_get_stats.argtypes = [vp, C.POINTER(C.c_int32 * 2), C.POINTER(A.RayStats)]
    L.incr_renderer_set_profiling.restype = C.c_int; L.incr_renderer_set_profiling.argtypes = [vp, C.c_int]
    L.incr_renderer_get_pass_times.restype = C.c_int; L.incr_renderer_get_pass_times.argtypes = [vp, vp, vp, u32, C.POINTER(u32)]
    L.incr_renderer_tick.restype = C.c_double; L.incr_renderer_tick.argtypes = [vp, C.c_float]
    L.incr_renderer_get_frame_info.restype = C.c_int
    L.incr_renderer_get_frame_info.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.incr_visibility_get_stats.restype = C.c_int; L.incr_visibility_get_stats.argtypes = [vp, C.POINTER(u32)]
    L.incr_update_dynamic_sky.restype = C.c_int; L.incr_update_dynamic_sky.argtypes = [vp, vp, u32, C.c_float, C.c_int32, vp, C.POINTER(vp)]
    L.incr_sky_ibl_get_info.restype = C.c_int; L.incr_sky_ibl_get_info.argtypes = [vp, C.POINTER(A.SkyIblInfo)]
    L.incr_cache_hdri.restype = C.c_int; L.incr_cache_hdri.argtypes = [vp, vp, C.c_size_t, C.c_int32, vp, C.POINTER(C.c_void_p)]
    L.incr_decode_hdr.restype = C.c_int; L.incr_decode_hdr.argtypes = [vp, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp, C.c_size_t]
    L.incr_sky_ibl_destroy.restype = C.c_int; L.incr_sky_ibl_destroy.argtypes = [vp, vp]
    L.incr_generate_env_brdf.restype = C.c_int; L.incr_generate_env_brdf.argtypes = [vp, u32, u32, C.c_int32, vp, vp]
    L.incr_compressed_mip_count.restype = u32; L.incr_compressed_mip_count.argtypes = [u32, u32]
    L.incr_mip_chain_texels.restype = u64; L.incr_mip_chain_texels.argtypes = [u32, u32, u32]
    L.incr_prefilter_normal_roughness.restype = C.c_int; L.incr_prefilter_normal_roughness.argtypes = [vp, C.POINTER(A.PrefilterNormalRoughnessDesc), vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise IncrError(rc, load().incr_last_error().decode())


class Context:
    """incr_context: one per process / GPU."""

    def __init__(self, device_index=0):
        self.lib = load()
        self.handle = C.c_void_p()
        check(self.lib.incr_initialize(device_index, C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.incr_teardown(self.handle)
            self.handle = C.c_void_p()

    # -- resource layer
    def buffer(self, nbytes):
        p = C.c_void_p()
        check(self.lib.incr_buffer_create(self.handle, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self.lib.incr_buffer_destroy(self.handle, ptr))

    def upload(self, ptr, array, stream=None):
        a = np.ascontiguousarray(array)
        check(self.lib.incr_buffer_upload(self.handle, ptr, a.ctypes.data, a.nbytes, stream))

    def download(self, ptr, shape, dtype, stream=None):
        out = np.empty(shape, dtype=dtype)
        check(self.lib.incr_buffer_download(self.handle, out.ctypes.data, ptr, out.nbytes, stream))
        return out

    def clear(self, ptr, nbytes, value32=0, stream=None):
        check(self.lib.incr_buffer_clear(self.handle, ptr, value32, nbytes, stream))

    def synchronize(self, stream=None):
        check(self.lib.incr_stream_synchronize(self.handle, stream))

    def set_pathtrace_pipelines(self, pipelines):
        check(self.lib.incr_set_pathtrace_pipelines(self.handle, pipelines))

    def set_pathtrace_overlap(self, enable):
        check(self.lib.incr_set_pathtrace_overlap(self.handle, 1 if enable else 0))

    def set_pathtrace_accumulate_placement(self, placement):
        check(self.lib.incr_set_pathtrace_accumulate_placement(self.handle, int(placement)))

    def profile_begin(self):
        check(self.lib.incr_profile_begin(self.handle))

    def profile_end(self):
        p = A.PathtraceProfile()
        check(self.lib.incr_profile_end(self.handle, C.byref(p)))
        return p

    # -- scene
    def create_scene(self, packed_scene):
        return Scene(self, packed_scene)

    # -- passes
    def pathtrace(self, desc, stream=None):
        check(self.lib.incr_pathtrace(self.handle, C.byref(desc), stream))

    def tonemap(self, desc, stream=None):
        check(self.lib.incr_tonemap(self.handle, C.byref(desc), stream))

    def unshard(self, dst, src, res_x, res_y, row_block_height, num_shards, texel_bytes, stream=None):
        check(self.lib.incr_unshard(self.handle, dst, src, res_x, res_y, row_block_height, num_shards, texel_bytes, stream))


class Scene:
    def __init__(self, ctx, packed_scene):
        self.ctx = ctx
        self.packed = packed_scene
        desc = packed_scene.to_desc()
        self.handle = C.c_void_p()
        check(ctx.lib.incr_scene_create(ctx.handle, C.byref(desc), C.byref(self.handle)))

    def info(self):
        i = A.SceneInfo()
        check(self.ctx.lib.incr_scene_get_info(self.ctx.handle, self.handle, C.byref(i)))
        return i

    def trace_rays(self, origins, dirs, tmin, tmax, any_hit=False):
        """incr_trace_rays on host arrays (uploads, launches, downloads): returns (ids [n,4] u32, tuv [n,3] f32)."""
        ctx = self.ctx
        o = np.ascontiguousarray(origins, dtype=np.float32); d = np.ascontiguousarray(dirs, dtype=np.float32)
        n = o.shape[0]
        bufs = [ctx.buffer(max(n, 1) * 12), ctx.buffer(max(n, 1) * 12), ctx.buffer(max(n, 1) * 16), ctx.buffer(max(n, 1) * 12)]
        try:
            ctx.upload(bufs[0], o); ctx.upload(bufs[1], d)
            check(ctx.lib.incr_trace_rays(ctx.handle, self.handle, n, bufs[0], bufs[1], tmin, tmax, int(any_hit), bufs[2], bufs[3], None))
            ctx.synchronize()
            return ctx.download(bufs[2], (n, 4), np.uint32), ctx.download(bufs[3], (n, 3), np.float32)
        finally:
            for b in bufs:
                ctx.free(b)

    def texture_chain(self, index):
        """(width, height, num_mips, uint32 texels of all levels) of scene texture `index` as the raster passes sample it."""
        w, h, m, n = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(self.ctx.lib.incr_scene_texture_info(self.ctx.handle, self.handle, index, C.byref(w), C.byref(h), C.byref(m), C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        check(self.ctx.lib.incr_scene_texture_download(self.ctx.handle, self.handle, index, out.ctypes.data, out.nbytes))
        return w.value, h.value, m.value, out

    def update_bvh(self, mesh_transforms=None, mesh_visibility=None, root_transform=None, stream=None):
        """incr_update_bvh (UpdateTLASPass): new per-mesh transforms (list of 4x4), visibility flags and / or root transform."""
        from . import scene_io as sio
        mt = (A.Matrix * len(mesh_transforms))(*[sio.to_matrix(m) for m in mesh_transforms]) if mesh_transforms is not None else None
        mv = (C.c_uint32 * len(mesh_visibility))(*[int(v) for v in mesh_visibility]) if mesh_visibility is not None else None
        rtm = sio.to_matrix(root_transform) if root_transform is not None else None
        check(self.ctx.lib.incr_update_bvh(self.ctx.handle, self.handle, mt, mv, C.byref(rtm) if rtm is not None else None, stream))

    def close(self):
        if self.handle:
            self.ctx.lib.incr_scene_destroy(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()


def make_pathtrace_desc(view, res_x, res_y, sample_index=1, shard=None, cfg=A.FConfig, storage_rounding=0, envmap=None, num_samples=1):
    """Fill an incr_pathtrace_desc from FConfig defaults (inc/common.h:6-48) and an FView.
    envmap = (pointer, resolution, levels) of the prefiltered sky cube (ibl.SkyIbl) or None for the analytic stand-in."""
    d = A.PathtraceDesc()
    if envmap is not None:
        d.envmap_r11g11b10, d.envmap_res, d.envmap_mips = envmap
    d.res_x, d.res_y = res_x, res_y
    if shard is not None:
        d.shard = shard
    d.current_sample_index = sample_index
    d.max_sample_count = cfg.MaxSampleCount
    d.view = view.to_ctypes()
    d.camera_aperture = cfg.Pathtracing_CameraAperture
    d.camera_focal_length = cfg.Pathtracing_CameraFocalLength
    d.env_sky_mode = cfg.EnvSkyMode
    d.direct_lighting = 1 if cfg.EnableDirectLighting else 0
    d.viewmode = cfg.Viewmode
    d.turbidity = cfg.Turbidity
    d.sky_brightness = cfg.SkyBrightness
    d.storage_rounding = storage_rounding
    d.num_samples = num_samples          # samples sample_index .. sample_index + num_samples - 1 as one wavefront (incr.h)
    return d


class PathTracer:
    """Convenience driver around incr_pathtrace with library-owned buffers (tests / smoke)."""

    def __init__(self, ctx, scene, res_x, res_y, shard=None):
        self.ctx, self.scene, self.res_x, self.res_y = ctx, scene, res_x, res_y
        self.shard = shard if shard is not None else A.Shard(0, 0, 1)
        self.rows = ctx.lib.incr_shard_local_rows(res_y, C.byref(self.shard))
        self.target = ctx.buffer(self.rows * res_x * 8)
        self.history = ctx.buffer(self.rows * res_x * 4)
        self.stats = ctx.buffer(C.sizeof(A.RayStats))
        ctx.clear(self.history, self.rows * res_x * 4)
        ctx.clear(self.stats, C.sizeof(A.RayStats))
        self.chain = None

    NOT_TRACED = 0xfffffffe      # hit-chain entries of path vertices that were never traced

    def render(self, view, sample_index=1, cfg=A.FConfig, storage_rounding=0, stream=None, count_rays=True, envmap=None, num_samples=1, hit_chain=False):
        d = make_pathtrace_desc(view, self.res_x, self.res_y, sample_index, self.shard, cfg, storage_rounding, envmap, num_samples)
        d.scene = self.scene.handle
        d.target_rgba16f, d.history_r11g11b10 = self.target, self.history
        d.ray_stats = self.stats if count_rays else None
        if hit_chain:       # incr_pathtrace_desc::debug_hit_chain (test instrumentation): read back with read_chain()
            if self.chain is None:
                self.chain = self.ctx.buffer(self.rows * self.res_x * 48)
            self.ctx.clear(self.chain, self.rows * self.res_x * 48, self.NOT_TRACED, stream)
            d.debug_hit_chain = self.chain
        self.ctx.pathtrace(d, stream)
        return d

    def read_chain(self):
        self.ctx.synchronize()
        return self.ctx.download(self.chain, (self.rows, self.res_x, 4, 3), np.uint32)

    def read(self):
        self.ctx.synchronize()
        target = self.ctx.download(self.target, (self.rows, self.res_x, 4), np.uint16)
        history = self.ctx.download(self.history, (self.rows, self.res_x), np.uint32)
        raw = self.ctx.download(self.stats, (C.sizeof(A.RayStats),), np.uint8)
        stats = A.RayStats.from_buffer_copy(raw.tobytes())
        return dict(target=target, history=history, stats=stats)

    def close(self):
        for p in (self.target, self.history, self.stats) + ((self.chain,) if self.chain is not None else ()):
            self.ctx.free(p)
