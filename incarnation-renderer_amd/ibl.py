"""Host wrapper of the sky / IBL precompute (include/incr.h "Sky and IBL precompute"): FScene::UpdateDynamicSky and
Renderer::GenerateEnvBrdfTexture of the reference (src/scene.cpp:1657-1790, src/renderer.cpp:169-245)."""
import ctypes as C

import numpy as np

from . import ctypes_api as A
from . import runtime as rt


def mip_count(w, h):
    n = 0
    while w >= 1 and h >= 1:
        n += 1; w >>= 1; h >>= 1
    return n


def level_shapes(res):
    """(lat-long level shapes, cube level sizes) for EnvmapResolution = res."""
    mips = mip_count(res, res)
    return [(max(res >> m, 1), max(2 * res >> m, 1)) for m in range(mips)], [max(res >> m, 1) for m in range(mips)]


class SkyIbl:
    """FScene::m_dynamicSkyEnvmap + m_dynamicSkySH and the intermediates, resident in HBM."""

    def __init__(self, ctx, scene, resolution=A.FConfig.EnvmapResolution, turbidity=A.FConfig.Turbidity, storage_rounding=0, stream=None):
        self.ctx, self.scene = ctx, scene
        self.resolution, self.turbidity, self.storage_rounding = resolution, turbidity, storage_rounding
        self.handle = C.c_void_p()
        self.update(stream)

    def update(self, stream=None):
        """Call after the sun moved (incr_scene_set_sun_dir): re-runs the whole chain into the same buffers."""
        rt.check(self.ctx.lib.incr_update_dynamic_sky(self.ctx.handle, self.scene.handle, self.resolution, self.turbidity, self.storage_rounding,
                                                      stream, C.byref(self.handle)))
        self.info = A.SkyIblInfo()
        rt.check(self.ctx.lib.incr_sky_ibl_get_info(self.handle, C.byref(self.info)))
        return self

    def download(self):
        i, ctx = self.info, self.ctx
        ctx.synchronize()
        return dict(latlong=ctx.download(i.latlong_r11g11b10, (i.latlong_texels,), np.uint32),
                    source_cube=ctx.download(i.source_cube_r11g11b10, (i.cube_texels,), np.uint32),
                    envmap=ctx.download(i.envmap_r11g11b10, (i.cube_texels,), np.uint32),
                    sh9=ctx.download(i.sh9_rgba, (9, 4), np.float32))

    def close(self):
        if self.handle:
            self.ctx.lib.incr_sky_ibl_destroy(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()


def decode_hdr(data):
    """Radiance .hdr bytes -> (H, W, 4) float32 through incr_decode_hdr (the reader of incr_cache_hdri; DirectX::LoadFromHDRFile in the reference, demo.cpp:313)."""
    L = rt.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_int32(0), C.c_int32(0)
    rt.check(L.incr_decode_hdr(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), np.float32)
    rt.check(L.incr_decode_hdr(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), out.ctypes.data, out.size))
    return out


class HdriProbe:
    """FTextureCache::CacheHDRI (demo.cpp:292-434): the light probe of EnvSkyMode::HDRI from the bytes of a Radiance .hdr file -- prefiltered cube + SH9, resident in HBM."""

    def __init__(self, ctx, hdr_bytes, storage_rounding=0, stream=None):
        self.ctx = ctx
        self.handle = C.c_void_p()
        buf = np.frombuffer(hdr_bytes, dtype=np.uint8)
        rt.check(ctx.lib.incr_cache_hdri(ctx.handle, buf.ctypes.data, buf.size, storage_rounding, stream, C.byref(self.handle)))
        self.info = A.SkyIblInfo()
        rt.check(ctx.lib.incr_sky_ibl_get_info(self.handle, C.byref(self.info)))

    def download(self):
        i, ctx = self.info, self.ctx
        ctx.synchronize()
        src_texels = sum(6 * (i.source_cube_res >> m) ** 2 for m in range(i.source_cube_mips))
        return dict(latlong=ctx.download(i.latlong_rgba32f, (i.latlong_texels, 4), np.float32),
                    source_cube=ctx.download(i.source_cube_r11g11b10, (src_texels,), np.uint32),
                    envmap=ctx.download(i.envmap_r11g11b10, (i.cube_texels,), np.uint32),
                    sh9=ctx.download(i.sh9_rgba, (9, 4), np.float32))

    def close(self):
        if self.handle:
            self.ctx.lib.incr_sky_ibl_destroy(self.ctx.handle, self.handle)
            self.handle = C.c_void_p()


def generate_env_brdf(ctx, width=512, height=512, storage_rounding=0, stream=None):
    """Renderer::GenerateEnvBrdfTexture: returns the device pointer of a width*height R16G16_FLOAT LUT (caller frees)."""
    buf = ctx.buffer(width * height * 4)
    rt.check(ctx.lib.incr_generate_env_brdf(ctx.handle, width, height, storage_rounding, buf, stream))
    return buf
