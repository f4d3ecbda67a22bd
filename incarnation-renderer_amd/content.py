"""Texture ingest, GPU step: vMF prefilter of normal + metallic-roughness maps (include/incr.h "Texture ingest";
reference FScene::PrefilterNormalRoughnessTextures, src/scene.cpp:1157-1320)."""
import ctypes as C

import numpy as np

from . import ctypes_api as A
from . import runtime as rt


def mip_count(width, height):
    return int(rt.load().incr_compressed_mip_count(width, height))


def split_levels(chain, width, height, mips):
    """Packed level chain (texels x 4 u8) -> list of (h_k, w_k, 4) arrays."""
    out, off = [], 0
    for k in range(mips):
        w, h = width >> k, height >> k
        out.append(chain[off:off + w * h].reshape(h, w, 4))
        off += w * h
    return out


def prefilter_normal_roughness(ctx, normalmap, metallic_roughness, stream=None):
    """normalmap, metallic_roughness: (H, W, 4) uint8 as decoded from the images.  Returns (normal_levels, metal_rough_levels),
    each a list of (H>>k, W>>k, 4) uint8 arrays; level 0 of the second is the .bgra swizzle of the input."""
    n = np.ascontiguousarray(normalmap, np.uint8); m = np.ascontiguousarray(metallic_roughness, np.uint8)
    if n.shape != m.shape or n.ndim != 3 or n.shape[2] != 4:
        raise ValueError("both maps must be (H, W, 4) uint8 of the same size (scene.cpp:1282)")
    H, W = n.shape[:2]
    lib = rt.load()
    mips = mip_count(W, H)
    if mips == 0:
        raise ValueError("texture %dx%d is below 4x4" % (W, H))
    texels = int(lib.incr_mip_chain_texels(W, H, mips))
    bufs = [ctx.buffer(W * H * 4), ctx.buffer(W * H * 4), ctx.buffer(texels * 4), ctx.buffer(texels * 4)]
    try:
        ctx.upload(bufs[0], n, stream); ctx.upload(bufs[1], m, stream)
        d = A.PrefilterNormalRoughnessDesc(normalmap=bufs[0], metallic_roughness=bufs[1], width=W, height=H, num_mips=mips,
                                           out_normalmap=bufs[2], out_metallic_roughness=bufs[3])
        rt.check(lib.incr_prefilter_normal_roughness(ctx.handle, C.byref(d), stream))
        on = ctx.download(bufs[2], (texels, 4), np.uint8, stream); om = ctx.download(bufs[3], (texels, 4), np.uint8, stream)
    finally:
        for b in bufs:
            ctx.free(b)
    return split_levels(on, W, H, mips), split_levels(om, W, H, mips)
