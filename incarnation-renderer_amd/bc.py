"""Block-compressed texture chains at the boundary (include/incr_types.h formats 4-7).

The reference compresses every texture with DirectXTex (src/scene.cpp:1090-1100, 1330-1420: BC3_UNORM_SRGB base colour / emissive,
BC4_UNORM AO / transmission / clearcoat, BC5_UNORM metallic-roughness, BC5_SNORM normals) and keeps the result as .dds files in its
content cache.  The library DECODES such chains (csrc/texture_mips.hip bc_decode_kernel; the decode is fixed by the format).  This
module holds what a test or a tool needs around that: a container for a compressed chain, a small deterministic ENCODER (range fit:
not DirectXTex's, any valid block stream will do for parity between the library and the oracle), and a reader / writer for the
DX10-header .dds files the reference's cache uses."""
import struct

import numpy as np

from . import ctypes_api as A

BLOCK_BYTES = {A.TEX_BC3_SRGB: 16, A.TEX_BC4_UNORM: 8, A.TEX_BC5_UNORM: 16, A.TEX_BC5_SNORM: 16}
DXGI = {A.TEX_BC3_SRGB: 78, A.TEX_BC4_UNORM: 80, A.TEX_BC5_UNORM: 83, A.TEX_BC5_SNORM: 84}        # DXGI_FORMAT_BC3_UNORM_SRGB, BC4_UNORM, BC5_UNORM, BC5_SNORM


class BcImage:
    """A compressed mip chain: `blocks` = the levels' blocks back to back (level l: ceil(w_l / 4) x ceil(h_l / 4), row-major)."""

    def __init__(self, blocks, width, height, mip_levels, fmt):
        self.blocks = np.ascontiguousarray(blocks, dtype=np.uint8).ravel()
        self.width, self.height, self.mip_levels, self.format = int(width), int(height), int(mip_levels), int(fmt)
        assert self.blocks.size == chain_bytes(width, height, mip_levels, fmt), (self.blocks.size, chain_bytes(width, height, mip_levels, fmt))

    @property
    def shape(self):                      # (H, W, 4): lets code that only asks an image for its size treat both kinds alike
        return (self.height, self.width, 4)


def level_dim(d, l):
    return max(d >> l, 1)


def chain_bytes(w, h, mips, fmt):
    return sum(((level_dim(w, l) + 3) // 4) * ((level_dim(h, l) + 3) // 4) * BLOCK_BYTES[fmt] for l in range(mips))


def _blocks_of(img):
    """(H, W, C) -> (by, bx, 16, C): 4x4 blocks, texels row-major inside a block; edges repeat the last row / column."""
    h, w = img.shape[:2]
    ph, pw = (h + 3) // 4 * 4, (w + 3) // 4 * 4
    if (ph, pw) != (h, w):
        img = np.pad(img, ((0, ph - h), (0, pw - w), (0, 0)), mode="edge")
    c = img.shape[2]
    return img.reshape(ph // 4, 4, pw // 4, 4, c).transpose(0, 2, 1, 3, 4).reshape(ph // 4, pw // 4, 16, c)


def _pack_indices(idx, bits):
    """(..., 16) small ints -> little-endian bit string of 16 * bits bits, as bytes (..., 2 * bits)."""
    v = np.zeros(idx.shape[:-1], dtype=np.uint64)
    for i in range(16):
        v |= idx[..., i].astype(np.uint64) << np.uint64(bits * i)
    return np.stack([((v >> np.uint64(8 * k)) & np.uint64(0xff)).astype(np.uint8) for k in range(2 * bits)], axis=-1)


def _encode_bc4_channel(vals, signed):
    """vals: (by, bx, 16) int (0..255, or -127..127 when signed) -> (by, bx, 8) bytes.  Endpoints = the block's max and min (eight-value
    mode when they differ), every texel takes the nearest palette entry."""
    v = vals.astype(np.int32)
    e0, e1 = v.max(axis=-1), v.min(axis=-1)                                   # e0 >= e1; equal -> six-value mode, all indices 0
    k = np.arange(1, 7, dtype=np.int32)
    pal8 = np.concatenate([e0[..., None], e1[..., None], ((7 - k) * e0[..., None] + k * e1[..., None])], axis=-1).astype(np.float64)
    pal8[..., 2:] /= 7.0
    idx = np.abs(v[..., :, None] - pal8[..., None, :]).argmin(axis=-1)
    idx = np.where((e0 == e1)[..., None], 0, idx)
    ends = np.stack([e0, e1], axis=-1)
    ends = (ends & 0xff).astype(np.uint8) if signed else ends.astype(np.uint8)
    return np.concatenate([ends, _pack_indices(idx, 3)], axis=-1)


def _encode_bc1_colour(rgb):
    """rgb: (by, bx, 16, 3) uint8 -> (by, bx, 8) bytes, four-colour mode (what BC3's colour block always is): endpoints = the corners of
    the block's bounding box in 5:6:5, indices by nearest palette colour."""
    v = rgb.astype(np.int32)
    hi, lo = v.max(axis=2), v.min(axis=2)

    def to565(c):
        r, g, b = (c[..., 0] * 31 + 127) // 255, (c[..., 1] * 63 + 127) // 255, (c[..., 2] * 31 + 127) // 255
        return (r << 11) | (g << 5) | b

    def from565(p):
        r, g, b = (p >> 11) & 31, (p >> 5) & 63, p & 31
        return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], axis=-1)
    c0, c1 = to565(hi), to565(lo)
    swap = c0 < c1
    c0, c1 = np.where(swap, c1, c0), np.where(swap, c0, c1)
    a, b = from565(c0), from565(c1)
    pal = np.stack([a, b, (2 * a + b + 1) // 3, (a + 2 * b + 1) // 3], axis=-2)                    # (by, bx, 4, 3)
    d = ((v[..., :, None, :] - pal[..., None, :, :]) ** 2).sum(axis=-1)
    idx = d.argmin(axis=-1)
    ends = np.stack([c0 & 0xff, c0 >> 8, c1 & 0xff, c1 >> 8], axis=-1).astype(np.uint8)
    return np.concatenate([ends, _pack_indices(idx, 2)], axis=-1)


def encode_level(img, fmt):
    """One level.  img: (H, W, 4) uint8 RGBA as the uncompressed formats carry it (BC5_UNORM takes its R, G from the B, G bytes -- the
    reference's B8G8R8A8 swizzle of the metallic-roughness image; BC5_SNORM maps c -> round((2 c / 255 - 1) * 127) as DirectXTex's
    UNORM -> SNORM conversion does; BC4 takes R)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    b = _blocks_of(img)
    if fmt == A.TEX_BC3_SRGB:
        out = np.concatenate([_encode_bc4_channel(b[..., 3], False), _encode_bc1_colour(b[..., :3])], axis=-1)
    elif fmt == A.TEX_BC4_UNORM:
        out = _encode_bc4_channel(b[..., 0], False)
    elif fmt == A.TEX_BC5_UNORM:
        out = np.concatenate([_encode_bc4_channel(b[..., 2], False), _encode_bc4_channel(b[..., 1], False)], axis=-1)
    elif fmt == A.TEX_BC5_SNORM:
        s = np.rint((2.0 * b[..., :2].astype(np.float64) / 255.0 - 1.0) * 127.0).astype(np.int32)
        out = np.concatenate([_encode_bc4_channel(s[..., 0], True), _encode_bc4_channel(s[..., 1], True)], axis=-1)
    else:
        raise ValueError("not a block-compressed format: %r" % fmt)
    return out.reshape(-1)


def box_mips(img, levels=None):
    """RGBA8 levels by 2x2 means on the bytes (a plain chain for test data; the reference builds its own before compressing)."""
    out = [np.ascontiguousarray(img, dtype=np.uint8)]
    while (levels is None and min(out[-1].shape[:2]) >= 8) or (levels is not None and len(out) < levels):
        s = out[-1].astype(np.uint32)
        h, w = max(s.shape[0] // 2, 1), max(s.shape[1] // 2, 1)
        ys, xs = np.minimum(np.arange(h) * 2 + 1, s.shape[0] - 1), np.minimum(np.arange(w) * 2 + 1, s.shape[1] - 1)
        y0, x0 = np.minimum(np.arange(h) * 2, s.shape[0] - 1), np.minimum(np.arange(w) * 2, s.shape[1] - 1)
        m = (s[y0][:, x0] + s[y0][:, xs] + s[ys][:, x0] + s[ys][:, xs] + 2) // 4
        out.append(m.astype(np.uint8))
    return out


def compress(img, fmt, levels=None):
    """RGBA8 image -> BcImage with a box-filtered chain (levels=None: down to 4x4 as the reference's chains, scene.cpp:1072-1081)."""
    chain = box_mips(img, levels)
    return BcImage(np.concatenate([encode_level(l, fmt) for l in chain]), img.shape[1], img.shape[0], len(chain), fmt)


# ---- .dds with the DX10 extension header (what DirectX::SaveToDDSFile writes for these formats; scene.cpp:1103-1115) ----
_DDS_MAGIC = 0x20534444


def write_dds(path, image):
    pitch = ((image.width + 3) // 4) * BLOCK_BYTES[image.format]
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | 0x20000 | 0x80000          # CAPS | HEIGHT | WIDTH | PIXELFORMAT | MIPMAPCOUNT | LINEARSIZE
    hdr = struct.pack("<7I44x", 124, flags, image.height, image.width, pitch * ((image.height + 3) // 4), 0, image.mip_levels)
    pf = struct.pack("<2I4s5I", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    caps = struct.pack("<4I4x", 0x1000 | (0x400008 if image.mip_levels > 1 else 0), 0, 0, 0)
    dx10 = struct.pack("<5I", DXGI[image.format], 3, 0, 1, 0)        # TEXTURE2D, array size 1
    with open(path, "wb") as f:
        f.write(struct.pack("<I", _DDS_MAGIC) + hdr + pf + caps + dx10 + image.blocks.tobytes())


def read_dds(path):
    raw = open(path, "rb").read()
    magic, size, _flags, height, width, _pitch, _depth, mips = struct.unpack_from("<8I", raw, 0)
    if magic != _DDS_MAGIC or size != 124 or raw[84:88] != b"DX10":
        raise ValueError("%s: not a DX10-header .dds" % path)
    dxgi = struct.unpack_from("<I", raw, 128)[0]
    fmt = {v: k for k, v in DXGI.items()}.get(dxgi)
    if fmt is None:
        raise ValueError("%s: DXGI format %d is not one of BC3_UNORM_SRGB / BC4_UNORM / BC5_UNORM / BC5_SNORM" % (path, dxgi))
    mips = max(mips, 1)
    n = chain_bytes(width, height, mips, fmt)
    return BcImage(np.frombuffer(raw, dtype=np.uint8, count=n, offset=148), width, height, mips, fmt)
