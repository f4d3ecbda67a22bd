"""Render a frame with the HIP library and write it as a PNG -- the headless stand-in for the reference's window
(exe/main.cpp presents the back buffer; this writes it).

    python examples/render.py --scene helmet --mode pathtrace --spp 64 --res 1024x1024 --out helmet.png
    python examples/render.py --scene sponza --mode raster --frames 16 --res 1920x1080 --hbao --out sponza.png

    python examples/render.py --scene helmet --mode pathtrace --spp 256 --res 1920x1080 --dump-hdr helmet_hdr.npy

--dump-hdr writes the frame's HDR surface as float32 [H, W, 3] (.npy): the R11G11B10_FLOAT path-trace history / TAA accumulation
decoded exactly (what the tonemap reads).  That is the array to hold against a capture of the reference's own history buffer
(tools/psnr_capture.py: PSNR on linear HDR clipped to [0, 65504] and on the tonemapped 8-bit image).

Needs an MI355X (there is no CPU fallback); the PNG encoder itself is plain Python (zlib)."""
import argparse
import importlib
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "incarnation-renderer_amd"


def write_png(path, rgba8):
    """rgba8: (H, W, 4) uint8, sRGB-encoded (the back buffer's R8G8B8A8_UNORM_SRGB bytes).  8-bit RGBA, no interlace."""
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = img.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 4)], axis=1).tobytes()      # filter type 0 per scanline

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path):
    """Inverse of write_png for its own files (filter 0, RGBA8): used by the tests."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8]); body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4)


def write_exr(path, rgb):
    """rgb: (H, W, 3) float32 -> OpenEXR 2.0 scan-line file, uncompressed, three FLOAT channels (the format a DX12 capture of the
    reference's HDR target would be exchanged in: `tools/psnr_capture.py` reads both this and .npy)."""
    img = np.ascontiguousarray(rgb, dtype="<f4")
    h, w = img.shape[:2]

    def attr(name, kind, data):
        return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chans = b"".join(n + b"\0" + struct.pack("<iBxxxii", 2, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"      # FLOAT, pLinear 0, sampling 1 x 1
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (attr("channels", "chlist", chans) + attr("compression", "compression", b"\0") + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
              + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    line_bytes = 3 * w * 4
    first = 8 + len(header) + 8 * h
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 20000630, 2) + header)
        f.write(struct.pack("<%dQ" % h, *[first + y * (8 + line_bytes) for y in range(h)]))
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes) + img[y, :, 2].tobytes() + img[y, :, 1].tobytes() + img[y, :, 0].tobytes())      # channels in alphabetical order


def read_exr(path):
    """Inverse of write_exr for uncompressed scan-line files with FLOAT B, G, R channels."""
    raw = open(path, "rb").read()
    assert struct.unpack_from("<II", raw, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while raw[pos] != 0:
        e = raw.index(b"\0", pos); name = raw[pos:e].decode(); pos = e + 1
        e = raw.index(b"\0", pos); pos = e + 1
        n = struct.unpack_from("<i", raw, pos)[0]; attrs[name] = raw[pos + 4:pos + 4 + n]; pos += 4 + n
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"] == b"\0" and attrs["channels"].startswith(b"B\0")
    offsets = struct.unpack_from("<%dQ" % h, raw, pos)
    out = np.zeros((h, w, 3), np.float32)
    for o in offsets:
        y, n = struct.unpack_from("<ii", raw, o)
        row = np.frombuffer(raw, "<f4", count=3 * w, offset=o + 8).reshape(3, w)
        out[y - y0] = row[::-1].T
    return out


def decode_r11g11b10(words):
    """R11G11B10_FLOAT -> float32 [..., 3]: unsigned, 5-bit exponents (bias 15), 6 / 6 / 5 mantissa bits, denormals, inf / NaN."""
    w = np.asarray(words, dtype=np.uint32)

    def part(bits, mant_bits):
        e = (bits >> mant_bits).astype(np.int32); m = (bits & ((1 << mant_bits) - 1)).astype(np.float32)
        scale = np.float32(1 << mant_bits)
        normal = np.ldexp(np.float32(1.0) + m / scale, e - 15).astype(np.float32)
        denorm = np.ldexp(m / scale, -14).astype(np.float32)
        special = np.where(m == 0, np.float32(np.inf), np.float32(np.nan))
        return np.where(e == 0, denorm, np.where(e == 31, special, normal)).astype(np.float32)
    return np.stack([part(w & 0x7ff, 6), part((w >> 11) & 0x7ff, 6), part(w >> 22, 5)], axis=-1)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="helmet", help="helmet | sponza | bistro | path to a .gltf")
    ap.add_argument("--mode", default="pathtrace", choices=("pathtrace", "raster"))
    ap.add_argument("--res", default="1280x720"); ap.add_argument("--spp", type=int, default=16, help="path trace: samples accumulated")
    ap.add_argument("--frames", type=int, default=8, help="raster: frames rendered (TAA converges over the jitter sequence)")
    ap.add_argument("--hbao", action="store_true"); ap.add_argument("--forward", action="store_true"); ap.add_argument("--viewmode", type=int, default=0)
    ap.add_argument("--texture-size", type=int, default=512); ap.add_argument("--out", default="frame.png")
    ap.add_argument("--max-anisotropy", type=int, default=0, help="raster: MaxAnisotropy of the material sampler (0 = the reference's 16, 1 = isotropic)")
    ap.add_argument("--block-compress", action="store_true", help="hand the textures over as block-compressed chains (bc.py's encoder)")
    ap.add_argument("--dump-hdr", default=None, help="also write the HDR surface the tonemap read as float32 [H, W, 3]: .npy, or OpenEXR when the name ends in .exr")
    a = ap.parse_args()
    A = importlib.import_module(PKG + ".ctypes_api"); sio = importlib.import_module(PKG + ".scene_io"); synth = importlib.import_module(PKG + ".synth")
    rt = importlib.import_module(PKG + ".runtime"); R = importlib.import_module(PKG + ".raster"); ibl = importlib.import_module(PKG + ".ibl")
    W, H = (int(v) for v in a.res.lower().split("x"))
    if a.scene == "helmet":
        packed = sio.load_gltf(os.path.join(ROOT, "assets/damaged-helmet/DamagedHelmet.gltf"))
    elif a.scene == "sponza":
        packed = synth.add_decals(synth.sponza_stat(texture_size=a.texture_size), texture_size=a.texture_size)
    elif a.scene == "bistro":
        packed = synth.bistro_scale(texture_size=min(a.texture_size, 256))
    else:
        packed = sio.load_gltf(a.scene)
    packed.max_anisotropy = a.max_anisotropy
    if a.block_compress:
        packed.compress_textures()
    packed = R.generate_meshlets(packed)

    class Cfg(A.FConfig):
        EnableHBAO = a.hbao
        UseBentNormals = a.hbao
        ForwardLighting = a.forward
        Viewmode = a.viewmode
        MaxSampleCount = max(A.FConfig.MaxSampleCount, a.spp + 1)
    ctx = rt.Context(0)
    scene = ctx.create_scene(packed)
    view = sio.FView().reset(packed, aspect=W / H)
    sky = ibl.SkyIbl(ctx, scene)
    lut = ibl.generate_env_brdf(ctx, 512, 512)
    hr = R.HostRenderer(ctx, scene, W, H, cfg=Cfg, path_trace=(a.mode == "pathtrace"), ibl=R.IblInputs.from_sky(sky, lut, (512, 512)))
    for _ in range(a.spp if a.mode == "pathtrace" else a.frames):
        hr.render(view)
    (culled_objects, culled_lights), rays = hr.stats()
    ldr = hr.download(A.SURFACE_BACK_BUFFER).view(np.uint8).reshape(H, W, 4).copy()
    ldr[..., 3] = 255
    write_png(a.out, ldr)
    if a.dump_hdr:
        surf = A.SURFACE_PATHTRACE_HISTORY if a.mode == "pathtrace" else (A.SURFACE_TAA_ACCUMULATION if Cfg.EnableTAA else A.SURFACE_HDR_RASTER)
        hdr = decode_r11g11b10(hr.download(surf).view(np.uint32).reshape(-1)[: W * H].reshape(H, W))
        if a.dump_hdr.lower().endswith(".exr"):
            write_exr(a.dump_hdr, hdr)
        else:
            np.save(a.dump_hdr, hdr)
    print("%s: %dx%d %s, %d rays traced, %d objects / %d lights culled in the last frame" % (a.out, W, H, a.mode, rays.total_rays(), culled_objects, culled_lights))
    hr.close(); sky.close(); ctx.free(lut); scene.close()


if __name__ == "__main__":
    main()
