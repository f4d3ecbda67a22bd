"""PSNR of a rendered HDR frame against a capture of the reference (SURVEY.md 8c "external ground truth", 8d "Image quality"):
both float32 [H, W, 3] arrays (.npy, uncompressed OpenEXR, or raw little-endian float32 with --res WxH) in linear radiance -- ours from
`examples/render.py --dump-hdr`, the capture from a dump of the reference's pathtraceHistoryBuffer / TAA accumulation surface.
Prints the PSNR on linear HDR clipped to [0, 65504] and on the tonemapped 8-bit image (Exposure(ev100) . ACES, tonemap.hlsl:30-59).

    python tools/psnr_capture.py ours.npy capture.npy [--exposure 13] [--res 1920x1080]

No such capture ships with the reference tree; until one is supplied the PSNR figures of this repo are HIP vs CPU oracle."""
import argparse
import numpy as np


def load(path, res):
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    if path.lower().endswith(".exr"):          # uncompressed scan-line OpenEXR with FLOAT R, G, B (examples/render.py --dump-hdr x.exr writes it)
        import os, sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
        import render
        return render.read_exr(path)
    w, h = (int(v) for v in res.lower().split("x"))
    return np.fromfile(path, dtype="<f4").reshape(h, w, -1)[..., :3]


def psnr(a, b, peak):
    fin = np.isfinite(a) & np.isfinite(b)
    mse = float(np.mean((np.clip(a[fin], 0, peak) - np.clip(b[fin], 0, peak)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def tonemap(hdr, exposure):
    x = hdr / np.float32(2.0 ** int(exposure) * 1.2)                     # pbr.hlsli:78-81 Exposure (ev100 truncated to int)
    ldr = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)       # ACESFilm, pbr.hlsli:84-92
    srgb = np.where(ldr <= 0.0031308, 12.92 * ldr, 1.055 * np.power(ldr, 1 / 2.4) - 0.055)
    return np.floor(srgb * 255 + 0.5)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ours"); ap.add_argument("capture"); ap.add_argument("--exposure", type=float, default=13.0); ap.add_argument("--res", default="1920x1080")
    a = ap.parse_args()
    x, y = load(a.ours, a.res), load(a.capture, a.res)
    assert x.shape == y.shape, (x.shape, y.shape)
    print("PSNR linear HDR (peak 65504): %.2f dB" % psnr(x, y, 65504.0))
    for c, n in enumerate("RGB"):
        print("   %s: %.2f dB" % (n, psnr(x[..., c], y[..., c], 65504.0)))
    print("PSNR tonemapped 8-bit: %.2f dB" % psnr(tonemap(x, a.exposure), tonemap(y, a.exposure), 255.0))


if __name__ == "__main__":
    main()
