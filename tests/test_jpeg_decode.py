"""incr_decode_jpeg (csrc/jpeg_decode.cpp): stb_image's published integer JPEG pipeline -- what the reference's loader runs (tinygltf::LoadImageData,
src/scene.cpp:40, 98-117).  PARITY UNPINNED: ext/tinygltf is an empty submodule in the reference checkout, so no byte of stb_image's own output exists here.
What pins the decoder instead:
  * a hand-derived known answer (an 8 x 8 DC-only block) and a 16 x 16 4:2:0 frame whose expected bytes are derived in this file, step by step, from the
    published formulas (12-bit islow IDCT, 3:1 + 3:1 chroma filter, 20-bit YCbCr) -- an independent restatement in Python, sharing no code with the C++;
  * synthetic files for every sampling layout (4:4:4, 4:2:2, 4:4:0, 4:2:0), ragged sizes and restart intervals against that restatement, bit for bit;
  * the shipped assets (baseline and progressive) against libjpeg-turbo (PIL): never more than 3 codes apart -- the two decoders differ only in rounding."""
import glob
import io
import os
import struct

import numpy as np
import pytest

from conftest import ROOT

DEZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


# ------------------------------------------------------------------ a minimal baseline JPEG writer for chosen quantised coefficients
class Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, value, nbits):
        for k in range(nbits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> k) & 1); self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xff:
                    self.out.append(0)
                self.acc = 0; self.n = 0

    def flush(self):
        while self.n:
            self.put(1, 1)


DC_SYMS = list(range(12))                                               # 12 codes of 4 bits: code = category
AC_SYMS = [0x00, 0xf0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]      # 162 codes of 8 bits: code = position in this list


def category(v):
    return 0 if v == 0 else int(abs(v)).bit_length()


def put_value(bw, v, s):
    bw.put(v if v >= 0 else v + (1 << s) - 1, s)


def encode_block(bw, coeffs_natural, pred):
    zz = [int(coeffs_natural[DEZIGZAG[k]]) for k in range(64)]
    diff = zz[0] - pred
    s = category(diff); bw.put(s, 4)
    if s:
        put_value(bw, diff, s)
    run = 0
    last = max([k for k in range(1, 64) if zz[k]], default=0)
    for k in range(1, last + 1):
        if zz[k] == 0:
            run += 1
            continue
        while run > 15:
            bw.put(AC_SYMS.index(0xf0), 8); run -= 16
        s = category(zz[k]); bw.put(AC_SYMS.index((run << 4) | s), 8); put_value(bw, zz[k], s); run = 0
    if last < 63:
        bw.put(AC_SYMS.index(0x00), 8)
    return zz[0]


def write_jpeg(width, height, comps, quant, blocks, restart=0):
    """comps: [(id, h, v)]; quant: 64 values (natural order) shared by all components; blocks[c][by][bx] = 64 quantised coefficients (natural order)."""
    hmax = max(c[1] for c in comps); vmax = max(c[2] for c in comps)
    mcux = (width + 8 * hmax - 1) // (8 * hmax); mcuy = (height + 8 * vmax - 1) // (8 * vmax)
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xdb" + struct.pack(">HB", 67, 0) + bytes(int(quant[DEZIGZAG[k]]) for k in range(64))
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * len(comps), 8, height, width, len(comps)) + b"".join(struct.pack("BBB", c[0], (c[1] << 4) | c[2], 0) for c in comps)
    out += b"\xff\xc4" + struct.pack(">HB", 19 + len(DC_SYMS), 0x00) + bytes([0, 0, 0, len(DC_SYMS)] + [0] * 12) + bytes(DC_SYMS)
    out += b"\xff\xc4" + struct.pack(">HB", 19 + len(AC_SYMS), 0x10) + bytes([0] * 7 + [len(AC_SYMS)] + [0] * 8) + bytes(AC_SYMS)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps)) + b"".join(struct.pack("BB", c[0], 0x00) for c in comps) + bytes([0, 63, 0])
    bw = Bits(); pred = [0] * len(comps); count = 0; rst = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and count and count % restart == 0:
                bw.flush(); out += bw.out + bytes([0xff, 0xd0 + (rst & 7)]); rst += 1
                bw = Bits(); pred = [0] * len(comps)
            for ci, (_, h, v) in enumerate(comps):
                for y in range(v):
                    for x in range(h):
                        pred[ci] = encode_block(bw, blocks[ci][my * v + y][mx * h + x], pred[ci])
            count += 1
    bw.flush()
    return bytes(out + bw.out + b"\xff\xd9")


# ------------------------------------------------------------------ the published formulas, restated (Python integers: arbitrary precision, checked to stay in 32 bits)
def f2f(x):
    return int(np.float32(x) * np.float32(4096) + 0.5)            # (int)(x * 4096 + 0.5): C truncates toward zero, as int() does


def idct_1d(s):
    s0, s1, s2, s3, s4, s5, s6, s7 = s
    p2, p3 = s2, s6
    p1 = (p2 + p3) * f2f(0.5411961)
    t2 = p1 + p3 * f2f(-1.847759065); t3 = p1 + p2 * f2f(0.765366865)
    t0 = (s0 + s4) * 4096; t1 = (s0 - s4) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s7, s5, s3, s1
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * f2f(1.175875602)
    t0 *= f2f(0.298631336); t1 *= f2f(2.053119869); t2 *= f2f(3.072711026); t3 *= f2f(1.501321110)
    p1 = p5 + p1 * f2f(-0.899976223); p2 = p5 + p2 * f2f(-2.562915447)
    p3 *= f2f(-1.961570560); p4 *= f2f(-0.390180644)
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3
    return x0, x1, x2, x3, t0, t1, t2, t3


def clamp(v):
    return 0 if v < 0 else 255 if v > 255 else v


def idct_block(d):
    d = [int(np.int16(v)) for v in d]
    val = [0] * 64
    for i in range(8):
        col = [d[i + 8 * k] for k in range(8)]
        if not any(col[1:]):
            for k in range(8):
                val[i + 8 * k] = col[0] * 4
        else:
            x0, x1, x2, x3, t0, t1, t2, t3 = idct_1d(col)
            x0 += 512; x1 += 512; x2 += 512; x3 += 512
            for k, v in ((0, x0 + t3), (7, x0 - t3), (1, x1 + t2), (6, x1 - t2), (2, x2 + t1), (5, x2 - t1), (3, x3 + t0), (4, x3 - t0)):
                val[i + 8 * k] = v >> 10
    out = np.zeros((8, 8), np.uint8)
    for r in range(8):
        x0, x1, x2, x3, t0, t1, t2, t3 = idct_1d(val[8 * r:8 * r + 8])
        b = 65536 + (128 << 17)
        for k, v in ((0, x0 + t3), (7, x0 - t3), (1, x1 + t2), (6, x1 - t2), (2, x2 + t1), (5, x2 - t1), (3, x3 + t0), (4, x3 - t0)):
            out[r, k] = clamp((v + b) >> 17)
    return out


def row_hv2(near, far):
    w = len(near); t = [3 * int(near[i]) + int(far[i]) for i in range(w)]
    if w == 1:
        return [(t[0] + 2) >> 2] * 2
    out = [0] * (2 * w); out[0] = (t[0] + 2) >> 2; out[-1] = (t[-1] + 2) >> 2
    for i in range(1, w):
        out[2 * i - 1] = (3 * t[i - 1] + t[i] + 8) >> 4; out[2 * i] = (3 * t[i] + t[i - 1] + 8) >> 4
    return out


def row_h2(a):
    a = [int(v) for v in a]; w = len(a)
    if w == 1:
        return [a[0], a[0]]
    out = [0] * (2 * w); out[0] = a[0]; out[1] = (3 * a[0] + a[1] + 2) >> 2
    for i in range(1, w - 1):
        out[2 * i] = (3 * a[i] + 2 + a[i - 1]) >> 2; out[2 * i + 1] = (3 * a[i] + 2 + a[i + 1]) >> 2
    out[2 * (w - 1)] = (3 * a[w - 2] + a[w - 1] + 2) >> 2; out[2 * w - 1] = a[w - 1]
    return out


def expected_image(width, height, comps, quant, blocks):
    hmax = max(c[1] for c in comps); vmax = max(c[2] for c in comps)
    mcux = (width + 8 * hmax - 1) // (8 * hmax); mcuy = (height + 8 * vmax - 1) // (8 * vmax)
    planes = []
    for ci, (_, h, v) in enumerate(comps):
        p = np.zeros((mcuy * v * 8, mcux * h * 8), np.uint8)
        for by in range(mcuy * v):
            for bx in range(mcux * h):
                p[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = idct_block([int(blocks[ci][by][bx][k]) * int(quant[k]) for k in range(64)])
        planes.append(p)
    rows_of = []
    for ci, (_, h, v) in enumerate(comps):                                  # load_jpeg_image's row stepping
        hs, vs = hmax // h, vmax // v
        cy = (height * v + vmax - 1) // vmax; wl = (width + hs - 1) // hs
        line0 = line1 = 0; ystep = vs >> 1; ypos = 0; rows = []
        for j in range(height):
            ybot = ystep >= (vs >> 1)
            near, far = (planes[ci][line1], planes[ci][line0]) if ybot else (planes[ci][line0], planes[ci][line1])
            near, far = near[:wl], far[:wl]
            if hs == 1 and vs == 1:
                r = [int(x) for x in near]
            elif hs == 1 and vs == 2:
                r = [(3 * int(a) + int(b) + 2) >> 2 for a, b in zip(near, far)]
            elif hs == 2 and vs == 1:
                r = row_h2(near)
            else:
                r = row_hv2(near, far)
            rows.append(r[:width])
            ystep += 1
            if ystep >= vs:
                ystep = 0; line0 = line1; ypos += 1
                if ypos < cy:
                    line1 += 1
        rows_of.append(rows)
    out = np.zeros((height, width, 4), np.uint8); out[..., 3] = 255
    fx = lambda x: int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8
    for j in range(height):
        for i in range(width):
            if len(comps) == 1:
                out[j, i, :3] = rows_of[0][j][i]
                continue
            y, cb, cr = rows_of[0][j][i], rows_of[1][j][i] - 128, rows_of[2][j][i] - 128
            yf = (y << 20) + (1 << 19)
            g_cb = (cb * -fx(0.34414)) & ~0xffff                              # "& 0xffff0000" on a 32-bit two's-complement int
            out[j, i, 0] = clamp((yf + cr * fx(1.40200)) >> 20)
            out[j, i, 1] = clamp((yf + cr * -fx(0.71414) + g_cb) >> 20)
            out[j, i, 2] = clamp((yf + cb * fx(1.77200)) >> 20)
    return out


def decode(sio, data):
    return sio.decode_jpeg(data)


def test_dc_only_block_by_hand(sio):
    """One 8 x 8 grey block, quantiser 8, DC difference 22: coefficient 176; the column pass of an all-zero-AC column gives 176 x 4 = 704 in every row; the row
    pass of (704, 0, ..., 0) gives 704 x 4096 for all eight outputs; (704 x 4096 + 65536 + (128 << 17)) >> 17 = floor(22 + 0.5 + 128) = 150."""
    blocks = [[[[22] + [0] * 63]]]
    img = decode(sio, write_jpeg(8, 8, [(1, 1, 1)], [8] * 64, blocks))
    assert img.shape == (8, 8, 4)
    assert (img[..., :3] == 150).all() and (img[..., 3] == 255).all()
    # and a saturating one: DC 127 x 8 -> 127 + 128.5 -> 255; DC -128 x 8 -> 0
    assert (decode(sio, write_jpeg(8, 8, [(1, 1, 1)], [8] * 64, [[[[127] + [0] * 63]]]))[..., 0] == 255).all()
    assert (decode(sio, write_jpeg(8, 8, [(1, 1, 1)], [8] * 64, [[[[-130] + [0] * 63]]]))[..., 0] == 0).all()


def test_420_frame_derived_step_by_step(sio):
    """16 x 16, 4:2:0: four luma blocks with different DC terms, Cb with a horizontal AC term, Cr with a vertical one.  Spot values by hand, the frame by the restatement."""
    z = [0] * 64
    ya = [[[10] + z[1:], [-20] + z[1:]], [[30] + z[1:], [5] + z[1:]]]
    cb = [[[4, 6] + z[2:]]]                                             # natural index 1: one horizontal cosine
    crb = [0] * 64; crb[0] = -3; crb[8] = 5                             # natural index 8: one vertical cosine
    q = [4] * 64
    comps = [(1, 2, 2), (2, 1, 1), (3, 1, 1)]
    blocks = [ya, cb, [[crb]]]
    img = decode(sio, write_jpeg(16, 16, comps, q, blocks))
    exp = expected_image(16, 16, comps, q, blocks)
    np.testing.assert_array_equal(img, exp)
    # by hand, pixel (0, 0): Y = 10 x 4 x 4 / 32 + 128.5 -> 133.  Cb row pass of (64, 96, 0 ...): even part 64 x 4096 = 262144; odd part t3 = 96 x (f2f(1.501321110) +
    # f2f(1.175875602) + f2f(-0.899976223) + f2f(-0.390180644)) = 96 x (6149 + 4816 - 3685 - 1597) = 96 x 5683 = 545568; (262144 + 545568 + 65536 + 16777216) >> 17 = 134.
    assert f2f(1.501321110) + f2f(1.175875602) + f2f(-0.899976223) + f2f(-0.390180644) == 5683
    assert idct_block([16, 24] + [0] * 62)[0, 0] == 134
    assert img[0, 0, 3] == 255 and int(exp[0, 0, 0]) == clamp(((133 << 20) + (1 << 19) + (int(idct_block([crb[k] * 4 for k in range(64)])[0, 0]) - 128) * (f2f(1.402) << 8)) >> 20)


@pytest.mark.parametrize("layout,w,h,restart", [("444", 24, 16, 0), ("422", 40, 24, 0), ("440", 24, 40, 0), ("420", 48, 48, 0), ("420", 37, 21, 0),
                                                ("420", 50, 35, 2), ("422", 19, 9, 3), ("grey", 21, 13, 0), ("420", 1, 1, 0), ("420", 2, 3, 0)])
def test_every_sampling_layout_against_the_restatement(sio, layout, w, h, restart):
    comps = {"444": [(1, 1, 1), (2, 1, 1), (3, 1, 1)], "422": [(1, 2, 1), (2, 1, 1), (3, 1, 1)], "440": [(1, 1, 2), (2, 1, 1), (3, 1, 1)],
             "420": [(1, 2, 2), (2, 1, 1), (3, 1, 1)], "grey": [(7, 1, 1)]}[layout]
    rng = np.random.default_rng(hash((layout, w, h)) & 0xffff)
    hmax = max(c[1] for c in comps); vmax = max(c[2] for c in comps)
    mcux = (w + 8 * hmax - 1) // (8 * hmax); mcuy = (h + 8 * vmax - 1) // (8 * vmax)
    q = [int(v) for v in rng.integers(1, 24, 64)]
    blocks = []
    for _, ch, cv in comps:
        rows = []
        for by in range(mcuy * cv):
            row = []
            for bx in range(mcux * ch):
                c = np.zeros(64, np.int64)
                c[0] = rng.integers(-40, 40)
                nz = rng.integers(0, 12)
                idx = rng.choice(np.arange(1, 64), size=nz, replace=False)
                c[idx] = rng.integers(-30, 31, nz) // (1 + idx // 8)          # energy falls with frequency, some saturation stays
                row.append([int(v) for v in c])
            rows.append(row)
        blocks.append(rows)
    img = decode(sio, write_jpeg(w, h, comps, q, blocks, restart))
    np.testing.assert_array_equal(img, expected_image(w, h, comps, q, blocks))


def test_shipped_assets_against_libjpeg_turbo(sio):
    """Baseline, progressive (the helmet's metal-roughness map is SOF2) and restart-interval files: the two decoders read the same coefficients and differ only
    in rounding -- at most 3 codes on any byte, more than 96 % of the bytes equal."""
    from PIL import Image
    files = sorted(glob.glob(os.path.join(ROOT, "assets", "**", "*.jpg"), recursive=True))
    assert len(files) >= 8
    seen_progressive = False
    for f in files:
        with open(f, "rb") as fh:
            data = fh.read()
        seen_progressive |= b"\xff\xc2" in data[:4096]
        a = decode(sio, data)
        b = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert a.shape == b.shape and d.max() <= 3 and (d == 0).mean() > 0.96, (f, int(d.max()), float((d == 0).mean()))
    assert seen_progressive


def test_rejects_what_it_does_not_decode(sio, rt):
    with pytest.raises(rt.IncrError):
        decode(sio, b"\x89PNG\r\n\x1a\n" + bytes(64))
    good = write_jpeg(8, 8, [(1, 1, 1)], [8] * 64, [[[[22] + [0] * 63]]])
    with pytest.raises(rt.IncrError):
        decode(sio, good[:20])                                          # truncated before the frame header
    with pytest.raises(rt.IncrError):
        decode(sio, good.replace(b"\xff\xc0", b"\xff\xc9"))             # arithmetic coding
