"""Deterministic synthetic benchmark scenes.

`sponza_stat()`  -- stand-in for BASELINE configs 3/4: the reference checkout has Sponza.gltf but NOT
    Sponza.bin (.MISSING_LARGE_BLOBS), so the geometry is procedural while every *statistic* follows
    content/models/sponza/Sponza.gltf: 1 mesh with node scale 0.008, 25 primitives whose triangle counts are
    the glTF's index-accessor counts / 3 (sum 262,267), u16 indices, POSITION/NORMAL/TANGENT/TEXCOORD_0 with
    strides 12/12/16/8, 25 materials (base colour + metal/rough + normal textures) of which primitives
    0, 1 and 20 are alphaMode MASK + doubleSided, the glTF's camera node, and 0 lights in the asset itself
    -> a synthetic rig (1 sun + 64 point lights, range 4) is injected so clustered shading has work to do.
`bistro_scale()` -- stand-in for config 5 (Bistro is not in the reference repo): many instanced meshes, u32 indices.

All randomness is splitmix64 keyed by the stated seed, evaluated with numpy uint64 arithmetic, so every
box produces identical bytes.
"""
import math

import numpy as np

from . import ctypes_api as A
from . import scene_io as sio

f32 = np.float32

SPONZA_SEED = 0x5A0B2A
BISTRO_SEED = 0xB157A0

# index-accessor counts of Sponza.gltf primitives 0..24 (sum / 3 = 262,267 triangles)
SPONZA_INDEX_COUNTS = [94308, 10416, 53064, 12258, 2388, 30504, 17628, 8448, 63, 21264, 2640, 69624, 54, 49488, 49536,
                       49536, 33024, 43008, 56832, 43008, 96, 59484, 27552, 9126, 43452]
SPONZA_MASKED = (0, 1, 20)
SPONZA_NODE_SCALE = 0.00800000037997961
SPONZA_CAMERA = dict(translation=[-10.216249465942383, 1.733044981956482, -0.825340986251831],
                     rotation=[0.4627925753593445, -0.5438889265060425, 0.492188036441803, 0.49775373935699463],
                     child_rotation=[-0.7071067690849304, 0, 0, 0.7071067690849304],
                     yfov=0.39959652046304894, aspect=1.7777777777777777)


# ------------------------------------------------------------------ splitmix64 (vectorised)
def splitmix64(seed, n, stream=0):
    """n uint64 values: splitmix64 sequence started at seed + stream * 2^32."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(stream) * np.uint64(1 << 32) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, stream=0):
    """float64 in [0,1) with 53 random bits."""
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


# ------------------------------------------------------------------ procedural textures
def value_noise(seed, stream, size, octaves=4, base=4):
    """Tileable value noise in [0,1]: sum of bilinearly interpolated random lattices."""
    out = np.zeros((size, size), dtype=np.float64)
    amp, total = 1.0, 0.0
    for o in range(octaves):
        cells = min(base << o, size)
        lat = uniform01(seed, cells * cells, stream * 16 + o).reshape(cells, cells)
        t = (np.arange(size, dtype=np.float64) + 0.5) * cells / size
        i0 = np.floor(t).astype(np.int64) % cells
        i1 = (i0 + 1) % cells
        fr = t - np.floor(t)
        fr = fr * fr * (3 - 2 * fr)
        rows = lat[i0][:, i0] * (1 - fr)[None, :] + lat[i0][:, i1] * fr[None, :]
        rows1 = lat[i1][:, i0] * (1 - fr)[None, :] + lat[i1][:, i1] * fr[None, :]
        out += amp * (rows * (1 - fr)[:, None] + rows1 * fr[:, None])
        total += amp
        amp *= 0.5
    return out / total


def make_material_textures(seed, mat_index, size, masked):
    """(basecolor RGBA8 sRGB, metalrough RGBA8 glTF layout G=rough B=metal, normal RGBA8) for one material."""
    h = value_noise(seed, mat_index * 4 + 0, size, 5, 4)
    c = value_noise(seed, mat_index * 4 + 1, size, 3, 2)
    tint = uniform01(seed, 3, 1000 + mat_index)
    base = np.empty((size, size, 4), dtype=np.uint8)
    for k in range(3):
        v = 0.05 + 0.85 * np.clip((0.35 + 0.65 * tint[k]) * (0.25 + 0.75 * h) * (0.6 + 0.4 * c), 0, 1)
        base[..., k] = np.round(v * 255).astype(np.uint8)
    if masked:   # alpha lattice: opaque strands on a transparent ground
        y, x = np.mgrid[0:size, 0:size]
        cell = max(size // 16, 2)
        strands = ((x % cell) < cell * 0.45) | ((y % cell) < cell * 0.3) | (h > 0.62)
        base[..., 3] = np.where(strands, 255, 0).astype(np.uint8)
    else:
        base[..., 3] = 255
    mr = np.zeros((size, size, 4), dtype=np.uint8)
    rough = 0.2 + 0.7 * value_noise(seed, mat_index * 4 + 2, size, 3, 4)
    metal = 255 if (mat_index % 5 == 3) else 0
    mr[..., 1] = np.round(rough * 255).astype(np.uint8)
    mr[..., 2] = metal
    mr[..., 3] = 255
    # normal map from the height-field gradient
    gx = (np.roll(h, -1, axis=1) - np.roll(h, 1, axis=1)) * size / 64.0
    gy = (np.roll(h, -1, axis=0) - np.roll(h, 1, axis=0)) * size / 64.0
    n = np.stack([-gx, -gy, np.ones_like(h)], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    nm = np.empty((size, size, 4), dtype=np.uint8)
    nm[..., :3] = np.round((n * 0.5 + 0.5) * 255).astype(np.uint8)
    nm[..., 3] = 255
    return base, mr, nm


# ------------------------------------------------------------------ parametric sheets
def _grid_dims(ntri, aspect):
    """Smallest nu x nv quad grid with 2*nu*nv >= ntri and nu/nv ~ aspect."""
    quads = max((ntri + 1) // 2, 1)
    nv = max(int(round(math.sqrt(quads / max(aspect, 1e-3)))), 1)
    nu = max((quads + nv - 1) // nv, 1)
    return nu, nv


def sheet(fn, ntri, aspect=1.0, uv_tile=(4.0, 4.0)):
    """Tessellate fn(u, v) -> (x, y, z) over [0,1]^2 into EXACTLY ntri triangles (grid truncated)."""
    nu, nv = _grid_dims(ntri, aspect)
    u = np.linspace(0, 1, nu + 1)
    v = np.linspace(0, 1, nv + 1)
    U, V = np.meshgrid(u, v, indexing="xy")           # rows = v
    U, V = U.ravel(), V.ravel()
    P = np.stack(fn(U, V), axis=-1)
    e = 1e-3
    Pu = (np.stack(fn(np.clip(U + e, 0, 1), V), axis=-1) - np.stack(fn(np.clip(U - e, 0, 1), V), axis=-1))
    Pv = (np.stack(fn(U, np.clip(V + e, 0, 1)), axis=-1) - np.stack(fn(U, np.clip(V - e, 0, 1)), axis=-1))
    N = np.cross(Pu, Pv)
    ln = np.linalg.norm(N, axis=-1, keepdims=True); N = np.where(ln > 1e-20, N / np.maximum(ln, 1e-20), np.array([0, 1.0, 0]))
    T = Pu / np.maximum(np.linalg.norm(Pu, axis=-1, keepdims=True), 1e-20)
    T4 = np.concatenate([T, np.ones((T.shape[0], 1))], axis=-1)
    UV = np.stack([U * uv_tile[0], V * uv_tile[1]], axis=-1)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    a = (j * (nu + 1) + i).ravel(); b = a + 1; c = a + (nu + 1); d = c + 1
    tris = np.stack([np.stack([a, b, d], -1), np.stack([a, d, c], -1)], axis=1).reshape(-1, 3)[:ntri]
    return P, N, T4, UV, tris


def assemble(sheets, ntri):
    """sheets: list of (fn, weight, aspect, uv_tile); split the triangle budget by weight, last sheet takes the rest."""
    wsum = sum(s[1] for s in sheets)
    budgets, left = [], ntri
    for k, s in enumerate(sheets):
        b = left if k == len(sheets) - 1 else min(left, max(int(ntri * s[1] / wsum), 1 if left > len(sheets) - k - 1 else 0))
        budgets.append(b); left -= b
    Ps, Ns, Ts, UVs, Is, base = [], [], [], [], [], 0
    for s, b in zip(sheets, budgets):
        if b <= 0:
            continue
        P, N, T4, UV, tr = sheet(s[0], b, s[2], s[3])
        Ps.append(P); Ns.append(N); Ts.append(T4); UVs.append(UV); Is.append(tr + base); base += P.shape[0]
    P = np.concatenate(Ps); tr = np.concatenate(Is)
    assert tr.shape[0] == ntri and P.shape[0] <= 65535, (tr.shape, P.shape, ntri)
    return (P.astype(f32), np.concatenate(Ns).astype(f32), np.concatenate(Ts).astype(f32), np.concatenate(UVs).astype(f32), tr.astype(np.uint16))


# surfaces are written in metres and divided by the node scale at the end; glTF right-handed, Y up.
def plane(o, du, dv, bump=0.0, freq=(3, 3)):
    o, du, dv = np.asarray(o, float), np.asarray(du, float), np.asarray(dv, float)
    n = np.cross(du, dv); n = n / np.linalg.norm(n)

    def fn(u, v):
        b = bump * np.sin(2 * math.pi * freq[0] * u) * np.sin(2 * math.pi * freq[1] * v)
        return tuple(o[k] + du[k] * u + dv[k] * v + n[k] * b for k in range(3))
    return fn


def cylinder(c, r, h, flutes=0, depth=0.0, outward=True, arc=(0.0, 1.0)):
    c = np.asarray(c, float)

    def fn(u, v):
        th = 2 * math.pi * (arc[0] + (arc[1] - arc[0]) * u)
        rr = r + depth * np.cos(flutes * th) if flutes else r + 0 * u
        s = 1.0 if outward else -1.0
        return (c[0] + rr * np.cos(s * -th), c[1] + h * v, c[2] + rr * np.sin(s * -th))
    return fn


def arch(c, r, width, axis="z", thick=0.0):
    """Half-cylinder vault spanning `width` along `axis`, visible from below (normal points down/inward)."""
    c = np.asarray(c, float)

    def fn(u, v):
        th = math.pi * u
        a, b = r * np.cos(th), r * np.sin(th)
        if axis == "z":
            return (c[0] + a, c[1] + b, c[2] + width * v)
        return (c[0] + width * (1 - v), c[1] + b, c[2] + a)
    return fn


def blob(c, r, bump, k):
    c = np.asarray(c, float)

    def fn(u, v):
        th, ph = 2 * math.pi * u, math.pi * (0.02 + 0.96 * v)
        rr = r * (1 + bump * np.sin(k * th) * np.sin(k * ph))
        return (c[0] + rr * np.sin(ph) * np.cos(-th), c[1] - rr * np.cos(ph), c[2] + rr * np.sin(ph) * np.sin(-th))
    return fn


def cloth(o, du, drop, wave, nw):
    o, du = np.asarray(o, float), np.asarray(du, float)
    n = np.cross(du, [0, -1.0, 0]); n = n / np.linalg.norm(n)

    def fn(u, v):
        w = wave * np.sin(2 * math.pi * nw * u) * (0.3 + 0.7 * v)
        return tuple(o[k] + du[k] * u + (-drop * v if k == 1 else 0.0) + n[k] * w for k in range(3))
    return fn


def _sponza_parts(seed):
    """25 part descriptions (list of sheets each).  Atrium: x in [-14, 13], z in [-9, 8.5], y in [0, 11.4] (metres)."""
    X0, X1, Z0, Z1, H = -14.0, 13.0, -9.0, 8.5, 11.4
    cx = [-11.0 + 3.6 * i for i in range(7)]           # column rows along x
    zin = (-4.2, 3.8)                                    # inner colonnade lines (aisles between these and the walls)
    rnd = uniform01(seed, 4096, 7)
    parts = [None] * 25
    # 0: plants (MASK, double sided): leaf fans around 12 vases
    leaves = []
    for i in range(12):
        bx, bz = cx[i % 6] + 1.8, (-1.6 if i < 6 else 1.4)
        for l in range(20):
            ang = 2 * math.pi * (l / 20.0) + rnd[i * 20 + l]
            tip = np.array([math.cos(ang), 0.9 + 0.8 * rnd[300 + i * 20 + l], math.sin(ang)]) * (0.7 + 0.5 * rnd[600 + i * 20 + l])
            side = np.array([-math.sin(ang), 0.0, math.cos(ang)]) * 0.22
            leaves.append((plane([bx, 0.55, bz] - side / 2, side, tip, 0.05, (1, 2)), 1.0, 0.4, (1.0, 2.0)))
    parts[0] = leaves
    # 1: chains (MASK): thin vertical strips under the upper arches
    parts[1] = [(plane([x + 1.8, 9.2, z], [0.06, 0, 0], [0, -3.2, 0]), 1.0, 0.05, (0.25, 8.0)) for x in cx[:6] for z in (-3.9, 3.5)]
    # 2: ground-floor arcade vaults along both aisles
    parts[2] = [(arch([x + 1.8, 3.6, z0], 1.75, zw, "z"), 1.0, 1.2, (2.0, 2.0)) for x in cx[:6] for (z0, zw) in ((Z0, zin[0] - Z0), (zin[1], Z1 - zin[1]))]
    # 3: flag poles
    parts[3] = [(cylinder([x + 1.8, 5.2, z], 0.04, 2.6), 1.0, 0.3, (1.0, 4.0)) for x in cx[:6] for z in (-3.6, 3.2)]
    # 4: outer shell far away (what secondary rays see through the open roof edges)
    parts[4] = [(plane([-15, -1, -9.5], [30, 0, 0], [0, 0, 19], 0.0), 1.0, 1.5, (4, 4)),
                (plane([-15.0, 11.6, -9.5], [0, 0, 19], [30, 0, 0]), 0.3, 1.5, (4, 4))]
    # 5: first-floor walls (inner faces)
    parts[5] = [(plane([X0, 4.6, Z0], [X1 - X0, 0, 0], [0, 5.6, 0], 0.03, (9, 3)), 1.0, 4.0, (4, 2)),
                (plane([X1, 4.6, Z1], [X0 - X1, 0, 0], [0, 5.6, 0], 0.03, (9, 3)), 1.0, 4.0, (4, 2))]
    # 6: first-floor arcade vaults
    parts[6] = [(arch([x + 1.8, 8.4, z0], 1.7, zw, "z"), 1.0, 1.2, (2.0, 2.0)) for x in cx[:6] for (z0, zw) in ((Z0, zin[0] - Z0), (zin[1], Z1 - zin[1]))]
    # 7: column bases
    parts[7] = [(cylinder([x, 0.0, z], 0.62, 0.5, 8, 0.03), 1.0, 3.0, (2, 0.5)) for x in cx for z in zin]
    # 8, 12, 20: tiny details
    parts[8] = [(plane([X0 + 0.02, 1.0, -0.8], [0, 0, 1.6], [0, 1.2, 0]), 1.0, 1.3, (1, 1))]
    parts[12] = [(plane([-0.9, 2.05, Z1 - 0.03], [-0.9, 0, 0], [0, 0.75, 0]), 1.0, 1.2, (1, 1))]
    parts[20] = [(plane([x, 1.1, Z0 + 0.05], [0.9, 0, 0], [0, 0.7, 0]), 1.0, 1.3, (2, 2)) for x in (-5.0, 4.0)]
    # 9, 13: first-floor columns
    parts[9] = [(cylinder([x, 4.6, zin[0]], 0.33, 3.0, 12, 0.02), 1.0, 0.7, (2, 3)) for x in cx]
    parts[13] = [(cylinder([x, 4.6, zin[1]], 0.33, 3.0, 12, 0.02), 1.0, 0.7, (2, 3)) for x in cx]
    # 10: end door
    parts[10] = [(plane([X1 - 0.02, 0, 1.2], [0, 0, -2.4], [0, 3.4, 0], 0.02, (2, 3)), 1.0, 0.7, (1, 2))]
    # 11: cornices (long bumpy bands at both floor levels on all inner faces)
    parts[11] = [(plane([X0, y, zin[0] + 0.45], [X1 - X0, 0, 0], [0, 0.55, 0], 0.05, (40, 1)), 1.0, 40.0, (16, 0.5)) for y in (4.05, 8.9)] + \
                [(plane([X1, y, zin[1] - 0.45], [X0 - X1, 0, 0], [0, 0.55, 0], 0.05, (40, 1)), 1.0, 40.0, (16, 0.5)) for y in (4.05, 8.9)]
    # 14, 15, 16: hanging fabrics between first-floor columns (facing the courtyard)
    parts[14] = [(cloth([cx[i] + 0.4, 8.2, zin[0] + 0.35], [2.8, 0, 0], 3.2, 0.18, 5), 1.0, 0.9, (2, 2)) for i in (0, 2, 4)]
    parts[15] = [(cloth([cx[i] + 3.2, 8.2, zin[1] - 0.35], [-2.8, 0, 0], 3.2, 0.18, 5), 1.0, 0.9, (2, 2)) for i in (0, 2, 4)]
    parts[16] = [(cloth([cx[i] + 0.4, 8.2, zin[0] + 0.35], [2.8, 0, 0], 3.2, 0.15, 4), 1.0, 0.9, (2, 2)) for i in (1, 3)]
    # 17, 18, 19: ground-floor fluted columns
    parts[17] = [(cylinder([x, 0.5, zin[0]], 0.45, 3.4, 16, 0.03), 1.0, 0.8, (2, 3)) for x in cx[:3]] + [(cylinder([x, 0.5, zin[1]], 0.45, 3.4, 16, 0.03), 1.0, 0.8, (2, 3)) for x in cx[:2]]
    parts[18] = [(cylinder([x, 0.5, zin[0]], 0.45, 3.4, 16, 0.03), 1.0, 0.8, (2, 3)) for x in cx[3:]] + [(cylinder([x, 0.5, zin[1]], 0.45, 3.4, 16, 0.03), 1.0, 0.8, (2, 3)) for x in cx[2:4]]
    parts[19] = [(cylinder([x, 0.5, zin[1]], 0.45, 3.4, 16, 0.03), 1.0, 0.8, (2, 3)) for x in cx[4:]] + [(cylinder([cx[3] + 1.8, 0.5, z], 0.3, 1.2, 10, 0.04), 0.4, 1.0, (2, 1)) for z in (-1.6, 1.4)]
    # 21: reliefs (bumpy blobs) on the end walls + vases
    parts[21] = [(blob([X0 + 0.3, 2.6, z], 0.55, 0.18, 7), 1.0, 2.0, (2, 1)) for z in (-6.0, -2.0, 1.6, 5.6)] + \
                [(blob([cx[i % 6] + 1.8, 0.3, (-1.6 if i < 6 else 1.4)], 0.32, 0.1, 5), 0.5, 2.0, (2, 1)) for i in range(12)]
    # 22: floor
    parts[22] = [(plane([X0, 0, Z1], [X1 - X0, 0, 0], [0, 0, Z0 - Z1], 0.004, (24, 16)), 1.0, 1.55, (12, 8))]
    # 23: ground-floor walls + end walls
    parts[23] = [(plane([X0, 0, Z0], [X1 - X0, 0, 0], [0, 4.6, 0], 0.02, (9, 2)), 1.0, 5.0, (6, 1.5)),
                 (plane([X1, 0, Z1], [X0 - X1, 0, 0], [0, 4.6, 0], 0.02, (9, 2)), 1.0, 5.0, (6, 1.5)),
                 (plane([X0, 0, Z1], [0, 0, Z0 - Z1], [0, H, 0], 0.02, (5, 4)), 1.0, 1.6, (4, 3)),
                 (plane([X1, 0, Z0], [0, 0, Z1 - Z0], [0, H, 0], 0.02, (5, 4)), 1.0, 1.6, (4, 3))]
    # 24: roofs over both aisles and the first-floor decks (courtyard stays open to the sky)
    parts[24] = [(plane([X0, H - 1.2, Z0], [0, 0, zin[0] - Z0 + 0.4], [X1 - X0, 0, 0], 0.05, (2, 30)), 1.0, 0.2, (1, 8)),
                 (plane([X0, H - 1.2, zin[1] - 0.4], [0, 0, Z1 - zin[1] + 0.4], [X1 - X0, 0, 0], 0.05, (2, 30)), 1.0, 0.2, (1, 8)),
                 (plane([X0, 4.5, Z0], [0, 0, zin[0] - Z0 + 0.5], [X1 - X0, 0, 0], 0.01, (2, 30)), 0.7, 0.2, (1, 8)),
                 (plane([X0, 4.5, zin[1] - 0.5], [0, 0, Z1 - zin[1] + 0.5], [X1 - X0, 0, 0], 0.01, (2, 30)), 0.7, 0.2, (1, 8)),
                 (plane([X0, 4.6, zin[0] + 0.5], [X1 - X0, 0, 0], [0, 0, Z0 - zin[0] - 0.5], 0.01, (30, 2)), 0.7, 5.0, (8, 1)),
                 (plane([X0, 4.6, Z1], [X1 - X0, 0, 0], [0, 0, zin[1] - 0.5 - Z1], 0.01, (30, 2)), 0.7, 5.0, (8, 1))]
    return parts


def _pack_geometry(s, prim_arrays, material_of):
    """Lay the primitives out in one glTF-style buffer: per primitive POSITION, NORMAL, TANGENT, TEXCOORD_0, indices."""
    chunks, off, prims = [], 0, []
    for k, (P, N, T4, UV, I) in enumerate(prim_arrays):
        acc = []
        for arr, stride in ((P, 12), (N, 12), (T4, 16), (UV, 8), (I.reshape(-1), I.dtype.itemsize)):
            raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
            pad = (-raw.size) % 4
            s.buffer_views.append((0, off, raw.size))
            s.accessors.append((len(s.buffer_views) - 1, 0, stride))
            acc.append(len(s.accessors) - 1)
            chunks.append(raw)
            if pad:
                chunks.append(np.zeros(pad, dtype=np.uint8))
            off += raw.size + pad
        prims.append(dict(position=acc[0], normal=acc[1], tangent=acc[2], uv=acc[3], index=acc[4], material=material_of(k),
                          index_count=int(I.size), sphere=sio._bounding_sphere(P)))
    chunks.append(np.zeros(32, dtype=np.uint8))
    s.buffers.append(np.concatenate(chunks))
    return prims


def add_light_rig(s, seed, num_point=64, bounds=((-13.0, 12.0), (0.6, 9.5), (-8.0, 7.5))):
    """Synthetic rig of SURVEY.md 8d config 3: 1 directional sun (time of day) + point lights, range 4, U[50,300] cd."""
    sun = A.Light(); sun.type = A.LIGHT_DIRECTIONAL
    sun.color[0], sun.color[1], sun.color[2] = 1.0, 0.96, 0.9
    sun.intensity, sun.range = 150.0, 0.0
    s.global_lights.append(sun); s.light_indices.append(0); s.light_transforms.append(sio.mat_identity())
    r = uniform01(seed, num_point * 8, 99)
    for i in range(num_point):
        l = A.Light(); l.type = A.LIGHT_POINT
        for k in range(3):
            l.color[k] = float(f32(0.5 + 0.5 * r[i * 8 + k]))
        l.intensity = float(f32(50.0 + 250.0 * r[i * 8 + 3])); l.range = 4.0
        pos = [bounds[k][0] + (bounds[k][1] - bounds[k][0]) * r[i * 8 + 4 + k] for k in range(3)]
        pos[2] = -pos[2]                                    # generator space is glTF right-handed; lights live in world (LH) space
        s.global_lights.append(l); s.light_indices.append(i + 1); s.light_transforms.append(sio.mat_translation(*pos))
    s.finalize_sun_light()


def add_spot_rig(s, seed, num_spot=8, bounds=((-13.0, 12.0), (2.5, 9.0), (-8.0, 7.5))):
    """Spot lights (lighting/common.hlsli:45-83) appended to the scene's rig: even ones unbounded (range 0: the only spots
    that light anything), odd ones with range 6 (quirk Q9: the shadowed redeclaration of radialAttenuation leaves 0, so a
    ranged spot contributes nothing and asks for no shadow ray).  Cones point mostly down with inner / outer half-angles
    of 0.25-0.45 / 0.5-0.8 rad; position = row 3 of the light transform, axis = -row 2 (SimpleMath CreateWorld)."""
    r = uniform01(seed, num_spot * 16, 177)
    for i in range(num_spot):
        l = A.Light(); l.type = A.LIGHT_SPOT
        for k in range(3):
            l.color[k] = float(f32(0.6 + 0.4 * r[i * 16 + k]))
        l.intensity = float(f32(4.0 + 12.0 * r[i * 16 + 3]))
        l.range = 0.0 if i % 2 == 0 else 6.0
        l.spotAngles[0] = float(f32(0.25 + 0.2 * r[i * 16 + 4])); l.spotAngles[1] = float(f32(0.5 + 0.3 * r[i * 16 + 5]))
        pos = [bounds[k][0] + (bounds[k][1] - bounds[k][0]) * r[i * 16 + 6 + k] for k in range(3)]
        pos[2] = -pos[2]
        fwd = sio.normalize3([r[i * 16 + 9] - 0.5, -1.0, r[i * 16 + 10] - 0.5])
        gi = len(s.global_lights)
        s.global_lights.append(l); s.light_indices.append(gi); s.light_transforms.append(sio.mat_create_world(pos, fwd, (0, 0, 1)))
    return s


def sponza_stat(texture_size=1024, with_lights=True, tri_scale=1.0, seed=SPONZA_SEED):
    """The 'Sponza-stat' scene.  tri_scale < 1 shrinks every primitive's triangle count (test-sized variants)."""
    s = sio.PackedScene()
    parts = _sponza_parts(seed)
    prim_arrays = []
    for k, ic in enumerate(SPONZA_INDEX_COUNTS):
        ntri = max(int(round(ic // 3 * tri_scale)), 2) if tri_scale != 1.0 else ic // 3
        P, N, T4, UV, I = assemble(parts[k], ntri)
        prim_arrays.append(((P / f32(SPONZA_NODE_SCALE)).astype(f32), N, T4, UV, I))
    prims = _pack_geometry(s, prim_arrays, lambda k: k)
    s.mesh_names.append("Mesh_0"); s.mesh_primitives.append(prims)
    RH2LH = sio.mat_identity(); RH2LH[2, 2] = f32(-1)
    sc = f32(SPONZA_NODE_SCALE)
    s.mesh_transforms.append(sio.mat_mul(sio.mat_scale(sc, sc, sc), RH2LH))
    for k in range(25):
        masked = k in SPONZA_MASKED
        base, mr, nm = make_material_textures(seed, k, texture_size, masked)
        m = A.Material()
        for n in A._MAT_TEX:
            setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
        bc = 0.5879999995231628 if k == 0 else 1.0          # Material_0 of Sponza.gltf has baseColorFactor 0.588
        for i in range(3):
            m.baseColorFactor[i] = bc; m.emissiveFactor[i] = 0.0
        m.metallicFactor, m.roughnessFactor, m.aoStrength = 1.0, 1.0, 1.0
        m.baseColorTextureIndex = s.add_texture(base, A.TEX_RGBA8_SRGB)
        if k != 12:                                          # Material_12 has only a base-colour texture
            m.metallicRoughnessTextureIndex = s.add_texture(mr, A.TEX_BGRA8_SWIZZLE_RG)
            m.normalTextureIndex = s.add_texture(nm, A.TEX_RG_SNORM)
        m.alphaMode = A.ALPHA_MASKED if masked else A.ALPHA_OPAQUE
        m.doubleSided = 1 if masked else 0
        s.materials.append(m)
    cam = SPONZA_CAMERA
    parent = sio.mat_mul(sio.mat_mul(sio.mat_identity(), sio.mat_from_quaternion(*cam["rotation"])), sio.mat_translation(*cam["translation"]))
    parent = sio.mat_mul(parent, RH2LH)
    child = sio.mat_mul(sio.mat_from_quaternion(*cam["child_rotation"]), parent)
    s.cameras.append({"transform": child, "yfov": cam["yfov"], "aspect": cam["aspect"]})
    if with_lights:
        add_light_rig(s, seed)
    return s


def bistro_scale(num_instances=1024, tris_per_mesh=2800, num_materials=100, texture_size=256, seed=BISTRO_SEED):
    """'Bistro-scale' scene: a street of instanced props; >= 2.8 M triangles at the defaults, u32 indices."""
    s = sio.PackedScene()
    r = uniform01(seed, num_instances * 8 + 64, 3)
    RH2LH = sio.mat_identity(); RH2LH[2, 2] = f32(-1)
    kinds = 16
    protos = []
    for k in range(kinds):
        rr = 0.5 + 0.5 * r[k]
        sheets = [(blob([0, 1.0 + rr, 0], 0.8 * rr + 0.4, 0.12 + 0.1 * (k % 3), 3 + k % 5), 1.0, 2.0, (2, 1)),
                  (cylinder([0, 0, 0], 0.25 + 0.1 * (k % 4), 1.2, 8, 0.03), 0.5, 1.0, (2, 2))]
        protos.append(assemble(sheets, tris_per_mesh))
    # one shared geometry buffer per prototype kind; every instance is its own mesh entry (the reference has no real instancing)
    prim_arrays = [(P, N, T4, UV, I.astype(np.uint32)) for (P, N, T4, UV, I) in protos]
    base_prims = _pack_geometry(s, prim_arrays, lambda k: 0)
    side = int(math.ceil(math.sqrt(num_instances)))
    for i in range(num_instances):
        k = i % kinds
        p = dict(base_prims[k]); p["material"] = 1 + (i % (num_materials - 1))
        gx, gz = (i % side) * 3.0 - side * 1.5, (i // side) * 3.0 - side * 1.5
        sc = 0.8 + 0.6 * r[64 + i * 8]
        ang = 2 * math.pi * r[64 + i * 8 + 1]
        q = (0.0, math.sin(ang / 2), 0.0, math.cos(ang / 2))
        t = sio.mat_mul(sio.mat_mul(sio.mat_scale(sc, sc, sc), sio.mat_from_quaternion(*q)), sio.mat_translation(gx, 0.0, gz))
        s.mesh_names.append("prop_%d" % i); s.mesh_primitives.append([p]); s.mesh_transforms.append(sio.mat_mul(t, RH2LH))
    # ground (material 0) as one more mesh
    ground = assemble([(plane([-side * 1.6, 0, side * 1.6], [side * 3.2, 0, 0], [0, 0, -side * 3.2], 0.02, (20, 20)), 1.0, 1.0, (64, 64))], 20000)
    gp = _pack_geometry(s, [(ground[0], ground[1], ground[2], ground[3], ground[4].astype(np.uint32))], lambda k: 0)
    # _pack_geometry appended a second buffer; re-point its views
    nb = len(s.buffers) - 1
    for vi in range(len(s.buffer_views) - 5, len(s.buffer_views)):
        b, o, l = s.buffer_views[vi]; s.buffer_views[vi] = (nb, o, l)
    s.mesh_names.append("ground"); s.mesh_primitives.append(gp); s.mesh_transforms.append(RH2LH.copy())
    for k in range(num_materials):
        masked = (k % 25 == 7)
        base, mr, nm = make_material_textures(seed, k, texture_size, masked)
        m = A.Material()
        for n in A._MAT_TEX:
            setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
        for i in range(3):
            m.baseColorFactor[i] = 1.0; m.emissiveFactor[i] = 0.0
        m.metallicFactor, m.roughnessFactor, m.aoStrength = 1.0, 1.0, 1.0
        m.baseColorTextureIndex = s.add_texture(base, A.TEX_RGBA8_SRGB)
        m.metallicRoughnessTextureIndex = s.add_texture(mr, A.TEX_BGRA8_SWIZZLE_RG)
        m.normalTextureIndex = s.add_texture(nm, A.TEX_RG_SNORM)
        m.alphaMode = A.ALPHA_MASKED if masked else A.ALPHA_OPAQUE
        m.doubleSided = 1 if masked else 0
        s.materials.append(m)
    eye = sio.mat_translation(-side * 1.2, 2.2, -side * 1.2)
    rot = sio.mat_from_quaternion(0.0, math.sin(math.pi / 8), 0.0, math.cos(math.pi / 8))
    cam_world = sio.mat_mul(sio.mat_mul(rot, eye), sio.mat_identity())
    cam_world[2, :3] = -cam_world[2, :3]                     # FView::Reset takes look = -row 2
    s.cameras.append({"transform": cam_world, "yfov": 0.7, "aspect": 16 / 9})
    add_light_rig(s, seed, 64, ((-side * 1.4, side * 1.4), (0.8, 4.0), (-side * 1.4, side * 1.4)))
    return s


def add_decals(s, seed=SPONZA_SEED, texture_size=64):
    """Decal meshes for the Sponza-stat atrium (the reference keeps meshes named "decal*" in FScene::m_sceneMeshDecals,
    scene.cpp:281): four primitives over the nave floor in front of the camera, overlapping each other, with the material
    combinations the decal pass distinguishes -- base colour only, + normal map, + metallic-roughness map, double-sided."""
    sc = f32(SPONZA_NODE_SCALE)
    parts = [
        # (sheet function, triangles, aspect, uv tile, material flags (normal map, mr map, double sided), alpha stream)
        (plane([-7.5, 0.03, 1.2], [5.0, 0, 0], [0, 0, -2.6]), 2, 2.0, (1.0, 1.0), (False, False, False)),
        (plane([-5.5, 0.05, 2.0], [4.0, 0, 0], [0, 0, -3.2]), 8, 1.2, (2.0, 2.0), (True, True, False)),
        (plane([-4.0, 0.9, 1.6], [3.0, 0.4, 0], [0, 0.2, -3.0], 0.25, (2, 2)), 200, 1.0, (1.0, 1.0), (True, False, True)),
        (plane([-9.0, 0.04, -0.4], [6.0, 0, 0], [0, 0, -1.0]), 2, 6.0, (3.0, 1.0), (False, True, False)),
    ]
    prim_arrays, flags = [], []
    for fn, ntri, aspect, tile, fl in parts:
        P, N, T4, UV, I = sheet(fn, ntri, aspect, tile)
        prim_arrays.append(((P / sc).astype(f32), N.astype(f32), T4.astype(f32), UV.astype(f32), I.astype(np.uint16)))
        flags.append(fl)
    first_material = len(s.materials)
    prims = _pack_geometry(s, prim_arrays, lambda k: first_material + k)
    # _pack_geometry appends a NEW buffer: its views must point at it
    new_buffer = len(s.buffers) - 1
    nviews = 5 * len(prim_arrays)
    s.buffer_views[-nviews:] = [(new_buffer, off, size) for (_, off, size) in s.buffer_views[-nviews:]]
    for k, (has_nm, has_mr, double_sided) in enumerate(flags):
        base, mr, nm = make_material_textures(seed, 200 + k, texture_size, False)
        alpha = value_noise(seed, 900 + k, texture_size, 3, 2)
        base[..., 3] = np.round(np.clip(0.15 + 0.85 * alpha, 0, 1) * 255).astype(np.uint8)      # opacity = base colour alpha
        m = A.Material()
        for n in A._MAT_TEX:
            setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
        for i in range(3):
            m.baseColorFactor[i] = 1.0; m.emissiveFactor[i] = 0.0
        m.metallicFactor, m.roughnessFactor, m.aoStrength = 1.0, 1.0, 1.0
        m.baseColorTextureIndex = s.add_texture(base, A.TEX_RGBA8_SRGB)
        if has_mr:
            m.metallicRoughnessTextureIndex = s.add_texture(mr, A.TEX_BGRA8_SWIZZLE_RG)
        if has_nm:
            m.normalTextureIndex = s.add_texture(nm, A.TEX_RG_SNORM)
        m.alphaMode, m.doubleSided = A.ALPHA_BLEND if hasattr(A, "ALPHA_BLEND") else A.ALPHA_MASKED, 1 if double_sided else 0
        s.materials.append(m)
    world = s.mesh_transforms[0]
    s.decal_primitives.extend((p, world) for p in prims)
    return s


def triangle_soup(seed, ntri, kind="plain", extent=3.0):
    """Random triangle soup in front of the default camera (edge-case scenes for the parity tests): kind = plain |
    degenerate (half the vertices coincide: zero-area triangles, duplicate positions) | flat (all in the plane z = 0:
    coplanar, depth ties everywhere) | u32 (32-bit indices)."""
    rng = np.random.default_rng(seed)
    s = sio.PackedScene()
    nv = max(3, ntri // 2 + 3)
    P = rng.uniform(-extent, extent, (nv, 3)).astype(f32)
    if kind == "progression":      # triangle k sits at x = 1.2^k and is as large as it is far out: the SAH peels one triangle per level (a 60-level binary tree)
        nv = 3 * ntri
        x = (f32(1.2) ** np.arange(ntri, dtype=f32)) * f32(1e-3)
        P = np.zeros((nv, 3), f32)
        P[0::3] = np.stack([x, -0.2 * x, 0.1 * x], 1); P[1::3] = np.stack([x, 0.3 * x, -0.2 * x], 1); P[2::3] = np.stack([1.1 * x, 0.1 * x, 0.35 * x], 1)
    if kind == "degenerate":
        P[: nv // 2] = P[0]
    elif kind == "flat":
        P[:, 2] = 0.0
    I = rng.integers(0, nv, (ntri, 3)).astype(np.uint32 if kind == "u32" else np.uint16)
    if kind == "progression":
        I = np.arange(nv, dtype=np.uint16).reshape(ntri, 3)
    N = rng.normal(size=(nv, 3)).astype(f32); N /= np.linalg.norm(N, axis=1, keepdims=True)
    T4 = np.tile(np.array([1, 0, 0, 1], f32), (nv, 1))
    UV = rng.uniform(0, 1, (nv, 2)).astype(f32)
    prims = _pack_geometry(s, [(P, N, T4, UV, I)], lambda k: 0)
    s.mesh_names.append("soup"); s.mesh_primitives.append(prims); s.mesh_transforms.append(sio.mat_identity())
    m = A.Material()
    for n in A._MAT_TEX:
        setattr(m, n + "TextureIndex", -1); setattr(m, n + "SamplerIndex", -1)
    for i in range(3):
        m.baseColorFactor[i] = 0.7; m.emissiveFactor[i] = 0.0
    m.metallicFactor, m.roughnessFactor, m.aoStrength = 0.2, 0.6, 1.0
    m.alphaMode, m.doubleSided = A.ALPHA_OPAQUE, 1 if kind == "flat" else 0
    s.materials.append(m)
    sun = A.Light(); sun.type = A.LIGHT_DIRECTIONAL
    sun.color[0], sun.color[1], sun.color[2] = 1.0, 1.0, 1.0
    sun.intensity, sun.range = 100.0, 0.0
    s.global_lights.append(sun); s.light_indices.append(0); s.light_transforms.append(sio.mat_identity())
    s.finalize_sun_light()
    return s
