"""Host-side scene packing: glTF -> the packed arrays of `incr_scene_desc`.

Mirrors the load-time half of the reference's FScene (reference, relative to source/demo-dll/):
  FScene::ReloadModel      src/scene.cpp:91-219   (RH2LH root, node walk, sun direction)
  FScene::LoadNode         src/scene.cpp:228-276  (node = S*R*T, row vectors, child * parent)
  FScene::LoadMesh         src/scene.cpp:278-369  (primitive accessors, bounds)
  LoadMeshBufferViews / LoadMeshAccessors  src/scene.cpp:408-478
  FScene::LoadMaterial     src/scene.cpp:887-1005 (texture formats: sRGB / UNORM / BC5_SNORM / B<->R swizzle)
  FScene::LoadCamera       src/scene.cpp:1440-1460
  FScene::LoadLights       src/scene.cpp:1462-1486
  FScene::UpdateSunDirection src/scene.cpp:1594-1655
  FView::Reset/UpdateViewTransform  src/demo.cpp:186-230, projection src/demo.cpp:94-102
All arithmetic is done in float32 like the reference's SimpleMath code.
"""
import ctypes as C
import json
import math
import os

import numpy as np

from . import ctypes_api as A

f32 = np.float32


# ------------------------------------------------------------------ SimpleMath-style helpers (row vectors)
def mat_identity():
    return np.eye(4, dtype=f32)


def mat_mul(a, b):
    """c_ij = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j in float32."""
    a = a.astype(f32)
    b = b.astype(f32)
    out = np.zeros((4, 4), dtype=f32)
    for i in range(4):
        for j in range(4):
            out[i, j] = f32(f32(f32(a[i, 0] * b[0, j]) + f32(a[i, 1] * b[1, j])) + f32(a[i, 2] * b[2, j])) + f32(a[i, 3] * b[3, j])
    return out


def mat_translation(x, y, z):
    m = mat_identity()
    m[3, 0], m[3, 1], m[3, 2] = f32(x), f32(y), f32(z)
    return m


def mat_scale(x, y, z):
    m = mat_identity()
    m[0, 0], m[1, 1], m[2, 2] = f32(x), f32(y), f32(z)
    return m


def mat_from_quaternion(qx, qy, qz, qw):
    """Matrix::CreateFromQuaternion == XMMatrixRotationQuaternion (row-vector form)."""
    x, y, z, w = f32(qx), f32(qy), f32(qz), f32(qw)
    one, two = f32(1), f32(2)
    m = mat_identity()
    m[0, 0] = one - two * (y * y + z * z); m[0, 1] = two * (x * y + z * w); m[0, 2] = two * (x * z - y * w)
    m[1, 0] = two * (x * y - z * w); m[1, 1] = one - two * (x * x + z * z); m[1, 2] = two * (y * z + x * w)
    m[2, 0] = two * (x * z + y * w); m[2, 1] = two * (y * z - x * w); m[2, 2] = one - two * (x * x + y * y)
    return m


def mat_create_world(position, forward, up):
    """SimpleMath Matrix::CreateWorld: zaxis = normalize(-forward), xaxis = normalize(cross(up, zaxis)), yaxis = cross(zaxis, xaxis)."""
    z = normalize3(-np.asarray(forward, dtype=f32))
    x = normalize3(np.cross(np.asarray(up, dtype=f32), z).astype(f32))
    y = np.cross(z, x).astype(f32)
    m = mat_identity()
    m[0, :3], m[1, :3], m[2, :3], m[3, :3] = x, y, z, np.asarray(position, dtype=f32)
    return m


def normalize3(v):
    v = np.asarray(v, dtype=f32)
    l = f32(np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))
    return (v / l).astype(f32)


def to_matrix(m):
    r = A.Matrix()
    flat = np.ascontiguousarray(m, dtype=f32).reshape(16)
    for i in range(16):
        r.m[i] = float(flat[i])
    return r


def reverse_z_infinite_perspective_fov_lh(fov, r, n):
    """Demo::Utils::GetReverseZInfinitePerspectiveFovLH (src/demo.cpp:94-102)."""
    t = f32(math.tan(f32(fov) / f32(2.0)))   # std::tan on float
    t = f32(np.tan(f32(f32(fov) / f32(2.0))))
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = f32(1.0) / (f32(r) * t)
    m[1, 1] = f32(1.0) / t
    m[2, 3] = f32(1.0)
    m[3, 2] = f32(n)
    return m


class FView:
    """FView (inc/renderer.h; src/demo.cpp:172-230)."""

    def __init__(self):
        self.position = np.array([0, 0, -15], dtype=f32)
        self.right = np.array([1, 0, 0], dtype=f32)
        self.up = np.array([0, 1, 0], dtype=f32)
        self.look = np.array([0, 0, 1], dtype=f32)
        self.fov = f32(A.FConfig.Fov)
        self.view_transform = mat_identity()
        self.projection_transform = mat_identity()

    def update_view_transform(self):
        self.look = normalize3(self.look)
        self.up = normalize3(np.cross(self.look, self.right).astype(f32))
        self.right = np.cross(self.up, self.look).astype(f32)
        p, r, u, l = self.position, self.right, self.up, self.look
        dot = lambda a, b: f32(f32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])
        t = (dot(p, r), dot(p, u), dot(p, l))
        m = mat_identity()
        m[0, 0], m[1, 0], m[2, 0], m[3, 0] = r[0], r[1], r[2], -t[0]
        m[0, 1], m[1, 1], m[2, 1], m[3, 1] = u[0], u[1], u[2], -t[1]
        m[0, 2], m[1, 2], m[2, 2], m[3, 2] = l[0], l[1], l[2], -t[2]
        self.view_transform = m

    def reset(self, scene=None, aspect=None):
        """FView::Reset: glTF camera 0 if present, else the default camera."""
        if scene is not None and scene.cameras:
            cam = scene.cameras[0]
            w = cam["transform"]
            self.position = w[3, :3].copy()
            self.right = w[0, :3].copy()
            self.up = w[1, :3].copy()
            self.look = (-w[2, :3]).astype(f32)          # SimpleMath Matrix::Forward() = -row 3
            self.update_view_transform()
            a = cam["aspect"] if aspect is None else aspect
            self.fov = f32(cam["yfov"])
            self.projection_transform = reverse_z_infinite_perspective_fov_lh(cam["yfov"], a, A.FConfig.CameraNearPlane)
        else:
            self.__init__()
            self.update_view_transform()
            self.projection_transform = reverse_z_infinite_perspective_fov_lh(self.fov, 16.0 / 9.0 if aspect is None else aspect, 1.0)
        return self

    def to_ctypes(self):
        v = A.View()
        v.view_transform = to_matrix(self.view_transform)
        v.projection_transform = to_matrix(self.projection_transform)
        for i in range(3):
            v.position[i] = float(self.position[i]); v.right[i] = float(self.right[i])
            v.up[i] = float(self.up[i]); v.look[i] = float(self.look[i])
        v.fov = float(self.fov)
        return v


def sun_direction_from_time_of_day(decimal_hours=A.FConfig.ToD_DecimalHours, julian=A.FConfig.ToD_JulianDate,
                                   latitude_deg=A.FConfig.ToD_Latitude):
    """FScene::UpdateSunDirection (src/scene.cpp:1594-1641), float32 like the reference."""
    XM_PI, XM_2PI = f32(3.141592654), f32(6.283185307)
    l = f32(f32(latitude_deg) * f32(XM_PI / f32(180.0)))
    delta = f32(0.4093) * f32(np.sin(f32(XM_2PI * f32(f32(julian) - f32(81.0)) / f32(368.0))))
    sin_l, cos_l = f32(np.sin(l)), f32(np.cos(l))
    sin_d, cos_d = f32(np.sin(delta)), f32(np.cos(delta))
    t = f32(XM_PI * f32(decimal_hours) / f32(12.0))
    sin_t, cos_t = f32(np.sin(t)), f32(np.cos(t))
    theta = f32(f32(0.5) * XM_PI - f32(np.arcsin(f32(sin_l * sin_d - cos_l * cos_d * cos_t))))
    phi = f32(np.arctan(f32(-cos_d * sin_t / f32(cos_l * sin_d - sin_l * cos_d * cos_t))))
    st, ct, sp, cp = f32(np.sin(theta)), f32(np.cos(theta)), f32(np.sin(phi)), f32(np.cos(phi))
    return normalize3(np.array([st * cp, ct, st * sp], dtype=f32))


# ------------------------------------------------------------------ packed scene container
class PackedScene:
    """Owns the numpy arrays behind an `incr_scene_desc` (keeps them alive for ctypes)."""

    def __init__(self):
        self.buffers = []            # list of np.uint8 arrays (raw glTF buffers)
        self.buffer_views = []       # (buffer, byteOffset, byteLength)
        self.accessors = []          # (bufferView, byteOffset, byteStride)
        self.mesh_names = []
        self.mesh_primitives = []    # per mesh: list of dict(index, position, uv, normal, tangent, material, index_count, sphere)
        self.mesh_transforms = []    # per mesh 4x4
        self.decal_primitives = []   # FScene::m_sceneMeshDecals flattened: list of (primitive dict, mesh 4x4), in draw order
        self.mesh_visibility = None
        self.materials = []          # list of A.Material
        self.textures = []           # list of (np.uint8 HxWx4, format)
        self.global_lights = []      # list of A.Light
        self.light_indices = []      # scene lights -> global light
        self.light_transforms = []   # per scene light 4x4
        self.cameras = []
        self.root_transform = mat_identity()
        self.sun_dir = sun_direction_from_time_of_day()
        self.max_anisotropy = 0      # MaxAnisotropy of the raster passes' sampler: 0 = the reference's 16, 1 = isotropic trilinear
        self.meshlets = None         # filled by meshlet generation: (GpuMeshlet array, vertex indices, packed triangles)
        self._keep = []
        self._desc = None

    # -- counts
    @property
    def num_triangles(self):
        return sum(p["index_count"] // 3 for prims in self.mesh_primitives for p in prims)

    def add_texture(self, rgba8, fmt):
        self.textures.append((np.ascontiguousarray(rgba8, dtype=np.uint8), int(fmt)))
        return len(self.textures) - 1

    def add_texture_bc(self, image):
        """A block-compressed mip chain (bc.BcImage: what the reference uploads after DirectX::Compress) in place of RGBA8 texels."""
        self.textures.append((image, int(image.format)))
        return len(self.textures) - 1

    def compress_textures(self):
        """Replace every RGBA8 texture by a block-compressed chain of the format the reference picks for its slot (scene.cpp:902-904,
        932-933, 1162-1163): sRGB -> BC3_UNORM_SRGB, normals -> BC5_SNORM, metallic-roughness -> BC5_UNORM, the rest -> BC4_UNORM.  The
        encoder is bc.py's range fit, not DirectXTex's: this produces VALID compressed input, not the reference's bytes."""
        from . import bc
        to_bc = {A.TEX_RGBA8_SRGB: A.TEX_BC3_SRGB, A.TEX_RG_SNORM: A.TEX_BC5_SNORM, A.TEX_BGRA8_SWIZZLE_RG: A.TEX_BC5_UNORM, A.TEX_RGBA8_UNORM: A.TEX_BC4_UNORM}
        self.textures = [(bc.compress(img, to_bc[fmt]), to_bc[fmt]) if fmt in to_bc else (img, fmt) for img, fmt in self.textures]
        self._desc = None
        return self

    def accessor_array(self, accessor_index, dtype, ncomp, count):
        """Read `count` elements through the accessor indirection (host-side helper)."""
        view_i, a_off, stride = self.accessors[accessor_index]
        buf_i, v_off, _ = self.buffer_views[view_i]
        raw = self.buffers[buf_i]
        base = v_off + a_off
        item = np.dtype(dtype).itemsize * ncomp
        out = np.empty((count, ncomp), dtype=dtype)
        for k in range(ncomp):
            out[:, k] = np.ndarray((count,), dtype=dtype, buffer=raw.data, offset=base + k * np.dtype(dtype).itemsize, strides=(stride,))
        assert stride >= item
        return out

    def finalize_sun_light(self):
        """FScene::UpdateSunDirection: a directional light follows the time-of-day sun (scene.cpp:1636-1641)."""
        for si, gi in enumerate(self.light_indices):
            if self.global_lights[gi].type == A.LIGHT_DIRECTIONAL:
                # the reference indexes m_transformList with the GLOBAL light index returned by GetDirectionalLight
                idx = gi if gi < len(self.light_transforms) else si
                self.light_transforms[idx] = mat_create_world((0, 0, 0), self.sun_dir, (0, 1, 0))
                break

    def to_desc(self):
        d = A.SceneDesc()
        keep = self._keep = []

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            keep.append(a)
            return a

        d.num_buffers = len(self.buffers)
        ptrs = arr(C.c_void_p, [b.ctypes.data for b in self.buffers])
        d.buffers = C.cast(ptrs, C.POINTER(C.c_void_p))
        d.buffer_sizes = C.cast(arr(C.c_uint64, [b.nbytes for b in self.buffers]), C.POINTER(C.c_uint64))
        d.num_buffer_views = len(self.buffer_views)
        d.buffer_views = C.cast(arr(A.MeshBufferView, [A.MeshBufferView(*v) for v in self.buffer_views]), C.POINTER(A.MeshBufferView))
        d.num_accessors = len(self.accessors)
        d.accessors = C.cast(arr(A.MeshAccessor, [A.MeshAccessor(*a) for a in self.accessors]), C.POINTER(A.MeshAccessor))
        d.num_meshes = len(self.mesh_primitives)
        d.primitive_counts = C.cast(arr(C.c_uint32, [len(p) for p in self.mesh_primitives]), C.POINTER(C.c_uint32))
        d.mesh_transforms = C.cast(arr(A.Matrix, [to_matrix(m) for m in self.mesh_transforms]), C.POINTER(A.Matrix))
        if self.mesh_visibility is not None:
            d.mesh_visibility = C.cast(arr(C.c_uint32, [int(v) for v in self.mesh_visibility]), C.POINTER(C.c_uint32))
        prims = []
        for mesh_index, plist in enumerate(self.mesh_primitives):
            for p in plist:
                g = A.GpuPrimitive()
                for i in range(4):
                    g.boundingSphere[i] = float(p["sphere"][i])
                g.meshIndex = mesh_index
                g.indexAccessor, g.positionAccessor = p["index"], p["position"]
                g.uvAccessor, g.normalAccessor, g.tangentAccessor = p["uv"], p["normal"], p["tangent"]
                g.materialIndex, g.indicesPerTriangle, g.indexCount = p["material"], 3, p["index_count"]
                prims.append(g)
        d.num_primitives = len(prims)
        d.primitives = C.cast(arr(A.GpuPrimitive, prims), C.POINTER(A.GpuPrimitive))
        d.num_materials = len(self.materials)
        d.materials = C.cast(arr(A.Material, self.materials), C.POINTER(A.Material))
        texs = []
        for img, fmt in self.textures:
            t = A.TextureDesc()
            if hasattr(img, "blocks"):           # bc.BcImage: the compressed chain goes over as it is
                t.texels = img.blocks.ctypes.data
                t.mip_levels = img.mip_levels
            else:
                t.texels = img.ctypes.data
            t.height, t.width = img.shape[0], img.shape[1]
            t.format = fmt
            texs.append(t)
        d.num_textures = len(texs)
        d.textures = C.cast(arr(A.TextureDesc, texs), C.POINTER(A.TextureDesc))
        if self.meshlets is not None:
            ml, vi, tri = self.meshlets
            keep.extend([ml, vi, tri])
            d.num_meshlets = len(ml)
            d.meshlets = C.cast(ml, C.POINTER(A.GpuMeshlet))
            d.num_meshlet_vertex_indices = vi.size
            d.meshlet_vertex_indices = C.cast(vi.ctypes.data, C.POINTER(C.c_uint32))
            d.num_meshlet_triangles = tri.size
            d.meshlet_packed_triangles = C.cast(tri.ctypes.data, C.POINTER(C.c_uint32))
        d.num_global_lights = len(self.global_lights)
        d.global_lights = C.cast(arr(A.Light, self.global_lights), C.POINTER(A.Light))
        d.num_scene_lights = len(self.light_indices)
        d.light_indices = C.cast(arr(C.c_int32, list(self.light_indices)), C.POINTER(C.c_int32))
        d.light_transforms = C.cast(arr(A.Matrix, [to_matrix(m) for m in self.light_transforms]), C.POINTER(A.Matrix))
        d.root_transform = to_matrix(self.root_transform)
        for i in range(3):
            d.sun_dir[i] = float(self.sun_dir[i])
        d.max_anisotropy = int(self.max_anisotropy)
        if self.decal_primitives:
            dps = []
            for p, _ in self.decal_primitives:
                g = A.GpuPrimitive()
                for i in range(4):
                    g.boundingSphere[i] = float(p["sphere"][i])
                g.meshIndex = 0
                g.indexAccessor, g.positionAccessor = p["index"], p["position"]
                g.uvAccessor, g.normalAccessor, g.tangentAccessor = p["uv"], p["normal"], p["tangent"]
                g.materialIndex, g.indicesPerTriangle, g.indexCount = p["material"], 3, p["index_count"]
                dps.append(g)
            d.num_decal_primitives = len(dps)
            d.decal_primitives = C.cast(arr(A.GpuPrimitive, dps), C.POINTER(A.GpuPrimitive))
            d.decal_transforms = C.cast(arr(A.Matrix, [to_matrix(m) for _, m in self.decal_primitives]), C.POINTER(A.Matrix))
        d._keep = keep          # whoever holds the descriptor holds the arrays its pointers name (a later to_desc() makes new ones)
        self._desc = d
        return d


# ------------------------------------------------------------------ glTF
_COMPONENT_SIZE = {5120: 1, 5121: 1, 5122: 2, 5123: 2, 5125: 4, 5126: 4}
_TYPE_COUNT = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


def _bounding_sphere(pos):
    """BoundingBox::CreateFromPoints -> BoundingSphere::CreateFromBoundingBox (scene.cpp:284-298,364)."""
    lo, hi = pos.min(axis=0).astype(f32), pos.max(axis=0).astype(f32)
    c = ((lo + hi) * f32(0.5)).astype(f32)
    e = ((hi - lo) * f32(0.5)).astype(f32)
    r = f32(np.sqrt(f32(f32(e[0] * e[0] + e[1] * e[1]) + e[2] * e[2])))
    return np.array([c[0], c[1], c[2], r], dtype=f32)


def decode_jpeg(data):
    """JPEG bytes -> RGBA8 array through incr_decode_jpeg: stb_image's published integer pipeline (what tinygltf::LoadImageData runs, scene.cpp:40, 98-117)."""
    import ctypes as C
    from . import runtime as _rt
    L = _rt.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h = C.c_int32(0), C.c_int32(0)
    _rt.check(L.incr_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), np.uint8)
    _rt.check(L.incr_decode_jpeg(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), out.ctypes.data, out.size))
    return out


def load_gltf(path, load_images=True, max_texture_size=None, ignore_tangents=False, use_content_cache=False, texture_cache_dir=None, jpeg_decoder="stb"):
    """glTF 2.0 loader -> PackedScene: .gltf (external or `data:` buffers and images) and binary .glb (embedded BIN chunk,
    bufferView images) -- the two containers tinygltf's LoadASCIIFromFile / LoadBinaryFromFile accept (scene.cpp:113-130)."""
    import base64
    import io
    import struct
    from PIL import Image

    base = os.path.dirname(os.path.abspath(path))
    with open(path, "rb") as f:
        blob = f.read()
    glb_bin = None
    if blob[:4] == b"glTF":                                   # GLB: 12-byte header, then (length, type, payload) chunks
        magic, version, total = struct.unpack_from("<4sII", blob, 0)
        if version != 2 or total > len(blob):
            raise ValueError("%s: not a glTF 2 binary container" % path)
        off, g = 12, None
        while off + 8 <= total:
            clen, ctype = struct.unpack_from("<II", blob, off)
            chunk = blob[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A and g is None:             # 'JSON'
                g = json.loads(chunk.decode("utf-8"))
            elif ctype == 0x004E4942 and glb_bin is None:     # 'BIN\0'
                glb_bin = chunk
            off += 8 + ((clen + 3) & ~3)
        if g is None:
            raise ValueError("%s: GLB without a JSON chunk" % path)
    else:
        g = json.loads(blob.decode("utf-8"))

    def uri_bytes(uri):
        if uri.startswith("data:"):
            return base64.b64decode(uri.split(",", 1)[1])
        with open(os.path.join(base, uri), "rb") as f:
            return f.read()

    s = PackedScene()
    for b in g.get("buffers", []):
        if "uri" in b:
            raw = np.frombuffer(uri_bytes(b["uri"]), dtype=np.uint8)[:b["byteLength"]].copy()
        else:
            if glb_bin is None:
                raise ValueError("%s: buffer without uri outside a GLB" % path)
            raw = np.frombuffer(glb_bin, dtype=np.uint8)[:b["byteLength"]].copy()
        pad = (-raw.size) % 16 + 16          # slack so dword-aligned 16-bit index unpack never reads past the end
        s.buffers.append(np.concatenate([raw, np.zeros(pad, dtype=np.uint8)]))
    for v in g.get("bufferViews", []):
        s.buffer_views.append((v["buffer"], v.get("byteOffset", 0), v["byteLength"]))
    for a in g.get("accessors", []):
        view = g["bufferViews"][a["bufferView"]]
        stride = view.get("byteStride", 0) or _COMPONENT_SIZE[a["componentType"]] * _TYPE_COUNT[a["type"]]
        s.accessors.append((a["bufferView"], a.get("byteOffset", 0), stride))

    # materials (FScene::LoadMaterial)
    image_cache = {}
    content_cache_stats = {"hits": 0, "misses": 0}

    def image_rgba(image_index):
        if image_index not in image_cache:
            im = g["images"][image_index]
            if "uri" in im:
                data = uri_bytes(im["uri"])
            else:                                              # embedded: the encoded file sits in a bufferView
                bi, bo, bl = s.buffer_views[im["bufferView"]]
                data = s.buffers[bi][bo:bo + bl].tobytes()
            # Content cache (scene.cpp:13-89 LoadImageCallback / GetContentCachePath, FConfig::UseContentCache): the reference keeps the
            # processed image next to the model under .content-cache/ and skips the decode when it is there.  Here the cached form is
            # the decoded RGBA8 array (.npy), keyed by the encoded bytes' hash and the size limit -- a stale entry cannot be picked up.
            cached = None
            if use_content_cache:
                import hashlib
                cdir = os.path.join(os.path.dirname(os.path.abspath(path)), ".content-cache")
                stem = os.path.splitext(os.path.basename(im["uri"]))[0] if "uri" in im and not im["uri"].startswith("data:") else "image%d" % image_index
                cached = os.path.join(cdir, "%s.%s.%s%s.npy" % (stem, hashlib.sha256(data).hexdigest()[:16], max_texture_size or "full", "" if jpeg_decoder == "stb" else "." + jpeg_decoder))
                if os.path.exists(cached):
                    image_cache[image_index] = np.load(cached)
                    content_cache_stats["hits"] += 1
                    return image_cache[image_index]
            if data[:2] == b"\xff\xd8" and jpeg_decoder == "stb":
                img = Image.fromarray(decode_jpeg(data), "RGBA")      # stb_image's integer pipeline (csrc/jpeg_decode.cpp): the reference's texel bytes
            else:
                img = Image.open(io.BytesIO(data)).convert("RGBA")    # "Source Images are always 4 channel 8bpp" (scene.cpp:1167); PNG is lossless
            if max_texture_size and max(img.size) > max_texture_size:
                img = img.resize((max_texture_size, max_texture_size), Image.BOX)
            image_cache[image_index] = np.asarray(img, dtype=np.uint8).copy()
            if cached:
                os.makedirs(os.path.dirname(cached), exist_ok=True)
                tmp = cached + ".tmp%d" % os.getpid()
                with open(tmp, "wb") as f:
                    np.save(f, image_cache[image_index])
                os.replace(tmp, cached)
                content_cache_stats["misses"] += 1
        return image_cache[image_index]

    tex_cache = {}

    def tex(texture_ref, fmt):
        if texture_ref is None or not load_images:
            return -1
        src = g["textures"][texture_ref["index"]]["source"]
        key = (src, fmt)
        if key not in tex_cache:
            # The reference's texture cache (scene.cpp:1046-1070, 1103-1115: `<cache>/<image stem>.dds`, the block-compressed mip chain
            # DirectX::Compress wrote): when such a file exists and holds the format this slot compresses to, ITS blocks are handed over
            # -- the texels the reference's GPU samples -- instead of the decoded image.
            uri = g["images"][src].get("uri", "")
            dds = os.path.join(texture_cache_dir, os.path.splitext(os.path.basename(uri))[0] + ".dds") if texture_cache_dir and uri and not uri.startswith("data:") else None
            if dds and os.path.exists(dds):
                from . import bc
                image = bc.read_dds(dds)
                slot_format = {A.TEX_RGBA8_SRGB: A.TEX_BC3_SRGB, A.TEX_RG_SNORM: A.TEX_BC5_SNORM, A.TEX_BGRA8_SWIZZLE_RG: A.TEX_BC5_UNORM, A.TEX_RGBA8_UNORM: A.TEX_BC4_UNORM}[fmt]
                if image.format == slot_format:
                    tex_cache[key] = s.add_texture_bc(image)
                    return tex_cache[key]
            tex_cache[key] = s.add_texture(image_rgba(src), fmt)
        return tex_cache[key]

    for m in g.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        mat = A.Material()
        ef = m.get("emissiveFactor", [0, 0, 0]); bc = pbr.get("baseColorFactor", [1, 1, 1, 1])
        for i in range(3):
            mat.emissiveFactor[i] = ef[i]; mat.baseColorFactor[i] = bc[i]
        mat.metallicFactor = pbr.get("metallicFactor", 1.0)
        mat.roughnessFactor = pbr.get("roughnessFactor", 1.0)
        occ = m.get("occlusionTexture")
        mat.aoStrength = occ.get("strength", 1.0) if occ else 1.0
        for n in A._MAT_TEX:
            setattr(mat, n + "TextureIndex", -1); setattr(mat, n + "SamplerIndex", -1)
        mat.emissiveTextureIndex = tex(m.get("emissiveTexture"), A.TEX_RGBA8_SRGB)
        mat.baseColorTextureIndex = tex(pbr.get("baseColorTexture"), A.TEX_RGBA8_SRGB)
        mat.aoTextureIndex = tex(occ, A.TEX_RGBA8_UNORM)
        mat.metallicRoughnessTextureIndex = tex(pbr.get("metallicRoughnessTexture"), A.TEX_BGRA8_SWIZZLE_RG)
        mat.normalTextureIndex = tex(m.get("normalTexture"), A.TEX_RG_SNORM)
        ext = m.get("extensions", {})
        # KHR_materials_transmission / KHR_materials_clearcoat (scene.cpp:939-1001): factors default to 0, the single-channel maps go to the
        # BC4 slots (source format R8G8B8A8_UNORM), clearcoat roughness / normal to the BC5 slots of their base-material counterparts
        tr = ext.get("KHR_materials_transmission", {}); cc = ext.get("KHR_materials_clearcoat", {})
        mat.transmissionFactor = tr.get("transmissionFactor", 0.0)
        mat.transmissionTextureIndex = tex(tr.get("transmissionTexture"), A.TEX_RGBA8_UNORM)
        mat.clearcoatFactor = cc.get("clearcoatFactor", 0.0)
        mat.clearcoatRoughnessFactor = cc.get("clearcoatRoughnessFactor", 0.0)
        mat.clearcoatTextureIndex = tex(cc.get("clearcoatTexture"), A.TEX_RGBA8_UNORM)
        mat.clearcoatRoughnessTextureIndex = tex(cc.get("clearcoatRoughnessTexture"), A.TEX_BGRA8_SWIZZLE_RG)
        mat.clearcoatNormalTextureIndex = tex(cc.get("clearcoatNormalTexture"), A.TEX_RG_SNORM)
        mat.alphaMode = {"OPAQUE": A.ALPHA_OPAQUE, "MASK": A.ALPHA_MASKED, "BLEND": A.ALPHA_BLEND}[m.get("alphaMode", "OPAQUE")]
        mat.doubleSided = 1 if m.get("doubleSided", False) else 0
        s.materials.append(mat)

    # lights (FScene::LoadLights)
    for L in g.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", []):
        l = A.Light()
        col = L.get("color", [1, 1, 1])
        for i in range(3):
            l.color[i] = col[i]
        inten = L.get("intensity", 1.0)
        l.intensity = inten if inten > 0 else 150
        l.range = L.get("range", 0.0)
        spot = L.get("spot", {})
        l.spotAngles[0] = spot.get("innerConeAngle", 0.0); l.spotAngles[1] = spot.get("outerConeAngle", math.pi / 4)
        l.type = {"directional": A.LIGHT_DIRECTIONAL, "point": A.LIGHT_POINT, "spot": A.LIGHT_SPOT}[L["type"]]
        s.global_lights.append(l)

    RH2LH = mat_identity(); RH2LH[2, 2] = f32(-1)

    # MeshUtils::FixupMeshes (mesh-utils.cpp:560-638): a primitive whose material has a normal map but which ships no TANGENT gets
    # MikkTSpace tangents (incr_generate_tangents, csrc/mikktspace_gen.cpp) in a new buffer / view / accessor, as the reference appends
    # them to the model.  ignore_tangents: treat every primitive as if it shipped none (tests).
    generated = {}

    def fixup_tangents(mesh_index, prim_index, p, at):
        key = (mesh_index, prim_index)
        if key in generated:
            return generated[key]
        acc = -1 if ignore_tangents else at.get("TANGENT", -1)
        mat_i = p.get("material", 0)
        needs = acc == -1 and mat_i < len(g.get("materials", [])) and "normalTexture" in g["materials"][mat_i]      # material.normalTexture.index != -1 (:576)
        if needs and "NORMAL" in at and "TEXCOORD_0" in at and "indices" in p:
            from . import runtime as _rt
            nv = int(g["accessors"][at["POSITION"]]["count"])
            pos = np.ascontiguousarray(s.accessor_array(at["POSITION"], np.float32, 3, nv))
            nrm = np.ascontiguousarray(s.accessor_array(at["NORMAL"], np.float32, 3, nv))
            uv = np.ascontiguousarray(s.accessor_array(at["TEXCOORD_0"], np.float32, 2, nv))
            ia = g["accessors"][p["indices"]]
            idt = {5121: np.uint8, 5123: np.uint16, 5125: np.uint32}[ia["componentType"]]
            idx = np.ascontiguousarray(s.accessor_array(p["indices"], idt, 1, ia["count"]).astype(np.uint32).reshape(-1, 3))
            out = np.zeros((nv, 4), np.float32)
            _rt.check(_rt.load().incr_generate_tangents(pos.ctypes.data, nrm.ctypes.data, uv.ctypes.data, idx.ctypes.data, nv, idx.shape[0], out.ctypes.data))
            s.buffers.append(np.concatenate([out.view(np.uint8).reshape(-1), np.zeros(32, np.uint8)]))
            s.buffer_views.append((len(s.buffers) - 1, 0, nv * 16))
            s.accessors.append((len(s.buffer_views) - 1, 0, 16))
            acc = len(s.accessors) - 1
        generated[key] = acc
        return acc

    def load_node(ni, parent):
        n = g["nodes"][ni]
        t = mat_identity()
        if "matrix" in n:
            t = np.array(n["matrix"], dtype=f32).reshape(4, 4)
        elif any(k in n for k in ("translation", "rotation", "scale")):
            T = mat_translation(*n["translation"]) if "translation" in n else mat_identity()
            R = mat_from_quaternion(*n["rotation"]) if "rotation" in n else mat_identity()
            S = mat_scale(*n["scale"]) if "scale" in n else mat_identity()
            t = mat_mul(mat_mul(S, R), T)
        world = mat_mul(t, parent)
        if "camera" in n:
            cam = g["cameras"][n["camera"]]
            if cam["type"] == "perspective":
                p = cam["perspective"]
                s.cameras.append({"transform": world, "yfov": p["yfov"], "aspect": p.get("aspectRatio", 16 / 9)})
        if "mesh" in n:
            mesh = g["meshes"][n["mesh"]]
            prims = []
            for pi, p in enumerate(mesh["primitives"]):
                at = p["attributes"]
                pos_acc = g["accessors"][at["POSITION"]]
                pos = s.accessor_array(at["POSITION"], np.float32, 3, pos_acc["count"])
                prims.append(dict(index=p["indices"], position=at["POSITION"], uv=at.get("TEXCOORD_0", -1),
                                  normal=at.get("NORMAL", -1), tangent=fixup_tangents(n["mesh"], pi, p, at), material=p.get("material", 0),
                                  index_count=g["accessors"][p["indices"]]["count"], vertex_count=int(pos_acc["count"]), sphere=_bounding_sphere(pos)))
            if mesh.get("name", "").startswith("decal"):          # scene.cpp:281: decal meshes live in their own entity list
                s.decal_primitives.extend((p, world) for p in prims)
            else:
                s.mesh_names.append(mesh.get("name", ""))
                s.mesh_primitives.append(prims)
                s.mesh_transforms.append(world)
        light_ext = n.get("extensions", {}).get("KHR_lights_punctual")
        if light_ext is not None:
            s.light_indices.append(light_ext["light"])
            s.light_transforms.append(world)
        for c in n.get("children", []):
            load_node(c, world)

    for sc in g.get("scenes", []):
        for ni in sc["nodes"]:
            load_node(ni, RH2LH)
    s.finalize_sun_light()
    s.content_cache_stats = content_cache_stats        # hits / misses of this load (use_content_cache)
    return s
