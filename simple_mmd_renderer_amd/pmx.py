"""PMX 2.0 files: a writer for synthetic fixtures and the Python face of the C-ABI loader.

The reference ships no .pmx asset, so fixtures are written here from a `FlatModel` and read back by
(a) this repo's from-scratch parser (csrc/pmx.cpp, `load_pmx`) and (b) the reference's own
`PmxReader` through oracle/ref_harness.cpp -- a write -> read round trip that pins the loader
against libmmd (tests/test_pmx_loader.py).  File layout per the PMX 2.0 specification, as the
reference's reader consumes it (L/reader/pmx_reader_impl.inl:16-449).
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _capi as api
from .synth import BDEF1, BDEF2, BDEF4, SDEF, FlatModel

# bone flag bits (PMX 2.0)
BONE_CHILD_USE_ID, BONE_ROTATABLE, BONE_MOVABLE, BONE_VISIBLE, BONE_CONTROLLABLE, BONE_HAS_IK = \
    0x0001, 0x0002, 0x0004, 0x0008, 0x0010, 0x0020
BONE_APPEND_ROTATE, BONE_APPEND_TRANSLATE, BONE_AXIS_FIXED, BONE_LOCAL_AXIS, BONE_POST_PHYSICS, \
    BONE_RECEIVE_TRANSFORM = 0x0100, 0x0200, 0x0400, 0x0800, 0x1000, 0x2000


@dataclass
class PmxWriteOptions:
    utf8: bool = False                 # text encoding (False = UTF-16LE, what MMD tools write)
    extra_uv: int = 0
    index_width: Tuple[int, int, int, int, int, int] = (0, 1, 1, 0, 0, 1)   # 0 = smallest that fits
    n_textures: int = 2
    n_materials: int = 3
    bone_flag_variety: bool = True     # exercise every optional bone field (IK, append, axes, ...)
    display_frames: int = 2            # sections AFTER the morphs (our parser stops before them)
    rigid_bodies: int = 1
    triangles: Optional[np.ndarray] = None
    bone_names: Optional[List[str]] = None
    morph_names: Optional[List[str]] = None
    rig: Optional[tuple] = None        # synth.make_ik_rig(...) output: levels, flag bits, append and IK records
    morph_rotation: Optional[np.ndarray] = None   # f32 [E,4]: rotation of bone-morph entries (default identity)
    model_name: str = "合成モデル synthetic"


def _fit(width: int, count: int, signed: bool) -> int:
    if width:
        return width
    if signed:      # "none" = -1 must stay representable
        return 1 if count <= 127 else (2 if count <= 32767 else 4)
    return 1 if count <= 255 else (2 if count <= 65535 else 4)


def write_pmx(m: FlatModel, opt: Optional[PmxWriteOptions] = None) -> bytes:
    opt = opt or PmxWriteOptions()
    nv, nb, nm = m.nv, m.nb, m.nm
    w_vertex = _fit(opt.index_width[0], nv, False)
    w_texture = _fit(opt.index_width[1], opt.n_textures, True)
    w_material = _fit(opt.index_width[2], opt.n_materials, True)
    w_bone = _fit(opt.index_width[3], nb, True)
    w_morph = _fit(opt.index_width[4], nm, True)
    w_rigid = _fit(opt.index_width[5], opt.rigid_bodies, True)
    out = bytearray()

    def text(s: str):
        b = s.encode("utf-8" if opt.utf8 else "utf-16-le")
        out.extend(struct.pack("<i", len(b)))
        out.extend(b)

    def index(v: int, width: int, unsigned: bool = False):
        v = int(v)
        if width == 1:
            out.extend(struct.pack("<B", v & 0xFF))
        elif width == 2:
            out.extend(struct.pack("<H", v & 0xFFFF))
        else:
            out.extend(struct.pack("<i", v if v < 2 ** 31 else v - 2 ** 32))

    f = lambda *v: out.extend(struct.pack("<%df" % len(v), *[float(x) for x in v]))  # noqa: E731

    out.extend(b"PMX ")
    f(2.0)
    out.extend(bytes([8, 1 if opt.utf8 else 0, opt.extra_uv, w_vertex, w_texture, w_material, w_bone,
                      w_morph, w_rigid]))
    text(opt.model_name); text("synthetic"); text("made by simple_mmd_renderer_amd.pmx.write_pmx"); text("")

    # vertices
    out.extend(struct.pack("<i", nv))
    rng = np.random.RandomState(12345)
    for i in range(nv):
        out.extend(np.asarray(m.positions[i], "<f4").tobytes())
        out.extend(np.asarray(m.normals[i], "<f4").tobytes())
        out.extend(np.asarray(m.uvs[i], "<f4").tobytes())
        for _ in range(opt.extra_uv):
            f(*rng.uniform(0, 1, 4))
        t = int(m.skin_type[i])
        ids, w = m.bone_ids[i], m.bone_weights[i]
        if t not in (BDEF1, BDEF2, BDEF4, SDEF):
            raise ValueError("PMX 2.0 has no deform type %d" % t)
        out.extend(struct.pack("<b", t))
        if t == BDEF1:
            index(ids[0], w_bone)
        elif t == BDEF2:
            index(ids[0], w_bone); index(ids[1], w_bone); out.extend(np.float32(w[0]).tobytes())
        elif t == BDEF4:
            for k in range(4):
                index(ids[k], w_bone)
            out.extend(np.asarray(w, "<f4").tobytes())
        else:
            index(ids[0], w_bone); index(ids[1], w_bone); out.extend(np.float32(w[0]).tobytes())
            sd = m.sdef[i] if m.sdef is not None else np.zeros(9, np.float32)
            out.extend(np.asarray(sd, "<f4").tobytes())
        f(1.0)      # edge scale

    # triangles
    tris = opt.triangles
    if tris is None:
        k = max(nv // 3, 0) * 3
        tris = np.arange(k, dtype=np.uint32)
    tris = np.asarray(tris, np.uint32).reshape(-1)
    out.extend(struct.pack("<i", tris.size))
    for t in tris:
        index(int(t), w_vertex, unsigned=True)

    # textures, materials
    out.extend(struct.pack("<i", opt.n_textures))
    for i in range(opt.n_textures):
        text("tex/テクスチャ%d.png" % i)
    out.extend(struct.pack("<i", opt.n_materials))
    per = (tris.size // 3 // max(opt.n_materials, 1)) * 3
    for i in range(opt.n_materials):
        text("材質%d" % i); text("material%d" % i)
        f(0.8, 0.7, 0.6, 1.0); f(0.2, 0.2, 0.2); f(5.0); f(0.4, 0.4, 0.4)
        out.extend(bytes([0x1F if i % 2 else 0x01]))
        f(0, 0, 0, 1); f(1.0)
        index(i % max(opt.n_textures, 1) if opt.n_textures else -1, w_texture)
        index(-1, w_texture)
        out.extend(bytes([0]))
        if i % 2:
            out.extend(bytes([1, 3]))                       # shared toon 3
        else:
            out.extend(bytes([0])); index(-1, w_texture)
        text("memo")
        cnt = per if i + 1 < opt.n_materials else tris.size - per * (opt.n_materials - 1)
        out.extend(struct.pack("<i", cnt))

    # bones
    out.extend(struct.pack("<i", nb))
    names_b = opt.bone_names or ["ボーン%d" % b for b in range(nb)]
    for b in range(nb):
        text(names_b[b]); text("bone%d" % b)
        out.extend(np.asarray(m.bone_pos[b], "<f4").tobytes())
        index(int(m.bone_parent[b]), w_bone)
        if opt.rig is not None:
            _, _, r_level, r_flags, r_ap, r_ar, r_ik = opt.rig
            out.extend(struct.pack("<i", int(r_level[b])))
            flags = BONE_ROTATABLE | BONE_VISIBLE | int(r_flags[b])
            out.extend(struct.pack("<H", flags))
            f(0.0, 1.0, 0.0)
            if flags & (BONE_APPEND_ROTATE | BONE_APPEND_TRANSLATE):
                index(int(r_ap[b]), w_bone); f(r_ar[b])
            if flags & BONE_HAS_IK:
                index(int(r_ik["target"][b]), w_bone)
                out.extend(struct.pack("<i", int(r_ik["loop"][b]))); f(r_ik["angle"][b])
                l0, l1 = int(r_ik["link_off"][b]), int(r_ik["link_off"][b + 1])
                out.extend(struct.pack("<i", l1 - l0))
                for l in range(l0, l1):
                    index(int(r_ik["link_bone"][l]), w_bone)
                    out.extend(struct.pack("<b", int(r_ik["link_limited"][l])))
                    if r_ik["link_limited"][l]:
                        f(*r_ik["link_lo"][l]); f(*r_ik["link_hi"][l])
            continue
        out.extend(struct.pack("<i", 0))
        flags = BONE_ROTATABLE | BONE_VISIBLE | BONE_CONTROLLABLE
        if opt.bone_flag_variety:
            flags |= [0, BONE_CHILD_USE_ID, BONE_MOVABLE, BONE_AXIS_FIXED, BONE_LOCAL_AXIS,
                      BONE_APPEND_ROTATE, BONE_APPEND_TRANSLATE | BONE_APPEND_ROTATE,
                      BONE_RECEIVE_TRANSFORM][b % 8]
            # no IK in fixtures whose palette comes from libmmd's own bone solve (keeps it FK);
            # the IK record layout is exercised on the last bone only, flagged but with 0 iterations
            if b == nb - 1 and nb > 2:
                flags |= BONE_HAS_IK
        out.extend(struct.pack("<H", flags))
        if flags & BONE_CHILD_USE_ID:
            index(min(b + 1, nb - 1), w_bone)
        else:
            f(0.0, 1.0, 0.0)
        if flags & (BONE_APPEND_ROTATE | BONE_APPEND_TRANSLATE):
            index(-1, w_bone); f(0.5)                        # append parent "none": ignored by the solve
        if flags & BONE_AXIS_FIXED:
            f(0, 1, 0)
        if flags & BONE_LOCAL_AXIS:
            f(1, 0, 0); f(0, 0, 1)
        if flags & BONE_RECEIVE_TRANSFORM:
            out.extend(struct.pack("<i", 7))
        if flags & BONE_HAS_IK:
            index(0, w_bone); out.extend(struct.pack("<i", 0)); f(1.0)
            out.extend(struct.pack("<i", 2))
            index(1 % nb, w_bone); out.extend(struct.pack("<b", 1)); f(-1, -1, -1); f(1, 1, 1)
            index(0, w_bone); out.extend(struct.pack("<b", 0))

    # morphs
    out.extend(struct.pack("<i", nm))
    names_m = opt.morph_names or ["モーフ%d" % k for k in range(nm)]
    for k in range(nm):
        text(names_m[k]); text("morph%d" % k)
        t = int(m.morph_type[k])
        lo, hi = int(m.morph_off[k]), int(m.morph_off[k + 1])
        out.extend(bytes([1 + k % 4, t]))
        out.extend(struct.pack("<i", hi - lo))
        for e in range(lo, hi):
            idx, v = int(m.morph_index[e]), m.morph_value[e]
            if t == 0:
                index(idx, w_morph); f(v[0])
            elif t == 1:
                index(idx, w_vertex, unsigned=True); out.extend(np.asarray(v, "<f4").tobytes())
            elif t == 2:
                index(idx, w_bone); out.extend(np.asarray(v, "<f4").tobytes())
                f(*(opt.morph_rotation[e] if opt.morph_rotation is not None else (0, 0, 0, 1)))
            elif 3 <= t <= 7:
                index(idx, w_vertex, unsigned=True); out.extend(np.asarray(v, "<f4").tobytes()); f(0.0)
            elif t == 8:
                index(idx, w_material); out.extend(bytes([0])); out.extend(b"\0" * 112)
            else:
                raise ValueError("unknown morph type %d" % t)

    # sections behind the morphs: display frames, rigid bodies, joints
    out.extend(struct.pack("<i", opt.display_frames))
    for d in range(opt.display_frames):
        text("枠%d" % d); text("frame%d" % d); out.extend(bytes([1 if d == 0 else 0]))
        items = [(0, 0)] + ([(1, 0)] if nm else [])
        out.extend(struct.pack("<i", len(items)))
        for is_morph, idx in items:
            out.extend(bytes([is_morph])); index(idx, w_morph if is_morph else w_bone)
    out.extend(struct.pack("<i", opt.rigid_bodies))
    for r in range(opt.rigid_bodies):
        text("剛体%d" % r); text("rigid%d" % r); index(0, w_bone)
        out.extend(struct.pack("<BHB", 0, 0xFFFF, 0)); f(1, 1, 1); f(0, 0, 0); f(0, 0, 0)
        f(1.0, 0.5, 0.5, 0.0, 0.5); out.extend(bytes([0]))
    out.extend(struct.pack("<i", 0))        # joints
    return bytes(out)


@dataclass
class PmxModel:
    flat: FlatModel
    info: dict
    triangles: np.ndarray
    material_index_count: np.ndarray
    bone_transform_level: np.ndarray
    bone_flags: np.ndarray
    name: str
    bone_names: List[str] = field(default_factory=list)
    morph_names: List[str] = field(default_factory=list)
    append_parent: Optional[np.ndarray] = None
    append_ratio: Optional[np.ndarray] = None
    ik: Optional[dict] = None          # target, loop, angle, link_off, link_bone, link_limited, link_lo, link_hi
    morph_rotation: Optional[np.ndarray] = None   # f32 [E,4], identity except for bone-morph entries

    def skeleton(self):
        """The device bone solver for this model's rig (vmd.Skeleton)."""
        from .vmd import Skeleton
        f = self.flat
        morphs = dict(type=f.morph_type, offset=f.morph_off, index=f.morph_index, value=f.morph_value,
                      rotation=self.morph_rotation) if f.nm else None
        return Skeleton(f.bone_pos, f.bone_parent, self.bone_transform_level, self.bone_flags,
                        self.append_parent, self.append_ratio, self.ik, morphs)


class PmxInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_vertices", C.c_uint32), ("n_indices", C.c_uint32),
                ("n_textures", C.c_uint32), ("n_materials", C.c_uint32), ("n_bones", C.c_uint32),
                ("n_morphs", C.c_uint32), ("n_morph_entries", C.c_uint32), ("extra_uv", C.c_uint32),
                ("utf8", C.c_uint32), ("index_width", C.c_uint8 * 8), ("bytes_consumed", C.c_uint64)]


class PmxArrays(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
                ("triangles", C.POINTER(C.c_uint32)), ("material_index_count", C.POINTER(C.c_uint32)),
                ("bone_rest_position", C.POINTER(C.c_float)), ("bone_parent", C.POINTER(C.c_int32)),
                ("bone_transform_level", C.POINTER(C.c_int32)), ("bone_flags", C.POINTER(C.c_uint16)),
                ("morph_panel", C.POINTER(C.c_uint8)), ("edge_scale", C.POINTER(C.c_float))]


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def load_pmd(source) -> PmxModel:
    """Parse a PMD 1.0 file (path or bytes) with the C-ABI loader (csrc/pmd.cpp); same result type."""
    return load_pmx(source, fmt="pmd")


def load_pmx(source, fmt: str = "pmx") -> PmxModel:
    """Parse a PMX 2.0 (or, fmt="pmd", a PMD 1.0) file (path or bytes) with the C-ABI loader."""
    lib = api.lib()
    h = C.c_void_p()
    parse, load = (lib.mmdx_pmx_parse, lib.mmdx_pmx_load_file) if fmt == "pmx" else (lib.mmdx_pmd_parse, lib.mmdx_pmd_load_file)
    if isinstance(source, (bytes, bytearray)):
        buf = (C.c_char * len(source)).from_buffer_copy(bytes(source))
        api.check(parse(buf, len(source), C.byref(h)))
    else:
        api.check(load(str(source).encode("utf-8"), C.byref(h)))
    try:
        info = PmxInfo()
        info.struct_size = C.sizeof(PmxInfo)
        api.check(lib.mmdx_pmx_get_info(h, C.byref(info)))
        d = api.ModelDesc()
        api.check(lib.mmdx_pmx_get_model_desc(h, C.byref(d)))
        a = PmxArrays()
        a.struct_size = C.sizeof(PmxArrays)
        api.check(lib.mmdx_pmx_get_arrays(h, C.byref(a)))
        nv, nb, nm, ne = info.n_vertices, info.n_bones, info.n_morphs, info.n_morph_entries
        f32, i32, u32 = np.float32, np.int32, np.uint32
        flat = FlatModel(
            positions=_arr(d.positions, nv * 3, f32).reshape(nv, 3),
            normals=_arr(d.normals, nv * 3, f32).reshape(nv, 3),
            uvs=_arr(d.uvs, nv * 2, f32).reshape(nv, 2),
            skin_type=_arr(d.skin_type, nv, i32),
            bone_ids=_arr(d.bone_ids, nv * 4, i32).reshape(nv, 4),
            bone_weights=_arr(d.bone_weights, nv * 4, f32).reshape(nv, 4),
            bone_pos=_arr(a.bone_rest_position, nb * 3, f32).reshape(nb, 3),
            bone_parent=_arr(d.bone_parent, nb, i32),
            morph_type=_arr(d.morph_type, nm, i32),
            morph_off=_arr(d.morph_offset, nm + 1, u32),
            morph_index=_arr(d.morph_index, ne, u32),
            morph_value=_arr(d.morph_value, ne * 3, f32).reshape(ne, 3),
            sdef=_arr(d.sdef_params, nv * 9, f32).reshape(nv, 9))

        from .vmd import SkeletonDesc
        sd = SkeletonDesc()
        api.check(lib.mmdx_pmx_get_skeleton_desc(h, C.byref(sd)))

        def vp(ptr, n, dtype):
            if n == 0 or not ptr:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()
        link_off = vp(sd.ik_link_offset, nb + 1, u32)
        nl = int(link_off[-1]) if nb else 0
        ik = dict(target=vp(sd.ik_target, nb, i32), loop=vp(sd.ik_loop_count, nb, i32), angle=vp(sd.ik_angle_limit, nb, f32),
                  link_off=link_off, link_bone=vp(sd.ik_link_bone, nl, i32), link_limited=vp(sd.ik_link_limited, nl, np.uint8),
                  link_lo=vp(sd.ik_link_lo, nl * 3, f32).reshape(nl, 3), link_hi=vp(sd.ik_link_hi, nl * 3, f32).reshape(nl, 3))
        append_parent, append_ratio = vp(sd.append_parent, nb, i32), vp(sd.append_ratio, nb, f32)
        morph_rotation = vp(sd.morph_rotation, ne * 4, f32).reshape(ne, 4)

        def name(kind, i):
            buf = C.create_string_buffer(1024)
            api.check(lib.mmdx_pmx_get_name(h, kind, i, buf, 1024))
            return buf.value.decode("utf-8", "replace")

        return PmxModel(
            flat=flat,
            info={k: getattr(info, k) for k, _ in PmxInfo._fields_ if k not in ("index_width",)} |
                 {"index_width": list(info.index_width)[:6]},
            triangles=_arr(a.triangles, info.n_indices, u32),
            material_index_count=_arr(a.material_index_count, info.n_materials, u32),
            bone_transform_level=_arr(a.bone_transform_level, nb, i32),
            bone_flags=_arr(a.bone_flags, nb, np.uint16),
            name=name(0, 0), bone_names=[name(1, i) for i in range(nb)],
            morph_names=[name(2, i) for i in range(nm)],
            append_parent=append_parent, append_ratio=append_ratio, ik=ik, morph_rotation=morph_rotation)
    finally:
        lib.mmdx_pmx_destroy(h)
