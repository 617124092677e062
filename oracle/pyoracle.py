"""ctypes front-ends for the two CHECKER libraries -- TEST INFRASTRUCTURE ONLY.

  Oracle    -> oracle/libmmdx_oracle.so   (from-scratch C restatement, oracle/mmdx_oracle.c)
  Reference -> oracle/_ref/libmmd_ref.so  (the real libmmd, oracle/ref_harness.cpp; build container
                                           only -- the prebuilt .so travels to the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product (simple_mmd_renderer_amd/) never does; it fails loudly when its HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libmmdx_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libmmd_ref.so")


def build(quiet: bool = True) -> None:
    """Compile the restatement (and the reference harness when /root/reference is mounted)."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Oracle:
    """The C restatement.  All methods take/return numpy arrays; nothing is cached."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.mmdx_oracle_time_crowd.restype = C.c_double

    def normalize(self, model):
        """Model::Normalize retagging; returns (type i32[NV], ids i64[NV,4], w f32[NV,4]) copies."""
        t = _c(model.skin_type, np.int32).copy()
        ids = _c(model.bone_ids, np.int64).copy()
        w = _c(model.bone_weights, np.float32).copy()
        par = _c(model.bone_parent, np.int64)
        self.lib.mmdx_oracle_normalize(C.c_uint32(model.nv), _p(t, C.c_int32), _p(ids, C.c_int64),
                                       _p(w, C.c_float), _p(par, C.c_int64))
        return t, ids, w

    def morph(self, model, rates):
        vimg = np.zeros((model.nv, 3), np.float32)
        rates = _c(rates, np.float32)
        assert rates.shape == (model.nm,)
        self.lib.mmdx_oracle_morph(
            C.c_uint32(model.nv), C.c_uint32(model.nm), _p(_c(model.morph_type, np.int32), C.c_int32),
            _p(_c(model.morph_off, np.uint32), C.c_uint32),
            _p(_c(model.morph_index, np.uint32), C.c_uint32),
            _p(_c(model.morph_value, np.float32), C.c_float), _p(rates, C.c_float),
            _p(vimg, C.c_float))
        return vimg

    def skin(self, model, palette, vimg=None, skin=None, positions=None):
        """Poser::Deform.  `skin` = (type, ids, w) as returned by normalize(), else the raw tags."""
        t, ids, w = skin if skin is not None else (
            _c(model.skin_type, np.int32), _c(model.bone_ids, np.int64),
            _c(model.bone_weights, np.float32))
        pos = _c(model.positions if positions is None else positions, np.float32)
        nrm = _c(model.normals, np.float32)
        pal = _c(palette, np.float32).reshape(-1, 16)
        assert pal.shape[0] == model.nb
        out_p = np.empty((model.nv, 3), np.float32)
        out_n = np.empty((model.nv, 3), np.float32)
        vi = _c(vimg, np.float32) if vimg is not None else None
        self.lib.mmdx_oracle_skin(C.c_uint32(model.nv), _p(pos, C.c_float), _p(nrm, C.c_float),
                                  _p(vi, C.c_float), _p(_c(t, np.int32), C.c_int32),
                                  _p(_c(ids, np.int64), C.c_int64), _p(_c(w, np.float32), C.c_float),
                                  _p(pal, C.c_float), _p(out_p, C.c_float), _p(out_n, C.c_float))
        return out_p, out_n

    def repack32(self, model, pos, nrm, pos_scale):
        out = np.empty((model.nv, 8), np.float32)
        self.lib.mmdx_oracle_repack32(C.c_uint32(model.nv), _p(_c(pos, np.float32), C.c_float),
                                      _p(_c(nrm, np.float32), C.c_float),
                                      _p(_c(model.uvs, np.float32), C.c_float),
                                      C.c_float(pos_scale), _p(out, C.c_float))
        return out

    def deform(self, model, rates, palette, normalize=True):
        """Whole path: (optional) Normalize -> morph pass -> Deform.  Returns (pos, nrm)."""
        skin = self.normalize(model) if normalize else None
        vimg = self.morph(model, rates)
        return self.skin(model, palette, vimg, skin)

    def morph_tracks(self, key_off, frames, weights, at):
        """Motion::GetMorphPose for every model morph at the frames `at` -> f32 [len(at), NM]."""
        key_off = _c(key_off, np.uint32)
        at = _c(at, np.uint32).reshape(-1)
        nm = key_off.size - 1
        out = np.zeros((at.size, nm), np.float32)
        fr, w = _c(frames, np.uint32), _c(weights, np.float32)
        if fr.size == 0:
            fr, w = np.zeros(1, np.uint32), np.zeros(1, np.float32)
        self.lib.mmdx_oracle_morph_tracks(C.c_uint32(nm), _p(key_off, C.c_uint32), _p(fr, C.c_uint32),
                                          _p(w, C.c_float), C.c_uint32(at.size), _p(at, C.c_uint32),
                                          _p(out, C.c_float))
        return out

    def bone_pose(self, frames, tr, rot, interp, frame):
        """Motion::GetBonePose for ONE track (keys ascending; interp int8 [n,64]) -> f32 [8]."""
        fr = _c(frames, np.uint32).reshape(-1)
        n = fr.size
        tr = _c(tr, np.float32).reshape(n, 3) if n else np.zeros((1, 3), np.float32)
        rot = _c(rot, np.float32).reshape(n, 4) if n else np.zeros((1, 4), np.float32)
        ip = _c(interp, np.int8).reshape(n, 64) if n else np.zeros((1, 64), np.int8)
        if n == 0:
            fr = np.zeros(1, np.uint32)
        out = np.zeros(8, np.float32)
        self.lib.mmdx_oracle_bone_pose(C.c_uint32(n), _p(fr, C.c_uint32), _p(tr, C.c_float), _p(rot, C.c_float),
                                       _p(ip, C.c_int8), C.c_uint32(int(frame)), _p(out, C.c_float))
        return out

    def bone_solve(self, rest, parent, poses, level=None, flags=None):
        """Bone solve without IK / append: poses f32 [NB,8] -> palette f32 [NB,16]."""
        rest = _c(rest, np.float32).reshape(-1, 3)
        nb = rest.shape[0]
        parent = _c(parent, np.int64).reshape(nb)
        poses = _c(poses, np.float32).reshape(nb, 8)
        lv = _c(level, np.int32).reshape(nb) if level is not None else None
        fl = _c(flags, np.uint16).reshape(nb) if flags is not None else None
        out = np.zeros((nb, 16), np.float32)
        scratch = np.zeros(nb * 17 + 4, np.float32)
        self.lib.mmdx_oracle_bone_solve.restype = C.c_int
        rc = self.lib.mmdx_oracle_bone_solve(
            C.c_uint32(nb), _p(rest, C.c_float), _p(parent, C.c_int64),
            _p(lv, C.c_int32) if lv is not None else None, _p(fl, C.c_uint16) if fl is not None else None,
            _p(poses, C.c_float), _p(out, C.c_float), scratch.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("oracle bone solve: IK / append bones are not restated")
        return out

    def bone_solve_full(self, rest, parent, poses, level=None, flags=None, append_parent=None, append_ratio=None,
                        ik=None, morphs=None, rates=None):
        """The whole bone solve incl. append bones and CCD-IK: poses f32 [NB,8] -> palette f32 [NB,16]."""
        rest = _c(rest, np.float32).reshape(-1, 3)
        nb = rest.shape[0]
        arrs = [rest, _c(parent, np.int64).reshape(nb),
                _c(level, np.int32).reshape(nb) if level is not None else None,
                _c(flags, np.uint16).reshape(nb) if flags is not None else np.zeros(nb, np.uint16),
                _c(append_parent, np.int64).reshape(nb) if append_parent is not None else None,
                _c(append_ratio, np.float32).reshape(nb) if append_ratio is not None else None] + ik_arrays(ik)
        types = [C.c_float, C.c_int64, C.c_int32, C.c_uint16, C.c_int64, C.c_float] + IK_TYPES
        poses = _c(poses, np.float32).reshape(nb, 8)
        out = np.zeros((nb, 16), np.float32)
        scratch = np.zeros(nb * 180 + 64, np.uint8)
        nm, marr = morph_arrays(morphs if rates is not None else None)
        r = _c(rates, np.float32).reshape(nm) if nm else None
        self.lib.mmdx_oracle_bone_solve_full.restype = C.c_int
        rc = self.lib.mmdx_oracle_bone_solve_full(
            C.c_uint32(nb), *[_p(a, t) if a is not None else None for a, t in zip(arrs, types)],
            C.c_uint32(nm), *[_p(a, t) if a is not None else None for a, t in zip(marr, MORPH_TYPES)],
            _p(r, C.c_float) if r is not None else None,
            _p(poses, C.c_float), _p(out, C.c_float), scratch.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("oracle bone solve: an index out of range")
        return out

    def bone_solve_physics(self, rest, parent, poses, over_bone, over_strict, over_skin, level=None, flags=None,
                           append_parent=None, append_ratio=None, ik=None, morphs=None, rates=None):
        """The bone solve with the physics reactor's writes between the two bone lists (main.cpp:1801-1810):
        over_skin f32 [K,16] replace the skinning matrices of over_bone [K] after the pre-physics list, Fix() is
        applied where over_strict [K] is non-zero.  Returns (palette f32 [NB,16], palette after the pre-physics list)."""
        rest = _c(rest, np.float32).reshape(-1, 3)
        nb = rest.shape[0]
        arrs = [rest, _c(parent, np.int64).reshape(nb),
                _c(level, np.int32).reshape(nb) if level is not None else None,
                _c(flags, np.uint16).reshape(nb) if flags is not None else np.zeros(nb, np.uint16),
                _c(append_parent, np.int64).reshape(nb) if append_parent is not None else None,
                _c(append_ratio, np.float32).reshape(nb) if append_ratio is not None else None] + ik_arrays(ik)
        types = [C.c_float, C.c_int64, C.c_int32, C.c_uint16, C.c_int64, C.c_float] + IK_TYPES
        poses = _c(poses, np.float32).reshape(nb, 8)
        out = np.zeros((nb, 16), np.float32)
        pre = np.zeros((nb, 16), np.float32)
        scratch = np.zeros(nb * 180 + 64, np.uint8)
        nm, marr = morph_arrays(morphs if rates is not None else None)
        r = _c(rates, np.float32).reshape(nm) if nm else None
        ob = _c(over_bone, np.int64).reshape(-1)
        k = ob.size
        os_ = _c(over_strict, np.uint8).reshape(k) if k else np.zeros(1, np.uint8)
        ok = _c(over_skin, np.float32).reshape(k, 16) if k else np.zeros((1, 16), np.float32)
        if not k:
            ob = np.zeros(1, np.int64)
        self.lib.mmdx_oracle_bone_solve_physics.restype = C.c_int
        rc = self.lib.mmdx_oracle_bone_solve_physics(
            C.c_uint32(nb), *[_p(a, t) if a is not None else None for a, t in zip(arrs, types)],
            C.c_uint32(nm), *[_p(a, t) if a is not None else None for a, t in zip(marr, MORPH_TYPES)],
            _p(r, C.c_float) if r is not None else None,
            _p(poses, C.c_float), _p(out, C.c_float), scratch.ctypes.data_as(C.c_void_p),
            C.c_uint32(k), _p(ob, C.c_int64), _p(os_, C.c_uint8), _p(ok, C.c_float), _p(pre, C.c_float))
        if rc != 0:
            raise ValueError("oracle bone solve: an index out of range")
        return out, pre

    def matrix_inverse(self, m):
        """Matrix4f::Inverse() restated (L/util/math_impl.inl:822-897): f32 [16] -> f32 [16]."""
        m = _c(m, np.float32).reshape(16)
        out = np.zeros(16, np.float32)
        self.lib.mmdx_oracle_matrix_inverse(_p(m, C.c_float), _p(out, C.c_float))
        return out

    def trace_libm(self, fn, capacity=1 << 22):
        """Run fn() with the bone solve's transcendental calls recorded; returns u32 [N,4] records (function id,
        argument bits, second argument bits, result bits) -- see mmdx_oracle_trace_libm."""
        buf = np.zeros((capacity, 4), np.uint32)
        self.lib.mmdx_oracle_trace_count.restype = C.c_size_t
        self.lib.mmdx_oracle_trace_libm(buf.ctypes.data_as(C.c_void_p), C.c_size_t(capacity))
        try:
            fn()
            n = int(self.lib.mmdx_oracle_trace_count())
        finally:
            self.lib.mmdx_oracle_trace_libm(None, C.c_size_t(0))
        if n > capacity:
            raise ValueError("libm trace overflow: %d calls" % n)
        return buf[:n].copy()

    def time_crowd(self, model, rates, palettes, normalize=True):
        """Seconds for one crowd step (shared morph pass + one skinning pass per palette)."""
        t, ids, w = self.normalize(model) if normalize else (
            _c(model.skin_type, np.int32), _c(model.bone_ids, np.int64),
            _c(model.bone_weights, np.float32))
        pal = _c(palettes, np.float32).reshape(-1, model.nb, 16)
        vimg = np.zeros((model.nv, 3), np.float32)
        op = np.empty((model.nv, 3), np.float32)
        on = np.empty((model.nv, 3), np.float32)
        return float(self.lib.mmdx_oracle_time_crowd(
            C.c_uint32(model.nv), C.c_uint32(model.nb), C.c_uint32(model.nm),
            _p(_c(model.positions, np.float32), C.c_float), _p(_c(model.normals, np.float32), C.c_float),
            _p(t, C.c_int32), _p(ids, C.c_int64), _p(w, C.c_float),
            _p(_c(model.morph_type, np.int32), C.c_int32), _p(_c(model.morph_off, np.uint32), C.c_uint32),
            _p(_c(model.morph_index, np.uint32), C.c_uint32),
            _p(_c(model.morph_value, np.float32), C.c_float), _p(_c(rates, np.float32), C.c_float),
            C.c_uint32(pal.shape[0]), _p(pal, C.c_float), _p(vimg, C.c_float), _p(op, C.c_float),
            _p(on, C.c_float)))


IK_TYPES = [C.c_int64, C.c_int32, C.c_float, C.c_uint32, C.c_int64, C.c_uint8, C.c_float, C.c_float]


def ik_arrays(ik):
    """The eight flat IK arrays in ABI order (None each when the rig has no IK)."""
    if ik is None:
        return [None] * 8
    return [_c(ik["target"], np.int64), _c(ik["loop"], np.int32), _c(ik["angle"], np.float32),
            _c(ik["link_off"], np.uint32), _c(ik["link_bone"], np.int64), _c(ik["link_limited"], np.uint8),
            _c(ik["link_lo"], np.float32), _c(ik["link_hi"], np.float32)]


MORPH_TYPES = [C.c_int32, C.c_uint32, C.c_uint32, C.c_float, C.c_float]


def morph_arrays(morphs):
    """(nm, [type, offset, index, value, rotation]) of a morph-table dict, or (0, five Nones)."""
    if morphs is None:
        return 0, [None] * 5
    nm = int(np.asarray(morphs["type"]).size)
    rot = morphs.get("rotation")
    return nm, [_c(morphs["type"], np.int32), _c(morphs["offset"], np.uint32), _c(morphs["index"], np.uint32),
                _c(morphs["value"], np.float32), _c(rot, np.float32) if rot is not None else None]


def reference_available() -> bool:
    return os.path.exists(REF_SO)


class ReferenceMotion:
    """libmmd's VmdReader + Motion (oracle/ref_harness.cpp)."""

    def __init__(self, path: str):
        if not reference_available():
            raise RuntimeError("oracle/_ref/libmmd_ref.so not built (needs /root/reference)")
        self.lib = C.CDLL(REF_SO)
        self.lib.mmdref_motion_load.restype = C.c_void_p
        self.lib.mmdref_last_error.restype = C.c_char_p
        self.lib.mmdref_motion_morph_weight.restype = C.c_float
        h = self.lib.mmdref_motion_load(str(path).encode())
        if not h:
            raise RuntimeError("libmmd VmdReader: " + (self.lib.mmdref_last_error() or b"").decode("utf-8", "replace"))
        self.h = C.c_void_p(h)

    def morph_weight(self, sjis_name: bytes, frame: int) -> float:
        """Motion::GetMorphPose(name, frame) for the track whose Shift-JIS name bytes are given; NaN when
        the motion has no such track."""
        return float(self.lib.mmdref_motion_morph_weight(self.h, sjis_name, C.c_uint32(frame)))

    def bone_pose(self, sjis_name: bytes, frame: int):
        """Motion::GetBonePose(name, frame) -> f32 [8] (t.xyz, 0, q.xyzw), or None without such a track."""
        out = np.zeros(8, np.float32)
        self.lib.mmdref_motion_bone_pose.restype = C.c_int
        ok = self.lib.mmdref_motion_bone_pose(self.h, sjis_name, C.c_uint32(frame), _p(out, C.c_float))
        return out if ok else None

    def time_motion_solve(self, skeleton_ref: "Reference", frames) -> float:
        """Seconds libmmd needs to turn this motion into palettes for len(frames) instances of a
        Reference.skeleton whose bone b is keyed "b<b>" (GetBonePose + SetBonePose + Pre/PostPhysicsPosing)."""
        fr = _c(frames, np.uint32).reshape(-1)
        self.lib.mmdref_time_motion_solve.restype = C.c_double
        return float(self.lib.mmdref_time_motion_solve(self.h, skeleton_ref.h, C.c_uint32(fr.size), _p(fr, C.c_uint32)))

    def names_match_model(self, ref_model: "Reference") -> int:
        """How many of the model's morph names libmmd finds in the motion (MotionPlayer's mapping)."""
        return int(self.lib.mmdref_motion_count_registered_morphs(self.h, ref_model.h))

    def close(self):
        if self.h:
            self.lib.mmdref_motion_destroy(self.h)
            self.h = None


class Reference:
    """The real libmmd Poser driven through oracle/ref_harness.cpp."""

    def __init__(self, model, normalize=True):
        if not reference_available():
            raise RuntimeError("oracle/_ref/libmmd_ref.so not built (needs /root/reference)")
        lib = C.CDLL(REF_SO)
        lib.mmdref_create.restype = C.c_void_p
        lib.mmdref_time_frames.restype = C.c_double
        lib.mmdref_time_crowd.restype = C.c_double
        self.lib, self.model = lib, model
        sdef = _c(model.sdef, np.float32) if model.sdef is not None else None
        self._keep = [
            _c(model.positions, np.float32), _c(model.normals, np.float32), _c(model.uvs, np.float32),
            _c(model.skin_type, np.int32), _c(model.bone_ids, np.int64),
            _c(model.bone_weights, np.float32), sdef, _c(model.bone_pos, np.float32),
            _c(model.bone_parent, np.int64), _c(model.morph_type, np.int32),
            _c(model.morph_off, np.uint32), _c(model.morph_index, np.uint32),
            _c(model.morph_value, np.float32)]
        k = self._keep
        self.h = C.c_void_p(lib.mmdref_create(
            C.c_uint32(model.nv), C.c_uint32(model.nb), C.c_uint32(model.nm),
            _p(k[0], C.c_float), _p(k[1], C.c_float), _p(k[2], C.c_float), _p(k[3], C.c_int32),
            _p(k[4], C.c_int64), _p(k[5], C.c_float), _p(k[6], C.c_float), _p(k[7], C.c_float),
            _p(k[8], C.c_int64), _p(k[9], C.c_int32), _p(k[10], C.c_uint32), _p(k[11], C.c_uint32),
            _p(k[12], C.c_float), C.c_int(1 if normalize else 0)))

    @classmethod
    def skeleton(cls, rest, parent, level=None, flags=None, append_parent=None, append_ratio=None, ik=None,
                 morphs=None) -> "Reference":
        """Bones-only libmmd model + Poser, for the bone solve (set_bone_pose / pose / get_palette).
        ik = dict(target i64[NB], loop i32[NB], angle f32[NB], link_off u32[NB+1], link_bone i64[L],
        link_limited u8[L], link_lo f32[L,3], link_hi f32[L,3]) or None."""
        if not reference_available():
            raise RuntimeError("oracle/_ref/libmmd_ref.so not built (needs /root/reference)")
        self = cls.__new__(cls)
        lib = C.CDLL(REF_SO)
        lib.mmdref_create_skeleton.restype = C.c_void_p
        rest = _c(rest, np.float32).reshape(-1, 3)
        nb = rest.shape[0]
        keep = [rest, _c(parent, np.int64).reshape(nb),
                _c(level, np.int32).reshape(nb) if level is not None else None,
                _c(flags, np.uint16).reshape(nb) if flags is not None else None,
                _c(append_parent, np.int64).reshape(nb) if append_parent is not None else None,
                _c(append_ratio, np.float32).reshape(nb) if append_ratio is not None else None]
        types = [C.c_float, C.c_int64, C.c_int32, C.c_uint16, C.c_int64, C.c_float]
        keep += ik_arrays(ik)
        types += IK_TYPES
        nm, marr = morph_arrays(morphs)
        h = lib.mmdref_create_skeleton(C.c_uint32(nb), *[_p(a, t) if a is not None else None
                                                         for a, t in zip(keep, types)],
                                       C.c_uint32(nm), *[_p(a, t) if a is not None else None
                                                         for a, t in zip(marr, MORPH_TYPES)])
        keep += marr
        self.lib, self._keep, self.h = lib, keep, C.c_void_p(h)

        class _Dims:
            pass
        self.model = _Dims()
        self.model.nv, self.model.nb, self.model.nm = 0, nb, nm
        return self

    def solve(self, poses, rates=None):
        """ResetPosing-equivalent state + the given local poses [NB,8] (+ morph rates [NM]) -> palette [NB,16]."""
        poses = _c(poses, np.float32).reshape(self.model.nb, 8)
        for b in range(self.model.nb):
            self.set_bone_pose(b, poses[b, 0:3], poses[b, 4:8])
        self.set_morphs(rates if rates is not None else np.zeros(self.model.nm, np.float32))
        self.pose()
        return self.get_palette()

    def solve_physics(self, poses, over_bone, over_strict, over_skin, rates=None):
        """The viewer's frame with a physics reactor in it: PrePhysicsPosing, the reactor's Synchronize / Fix writes
        (over_skin [K,16] for over_bone [K], Fix where over_strict [K]), PostPhysicsPosing.
        Returns (palette [NB,16], palette after PrePhysicsPosing)."""
        poses = _c(poses, np.float32).reshape(self.model.nb, 8)
        for b in range(self.model.nb):
            self.set_bone_pose(b, poses[b, 0:3], poses[b, 4:8])
        self.set_morphs(rates if rates is not None else np.zeros(self.model.nm, np.float32))
        ob = _c(over_bone, np.int64).reshape(-1)
        k = ob.size
        os_ = _c(over_strict, np.uint8).reshape(k) if k else np.zeros(1, np.uint8)
        ok = _c(over_skin, np.float32).reshape(k, 16) if k else np.zeros((1, 16), np.float32)
        if not k:
            ob = np.zeros(1, np.int64)
        pre = np.zeros((self.model.nb, 16), np.float32)
        self.lib.mmdref_pose_physics(self.h, C.c_uint32(k), _p(ob, C.c_int64), _p(os_, C.c_uint8), _p(ok, C.c_float),
                                     _p(pre, C.c_float))
        return self.get_palette(), pre

    @staticmethod
    def matrix_inverse(m):
        """libmmd's Matrix4f::Inverse() itself."""
        lib = C.CDLL(REF_SO)
        m = _c(m, np.float32).reshape(16)
        out = np.zeros(16, np.float32)
        lib.mmdref_matrix_inverse(_p(m, C.c_float), _p(out, C.c_float))
        return out

    @classmethod
    def from_pmd(cls, path: str) -> "Reference":
        """libmmd's own PmdReader + Poser on a file (the older format)."""
        return cls.from_pmx(path, entry="mmdref_create_from_pmd")

    @classmethod
    def from_pmx(cls, path: str, entry: str = "mmdref_create_from_pmx") -> "Reference":
        """Load a .pmx through the reference's own FileReader + PmxReader (+ Normalize) + Poser."""
        if not reference_available():
            raise RuntimeError("oracle/_ref/libmmd_ref.so not built (needs /root/reference)")
        self = cls.__new__(cls)
        lib = C.CDLL(REF_SO)
        create = getattr(lib, entry)
        create.restype = C.c_void_p
        lib.mmdref_last_error.restype = C.c_char_p
        lib.mmdref_time_frames.restype = C.c_double
        lib.mmdref_time_crowd.restype = C.c_double
        lib.mmdref_time_pmx_load.restype = C.c_double
        h = create(str(path).encode())
        if not h:
            raise RuntimeError("libmmd reader: " + (lib.mmdref_last_error() or b"").decode("utf-8", "replace"))
        self.lib, self.h = lib, C.c_void_p(h)
        nv, nb, nm, nt = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.mmdref_get_counts(self.h, C.byref(nv), C.byref(nb), C.byref(nm), C.byref(nt))

        class _Dims:          # just enough of a FlatModel for the accessors below
            pass
        self.model = _Dims()
        self.model.nv, self.model.nb, self.model.nm, self.model.ntri = nv.value, nb.value, nm.value, nt.value
        self._keep = []
        return self

    @staticmethod
    def time_pmx_load(path: str, repeats: int = 5) -> float:
        lib = C.CDLL(REF_SO)
        lib.mmdref_time_pmx_load.restype = C.c_double
        return float(lib.mmdref_time_pmx_load(str(path).encode(), C.c_int(repeats))) / repeats

    def close(self):
        if self.h:
            self.lib.mmdref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_skin(self):
        nv = self.model.nv
        t = np.empty(nv, np.int32)
        ids = np.empty((nv, 4), np.int64)
        w = np.empty((nv, 4), np.float32)
        self.lib.mmdref_get_skin(self.h, _p(t, C.c_int32), _p(ids, C.c_int64), _p(w, C.c_float))
        return t, ids, w

    def reset_posing(self):
        self.lib.mmdref_reset_posing(self.h)

    def set_bone_pose(self, i, t, q):
        t = _c(t, np.float32)
        q = _c(q, np.float32)
        self.lib.mmdref_set_bone_pose(self.h, C.c_uint32(i), _p(t, C.c_float), _p(q, C.c_float))

    def set_morphs(self, rates):
        for i, r in enumerate(np.asarray(rates, np.float32)):
            self.lib.mmdref_set_morph(self.h, C.c_uint32(i), C.c_float(float(r)))

    def pose(self):
        self.lib.mmdref_pose(self.h)

    def get_palette(self):
        out = np.empty((self.model.nb, 16), np.float32)
        self.lib.mmdref_get_palette(self.h, _p(out, C.c_float))
        return out

    def set_palette(self, pal):
        pal = _c(pal, np.float32).reshape(self.model.nb, 16)
        self.lib.mmdref_set_palette(self.h, _p(pal, C.c_float))

    def deform(self):
        self.lib.mmdref_deform(self.h)
        pos = np.empty((self.model.nv, 3), np.float32)
        nrm = np.empty((self.model.nv, 3), np.float32)
        self.lib.mmdref_get_pose_image(self.h, _p(pos, C.c_float), _p(nrm, C.c_float))
        return pos, nrm

    def repack32(self, pos_scale=0.1):
        out = np.empty((self.model.nv, 8), np.float32)
        self.lib.mmdref_repack32(self.h, C.c_float(pos_scale), _p(out, C.c_float))
        return out

    def run(self, rates, palette=None):
        """ResetPosing -> SetMorphPose x NM -> Pre/PostPhysicsPosing -> [inject palette] -> Deform.
        Returns (pos, nrm, palette_used)."""
        self.reset_posing()
        self.set_morphs(rates)
        self.pose()
        if palette is not None:
            self.set_palette(palette)
        pal = self.get_palette()
        pos, nrm = self.deform()
        return pos, nrm, pal

    def time_frames(self, rates, palettes=None):
        rates = _c(rates, np.float32).reshape(-1, self.model.nm)
        pal = _c(palettes, np.float32) if palettes is not None else None
        scratch = np.empty((self.model.nv, 8), np.float32)
        return float(self.lib.mmdref_time_frames(self.h, C.c_uint32(rates.shape[0]),
                                                 _p(rates, C.c_float), _p(pal, C.c_float),
                                                 _p(scratch, C.c_float)))

    def time_crowd(self, rates, palettes):
        pal = _c(palettes, np.float32).reshape(-1, self.model.nb, 16)
        return float(self.lib.mmdref_time_crowd(self.h, C.c_uint32(pal.shape[0]),
                                                _p(_c(rates, np.float32), C.c_float),
                                                _p(pal, C.c_float)))


# ---- the REAL physics reactor (libmmd's mmd-bullet binding + the vendored Bullet), oracle/ref_bullet_harness.cpp ------------
BULLET_REF_SO = os.path.join(_HERE, "_ref", "libmmd_bullet_ref.so")


def bullet_reference_available() -> bool:
    return os.path.exists(BULLET_REF_SO)


class BulletReference:
    """mmd::Poser + mmd::BulletPhysicsReactor on a model with rigid bodies and 6-DOF spring constraints, stepped the way the
    viewer's frame() does (main.cpp:1786-1821).  `rig` = dict(rest, parent, level, flags); `mesh` = dict(positions, normals,
    skin_type, bone_ids, bone_weights); `bodies` / `joints` = dicts of the PMX rigid-body / joint fields (see
    oracle/gen_golden_bullet.py)."""

    def __init__(self, rig, mesh, bodies, joints):
        if not bullet_reference_available():
            raise RuntimeError("oracle/_ref/libmmd_bullet_ref.so not built (needs /root/reference)")
        self.lib = C.CDLL(BULLET_REF_SO)
        self.lib.mmdbt_create.restype = C.c_void_p
        self.nb = int(np.asarray(rig["rest"]).shape[0])
        self.nv = int(np.asarray(mesh["positions"]).shape[0])
        self.nrb = int(np.asarray(bodies["bone"]).shape[0])
        nc = int(np.asarray(joints["body"]).shape[0])
        f32, i64 = np.float32, np.int64
        a = [(_c(rig["rest"], f32), C.c_float), (_c(rig["parent"], i64), C.c_int64), (_c(rig["level"], np.int32), C.c_int32),
             (_c(rig["flags"], np.uint16), C.c_uint16)]
        b = [(_c(mesh["positions"], f32), C.c_float), (_c(mesh["normals"], f32), C.c_float),
             (_c(mesh["skin_type"], np.int32), C.c_int32), (_c(mesh["bone_ids"], i64), C.c_int64),
             (_c(mesh["bone_weights"], f32), C.c_float)]
        c = [(_c(bodies["bone"], i64), C.c_int64), (_c(bodies["group"], np.uint8), C.c_uint8), (_c(bodies["mask"], np.uint16), C.c_uint16),
             (_c(bodies["shape"], np.uint8), C.c_uint8), (_c(bodies["dims"], f32), C.c_float), (_c(bodies["pos"], f32), C.c_float),
             (_c(bodies["rot"], f32), C.c_float), (_c(bodies["mass"], f32), C.c_float), (_c(bodies["tdamp"], f32), C.c_float),
             (_c(bodies["rdamp"], f32), C.c_float), (_c(bodies["restitution"], f32), C.c_float), (_c(bodies["friction"], f32), C.c_float),
             (_c(bodies["type"], np.uint8), C.c_uint8)]
        d = [(_c(joints["body"], i64), C.c_int64)] + [(_c(joints[k], f32), C.c_float)
                                                       for k in ("pos", "rot", "pos_lo", "pos_hi", "rot_lo", "rot_hi", "spring_t", "spring_r")]
        self._keep = a + b + c + d
        self.h = C.c_void_p(self.lib.mmdbt_create(
            C.c_uint32(self.nb), *[_p(x, t) for x, t in a], C.c_uint32(self.nv), *[_p(x, t) for x, t in b],
            C.c_uint32(self.nrb), *[_p(x, t) for x, t in c], C.c_uint32(nc), *[_p(x, t) for x, t in d]))

    def body_info(self):
        """(passive, strict, ghost) u8 [NRB] as PoserMotionState classified the bodies."""
        out = [np.zeros(self.nrb, np.uint8) for _ in range(3)]
        self.lib.mmdbt_body_info(self.h, *[_p(o, C.c_uint8) for o in out])
        return out

    def frame(self, poses, step=1.0 / 30.0):
        """One viewer frame.  Returns dict(palette_pre [NB,16], body_xf [NRB,16], palette [NB,16], pos [NV,3], nrm [NV,3])."""
        poses = _c(poses, np.float32).reshape(self.nb, 8)
        o = dict(palette_pre=np.zeros((self.nb, 16), np.float32), body_xf=np.zeros((self.nrb, 16), np.float32),
                 palette=np.zeros((self.nb, 16), np.float32), pos=np.zeros((self.nv, 3), np.float32),
                 nrm=np.zeros((self.nv, 3), np.float32))
        self.lib.mmdbt_frame(self.h, _p(poses, C.c_float), C.c_float(step), *[_p(o[k], C.c_float)
                                                                             for k in ("palette_pre", "body_xf", "palette", "pos", "nrm")])
        return o

    def close(self):
        if self.h:
            self.lib.mmdbt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
