"""VMD motion files: a writer for synthetic fixtures and the Python face of the C-ABI loader /
device evaluator (csrc/vmd.cpp + morph_track_eval_kernel).

File layout as the reference's reader consumes it (L/reader/vmd_reader_impl.inl:9-79,
L/reader/interprete/vmd_types.inl:17-37): 50-byte header, u32 count + 111-byte bone records,
u32 count + 23-byte morph records; names are Shift-JIS in 15-byte fields.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _capi as api

MAGIC = b"Vocaloid Motion Data 0002"


def _name15(name: str) -> bytes:
    b = name.encode("shift_jis")
    if len(b) > 15:
        raise ValueError("VMD names are at most 15 bytes of Shift-JIS: %r" % name)
    return b + b"\0" * (15 - len(b))


def write_vmd(bone_keys: Sequence[tuple], morph_keys: Sequence[Tuple[str, int, float]],
              model_name: str = "synthetic") -> bytes:
    """bone_keys: (name, frame, (tx,ty,tz), (qx,qy,qz,qw), interpolation[64] or None);
    morph_keys: (name, frame, weight).  Records are written in the given order (a later record for
    the same (name, frame) wins, as in the reference)."""
    out = bytearray()
    out += MAGIC + b"\0" * (30 - len(MAGIC))
    mn = model_name.encode("shift_jis")[:20]
    out += mn + b"\0" * (20 - len(mn))
    out += struct.pack("<I", len(bone_keys))
    for name, frame, t, q, interp in bone_keys:
        out += _name15(name) + struct.pack("<I", frame) + struct.pack("<3f", *t) + struct.pack("<4f", *q)
        ip = bytes(interp) if interp is not None else bytes([20, 20, 0, 0, 20, 20, 20, 20, 107, 107, 107, 107,
                                                               107, 107, 107, 107] * 4)
        assert len(ip) == 64
        out += ip
    out += struct.pack("<I", len(morph_keys))
    for name, frame, w in morph_keys:
        out += _name15(name) + struct.pack("<I", frame) + np.float32(w).tobytes()
    out += struct.pack("<I", 0) * 3        # camera, light, self-shadow sections: empty
    return bytes(out)


class VmdInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_bone_records", C.c_uint32), ("n_morph_records", C.c_uint32),
                ("n_bone_tracks", C.c_uint32), ("n_morph_tracks", C.c_uint32), ("n_bone_keys", C.c_uint32),
                ("n_morph_keys", C.c_uint32), ("max_frame", C.c_uint32), ("bytes_consumed", C.c_uint64)]


class VmdBoneKey(C.Structure):
    _fields_ = [("frame", C.c_uint32), ("translation", C.c_float * 3), ("rotation", C.c_float * 4),
                ("interpolation", C.c_int8 * 64)]


class Vmd:
    """A parsed VMD motion (mmdx_vmd_t)."""

    def __init__(self, source):
        lib = api.lib()
        self.h = C.c_void_p()
        if isinstance(source, (bytes, bytearray)):
            buf = (C.c_char * len(source)).from_buffer_copy(bytes(source))
            api.check(lib.mmdx_vmd_parse(buf, len(source), C.byref(self.h)))
        else:
            api.check(lib.mmdx_vmd_load_file(str(source).encode("utf-8"), C.byref(self.h)))
        info = VmdInfo()
        info.struct_size = C.sizeof(VmdInfo)
        api.check(lib.mmdx_vmd_get_info(self.h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in VmdInfo._fields_}

    def close(self):
        if getattr(self, "h", None):
            api.lib().mmdx_vmd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _name(self, is_morph: int, i: int) -> str:
        buf = C.create_string_buffer(256)
        api.check(api.lib().mmdx_vmd_track_name(self.h, is_morph, i, buf, 256))
        return buf.value.decode("utf-8", "replace")

    @property
    def morph_track_names(self) -> List[str]:
        return [self._name(1, i) for i in range(self.info["n_morph_tracks"])]

    @property
    def bone_track_names(self) -> List[str]:
        return [self._name(0, i) for i in range(self.info["n_bone_tracks"])]

    def morph_track(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        fr, w, n = C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.c_uint32()
        api.check(api.lib().mmdx_vmd_morph_track(self.h, i, C.byref(fr), C.byref(w), C.byref(n)))
        k = n.value
        return (np.ctypeslib.as_array(fr, (k,)).copy() if k else np.zeros(0, np.uint32),
                np.ctypeslib.as_array(w, (k,)).copy() if k else np.zeros(0, np.float32))

    def bone_track(self, i: int) -> List[dict]:
        keys, n = C.POINTER(VmdBoneKey)(), C.c_uint32()
        api.check(api.lib().mmdx_vmd_bone_track(self.h, i, C.byref(keys), C.byref(n)))
        return [dict(frame=keys[j].frame, translation=tuple(keys[j].translation), rotation=tuple(keys[j].rotation),
                     interpolation=bytes(bytearray(keys[j].interpolation))) for j in range(n.value)]

    def bind_morphs(self, morph_names: Sequence[str]) -> "MorphMotion":
        return MorphMotion(self, morph_names)

    def bind_bones(self, bone_names: Sequence[str]) -> "BoneMotion":
        return BoneMotion(self, bone_names)


class MorphMotion:
    """Morph tracks of a motion bound to a model's morph list; evaluated on the GPU."""

    def __init__(self, vmd: Vmd, morph_names: Sequence[str]):
        enc = [n.encode("utf-8") for n in morph_names]
        arr = (C.c_char_p * max(len(enc), 1))(*enc)
        self.h = C.c_void_p()
        api.check(api.lib().mmdx_vmd_bind_morphs(vmd.h, len(enc), arr, C.byref(self.h)))
        nm, mapped, keys = C.c_uint32(), C.c_uint32(), C.c_uint32()
        api.check(api.lib().mmdx_morph_motion_get_info(self.h, C.byref(nm), C.byref(mapped), C.byref(keys)))
        self.nm, self.n_mapped, self.n_keys = nm.value, mapped.value, keys.value

    def eval(self, frames, model=None) -> np.ndarray:
        """Host convenience: frames [NI] -> rates f32 [NI, NM] (device evaluation + D2H)."""
        fr = np.ascontiguousarray(frames, np.uint32).reshape(-1)
        out = np.empty((fr.size, self.nm), np.float32)
        api.check(api.lib().mmdx_morph_motion_eval(self.h, model.h if model is not None else None, fr.size,
                                                   fr.ctypes.data, 0, out.ctypes.data))
        return out

    def eval_device(self, n_instances: int, frames_ptr, out_ptr, model=None) -> None:
        """frames u32[NI] and out f32[NI][NM] resident in HBM; asynchronous on the model's stream."""
        api.check(api.lib().mmdx_morph_motion_eval(self.h, model.h if model is not None else None, n_instances,
                                                   frames_ptr, 1 | api.OUT_ON_DEVICE, out_ptr))

    def close(self):
        if getattr(self, "h", None):
            api.lib().mmdx_morph_motion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


FRAMES_ON_DEVICE = 1 << 0
POSES_ON_DEVICE = 1 << 4
POSE_FLOATS = 8


class BoneMotion:
    """Bone tracks of a motion bound to a model's bone list; local poses are evaluated on the GPU
    (Motion::GetBonePose for every (instance, bone): t.xyz, 0, q.xyzw)."""

    def __init__(self, vmd: Vmd, bone_names: Sequence[str]):
        enc = [n.encode("utf-8") for n in bone_names]
        arr = (C.c_char_p * max(len(enc), 1))(*enc)
        self.h = C.c_void_p()
        api.check(api.lib().mmdx_vmd_bind_bones(vmd.h, len(enc), arr, C.byref(self.h)))
        nb, mapped, keys, curves = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        api.check(api.lib().mmdx_bone_motion_get_info(self.h, C.byref(nb), C.byref(mapped), C.byref(keys),
                                                      C.byref(curves)))
        self.nb, self.n_mapped, self.n_keys, self.n_curves = nb.value, mapped.value, keys.value, curves.value

    def eval(self, frames, model=None) -> np.ndarray:
        """Host convenience: frames [NI] -> poses f32 [NI, NB, 8] (device evaluation + D2H)."""
        fr = np.ascontiguousarray(frames, np.uint32).reshape(-1)
        out = np.empty((fr.size, self.nb, POSE_FLOATS), np.float32)
        api.check(api.lib().mmdx_bone_motion_eval(self.h, model.h if model is not None else None, fr.size,
                                                  fr.ctypes.data, 0, out.ctypes.data))
        return out

    def eval_device(self, n_instances: int, frames_ptr, out_ptr, model=None) -> None:
        """frames u32[NI] and out f32[NI][NB][8] resident in HBM; asynchronous on the model's stream."""
        api.check(api.lib().mmdx_bone_motion_eval(self.h, model.h if model is not None else None, n_instances,
                                                  frames_ptr, FRAMES_ON_DEVICE | api.OUT_ON_DEVICE, out_ptr))

    def close(self):
        if getattr(self, "h", None):
            api.lib().mmdx_bone_motion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SkeletonDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_bones", C.c_uint32), ("rest_position", C.c_void_p),
                ("parent", C.c_void_p), ("transform_level", C.c_void_p), ("flags", C.c_void_p),
                ("append_parent", C.c_void_p), ("append_ratio", C.c_void_p),
                ("ik_target", C.c_void_p), ("ik_loop_count", C.c_void_p), ("ik_angle_limit", C.c_void_p),
                ("ik_link_offset", C.c_void_p), ("ik_link_bone", C.c_void_p), ("ik_link_limited", C.c_void_p),
                ("ik_link_lo", C.c_void_p), ("ik_link_hi", C.c_void_p),
                ("n_morphs", C.c_uint32), ("create_flags", C.c_uint32), ("morph_type", C.c_void_p),
                ("morph_offset", C.c_void_p), ("morph_index", C.c_void_p), ("morph_value", C.c_void_p),
                ("morph_rotation", C.c_void_p)]


class SkeletonInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_bones", C.c_uint32), ("n_pre_physics", C.c_uint32),
                ("n_post_physics", C.c_uint32), ("max_chain", C.c_uint32), ("solver", C.c_uint32),
                ("n_ik_bones", C.c_uint32), ("n_ik_links", C.c_uint32), ("n_append_bones", C.c_uint32),
                ("n_bone_morph_entries", C.c_uint32), ("n_solve_rounds", C.c_uint32),
                ("n_ik_rounds_16_lanes", C.c_uint32)]


SOLVER_PARALLEL_FK, SOLVER_SERIAL = 0, 1
SKELETON_PHYSICS_SEAM = 1
OVERRIDES_ON_DEVICE = 1 << 5


class PhysicsOverrides(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_bones", C.c_uint32), ("bone", C.c_void_p), ("strict", C.c_void_p),
                ("skinning", C.c_void_p)]


class Skeleton:
    """A model's bone hierarchy compiled for the device bone solve (local poses -> float[16] palettes,
    Poser::UpdateBoneTransform + UpdateBoneSkinningMatrix in the reference's evaluation order).
    ik = dict(target, loop, angle, link_off, link_bone, link_limited, link_lo, link_hi) as produced by
    synth.make_ik_rig / the PMX loader; None for a rig without IK."""

    def __init__(self, rest_position, parent, transform_level=None, flags=None, append_parent=None,
                 append_ratio=None, ik=None, morphs=None, physics_seam=False):
        """morphs = dict(type i32[NM], offset u32[NM+1], index u32[E], value f32[E,3], rotation f32[E,4] or None):
        the model's morph table; its group and bone morphs feed mmdx_skeleton_solve_morphed."""
        rest = np.ascontiguousarray(rest_position, np.float32).reshape(-1, 3)
        nb = rest.shape[0]

        def arr(a, dt, shape=None):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            return a.reshape(shape) if shape is not None else a
        keep = [rest, arr(parent, np.int32, nb), arr(transform_level, np.int32, nb), arr(flags, np.uint16, nb),
                arr(append_parent, np.int32, nb), arr(append_ratio, np.float32, nb)]
        if ik is not None:
            keep += [arr(ik["target"], np.int32, nb), arr(ik["loop"], np.int32, nb), arr(ik["angle"], np.float32, nb),
                     arr(ik["link_off"], np.uint32, nb + 1), arr(ik["link_bone"], np.int32), arr(ik["link_limited"], np.uint8),
                     arr(ik["link_lo"], np.float32), arr(ik["link_hi"], np.float32)]
        else:
            keep += [None] * 8
        nm = 0
        if morphs is not None:
            nm = int(np.asarray(morphs["type"]).size)
            mk = [arr(morphs["type"], np.int32, nm), arr(morphs["offset"], np.uint32, nm + 1), arr(morphs["index"], np.uint32),
                  arr(morphs["value"], np.float32), arr(morphs.get("rotation"), np.float32)]
        else:
            mk = [None] * 5
        ptr = lambda a: a.ctypes.data if a is not None else None          # noqa: E731
        d = SkeletonDesc(C.sizeof(SkeletonDesc), nb, *[ptr(a) for a in keep], nm, SKELETON_PHYSICS_SEAM if physics_seam else 0,
                         *[ptr(a) for a in mk])
        self._keep = keep + mk
        self.nm = nm
        self.h = C.c_void_p()
        api.check(api.lib().mmdx_skeleton_create(C.byref(d), C.byref(self.h)))
        info = SkeletonInfo()
        info.struct_size = C.sizeof(SkeletonInfo)
        api.check(api.lib().mmdx_skeleton_get_info(self.h, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in SkeletonInfo._fields_}
        self.nb = nb

    def solve(self, poses, model=None, morph_weights=None) -> np.ndarray:
        """Host convenience: poses f32 [NI, NB, 8] (+ morph rates [NI, NM] or shared [NM]) -> palettes
        f32 [NI, NB, 16]."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, self.nb, POSE_FLOATS)
        out = np.empty((p.shape[0], self.nb, 16), np.float32)
        if morph_weights is None:
            api.check(api.lib().mmdx_skeleton_solve(self.h, model.h if model is not None else None, p.shape[0],
                                                    p.ctypes.data, 0, out.ctypes.data))
        else:
            w = np.ascontiguousarray(morph_weights, np.float32)
            flags = api.WEIGHTS_SHARED if w.ndim == 1 else 0
            assert w.shape[-1] == self.nm and (w.ndim == 1 or w.shape[0] == p.shape[0])
            api.check(api.lib().mmdx_skeleton_solve_morphed(self.h, model.h if model is not None else None, p.shape[0],
                                                            p.ctypes.data, w.ctypes.data, flags, out.ctypes.data))
        return out

    # -- the physics seam: PrePhysicsPosing | the reactor's writes | PostPhysicsPosing (main.cpp:1801-1810) ------------
    def solve_pre(self, poses, model=None, morph_weights=None) -> np.ndarray:
        """Reset + bone morphs + the pre-physics bone list -> palettes [NI, NB, 16] (rows of post-physics bones are
        unspecified until solve_post)."""
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, self.nb, POSE_FLOATS)
        self._seam_out = np.empty((p.shape[0], self.nb, 16), np.float32)
        w, flags = None, 0
        if morph_weights is not None:
            w = np.ascontiguousarray(morph_weights, np.float32)
            flags = api.WEIGHTS_SHARED if w.ndim == 1 else 0
        api.check(api.lib().mmdx_skeleton_solve_pre(self.h, model.h if model is not None else None, p.shape[0],
                                                    p.ctypes.data, w.ctypes.data if w is not None else None, flags,
                                                    self._seam_out.ctypes.data))
        return self._seam_out.copy()

    def solve_post(self, bones, strict, skinning, model=None) -> np.ndarray:
        """The reactor's writes (skinning [NI, K, 16] for bones [K]; Fix where strict [K]) then the post-physics list
        -> the complete palettes [NI, NB, 16]."""
        b = np.ascontiguousarray(bones, np.int32).reshape(-1)
        k = b.size
        st = np.ascontiguousarray(strict, np.uint8).reshape(k)
        xf = np.ascontiguousarray(skinning, np.float32)
        xf = xf.reshape(-1, k, 16) if k else xf.reshape(xf.shape[0], 0, 16)
        ni = xf.shape[0]
        if getattr(self, "_seam_out", None) is None or self._seam_out.shape[0] != ni:
            self._seam_out = np.empty((ni, self.nb, 16), np.float32)     # the library rejects the call
        ov = PhysicsOverrides(C.sizeof(PhysicsOverrides), k, b.ctypes.data if k else None, st.ctypes.data if k else None,
                              xf.ctypes.data if k else None)
        api.check(api.lib().mmdx_skeleton_solve_post(self.h, model.h if model is not None else None, ni, C.byref(ov), 0,
                                                     self._seam_out.ctypes.data))
        return self._seam_out.copy()

    # -- bone tracks -> palettes in one call (one launch on parallel-FK skeletons: the poses stay in LDS) -------------
    def solve_motion(self, motion: "BoneMotion", frames, model=None) -> np.ndarray:
        """Host convenience: frame numbers [NI] -> palettes f32 [NI, NB, 16] (= solve(motion.eval(frames)))."""
        f = np.ascontiguousarray(frames, np.uint32).reshape(-1)
        out = np.empty((f.size, self.nb, 16), np.float32)
        api.check(api.lib().mmdx_skeleton_solve_motion(self.h, motion.h, model.h if model is not None else None, f.size,
                                                       f.ctypes.data, 0, out.ctypes.data))
        return out

    def solve_motion_device(self, motion: "BoneMotion", n_instances: int, frames_ptr, out_ptr, model=None) -> None:
        """Frame numbers and palettes resident in HBM; asynchronous on the model's stream."""
        api.check(api.lib().mmdx_skeleton_solve_motion(self.h, motion.h, model.h if model is not None else None, n_instances,
                                                       frames_ptr, FRAMES_ON_DEVICE | api.OUT_ON_DEVICE, out_ptr))

    def solve_device(self, n_instances: int, poses_ptr, out_ptr, model=None, weights_ptr=None, shared=False) -> None:
        """poses, palettes (and morph rates) resident in HBM; asynchronous on the model's stream."""
        flags = POSES_ON_DEVICE | api.OUT_ON_DEVICE | (api.WEIGHTS_ON_DEVICE if weights_ptr else 0) | \
            (api.WEIGHTS_SHARED if shared else 0)
        api.check(api.lib().mmdx_skeleton_solve_morphed(self.h, model.h if model is not None else None, n_instances,
                                                        poses_ptr, weights_ptr, flags, out_ptr))

    def close(self):
        if getattr(self, "h", None):
            api.lib().mmdx_skeleton_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
