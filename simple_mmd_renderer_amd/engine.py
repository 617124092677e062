"""Host-side Python front-end of the C ABI (include/mmdx.h): model handles, device buffers and the
deform calls.  Thin by design -- every number is produced by libmmdx.so on the GPU.

`Poser` mirrors the slice of the reference's `mmd::Poser` that sits on the hot path
(L/motion/poser.inl:17-43): SetMorphPose / ResetPosing / Deform / pose_image, plus the palette
injection the reference performs through PhysicsReactor::GetPoserBoneImage
(L/motion/physics.inl:32-40) and the viewer's UpdateDeformedVertices (main.cpp:821-863).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi as api
from .synth import FlatModel

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None and a.size else None


def device_count() -> int:
    n = C.c_int32(0)
    st = api.lib().mmdx_device_count(C.byref(n))
    return int(n.value) if st == api.OK else 0


def device_select(ordinal: int) -> None:
    api.check(api.lib().mmdx_device_select(ordinal))


def device_name(ordinal: int = 0) -> str:
    buf = C.create_string_buffer(256)
    api.check(api.lib().mmdx_device_name(ordinal, buf, 256))
    return buf.value.decode()


def device_synchronize() -> None:
    api.check(api.lib().mmdx_device_synchronize())


class DeviceBuffer:
    """A hipMalloc'ed range owned through the C ABI's memory helpers."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        api.check(api.lib().mmdx_device_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def adopt(cls, ptr: int, nbytes: int) -> "DeviceBuffer":
        """Wrap a device range allocated by the library (freed with mmdx_device_free like any other)."""
        b = cls.__new__(cls)
        b.nbytes, b.ptr = int(nbytes), int(ptr)
        return b

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.upload(a)
        return b

    def upload(self, a: np.ndarray, offset: int = 0) -> None:
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        api.check(api.lib().mmdx_memcpy_h2d(self.ptr + offset, a.ctypes.data, a.nbytes))

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        api.check(api.lib().mmdx_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def memset(self, value: int = 0) -> None:
        api.check(api.lib().mmdx_device_memset(self.ptr, value, self.nbytes))

    def free(self) -> None:
        if self.ptr:
            api.lib().mmdx_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A numpy view of page-locked host memory (mmdx_host_malloc): frame buffers that cross PCIe by DMA."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        api.check(api.lib().mmdx_host_malloc(C.byref(p), nbytes))
        self.ptr = p.value
        buf = (C.c_char * max(nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self) -> None:
        if self.ptr:
            self.array = None
            api.lib().mmdx_host_free(self.ptr)
            self.ptr = None


class Graph:
    """mmdx_graph_t: a recorded sequence of library calls on one model's stream, replayed with one submission."""

    def __init__(self, handle):
        self.h = handle

    def launch(self) -> None:
        api.check(api.lib().mmdx_graph_launch(self.h))

    def close(self) -> None:
        if self.h:
            api.lib().mmdx_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeformModel:
    """mmdx_model_t: the model compiled to the kernels' HBM layout, resident on one GPU."""

    def __init__(self, flat: FlatModel, normalize: bool = True, f16_positions: bool = False,
                 host_only: bool = False, fast_math: bool = False, tile_order: bool = False):
        """fast_math: MMDX_CREATE_FAST_MATH -- contracted multiply-adds, results within the stated tolerance of the
        reference's instead of bit-identical (include/mmdx.h).  tile_order: MMDX_CREATE_TILE_ORDER -- outputs in the
        engine's vertex order (vertex_order() gives the permutation), same values."""
        self.flat = flat
        self.f16 = bool(f16_positions)
        self._keep = dict(
            positions=_c(flat.positions, np.float32), normals=_c(flat.normals, np.float32),
            uvs=_c(flat.uvs, np.float32), skin_type=_c(flat.skin_type, np.int32),
            bone_ids=_c(flat.bone_ids, np.int32), bone_weights=_c(flat.bone_weights, np.float32),
            sdef=_c(flat.sdef, np.float32) if flat.sdef is not None else None,
            bone_parent=_c(flat.bone_parent, np.int32), morph_type=_c(flat.morph_type, np.int32),
            morph_offset=_c(flat.morph_off, np.uint32), morph_index=_c(flat.morph_index, np.uint32),
            morph_value=_c(flat.morph_value, np.float32))
        k = self._keep
        d = api.ModelDesc()
        d.struct_size = C.sizeof(api.ModelDesc)
        d.flags = ((api.CREATE_NORMALIZE if normalize else 0) |
                   (api.CREATE_F16_POSITIONS if f16_positions else 0) |
                   (api.CREATE_HOST_ONLY if host_only else 0) |
                   (api.CREATE_FAST_MATH if fast_math else 0) |
                   (api.CREATE_TILE_ORDER if tile_order else 0))
        d.n_vertices, d.n_bones, d.n_morphs = flat.nv, flat.nb, flat.nm
        d.positions = _ptr(k["positions"], _f32p)
        d.normals = _ptr(k["normals"], _f32p)
        d.uvs = _ptr(k["uvs"], _f32p)
        d.skin_type = _ptr(k["skin_type"], _i32p)
        d.bone_ids = _ptr(k["bone_ids"], _i32p)
        d.bone_weights = _ptr(k["bone_weights"], _f32p)
        d.sdef_params = _ptr(k["sdef"], _f32p) if k["sdef"] is not None else None
        d.bone_parent = _ptr(k["bone_parent"], _i32p)
        d.morph_type = _ptr(k["morph_type"], _i32p)
        d.morph_offset = _ptr(k["morph_offset"], _u32p)
        d.morph_index = _ptr(k["morph_index"], _u32p)
        d.morph_value = _ptr(k["morph_value"], _f32p)
        h = C.c_void_p()
        api.check(api.lib().mmdx_model_create(C.byref(d), C.byref(h)))
        self.h = h
        self._keep = None  # borrowed only for the duration of the call
        self.info = self._get_info()
        self.nv, self.nb, self.nm, self.ns = (self.info.n_vertices, self.info.n_bones,
                                              self.info.n_morphs, self.info.n_slots)

    # -- lifetime -----------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "h", None):
            api.lib().mmdx_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- queries ------------------------------------------------------------------------------
    def _get_info(self) -> api.ModelInfo:
        info = api.ModelInfo()
        info.struct_size = C.sizeof(api.ModelInfo)
        api.check(api.lib().mmdx_model_get_info(self.h, C.byref(info)))
        return info

    def get_skin(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        t = np.empty(self.nv, np.int32)
        ids = np.empty((self.nv, 4), np.int32)
        w = np.empty((self.nv, 4), np.float32)
        api.check(api.lib().mmdx_model_get_skin(self.h, _ptr(t, _i32p), _ptr(ids, _i32p), _ptr(w, _f32p)))
        return t, ids, w

    def vertex_order(self) -> Tuple[np.ndarray, np.ndarray]:
        """(engine_to_original, original_to_engine): position e of a tile-order model's outputs holds file vertex
        engine_to_original[e]."""
        e2o, o2e = np.empty(self.nv, np.uint32), np.empty(self.nv, np.uint32)
        u32p = C.POINTER(C.c_uint32)
        api.check(api.lib().mmdx_model_get_vertex_order(self.h, _ptr(e2o, u32p), _ptr(o2e, u32p)))
        return e2o, o2e

    def slot_weights(self, rates) -> np.ndarray:
        rates = _c(rates, np.float32)
        assert rates.shape == (self.nm,)
        out = np.zeros(max(self.ns, 1), np.float32)
        api.check(api.lib().mmdx_model_slot_weights(self.h, _ptr(rates, _f32p), _ptr(out, _f32p)))
        return out[:self.ns]

    # -- the hot path -------------------------------------------------------------------------
    def deform(self, rates, palette) -> Tuple[np.ndarray, np.ndarray]:
        """mmdx_deform: one instance, host in / host out -> pose_image (pos, nrm) f32 [NV,3]."""
        rates = _c(rates, np.float32).reshape(-1)
        pal = _c(palette, np.float32).reshape(-1)
        assert rates.size == self.nm and pal.size == self.nb * 16
        pos = np.empty((self.nv, 3), np.float32)
        nrm = np.empty((self.nv, 3), np.float32)
        api.check(api.lib().mmdx_deform(self.h, _ptr(rates, _f32p), _ptr(pal, _f32p),
                                        _ptr(pos, _f32p), _ptr(nrm, _f32p)))
        return pos, nrm

    def deform_vertex32(self, rates, palette, pos_scale: float = 0.1) -> np.ndarray:
        rates = _c(rates, np.float32).reshape(-1)
        pal = _c(palette, np.float32).reshape(-1)
        assert rates.size == self.nm and pal.size == self.nb * 16
        out = np.empty((self.nv, 8), np.float32)
        api.check(api.lib().mmdx_deform_vertex32(self.h, _ptr(rates, _f32p), _ptr(pal, _f32p),
                                                 C.c_float(pos_scale), out.ctypes.data))
        return out

    def out_sizes(self, layout: int, ni: int) -> Tuple[int, int]:
        nvi = ni * self.nv
        if layout == api.OUT_SOA:
            return nvi * 12, nvi * 12
        if layout == api.OUT_VERTEX32:
            return nvi * 32, 0
        return nvi * 6, nvi * 12

    def alloc_outputs(self, layout: int, ni: int, max_tries: int = 16):
        """The crowd's output arrays through mmdx_crowd_output_alloc (placement-aware on MI355X).
        Returns (a, b or None, info dict); info["store_flags"] is the MMDX_OUT_STORES_* hint to OR into the flags of the crowd
        calls that write these arrays (the library itself remembers nothing about them)."""
        class _Info(C.Structure):
            _fields_ = [("struct_size", C.c_uint32), ("tries", C.c_uint32), ("probed", C.c_uint32),
                        ("store_GBs", C.c_float), ("fill_GBs", C.c_float), ("store_flags", C.c_uint32)]
        info = _Info()
        info.struct_size = C.sizeof(_Info)
        pa, pb = C.c_void_p(), C.c_void_p()
        api.check(api.lib().mmdx_crowd_output_alloc(self.h, ni, layout, max_tries, C.byref(pa), C.byref(pb),
                                                    C.byref(info)))
        sa, sb = self.out_sizes(layout, ni)
        a = DeviceBuffer.adopt(pa.value, sa)
        b = DeviceBuffer.adopt(pb.value, sb) if pb.value else None
        return a, b, {"tries": info.tries, "probed": bool(info.probed), "store_GBs": info.store_GBs,
                      "fill_GBs": info.fill_GBs, "store_flags": int(info.store_flags)}

    def deform_batched_raw(self, ni: int, weights_ptr, palettes_ptr, out_a_ptr, out_b_ptr, layout: int,
                           flags: int, pos_scale: float = 1.0) -> None:
        a = api.DeformArgs()
        a.struct_size = C.sizeof(api.DeformArgs)
        a.flags, a.n_instances, a.out_layout = flags, ni, layout
        a.morph_weights, a.palettes = weights_ptr, palettes_ptr
        a.out_a, a.out_b = out_a_ptr, out_b_ptr
        a.pos_scale = pos_scale
        api.check(api.lib().mmdx_deform_batched(self.h, C.byref(a)))

    def deform_batched(self, weights, palettes, layout: int = api.OUT_SOA, shared_weights: bool = False,
                       pos_scale: float = 1.0):
        """Host arrays in, host arrays out (copies + sync inside the call).
        weights [NI,NM] (or [NM] with shared_weights); palettes [NI,NB,16]."""
        pal = _c(palettes, np.float32).reshape(-1, self.nb, 16)
        ni = pal.shape[0]
        w = _c(weights, np.float32)
        if self.nm:
            w = w.reshape(self.nm) if shared_weights else w.reshape(ni, self.nm)
        flags = api.WEIGHTS_SHARED if shared_weights else 0
        if layout == api.OUT_SOA:
            oa = np.empty((ni, self.nv, 3), np.float32)
            ob = np.empty((ni, self.nv, 3), np.float32)
        elif layout == api.OUT_VERTEX32:
            oa = np.empty((ni, self.nv, 8), np.float32)
            ob = None
        else:
            oa = np.empty((ni, self.nv, 3), np.float16)
            ob = np.empty((ni, self.nv, 3), np.float32)
        self.deform_batched_raw(ni, w.ctypes.data if w.size else None, pal.ctypes.data, oa.ctypes.data,
                                ob.ctypes.data if ob is not None else None, layout, flags, pos_scale)
        return (oa, ob) if ob is not None else oa

    def sync(self) -> None:
        api.check(api.lib().mmdx_sync(self.h))

    def morph_pass_stats(self) -> Tuple[int, int, int]:
        """(launches that walked the morph table, launches that found the rates unchanged on the device and skipped the walk,
        calls whose host-side comparison skipped the launch) of this model's shared morph passes."""
        w, d, h = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        api.check(api.lib().mmdx_debug_morph_pass_stats(self.h, C.byref(w), C.byref(d), C.byref(h)))
        return w.value, d.value, h.value

    def last_store_policy(self) -> str:
        """Store flavour the kernel of the last crowd call ran with: 'nt' (cached, non-temporal) or 'sc1 nt' (write-through)."""
        wt = C.c_int32(0)
        api.check(api.lib().mmdx_debug_last_store_policy(self.h, C.byref(wt)))
        return "sc1 nt" if wt.value else "nt"

    def timer_start(self) -> None:
        api.check(api.lib().mmdx_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        api.check(api.lib().mmdx_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    # -- HIP-graph replay of the device work of a frame ----------------------------------------------
    def graph_begin(self) -> None:
        api.check(api.lib().mmdx_graph_begin(self.h))

    def graph_end(self) -> "Graph":
        g = C.c_void_p()
        api.check(api.lib().mmdx_graph_end(self.h, C.byref(g)))
        return Graph(g)

    def profile_enable(self, on, every: int = 1) -> None:
        """Events around the morph kernels and the skinning kernel of every call, or of every `every`-th call."""
        api.check(api.lib().mmdx_profile_enable(self.h, max(int(every), 1) if on else 0))

    def profile_collect(self) -> Tuple[int, float, float]:
        """(calls, skin kernel ms total, morph pass ms total) since profile_enable / last collect."""
        n, s, m = C.c_uint32(0), C.c_float(0), C.c_float(0)
        api.check(api.lib().mmdx_profile_collect(self.h, C.byref(n), C.byref(s), C.byref(m)))
        return int(n.value), float(s.value), float(m.value)


class PoseImage:
    def __init__(self, nv: int):
        self.coordinates = np.zeros((nv, 3), np.float32)
        self.normals = np.zeros((nv, 3), np.float32)


class Poser:
    """The hot-path slice of mmd::Poser, GPU-backed.  The host keeps doing what the reference's
    frame() does upstream (motion seek, bone solve, physics) and hands over morph rates and the
    finished bone palette; Deform() then fills pose_image exactly as Poser::Deform() would."""

    def __init__(self, model: FlatModel, normalize: bool = True):
        self.model = model
        self._dm = DeformModel(model, normalize=normalize)
        self.pose_image = PoseImage(model.nv)
        self._rates = np.zeros(model.nm, np.float32)
        self._palette = np.tile(np.eye(4, dtype=np.float32).reshape(16), (model.nb, 1))
        # the reference's constructor ends with ResetPosing(); Deform()  (poser_impl.inl:126-127)
        self.Deform()

    def ResetPosing(self) -> None:
        self._rates[:] = 0.0

    def SetMorphPose(self, index: int, weight: float) -> None:
        self._rates[index] = np.float32(weight)

    def SetSkinningMatrix(self, bone: int, matrix16) -> None:
        self._palette[bone] = np.asarray(matrix16, np.float32).reshape(16)

    def SetSkinningMatrices(self, palette) -> None:
        self._palette[:] = np.asarray(palette, np.float32).reshape(self.model.nb, 16)

    def Deform(self) -> None:
        pos, nrm = self._dm.deform(self._rates, self._palette)
        self.pose_image.coordinates, self.pose_image.normals = pos, nrm

    def UpdateDeformedVertices(self, pos_scale: float = 0.1) -> np.ndarray:
        """Deform + repack in one pass on the GPU: the viewer's 32-byte vertex stream."""
        return self._dm.deform_vertex32(self._rates, self._palette, pos_scale)

    def close(self) -> None:
        self._dm.close()
