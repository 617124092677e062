"""PMD 1.0 files: a writer for synthetic fixtures (the reference ships no asset) -- the reader is the C-ABI
loader csrc/pmd.cpp, reached through simple_mmd_renderer_amd.pmx.load_pmd.

Record layouts as the reference's reader consumes them (L/reader/interprete/pmd_types.inl:17-125,
L/reader/pmd_reader_impl.inl:16-420): 283-byte header, 38-byte vertices, u16 indices, 70-byte materials,
39-byte bones, IK records (11 bytes + u16 chain), morphs (25-byte header + 16-byte entries), display lists.
Files end after the display lists (the "legacy 3.0" layout the reader accepts) unless `extended` asks for
the English-name block, toon names, rigid bodies and joints behind them.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .synth import FlatModel, make_model

BONE_ROTATE, BONE_ROTATE_TRANSLATE, BONE_IK, BONE_UNKNOWN, BONE_IK_LINK, BONE_ROTATE_EFFECT, BONE_IK_TO, \
    BONE_INVISIBLE, BONE_TWIST, BONE_ROTATE_RATIO = range(10)


def _sjis(name: str, size: int) -> bytes:
    b = name.encode("shift_jis")
    if len(b) > size:
        raise ValueError("name too long for a %d-byte Shift-JIS field: %r" % (size, name))
    return b + b"\0" * (size - len(b))


@dataclass
class PmdBone:
    name: str
    parent: int                 # -1 = none
    position: Sequence[float]
    type: int = BONE_ROTATE
    child: int = 0
    ik_number: int = 0          # type 5: append source bone; type 9: append ratio in percent


@dataclass
class PmdIk:
    bone: int
    target: int
    chain: List[int]
    loop: int = 40
    angle: float = 0.5          # the reader multiplies by 4


@dataclass
class PmdWriteOptions:
    bones: Optional[List[PmdBone]] = None      # default: one rotate bone per model bone, names ボーンN
    iks: List[PmdIk] = field(default_factory=list)
    morph_names: Optional[List[str]] = None
    base_morph: bool = True                    # write a "system" base morph and index the others through it
    n_materials: int = 2
    extended: bool = False
    model_name: str = "合成PMD"


def write_pmd(m: FlatModel, opt: Optional[PmdWriteOptions] = None) -> bytes:
    """Every vertex becomes a BDEF2 record (bone_ids[:, :2], weight rounded to a byte of percent); morphs must
    all be vertex morphs.  With `base_morph` a category-0 morph listing every morphed vertex once is written
    first and the others refer to its entries, as real PMD files do."""
    opt = opt or PmdWriteOptions()
    nv, nb = m.nv, m.nb
    out = bytearray()
    out += b"Pmd" + struct.pack("<f", 1.0) + _sjis(opt.model_name, 20) + _sjis("synthetic", 256)
    out += struct.pack("<I", nv)
    w100 = np.clip(np.rint(np.asarray(m.bone_weights[:, 0], np.float64) * 100), 0, 100).astype(np.uint8)
    for i in range(nv):
        out += np.asarray(m.positions[i], "<f4").tobytes() + np.asarray(m.normals[i], "<f4").tobytes()
        out += np.asarray(m.uvs[i], "<f4").tobytes()
        out += struct.pack("<hhBB", int(m.bone_ids[i, 0]), int(m.bone_ids[i, 1]), int(w100[i]), i % 2)
    tris = np.arange(nv - nv % 3, dtype=np.uint16)
    out += struct.pack("<I", tris.size) + tris.astype("<u2").tobytes()
    out += struct.pack("<I", opt.n_materials)
    per = tris.size // 3 // max(opt.n_materials, 1) * 3
    for k in range(opt.n_materials):
        cnt = per if k + 1 < opt.n_materials else tris.size - per * (opt.n_materials - 1)
        out += struct.pack("<4f", 0.8, 0.7, 0.6, 1.0) + struct.pack("<f", 5.0) + struct.pack("<3f", 0.2, 0.2, 0.2)
        out += struct.pack("<3f", 0.4, 0.4, 0.4) + struct.pack("<bB", k % 10, 1) + struct.pack("<I", cnt) + b"\0" * 20
    bones = opt.bones or [PmdBone("ボーン%d" % b, int(m.bone_parent[b]), m.bone_pos[b]) for b in range(nb)]
    out += struct.pack("<H", len(bones))
    for b in bones:
        out += _sjis(b.name, 20) + struct.pack("<hhBh", b.parent, b.child, b.type, b.ik_number)
        out += np.asarray(b.position, "<f4").tobytes()
    out += struct.pack("<H", len(opt.iks))
    for k in opt.iks:
        out += struct.pack("<hhBHf", k.bone, k.target, len(k.chain), k.loop, k.angle)
        out += np.asarray(k.chain, "<u2").tobytes()
    # morphs
    nm = m.nm
    names = opt.morph_names or ["モーフ%d" % k for k in range(nm)]
    assert all(int(t) == 1 for t in m.morph_type), "PMD holds vertex morphs only"
    if opt.base_morph and nm:
        verts = np.unique(np.asarray(m.morph_index))
        slot = {int(v): i for i, v in enumerate(verts)}
        out += struct.pack("<H", nm + 1)
        out += _sjis("base", 20) + struct.pack("<IB", verts.size, 0)
        for v in verts:
            out += struct.pack("<I", int(v)) + np.asarray(m.positions[int(v)], "<f4").tobytes()
    else:
        slot = None
        out += struct.pack("<H", nm)
    for k in range(nm):
        lo, hi = int(m.morph_off[k]), int(m.morph_off[k + 1])
        out += _sjis(names[k], 20) + struct.pack("<IB", hi - lo, 1 + k % 4)
        for e in range(lo, hi):
            v = int(m.morph_index[e])
            out += struct.pack("<I", slot[v] if slot is not None else v) + np.asarray(m.morph_value[e], "<f4").tobytes()
    # display lists
    out += struct.pack("<B", 0) + struct.pack("<B", 1) + _sjis("枠", 50) + struct.pack("<I", 1) + struct.pack("<HB", 0, 1)
    if opt.extended:
        out += struct.pack("<B", 1) + _sjis("model", 20) + _sjis("english", 256)
        out += b"".join(_sjis("bone%d" % i, 20) for i in range(len(bones)))
        n_written = nm + (1 if opt.base_morph and nm else 0)
        out += b"".join(_sjis("morph%d" % i, 20) for i in range(max(n_written - 1, 0)))
        out += _sjis("frame", 50)
        out += b"".join(_sjis("toon%02d.bmp" % (i + 1), 100) for i in range(10))
        out += struct.pack("<I", 0) + struct.pack("<I", 0)      # no rigid bodies, no joints
    return bytes(out)


def make_rigged_pmd(seed, nv=240, nb=16, extended=False, base_morph=True, knee=False):
    """A PMD file with every bone type the reader converts, two IK records on one bone (-> an appended bone),
    an IK bone whose children inherit its transform level, and vertex morphs behind a base morph."""
    m = make_model(nv, nb, 4, 30, seed=seed)
    m.bone_parent = np.asarray([-1, 0] + [b - 2 for b in range(2, nb)], m.bone_parent.dtype)   # two interleaved chains
    bones = [PmdBone("ボーン%d" % b, int(m.bone_parent[b]), m.bone_pos[b]) for b in range(nb)]

    def anc(b, n):
        out, p = [], int(m.bone_parent[b])
        while p > 0 and len(out) < n:
            out.append(p)
            p = int(m.bone_parent[p])
        return out
    deep = [b for b in range(nb - 5) if len(anc(b, 2)) == 2 and max(anc(b, 2)) < nb - 5]   # plain bones only
    t0, t1 = deep[0], deep[-1]
    ikb = nb - 1
    bones[ikb].type = BONE_IK
    bones[nb - 2].type, bones[nb - 2].ik_number = BONE_ROTATE_EFFECT, 3
    bones[nb - 3].type, bones[nb - 3].child, bones[nb - 3].ik_number = BONE_ROTATE_RATIO, 2, 50
    bones[nb - 4].type = BONE_TWIST
    bones[nb - 5].type = BONE_IK                     # IK type without an IK record: no links, target 0
    bones[1].type = BONE_ROTATE_TRANSLATE
    if knee:
        bones[anc(t0, 1)[0]].name = "左ひざ"
    iks = [PmdIk(ikb, t0, anc(t0, 2), loop=20, angle=0.3)]
    if t1 != t0 and anc(t1, 2)[0] != anc(t0, 2)[0]:
        iks.append(PmdIk(ikb, t1, anc(t1, 2), loop=5, angle=0.5))
    opt = PmdWriteOptions(bones=bones, iks=iks, extended=extended, base_morph=base_morph,
                              morph_names=["あ", "い", "う", "まばたき"])
    return write_pmd(m, opt), m, opt
