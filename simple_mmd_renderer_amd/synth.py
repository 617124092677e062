"""Synthetic PMX-like models, morph weights and bone palettes for tests and bench.

The reference ships no .pmx/.vmd asset, so every input is generated here,
deterministically, following the recipe of SURVEY.md section 8d: positions U(-10,10)xU(0,20)xU(-2,2),
unit normals, deform-type mix BDEF1 20 % / BDEF2 50 % / BDEF4 25 % / SDEF 5 % assigned i.i.d.
per vertex, bone ids drawn from a window of 16 consecutive ids around v*NB/NV, BDEF4 weights
normalised U(0.01,1)^4, vertex morphs of K distinct vertices with offsets U(-0.5,0.5)^3, and rigid
per-frame palettes from a forward-kinematics pass over a random hierarchy.

A model is a `FlatModel`: the flat, SoA description that crosses the C ABI (include/mmdx.h
`mmdx_model_desc`) -- the same information the reference keeps in mmd::Model's vertex store,
SkinningOperator union and Morph lists (L/model/model.inl:21-104, :334-517, :719-726).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# PMX / libmmd tag values (L/model/model.inl:23-28, :488-498)
BDEF1, BDEF2, BDEF4, SDEF = 0, 1, 2, 3
MORPH_GROUP, MORPH_VERTEX, MORPH_BONE, MORPH_UV, MORPH_MATERIAL = 0, 1, 2, 3, 8


@dataclass
class FlatModel:
    positions: np.ndarray      # f32 [NV,3]
    normals: np.ndarray        # f32 [NV,3]
    uvs: np.ndarray            # f32 [NV,2]
    skin_type: np.ndarray      # i32 [NV]   raw tag (unknown values take the BDEF2 branch)
    bone_ids: np.ndarray       # i32 [NV,4]
    bone_weights: np.ndarray   # f32 [NV,4] (BDEF2/SDEF: [:,0])
    bone_pos: np.ndarray       # f32 [NB,3] rest positions
    bone_parent: np.ndarray    # i32 [NB]   -1 = root
    morph_type: np.ndarray     # i32 [NM]
    morph_off: np.ndarray      # u32 [NM+1] entry ranges
    morph_index: np.ndarray    # u32 [E]    vertex (vertex morph) / morph (group) / bone / ...
    morph_value: np.ndarray    # f32 [E,3]  offset (vertex morph) / (rate,0,0) (group) / translation
    sdef: Optional[np.ndarray] = None   # f32 [NV,9] C,R0,R1 (stored, never evaluated: see DESIGN.md)
    meta: dict = field(default_factory=dict)

    @property
    def nv(self) -> int:
        return int(self.positions.shape[0])

    @property
    def nb(self) -> int:
        return int(self.bone_pos.shape[0])

    @property
    def nm(self) -> int:
        return int(self.morph_type.shape[0])

    def copy(self) -> "FlatModel":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.copy() if isinstance(v, np.ndarray) else (dict(v) if isinstance(v, dict) else v)
        return FlatModel(**kw)


def checksum64(a: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum of an array's bytes (position-salted multiply-xor fold).
    Used for the checksum-only fixtures (600-frame config 1) and for full-size GPU runs where
    shipping every output would be too much."""
    b = np.ascontiguousarray(a).ravel().view(np.uint8)
    pad = (-b.size) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    idx = np.arange(1, w.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mixed = (w ^ (idx * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0x100000001B3)
        mixed ^= mixed >> np.uint64(29)
        return int(np.bitwise_xor.reduce(mixed) ^ np.uint64(w.size))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_morphs(rng, nv: int, nm: int, k: int):
    """nm vertex morphs, each of k distinct vertices (file order = draw order)."""
    k = min(k, nv)
    morph_type = np.full(nm, MORPH_VERTEX, np.int32)
    morph_off = (np.arange(nm + 1, dtype=np.uint64) * k).astype(np.uint32)
    idx = np.empty((nm, k), np.uint32)
    for m in range(nm):
        if k * 8 < nv:
            # rejection-free partial shuffle is overkill: draw with margin, keep first k uniques
            while True:
                c = rng.randint(0, nv, size=int(k * 1.2) + 16)
                _, first = np.unique(c, return_index=True)
                if first.size >= k:
                    idx[m] = c[np.sort(first)[:k]]
                    break
        else:
            idx[m] = rng.permutation(nv)[:k]
    val = _f32(rng.uniform(-0.5, 0.5, size=(nm * k, 3)))
    return morph_type, morph_off, idx.reshape(-1), val


def make_model(nv: int, nb: int, nm: int, k: int, seed: int,
               mix=(0.20, 0.50, 0.25, 0.05), window: int = 16) -> FlatModel:
    rng = np.random.RandomState(seed)
    pos = np.stack([rng.uniform(-10, 10, nv), rng.uniform(0, 20, nv), rng.uniform(-2, 2, nv)], 1)
    nrm = rng.uniform(-1, 1, size=(nv, 3))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6)
    uv = rng.uniform(0, 1, size=(nv, 2))
    bone_pos = np.stack([rng.uniform(-10, 10, nb), rng.uniform(0, 20, nb), rng.uniform(-1, 1, nb)], 1)
    parent = np.full(nb, -1, np.int32)
    for b in range(1, nb):
        parent[b] = rng.randint(0, b)
    u = rng.uniform(0, 1, nv)
    edges = np.cumsum(mix)
    skin_type = np.where(u < edges[0], BDEF1,
                         np.where(u < edges[1], BDEF2,
                                  np.where(u < edges[2], BDEF4, SDEF))).astype(np.int32)
    win = min(window, nb)
    centre = (np.arange(nv, dtype=np.int64) * nb) // max(nv, 1)
    lo = np.clip(centre - win // 2, 0, nb - win)
    ids = (lo[:, None] + rng.randint(0, win, size=(nv, 4))).astype(np.int32)
    w = np.zeros((nv, 4), np.float32)
    w2 = rng.uniform(0.01, 0.99, nv).astype(np.float32)
    w4 = rng.uniform(0.01, 1.0, size=(nv, 4))
    w4 = (w4 / w4.sum(1, keepdims=True)).astype(np.float32)
    is4 = skin_type == BDEF4
    is2 = (skin_type == BDEF2) | (skin_type == SDEF)
    w[is4] = w4[is4]
    w[is2, 0] = w2[is2]
    sdef = _f32(rng.uniform(-1, 1, size=(nv, 9)))
    mt, mo, mi, mv = make_morphs(rng, nv, nm, k)
    return FlatModel(_f32(pos), _f32(nrm), _f32(uv), skin_type, np.ascontiguousarray(ids), w,
                     _f32(bone_pos), parent, mt, mo, mi, mv, sdef,
                     meta=dict(nv=nv, nb=nb, nm=nm, k=k, seed=seed))


def morph_weights(nm: int, frame) -> np.ndarray:
    """weight_m(frame) = clip(0.5 + 0.5 sin(2 pi (frame/90 + m/NM)), 0, 1); frame may be an array."""
    frame = np.atleast_1d(np.asarray(frame, np.float64))
    m = np.arange(nm, dtype=np.float64)
    w = 0.5 + 0.5 * np.sin(2 * np.pi * (frame[:, None] / 90.0 + m[None, :] / max(nm, 1)))
    return np.clip(w, 0.0, 1.0).astype(np.float32)


def _axis_angle_to_mat(axis, angle):
    """Row-vector rotation matrices [...,3,3] (y = x . R)."""
    axis = axis / np.maximum(np.linalg.norm(axis, axis=-1, keepdims=True), 1e-9)
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    c, s = np.cos(angle), np.sin(angle)
    t = 1 - c
    r = np.empty(axis.shape[:-1] + (3, 3))
    r[..., 0, 0] = t * x * x + c
    r[..., 0, 1] = t * x * y + s * z
    r[..., 0, 2] = t * x * z - s * y
    r[..., 1, 0] = t * x * y - s * z
    r[..., 1, 1] = t * y * y + c
    r[..., 1, 2] = t * y * z + s * x
    r[..., 2, 0] = t * x * z + s * y
    r[..., 2, 1] = t * y * z - s * x
    r[..., 2, 2] = t * z * z + c
    return r


def make_palettes(model: FlatModel, frames, seed: int = 7) -> np.ndarray:
    """Rigid skinning matrices f32 [len(frames), NB, 16], row-vector convention, translation in
    elements 12..14 -- the layout of Poser::BoneImage::skinning_matrix_ (L/motion/poser.inl:96,
    L/util/math.inl:383-395).  skinning = translate(-rest) x world(bone), world = local x world(parent)
    (the shape of UpdateBoneSkinningMatrix, L/motion/poser_impl.inl:320-326; our own FK, float64)."""
    frames = np.atleast_1d(np.asarray(frames, np.float64))
    nf, nb = frames.shape[0], model.nb
    rng = np.random.RandomState(seed)
    axis = rng.uniform(-1, 1, size=(nb, 3))
    trans = rng.uniform(-0.2, 0.2, size=(nb, 3))
    b = np.arange(nb, dtype=np.float64)
    angle = 0.5 * np.sin(2 * np.pi * (frames[:, None] / 60.0 + b[None, :] / nb))   # [nf, nb]
    rot = _axis_angle_to_mat(np.broadcast_to(axis, (nf, nb, 3)), angle)             # [nf, nb,3,3]
    rest = model.bone_pos.astype(np.float64)
    world = np.zeros((nf, nb, 4, 4))
    out = np.zeros((nf, nb, 4, 4))
    for i in range(nb):
        p = int(model.bone_parent[i])
        local = np.zeros((nf, 4, 4))
        local[:, :3, :3] = rot[:, i]
        off = rest[i] - (rest[p] if p >= 0 else 0.0)
        local[:, 3, :3] = trans[i] + off
        local[:, 3, 3] = 1.0
        world[:, i] = local @ world[:, p] if p >= 0 else local
        g = np.eye(4)
        g[3, :3] = -rest[i]
        out[:, i] = g @ world[:, i]
    return np.ascontiguousarray(out.reshape(nf, nb, 16), dtype=np.float32)


# ---- the five BASELINE.json configs ----------------------------------------
CONFIGS = {
    "config1_20k": dict(nv=20000, nb=150, nm=30, k=500, seed=20001),
    "config2_50k": dict(nv=50000, nb=300, nm=200, k=2048, seed=50002),
    "config3_crowd": dict(nv=50000, nb=300, nm=200, k=2048, seed=50002, instances=1024),
    "config5_256k": dict(nv=262144, nb=512, nm=1024, k=4096, seed=262005),
}


def presort_by_class(m: FlatModel, tile: int = 512) -> FlatModel:
    """The same model with its vertices reordered, inside each `tile` consecutive vertices, by deform class
    (BDEF1, BDEF2/SDEF, BDEF4) and by the number of morph entries that touch them -- the order the plan gives a
    tile's lanes.  SURVEY 8d's "bucketed" variant: the engine's lane -> output permutation becomes the identity."""
    st = np.asarray(m.skin_type)
    cls = np.where(st == BDEF1, 0, np.where(st == BDEF4, 2, 1))
    vm = np.concatenate([np.asarray(m.morph_index[int(m.morph_off[k]):int(m.morph_off[k + 1])], np.int64)
                         for k in range(m.nm) if int(m.morph_type[k]) == MORPH_VERTEX] or [np.zeros(0, np.int64)])
    cnt = np.bincount(vm, minlength=m.nv)
    order = np.concatenate([vs[np.lexsort((vs, -cnt[vs], cls[vs]))]
                            for vs in (np.arange(t0, min(t0 + tile, m.nv)) for t0 in range(0, m.nv, tile))])
    new_of = np.empty(m.nv, np.int64)
    new_of[order] = np.arange(m.nv)
    s = m.copy()
    for f in ("positions", "normals", "uvs", "skin_type", "bone_ids", "bone_weights"):
        setattr(s, f, np.ascontiguousarray(getattr(m, f)[order]))
    if m.sdef is not None:
        s.sdef = np.ascontiguousarray(m.sdef[order])
    idx = np.asarray(m.morph_index).copy()
    for k in range(m.nm):
        if int(m.morph_type[k]) == MORPH_VERTEX:
            lo, hi = int(m.morph_off[k]), int(m.morph_off[k + 1])
            idx[lo:hi] = new_of[idx[lo:hi]]
    s.morph_index = idx.astype(np.uint32)
    s.meta = dict(m.meta, presorted=True)
    return s


def make_config(name: str) -> FlatModel:
    c = CONFIGS[name]
    m = make_model(c["nv"], c["nb"], c["nm"], c["k"], c["seed"])
    m.meta["config"] = name
    return m


# ---- synthetic rigs and bone motions ------------------------------
def make_skeleton(nb: int, seed: int, forward_parents: int = 0, post_physics: float = 0.0, levels: int = 1):
    """rest f32 [NB,3], parent i32 [NB] (-1 = root), level i32 [NB], flags u16 [NB].
    Like the model generator: parent(b) uniform in [0,b), bone 0 the root.  `forward_parents` bones get
    a parent with a LARGER index (legal in PMX; evaluated later unless its level says otherwise),
    `post_physics` is the fraction of bones flagged 0x1000, `levels` spreads transform levels."""
    rng = np.random.RandomState(seed)
    rest = np.stack([rng.uniform(-10, 10, nb), rng.uniform(0, 20, nb), rng.uniform(-1, 1, nb)], 1).astype(np.float32)
    parent = np.full(nb, -1, np.int32)
    for b in range(1, nb):
        parent[b] = rng.randint(0, b)
    for b in rng.choice(np.arange(1, max(nb - 1, 2)), min(forward_parents, max(nb - 2, 0)), replace=False):
        parent[b] = rng.randint(b + 1, nb)
    level = rng.randint(0, levels, nb).astype(np.int32)
    flags = np.where(rng.uniform(size=nb) < post_physics, 0x1000, 0).astype(np.uint16)
    return rest, parent, level, flags


def make_bone_keys(names, seed: int, keys_per: int = 6, span: int = 240, curved: float = 0.7):
    """VMD bone records (name, frame, t, q, interpolation[64]) for simple_mmd_renderer_amd.vmd.write_vmd.
    Rotations are unit quaternions from both hemispheres (NLerp's sign flip), a few left un-normalised;
    `curved` is the fraction of keys with random Bezier control bytes (the rest keep the linear default)."""
    rng = np.random.RandomState(seed)
    keys = []
    for n in names:
        for f in sorted(rng.choice(span, keys_per, replace=False)):
            q = rng.normal(size=4)
            q = q / np.linalg.norm(q) * (1.0 if rng.uniform() < 0.9 else rng.uniform(0.5, 1.5))
            t = rng.uniform(-2, 2, 3)
            ip = None
            if rng.uniform() < curved:
                ip = rng.randint(0, 128, 64).astype(np.uint8)
                if rng.uniform() < 0.2:                      # one channel exactly linear (x == y)
                    c = 16 * rng.randint(0, 4)
                    ip[c + 4], ip[c + 12] = ip[c + 0], ip[c + 8]
                ip = bytes(ip)
            keys.append((n, int(f), tuple(np.float32(t).tolist()), tuple(np.float32(q).tolist()), ip))
    return keys


def make_ik_rig(nb: int, seed: int, n_ik: int = 4, n_append: int = 4, post_physics: float = 0.1, levels: int = 2):
    """A random rig with CCD-IK chains and append (inherit) bones, as flat arrays:
    (rest, parent, level, flags, append_parent i32[NB], append_ratio f32[NB], ik) with
    ik = dict(target i32[NB] (-1 = no IK), loop i32[NB], angle f32[NB], link_off u32[NB+1],
    link_bone i32[L], link_limited u8[L], link_lo / link_hi f32[L,3]).
    IK bones sit near their target's rest position; links are the target's 1-3 nearest ancestors,
    target-side first (the PMX convention); limits cover the knee-style X-only hinge, two-axis zeros,
    all-zero (fixed) links and general boxes, so every Euler order / fix type of the reference occurs."""
    rng = np.random.RandomState(seed)
    rest, parent, level, flags = make_skeleton(nb, seed, 0, post_physics, levels)
    flags = flags.copy()
    depth = np.zeros(nb, np.int32)
    for b in range(1, nb):
        depth[b] = depth[parent[b]] + 1
    target = np.full(nb, -1, np.int32)
    loop = np.zeros(nb, np.int32)
    angle = np.zeros(nb, np.float32)
    links_of = {}
    used = set()
    cands = [b for b in range(nb) if depth[b] >= 2]
    rng.shuffle(cands)
    ik_bones = []
    for t in cands:
        if len(ik_bones) >= n_ik:
            break
        chain, c = [], int(parent[t])
        for _ in range(rng.randint(1, 4)):
            if c <= 0:
                break
            chain.append(c)
            c = int(parent[c])
        if not chain or t in used or any(x in used for x in chain):
            continue
        free = [b for b in range(1, nb) if b not in used and b != t and b not in chain
                and not np.any(parent == b)]                       # a leaf becomes the IK bone
        if not free:
            break
        ikb = int(free[rng.randint(len(free))])
        used.update(chain + [t, ikb])
        ik_bones.append(ikb)
        flags[ikb] |= 0x0020
        rest[ikb] = rest[t] + rng.uniform(-0.3, 0.3, 3).astype(np.float32)
        target[ikb] = t
        loop[ikb] = int(rng.choice([0, 3, 15, 40, 300, -1]))
        angle[ikb] = np.float32(rng.choice([0.03, 0.5, 2.0, 4.0]))
        links_of[ikb] = chain
    link_off, link_bone, link_limited, lo, hi = [0], [], [], [], []
    kinds = ["knee", "free", "box", "fixed", "yonly", "zonly", "swapped", "ybox"]
    for b in range(nb):
        for lb in links_of.get(b, []):
            kind = kinds[rng.randint(len(kinds))]
            link_bone.append(lb)
            link_limited.append(0 if kind == "free" else 1)
            a, z = np.zeros(3, np.float32), np.zeros(3, np.float32)
            if kind == "knee":
                a[0], z[0] = -3.1415927, -0.008726646
            elif kind == "box":
                a, z = rng.uniform(-1.5, -0.1, 3).astype(np.float32), rng.uniform(0.1, 1.5, 3).astype(np.float32)
            elif kind == "yonly":
                a[1], z[1] = -1.0, 2.0
            elif kind == "zonly":
                a[2], z[2] = -0.5, 0.5
            elif kind == "swapped":                               # lo > hi: the ctor takes min / max
                a, z = rng.uniform(0.1, 3.0, 3).astype(np.float32), rng.uniform(-3.0, -0.1, 3).astype(np.float32)
            elif kind == "ybox":                                  # x range beyond +-pi/2 -> XYZ order
                a[:] = (-2.0, -1.0, -3.0)
                z[:] = (2.0, 1.0, 3.0)
            lo.append(a)
            hi.append(z)
        link_off.append(len(link_bone))
    append_parent = np.full(nb, -1, np.int32)
    append_ratio = np.zeros(nb, np.float32)
    plain = [b for b in range(1, nb) if b not in used]
    rng.shuffle(plain)
    for b in plain[:n_append]:
        flags[b] |= int(rng.choice([0x0100, 0x0200, 0x0300]))
        append_parent[b] = int(rng.randint(0, nb)) if rng.uniform() < 0.9 else nb + 5   # out of range = no append
        append_ratio[b] = np.float32(rng.choice([0.5, 1.0, -1.0, 0.25, 2.0]))
    ik = dict(target=target, loop=loop, angle=angle, link_off=np.asarray(link_off, np.uint32),
              link_bone=np.asarray(link_bone, np.int32).reshape(-1),
              link_limited=np.asarray(link_limited, np.uint8).reshape(-1),
              link_lo=np.asarray(lo, np.float32).reshape(-1, 3), link_hi=np.asarray(hi, np.float32).reshape(-1, 3))
    return rest, parent, level, flags, append_parent, append_ratio, ik


def make_nested_ik_rig(nb: int, seed: int, n_ik: int = 4, n_append: int = 3):
    """make_ik_rig with NESTED IK added (the reference's UpdateBoneTransform recurses into a link or target that is
    itself an IK bone, poser_impl.inl:196-206): IK bone A gets IK bone B as an extra, root-most link (B's whole chain is
    solved when A's solve places its links), and a new IK bone C takes IK bone B... as its TARGET (B's chain is solved
    every time C's loop re-places its target).  Loop counts are kept small: the cost multiplies."""
    rest, parent, level, flags, ap, ar, ik = make_ik_rig(nb, seed, n_ik=n_ik, n_append=n_append, post_physics=0.0, levels=1)
    rng = np.random.RandomState(seed + 7000)
    iks = [b for b in range(nb) if flags[b] & 0x0020]
    assert len(iks) >= 3
    a_, b_, c_ = iks[0], iks[1], iks[2]
    loop = ik["loop"].copy()
    loop[iks] = [int(rng.choice([2, 3, 5])) for _ in iks]
    loop[b_] = 4
    links = {b: list(ik["link_bone"][ik["link_off"][b]:ik["link_off"][b + 1]]) for b in range(nb)}
    lim = {b: list(ik["link_limited"][ik["link_off"][b]:ik["link_off"][b + 1]]) for b in range(nb)}
    lo = {b: [x for x in ik["link_lo"][ik["link_off"][b]:ik["link_off"][b + 1]]] for b in range(nb)}
    hi = {b: [x for x in ik["link_hi"][ik["link_off"][b]:ik["link_off"][b + 1]]] for b in range(nb)}
    # (1) B becomes a link of A
    links[a_].append(b_); lim[a_].append(0); lo[a_].append(np.zeros(3, np.float32)); hi[a_].append(np.zeros(3, np.float32))
    # (2) C's target becomes IK bone B; C keeps its links
    target = ik["target"].copy()
    target[c_] = b_
    link_off, link_bone, link_limited, llo, lhi = [0], [], [], [], []
    for b in range(nb):
        link_bone += links[b]; link_limited += lim[b]; llo += lo[b]; lhi += hi[b]
        link_off.append(len(link_bone))
    ik2 = dict(target=target, loop=loop, angle=ik["angle"], link_off=np.asarray(link_off, np.uint32),
               link_bone=np.asarray(link_bone, np.int32).reshape(-1), link_limited=np.asarray(link_limited, np.uint8).reshape(-1),
               link_lo=np.asarray(llo, np.float32).reshape(-1, 3), link_hi=np.asarray(lhi, np.float32).reshape(-1, 3))
    return rest, parent, level, flags, ap, ar, ik2


def make_bone_morphs(nb: int, seed: int, n_bone: int = 5, n_group: int = 3, n_other: int = 2):
    """A morph table with bone morphs (translation + rotation, several bones each, a bone hit by several morphs),
    group morphs over them (depth 2, a bone morph reached directly and through a group) and a few vertex / uv
    morphs that the bone solve must ignore.  dict(type, offset, index, value, rotation)."""
    rng = np.random.RandomState(seed)
    types, off, index, value, rot = [], [0], [], [], []
    nm = n_other + n_bone + n_group
    for _ in range(n_other):                                   # ignored types first (indices must still line up)
        types.append(int(rng.choice([1, 3, 8])))
        for _ in range(rng.randint(0, 3)):
            index.append(0); value.append([0.1, 0.2, 0.3]); rot.append([0, 0, 0, 1])
        off.append(len(index))
    for _ in range(n_bone):
        types.append(2)
        for _ in range(rng.randint(1, 4)):
            q = rng.normal(size=4)
            q = q / np.linalg.norm(q) * (1.0 if rng.uniform() < 0.8 else 0.9)
            if rng.uniform() < 0.15:
                q = np.array([0, 0, 0, 1.0])                   # pure translation morph
            index.append(int(rng.randint(0, min(nb, 6)) if rng.uniform() < 0.5 else rng.randint(0, nb)))
            value.append(rng.uniform(-0.5, 0.5, 3)); rot.append(q)
        off.append(len(index))
    for g in range(n_group):
        types.append(0)
        lo = n_other if g < n_group - 1 else n_other + n_bone    # the last group also nests the other groups
        for _ in range(rng.randint(1, 4)):
            index.append(int(rng.randint(lo, n_other + n_bone + g) if g == n_group - 1 and g > 0
                             else rng.randint(n_other, n_other + n_bone)))
            value.append([float(rng.choice([0.5, 1.0, 1e-4, 2.0])), 0, 0]); rot.append([0, 0, 0, 1])
        off.append(len(index))
    assert len(types) == nm
    return dict(type=np.asarray(types, np.int32), offset=np.asarray(off, np.uint32), index=np.asarray(index, np.uint32),
                value=np.asarray(value, np.float32).reshape(-1, 3), rotation=np.asarray(rot, np.float32).reshape(-1, 4))
