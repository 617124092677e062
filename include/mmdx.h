/* mmdx.h -- C ABI of the MI355X-native per-frame deformation engine for MMD/PMX models.
 *
 * This is the drop-in boundary for ONE path of CU-Production/simple_mmd_renderer: the per-vertex
 * morph-target blend + BDEF1/2/4/SDEF linear-blend skinning that the vendored libmmd evaluates on
 * one CPU thread every frame, plus the viewer's repack into its 32-byte vertex stream:
 *
 *     reference (L/ = 3rd_party/libmmd/include/mmd/)              replaced by
 *     ----------------------------------------------------------  -------------------------------
 *     mmd::Model vertex/skin/morph stores  L/model/model.inl:21-104,  mmdx_model_create()
 *         :334-517, :719-726; Model::Normalize L/model/model_impl.inl:406-452
 *     Poser::SetMorphPose + morph part of PrePhysicsPosing +        morph_weights argument of
 *         UpdateMorphTransform  L/motion/poser_impl.inl:328-346,       mmdx_deform*()
 *         :362-365, :384-386, :478-480
 *     Poser::BoneImage::skinning_matrix_ (the palette, read through  palette argument of
 *         PhysicsReactor::GetPoserBoneImage L/motion/physics.inl:32-40) mmdx_deform*()
 *     Poser::Deform -> Poser::pose_image  L/motion/poser_impl.inl:396-461,  mmdx_deform*(), layout
 *         L/motion/poser.inl:17-20                                       MMDX_OUT_SOA
 *     UpdateDeformedVertices (struct Vertex, x0.1 scale, uv copy)    mmdx_deform*(), layout
 *         main.cpp:50-54, :821-863                                       MMDX_OUT_VERTEX32
 *     call site frame() main.cpp:1821 + :1824                        one mmdx_deform() call
 *
 * Everything else of the viewer (bone solve, VMD evaluation, Bullet, sokol draw loop) is untouched:
 * the host keeps producing morph rates and the bone palette and hands them over per frame.
 *
 * Conventions
 *   - Matrices: row-vector, row-major float[16], y = x * M, translation in elements 12..14
 *     (L/util/math.inl:383-395; identical to an OpenGL column-major float[16]).
 *   - No exceptions cross this boundary: every call returns an mmdx_status; the text of the last
 *     error on the calling thread is available from mmdx_last_error_string().
 *   - All indices are validated in mmdx_model_create(); mmdx_deform*() can only fail on argument
 *     errors or HIP runtime errors.
 *   - A model handle owns its device allocations and one HIP stream; calls on one handle are not
 *     re-entrant; different handles (and devices) may be driven from different host threads.
 *   - Results are bit-identical to the reference's CPU path (the kernels are compiled with
 *     -ffp-contract=off and keep the reference's operation order); see DESIGN.md.
 *   - Measurement and A/B helpers that no reference interface corresponds to (event timers, per-kernel
 *     profiling, copy / fill / store-pattern ceilings, launch-shape overrides) live in mmdx_bench.h.
 */
#ifndef MMDX_H_INCLUDED
#define MMDX_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MMDX_API __attribute__((visibility("default")))
#else
#define MMDX_API
#endif

/* 2: mmdx_skeleton_desc.create_flags (was reserved0), the physics seam and graph entry points, unknown flag bits are
 *    rejected, bench / debug entry points moved to mmdx_bench.h (same library). */
/* 3: mmdx_placement_info.store_flags (the library keeps no table of array addresses any more: the caller carries the probe's
 *    verdict into mmdx_deform_args.flags); shared morph rates that did not change are detected by the library itself. */
#define MMDX_ABI_VERSION 3u

typedef int32_t mmdx_status;
enum {
    MMDX_OK = 0,
    MMDX_ERR_INVALID_ARGUMENT = 1, /* NULL / size / flag problems                                  */
    MMDX_ERR_BAD_INDEX = 2,        /* a bone, vertex or morph index in the model is out of range   */
    MMDX_ERR_NO_DEVICE = 3,        /* no usable HIP device (the product has no CPU fallback)       */
    MMDX_ERR_HIP = 4,              /* a HIP runtime call failed; see mmdx_last_error_string()      */
    MMDX_ERR_OUT_OF_MEMORY = 5,
    MMDX_ERR_UNSUPPORTED = 6       /* e.g. morph group cycle, > 65535 bones in one vertex tile     */
};

/* Deform-type tags = PMX / libmmd values (L/model/model.inl:23-28).  Any other value is legal and
 * takes the reference's `default:` branch, i.e. is evaluated as BDEF2 (poser_impl.inl:417-418).
 * SDEF is evaluated as BDEF2 as well: the reference's spherical-deform code is commented out
 * ("UNDONE", poser_impl.inl:438-458). */
enum { MMDX_SKIN_BDEF1 = 0, MMDX_SKIN_BDEF2 = 1, MMDX_SKIN_BDEF4 = 2, MMDX_SKIN_SDEF = 3 };

/* Morph-type tags (L/model/model.inl:488-498).  Only GROUP and VERTEX move vertices; BONE feeds the
 * host's bone solve; UV / EXT_UV / MATERIAL are ignored by the reference path and by this engine. */
enum { MMDX_MORPH_GROUP = 0, MMDX_MORPH_VERTEX = 1, MMDX_MORPH_BONE = 2, MMDX_MORPH_UV = 3,
       MMDX_MORPH_MATERIAL = 8 };

/* mmdx_model_desc.flags */
enum {
    MMDX_CREATE_NORMALIZE = 1u << 0, /* apply Model::Normalize retagging (needs bone_parent)        */
    MMDX_CREATE_HOST_ONLY = 1u << 1, /* build + validate the plan only; touch no device (for tests
                                        and tools on machines without a GPU; deform then fails)     */
    MMDX_CREATE_F16_POSITIONS = 1u << 2, /* keep base positions and morph offsets as IEEE binary16 in
                                        HBM (rounded to nearest even once, here); arithmetic stays
                                        f32.  Bandwidth-stress configuration, not reference parity */
    MMDX_CREATE_FAST_MATH = 1u << 3, /* OPT-IN: the deform kernels of this model may contract a multiply and the add
                                        that consumes it into one fused multiply-add (one rounding instead of two), as
                                        any compiler does to L/motion/poser_impl.inl:396-437 at -ffp-contract=fast.
                                        Same operations in the same order, same epsilon tests and morph skips; results
                                        are then NOT bit-identical to libmmd's but within the tolerance stated and
                                        tested in tests/test_fast_math.py: |x - x_ref| <= 1e-5 * (1 + |x_ref|) per
                                        position component (measured on the benchmark models: 2.4e-6, a handful of
                                        binary32 ulps), <= 2e-6 per normal component (measured 1.8e-7), binary16
                                        positions within that plus one binary16 ulp.  Buys 5-8 % throughput.
                                        Without this flag (the default) every result is bit-identical to the reference's. */
    MMDX_CREATE_TILE_ORDER = 1u << 4 /* OPT-IN: every output array of this model (pose_image SoA, the 32-byte vertex, the
                                        f16 layout) holds the vertices in the ENGINE's order instead of the file's: inside
                                        each run of 512 consecutive file vertices they are sorted by deform class and
                                        morph-row length (mmdx_model_get_vertex_order returns the permutation).  Values
                                        are bit-identical to the default's, only their position in the array changes.
                                        The kernels then store straight from registers -- no on-chip transpose, no
                                        per-instance barrier: 3 % for the shared-morph crowd, 9 % for batches with per-instance morph weights
                                        (use with mmdx_crowd_output_alloc).  A renderer
                                        adopts it by remapping its index buffer once at load (main.cpp:781-787:
                                        index[i] = original_to_engine[index[i]]); INTEGRATION.md 1d''.                  */
};

/* Flat model description.  All pointers are host pointers, borrowed for the duration of the call. */
typedef struct mmdx_model_desc {
    uint32_t struct_size; /* = sizeof(mmdx_model_desc)                                             */
    uint32_t flags;       /* MMDX_CREATE_*                                                         */
    uint32_t n_vertices, n_bones, n_morphs;
    uint32_t reserved0;
    const float *positions;    /* [NV][3]                                                          */
    const float *normals;      /* [NV][3]                                                          */
    const float *uvs;          /* [NV][2] or NULL (zeros)                                          */
    const int32_t *skin_type;  /* [NV] MMDX_SKIN_* (raw PMX tag)                                    */
    const int32_t *bone_ids;   /* [NV][4]  BDEF1: [0]; BDEF2/SDEF: [0],[1]; BDEF4: [0..3]          */
    const float *bone_weights; /* [NV][4]  BDEF2/SDEF: [0] = weight of bone [0]; BDEF4: [0..3]     */
    const float *sdef_params;  /* [NV][9] C,R0,R1 or NULL -- accepted, validated for size only      */
    const int32_t *bone_parent; /* [NB] (-1 = none) or NULL; used only by MMDX_CREATE_NORMALIZE    */
    const int32_t *morph_type;  /* [NM] MMDX_MORPH_*                                               */
    const uint32_t *morph_offset; /* [NM+1] entry range of morph m = [off[m], off[m+1])            */
    const uint32_t *morph_index;  /* [E] vertex index (VERTEX) / morph index (GROUP) / other       */
    const float *morph_value;     /* [E][3] offset xyz (VERTEX) / {rate,-,-} (GROUP) / other       */
} mmdx_model_desc;

typedef struct mmdx_model_s *mmdx_model_t;

/* Output layouts */
enum {
    MMDX_OUT_SOA = 0,      /* out_a = f32 pos[NI][NV][3], out_b = f32 nrm[NI][NV][3]
                              (= Poser::pose_image.coordinates / .normals)                          */
    MMDX_OUT_VERTEX32 = 1, /* out_a = struct{f32 pos[3]; f32 normal[3]; f32 uv[2];}[NI][NV]
                              (= main.cpp:50-54); out_b unused                                     */
    MMDX_OUT_SOA_POS16 = 2 /* out_a = f16 pos[NI][NV][3], out_b = f32 nrm[NI][NV][3]               */
};

/* mmdx_deform_args.flags */
enum {
    MMDX_PALETTE_ON_DEVICE = 1u << 0, /* palettes is a device pointer (else host, copied per call)  */
    MMDX_WEIGHTS_ON_DEVICE = 1u << 1, /* morph_weights is a device pointer                          */
    MMDX_OUT_ON_DEVICE = 1u << 2,     /* out_a/out_b are device pointers (else host; D2H + sync)   */
    MMDX_WEIGHTS_SHARED = 1u << 3,    /* one morph_weights[NM] for all instances (crowd with shared
                                         facial state): the morph pass runs once per call          */
    MMDX_MORPH_UNCHANGED = 1u << 4,   /* with MMDX_WEIGHTS_SHARED and NI > 1: the morph weights are those of the
                                         previous such call on this model (a crowd whose facial state
                                         changes less often than its poses): the morphed positions of that
                                         call are reused and the morph pass is skipped; morph_weights is
                                         not read.  An error without such an earlier call.
                                         WITHOUT this flag the library finds out by itself (the reference's vertex_images_
                                         depends on morph_rates_ only, L/motion/poser_impl.inl:362-386): shared rates in host
                                         memory are compared with the ones of the pass whose result the handle holds and the
                                         pass (launch and upload) is skipped when they are bit for bit the same; shared rates
                                         in device memory are compared by the morph pass itself, on the device, which then
                                         skips its walk over the morph table (the launch remains: ~2 us instead of ~9).  The
                                         flag is the caller's promise and saves that launch too.                  */
    /* Hints for crowds whose outputs stream through the caches (device arrays of >= 512 MB per call): how the kernel
       writes them -- cached non-temporal stores (best where the arrays' physical backing is in the fast store mode, see
       mmdx_crowd_output_alloc) or write-through stores (~5 % faster everywhere else, 2 % slower in the fast mode).
       mmdx_crowd_output_alloc returns the right one for its arrays in mmdx_placement_info.store_flags; a caller who has
       measured its own arrays (mmdx_bench_store_pattern against mmdx_bench_fill, mmdx_bench.h) can say so too.  Without
       a hint such outputs are written through (six plain allocations in seven are not in the fast mode).  The decision
       is made from the call's arguments alone: the library remembers nothing about array addresses.  Results are
       identical either way; kernels without a write-through flavour ignore the hint; the two exclude each other.   */
    MMDX_OUT_STORES_WRITE_THROUGH = 1u << 5,
    MMDX_OUT_STORES_CACHED = 1u << 6
};

typedef struct mmdx_deform_args {
    uint32_t struct_size;  /* = sizeof(mmdx_deform_args)                                           */
    uint32_t flags;        /* MMDX_*_ON_DEVICE | MMDX_WEIGHTS_SHARED                               */
    uint32_t n_instances;  /* NI >= 1                                                              */
    uint32_t out_layout;   /* MMDX_OUT_*                                                           */
    const float *morph_weights; /* [NI][NM], or [NM] with MMDX_WEIGHTS_SHARED; may be NULL if NM==0 */
    const float *palettes;      /* [NI][NB][16]                                                    */
    void *out_a;
    void *out_b;
    float pos_scale; /* positions are multiplied by this AFTER the transform, as a separate f32
                        multiply (main.cpp:848-850 uses 0.1f); 1.0f = leave as pose_image          */
    uint32_t reserved0;
} mmdx_deform_args;

typedef struct mmdx_model_info {
    uint32_t struct_size;
    uint32_t n_vertices, n_bones, n_morphs;
    uint32_t n_slots;          /* vertex-morph applications in the reference's traversal order      */
    uint32_t n_entries;        /* vertex-morph entries after group expansion                       */
    uint32_t n_entries_padded; /* ... incl. the padding of the sliced-ELL gather table               */
    uint32_t n_tiles, tile_vertices;
    uint32_t n_bdef1, n_bdef2, n_bdef4; /* after optional Normalize; SDEF/unknown count as bdef2   */
    uint32_t max_tile_bones;
    uint64_t device_bytes;     /* static streams resident in HBM                                   */
    uint32_t device_ordinal;
    uint32_t flags;
    uint32_t reserved0;
} mmdx_model_info;

/* ---- library / device ------------------------------------------------------------------------ */
MMDX_API uint32_t mmdx_abi_version(void);
MMDX_API const char *mmdx_last_error_string(void);
MMDX_API mmdx_status mmdx_device_count(int32_t *count);
/* Device for the models, motions and buffers the CALLING THREAD creates from now on (like hipSetDevice), and the
 * default for threads that never select one.  A handle stays on the device it was created on and may be used
 * from any thread, one call at a time per handle; handles on different devices (or on the same one: each has its
 * own stream) run concurrently from different host threads -- the in-process form of the instance-sharded crowd. */
MMDX_API mmdx_status mmdx_device_select(int32_t ordinal);
MMDX_API mmdx_status mmdx_device_name(int32_t ordinal, char *buf, size_t buf_size);

/* ---- model ----------------------------------------------------------------------------------- */
MMDX_API mmdx_status mmdx_model_create(const mmdx_model_desc *desc, mmdx_model_t *out_model);
MMDX_API mmdx_status mmdx_model_destroy(mmdx_model_t model);
MMDX_API mmdx_status mmdx_model_get_info(mmdx_model_t model, mmdx_model_info *info);
/* Post-Normalize skin tags in ORIGINAL vertex order: type[NV] (0,1,2), ids[NV][4], weights[NV][4]. */
MMDX_API mmdx_status mmdx_model_get_skin(mmdx_model_t model, int32_t *type, int32_t *ids,
                                         float *weights);
/* The engine's vertex order (a permutation of [0, NV) that only moves vertices inside their tile of 512 consecutive file
 * vertices): engine_to_original[e] = file index of the vertex at position e of an MMDX_CREATE_TILE_ORDER model's outputs,
 * original_to_engine = its inverse (what an index buffer is remapped through).  Either pointer may be NULL.  Defined for every
 * model; only MMDX_CREATE_TILE_ORDER models write their outputs in this order. */
MMDX_API mmdx_status mmdx_model_get_vertex_order(mmdx_model_t model, uint32_t *engine_to_original /*[NV]*/,
                                                 uint32_t *original_to_engine /*[NV]*/);
/* Host-side flattening of group morphs: slot_weights[n_slots] for one set of morph rates, exactly
 * what the device consumes (a slot whose chain hits the reference's `rate < 1e-7` skip gets 0). */
MMDX_API mmdx_status mmdx_model_slot_weights(mmdx_model_t model, const float *morph_weights,
                                             float *slot_weights);
/* Borrow an external HIP stream (hipStream_t) instead of the handle's own; NULL restores it. */
MMDX_API mmdx_status mmdx_model_set_stream(mmdx_model_t model, void *hip_stream);

/* ---- HIP-graph replay of a frame's device work ------------------------------------------------------
 * Between mmdx_graph_begin and mmdx_graph_end everything the library enqueues on `model`'s stream is RECORDED instead
 * of executed: mmdx_deform_batched, mmdx_morph_motion_eval, mmdx_bone_motion_eval and mmdx_skeleton_solve* called with
 * this model and every operand in device memory (MMDX_*_ON_DEVICE).  mmdx_graph_launch then replays the whole sequence
 * with ONE submission (asynchronous, on the model's stream); the recorded calls read and write the same device
 * addresses on every replay, so a frame is "update the inputs in place, launch".  What it buys is host time: a
 * frame of motion -> poses -> palettes -> vertices is four to six launches (~4 us of host time each); the device-side
 * gaps between dependent kernels are the same either way.  Rules: run the same sequence once un-captured first (it
 * sizes the handles' scratch buffers; a call that would have to allocate while recording fails), no host operands,
 * no mmdx_profile_enable while recording, one recording per model at a time; mmdx_graph_end on the thread that called
 * mmdx_graph_begin.
 * Lifetime: a graph holds raw device addresses of the scratch buffers of every handle that took part in the recording
 * (the model, and skeletons / motions called with it).  While the graph is alive those handles are pinned: a later call
 * on them that would have to GROW such a buffer fails with MMDX_ERR_INVALID_ARGUMENT instead of moving it (size the
 * buffers with an un-recorded call of the largest shape first); destroying a pinned handle is allowed and invalidates
 * the graph -- mmdx_graph_launch then fails, it never replays into freed memory.  A skeleton or motion destroyed WHILE
 * a recording that used it is in progress poisons that recording (mmdx_graph_end reports it and returns no graph; its
 * device blocks are freed at mmdx_graph_end, the runtime refuses frees on a recording thread); a model destroyed in
 * the middle of its own recording ends the recording.  Recorded calls come from the thread that called
 * mmdx_graph_begin. */
typedef struct mmdx_graph_s *mmdx_graph_t;
MMDX_API mmdx_status mmdx_graph_begin(mmdx_model_t model);
MMDX_API mmdx_status mmdx_graph_end(mmdx_model_t model, mmdx_graph_t *out_graph);
MMDX_API mmdx_status mmdx_graph_launch(mmdx_graph_t graph);
MMDX_API void mmdx_graph_destroy(mmdx_graph_t graph);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* deform(model, morph_weights, bone_palette, out_verts): one instance, host pointers, synchronous.
 * out_pos/out_nrm = f32[NV][3] each = Poser::pose_image after Poser::Deform(). */
MMDX_API mmdx_status mmdx_deform(mmdx_model_t model, const float *morph_weights /*[NM]*/,
                                 const float *palette /*[NB][16]*/, float *out_pos, float *out_nrm);
/* Same, producing the viewer's interleaved 32-byte vertices (Deform + UpdateDeformedVertices). */
MMDX_API mmdx_status mmdx_deform_vertex32(mmdx_model_t model, const float *morph_weights,
                                          const float *palette, float pos_scale,
                                          void *out_vertices /*[NV] x 32 B*/);
/* General / crowd form.  Asynchronous on the handle's stream when every pointer is a device pointer;
 * otherwise returns after the copies have completed. */
MMDX_API mmdx_status mmdx_deform_batched(mmdx_model_t model, const mmdx_deform_args *args);
MMDX_API mmdx_status mmdx_sync(mmdx_model_t model);

/* ---- plain device-memory helpers (thin hipMalloc / hipMemcpy wrappers) ----------------------- */
/* So that C, C++ and ctypes callers can keep palettes and outputs resident in HBM without linking
 * the HIP runtime themselves. */
MMDX_API mmdx_status mmdx_device_malloc(void **ptr, size_t bytes);
MMDX_API mmdx_status mmdx_device_free(void *ptr);
/* Page-locked host memory: palettes / rates handed to mmdx_deform*() from here move over PCIe by DMA
 * without the runtime's staging copy, and OUTPUT buffers from here (or any hipHostMalloc / hipHostRegister
 * memory) are written by the kernel directly -- no device-side staging buffer, no device-to-host copy
 * command: the single-model drop-in path (the viewer's vertex buffer, main.cpp:735-863).  Pageable
 * memory works everywhere too: frame-sized inputs / outputs go through page-locked bounce buffers owned by
 * the model (one extra CPU memcpy), larger ones through the runtime's staging copies. */
MMDX_API mmdx_status mmdx_host_malloc(void **ptr, size_t bytes);
MMDX_API mmdx_status mmdx_host_free(void *ptr);
MMDX_API mmdx_status mmdx_memcpy_h2d(void *dst_device, const void *src_host, size_t bytes);
MMDX_API mmdx_status mmdx_memcpy_d2h(void *dst_host, const void *src_device, size_t bytes);
MMDX_API mmdx_status mmdx_device_memset(void *dst_device, int value, size_t bytes);
MMDX_API mmdx_status mmdx_device_synchronize(void);
/* Placement-aware allocation of a crowd's output arrays ([n_instances][NV] in `out_layout`; out_b stays
 * NULL for MMDX_OUT_VERTEX32).  On MI355X the store rate of the crowd's output pattern is bimodal in WHERE
 * the driver places the arrays (same virtual addresses, different physical backing: ~0.97 or ~0.75 of the
 * linear-fill rate, stable for the life of the allocation; tools/archive/probes/alloc_probe.py, alloc_kernel_probe.py),
 * and the deform kernel follows it.  This helper allocates, times the store-only replay of the pattern
 * against a linear fill, and retries up to `max_tries` times (about one placement in seven is the fast
 * one; ~5 ms per try), keeping the best placement seen, then waits until the driver's background wipe of
 * the freed candidates no longer shows.  max_tries <= 1: plain allocation.  Free both arrays with
 * mmdx_device_free() (or hipFree: the library keeps no record of the arrays).  What the probe found comes back in
 * mmdx_placement_info.store_flags: OR it into mmdx_deform_args.flags of the crowd calls that write these arrays
 * (MMDX_OUT_STORES_* above) -- a performance hint only, results never depend on it, and its lifetime is the caller's:
 * it describes the physical backing of THIS allocation and is void once the arrays are freed. */
typedef struct mmdx_placement_info {
    uint32_t struct_size;
    uint32_t tries;          /* allocations made                                                        */
    uint32_t probed;         /* 0: layout / vertex count not probe-able, plain allocation               */
    float store_GBs;         /* store-only replay on the returned arrays                                */
    float fill_GBs;          /* linear fill of the same bytes (the yardstick)                           */
    uint32_t store_flags;    /* MMDX_OUT_STORES_CACHED (fast store mode found), MMDX_OUT_STORES_WRITE_THROUGH
                                (not found), or 0 (not probed: the library's default applies)            */
} mmdx_placement_info;
MMDX_API mmdx_status mmdx_crowd_output_alloc(mmdx_model_t model, uint32_t n_instances, int32_t out_layout,
                                             uint32_t max_tries, void **out_a_device, void **out_b_device,
                                             mmdx_placement_info *info /* may be NULL */);

/* ---- PMX 2.0 loader (the data format on the input side of the path) --------------------------- */
/* From-scratch parser for the fields the deformation path consumes; replaces, for those fields,
 * PmxReader::ReadModel (L/reader/pmx_reader_impl.inl:16-449) + FileReader (L/util/dwarf_impl.inl:29-130)
 * with the same observable semantics (1/2-byte indices zero-extended, 4-byte sign-extended; UTF-16LE
 * or UTF-8 text; unknown deform / morph types are errors).  Display frames, rigid bodies and joints
 * are not read. */
typedef struct mmdx_pmx_s *mmdx_pmx_t;

typedef struct mmdx_pmx_info {
    uint32_t struct_size;
    uint32_t n_vertices, n_indices, n_textures, n_materials, n_bones, n_morphs, n_morph_entries;
    uint32_t extra_uv;        /* additional UV sets per vertex (skipped)                            */
    uint32_t utf8;            /* text encoding of the file: 1 = UTF-8, 0 = UTF-16LE                 */
    uint8_t index_width[8];   /* vertex, texture, material, bone, morph, rigid body; 2 spare        */
    uint64_t bytes_consumed;  /* file offset just behind the morph block                           */
} mmdx_pmx_info;

typedef struct mmdx_pmx_arrays {      /* pointers into the parsed model, valid until destroy       */
    uint32_t struct_size;
    uint32_t reserved0;
    const uint32_t *triangles;             /* [n_indices] original winding                         */
    const uint32_t *material_index_count;  /* [n_materials] consecutive index ranges               */
    const float *bone_rest_position;       /* [n_bones][3]                                         */
    const int32_t *bone_parent;            /* [n_bones], -1 = none                                 */
    const int32_t *bone_transform_level;   /* [n_bones]                                            */
    const uint16_t *bone_flags;            /* [n_bones] raw PMX bone flag word                     */
    const uint8_t *morph_panel;            /* [n_morphs]                                           */
    const float *edge_scale;               /* [n_vertices]                                         */
} mmdx_pmx_arrays;

enum { MMDX_PMX_NAME_MODEL = 0, MMDX_PMX_NAME_BONE = 1, MMDX_PMX_NAME_MORPH = 2,
       MMDX_PMX_NAME_MATERIAL = 3, MMDX_PMX_NAME_TEXTURE = 4 };

MMDX_API mmdx_status mmdx_pmx_parse(const void *data, size_t size, mmdx_pmx_t *out_pmx);
MMDX_API mmdx_status mmdx_pmx_load_file(const char *path, mmdx_pmx_t *out_pmx);
/* PMD 1.0, the older format: replaces PmdReader::ReadModel (L/reader/pmd_reader_impl.inl:16-566) for the same
 * fields and yields the same kind of handle, so every mmdx_pmx_* accessor below serves it (incl. the rig: PMD
 * bone types and IK list are converted to flag words, append and IK tables exactly as libmmd converts them;
 * a second IK record of one bone appends a bone, so n_bones can exceed the file's count).  All vertices
 * are BDEF2 with weight byte * 0.01f; morph indices are resolved through the base morph. */
MMDX_API mmdx_status mmdx_pmd_parse(const void *data, size_t size, mmdx_pmx_t *out_pmx);
MMDX_API mmdx_status mmdx_pmd_load_file(const char *path, mmdx_pmx_t *out_pmx);
MMDX_API void mmdx_pmx_destroy(mmdx_pmx_t pmx);
MMDX_API mmdx_status mmdx_pmx_get_info(mmdx_pmx_t pmx, mmdx_pmx_info *info);
/* Fills `desc` with pointers into `pmx` (valid until mmdx_pmx_destroy) and MMDX_CREATE_NORMALIZE,
 * ready for mmdx_model_create(): the reference's reader ends with model.Normalize() too. */
MMDX_API mmdx_status mmdx_pmx_get_model_desc(mmdx_pmx_t pmx, mmdx_model_desc *desc);
MMDX_API mmdx_status mmdx_pmx_get_arrays(mmdx_pmx_t pmx, mmdx_pmx_arrays *arrays);
/* Names as UTF-8 (bone and morph names are what VMD motion data is keyed by). */
MMDX_API mmdx_status mmdx_pmx_get_name(mmdx_pmx_t pmx, int32_t kind, uint32_t index, char *buf,
                                       size_t buf_size);

/* ---- VMD motion loader + morph-rate evaluation on the device (the per-frame input side) --------- */
/* Replaces, for morph tracks, VmdReader::ReadMotion (L/reader/vmd_reader_impl.inl:9-79),
 * MotionPlayer's name mapping (L/motion/poser_impl.inl:522-537) and Motion::GetMorphPose
 * (L/motion/motion_impl.inl:382-424): the rates a crowd needs -- [instances][morphs], every instance at
 * its own frame -- are produced in HBM, ready for mmdx_deform_batched(MMDX_WEIGHTS_ON_DEVICE).  Bone
 * keyframes are exposed raw (for a host bone solve) and evaluated on the device further below. */
typedef struct mmdx_vmd_s *mmdx_vmd_t;
typedef struct mmdx_morph_motion_s *mmdx_morph_motion_t;

typedef struct mmdx_vmd_info {
    uint32_t struct_size;
    uint32_t n_bone_records, n_morph_records; /* as stored in the file                              */
    uint32_t n_bone_tracks, n_morph_tracks;   /* distinct names                                     */
    uint32_t n_bone_keys, n_morph_keys;       /* after "last record for a (name, frame) wins"       */
    uint32_t max_frame;
    uint64_t bytes_consumed;                  /* camera / light / shadow sections follow; not read  */
} mmdx_vmd_info;

typedef struct mmdx_vmd_bone_key {            /* one 111-byte VMD bone record minus the name         */
    uint32_t frame;
    float translation[3];
    float rotation[4];                        /* quaternion x, y, z, w                               */
    int8_t interpolation[64];                 /* x, y, z, rotation: 16 bytes each, control points at
                                                 [0],[4],[8],[12] in units of 1/127                  */
} mmdx_vmd_bone_key;

enum { MMDX_FRAMES_ON_DEVICE = 1u << 0 };     /* with MMDX_OUT_ON_DEVICE for mmdx_morph_motion_eval  */

MMDX_API mmdx_status mmdx_vmd_parse(const void *data, size_t size, mmdx_vmd_t *out_vmd);
MMDX_API mmdx_status mmdx_vmd_load_file(const char *path, mmdx_vmd_t *out_vmd);
MMDX_API void mmdx_vmd_destroy(mmdx_vmd_t vmd);
MMDX_API mmdx_status mmdx_vmd_get_info(mmdx_vmd_t vmd, mmdx_vmd_info *info);
MMDX_API mmdx_status mmdx_vmd_track_name(mmdx_vmd_t vmd, int32_t is_morph, uint32_t track, char *buf,
                                         size_t buf_size);   /* UTF-8 */
MMDX_API mmdx_status mmdx_vmd_bone_track(mmdx_vmd_t vmd, uint32_t track, const mmdx_vmd_bone_key **keys,
                                         uint32_t *n_keys);  /* sorted by frame */
MMDX_API mmdx_status mmdx_vmd_morph_track(mmdx_vmd_t vmd, uint32_t track, const uint32_t **frames,
                                          const float **weights, uint32_t *n_keys);
/* Associate the motion's morph tracks with a model's morphs by name (UTF-8, e.g. from
 * mmdx_pmx_get_name); morphs without a track evaluate to 0. */
MMDX_API mmdx_status mmdx_vmd_bind_morphs(mmdx_vmd_t vmd, uint32_t n_morphs,
                                          const char *const *morph_names_utf8,
                                          mmdx_morph_motion_t *out_motion);
MMDX_API mmdx_status mmdx_morph_motion_get_info(mmdx_morph_motion_t motion, uint32_t *n_morphs,
                                                uint32_t *n_mapped, uint32_t *n_keys);
/* out_weights[i][m] = rate of model morph m at frames[i], i < n_instances.  Runs on `model`'s device
 * and stream when `model` is given (so a following mmdx_deform_batched sees the result), else on the
 * selected device's default stream.  flags: MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE. */
MMDX_API mmdx_status mmdx_morph_motion_eval(mmdx_morph_motion_t motion, mmdx_model_t model,
                                            uint32_t n_instances, const uint32_t *frames,
                                            uint32_t flags, float *out_weights);
MMDX_API void mmdx_morph_motion_destroy(mmdx_morph_motion_t motion);


/* ---- Bone tracks -> local bone poses on the device; skeleton -> bone palette ---------------------- */
/* The per-frame input side of the palette.  mmdx_bone_motion_eval replaces
 * Motion::GetBonePose (L/motion/motion_impl.inl:255-319: clamp to the first / last key, exact key hit,
 * else per-channel presampled-Bezier blend of the translation and NLerp of the rotation,
 * L/util/math_impl.inl:1260-1282, :1372-1428) together with VmdReader's control-point set-up
 * (L/reader/vmd_reader_impl.inl:31-60) and MotionPlayer's name mapping + SeekFrame
 * (L/motion/poser_impl.inl:522-548), for every (instance, bone) at once.  mmdx_skeleton_solve replaces
 * Poser::UpdateBoneTransform + UpdateBoneSkinningMatrix (L/motion/poser_impl.inl:142-166, :320-326) in
 * the reference's evaluation order (pre-physics bones, then post-physics bones, each sorted by transform
 * level then index, L/motion/poser_impl.inl:107-109, :500-510) and writes the float[16] palettes
 * mmdx_deform_batched(MMDX_PALETTE_ON_DEVICE) consumes.  Both are bit-exact against libmmd. */
typedef struct mmdx_bone_motion_s *mmdx_bone_motion_t;
typedef struct mmdx_skeleton_s *mmdx_skeleton_t;

enum { MMDX_POSES_ON_DEVICE = 1u << 4 };      /* with MMDX_OUT_ON_DEVICE (and MMDX_WEIGHTS_*) for
                                                 mmdx_skeleton_solve*                                 */
enum { MMDX_POSE_FLOATS = 8 };                /* one local pose: translation xyz, 0, quaternion xyzw  */

/* Associate the motion's bone tracks with a model's bones by name (UTF-8); bones without a track keep
 * the rest pose (zero translation, identity rotation), as after Poser::ResetPosing. */
MMDX_API mmdx_status mmdx_vmd_bind_bones(mmdx_vmd_t vmd, uint32_t n_bones,
                                         const char *const *bone_names_utf8,
                                         mmdx_bone_motion_t *out_motion);
MMDX_API mmdx_status mmdx_bone_motion_get_info(mmdx_bone_motion_t motion, uint32_t *n_bones,
                                               uint32_t *n_mapped, uint32_t *n_keys,
                                               uint32_t *n_curves);
/* out_poses[i][b][MMDX_POSE_FLOATS] = local pose of model bone b at frames[i].  Stream and device as
 * for mmdx_morph_motion_eval.  flags: MMDX_FRAMES_ON_DEVICE | MMDX_OUT_ON_DEVICE. */
MMDX_API mmdx_status mmdx_bone_motion_eval(mmdx_bone_motion_t motion, mmdx_model_t model,
                                           uint32_t n_instances, const uint32_t *frames,
                                           uint32_t flags, float *out_poses);
MMDX_API void mmdx_bone_motion_destroy(mmdx_bone_motion_t motion);

enum {                                        /* bits of the PMX bone flag word the skeleton reads    */
    MMDX_BONE_HAS_IK = 0x0020, MMDX_BONE_APPEND_ROTATE = 0x0100, MMDX_BONE_APPEND_TRANSLATE = 0x0200,
    MMDX_BONE_POST_PHYSICS = 0x1000
};

typedef struct mmdx_skeleton_desc {           /* e.g. straight from mmdx_pmx_get_arrays               */
    uint32_t struct_size;
    uint32_t n_bones;
    const float *rest_position;               /* [n_bones][3]                                         */
    const int32_t *parent;                    /* [n_bones]; outside [0, n_bones) = none               */
    const int32_t *transform_level;           /* [n_bones] or NULL (all 0); compared as unsigned, like
                                                 the reference's size_t cast                          */
    const uint16_t *flags;                    /* [n_bones] PMX bone flag word, or NULL (all 0)        */
    /* append (inherit) bones -- read where flags has MMDX_BONE_APPEND_*; NULL if no bone has them     */
    const int32_t *append_parent;             /* [n_bones]; outside [0, n_bones) = the bone does not
                                                 append (L/motion/poser_impl.inl:50-56)               */
    const float *append_ratio;                /* [n_bones]                                            */
    /* CCD-IK -- read where flags has MMDX_BONE_HAS_IK; NULL if no bone has it                         */
    const int32_t *ik_target;                 /* [n_bones]                                            */
    const int32_t *ik_loop_count;             /* [n_bones]; negative or > 256 means 256 (:94)         */
    const float *ik_angle_limit;              /* [n_bones] radians per link step                      */
    const uint32_t *ik_link_offset;           /* [n_bones+1] into the link arrays                     */
    const int32_t *ik_link_bone;              /* [L] target-side link first                           */
    const uint8_t *ik_link_limited;           /* [L]                                                  */
    const float *ik_link_lo, *ik_link_hi;     /* [L][3] Euler limits (either order; min/max is taken) */
    /* bone morphs (Poser::UpdateMorphTransform, MORPH_TYPE_BONE, L/motion/poser_impl.inl:347-354) -- the
     * model's morph table as in mmdx_model_desc; only group (0) and bone (2) morphs are read.  n_morphs
     * == 0 / NULL: no bone morphs (morph_rotation_ = identity, morph_translation_ = 0).                */
    uint32_t n_morphs;
    uint32_t create_flags;                    /* MMDX_SKELETON_*                                      */
    const int32_t *morph_type;                /* [n_morphs] PMX morph type                            */
    const uint32_t *morph_offset;             /* [n_morphs+1]                                         */
    const uint32_t *morph_index;              /* [E] group: morph index; bone: bone index             */
    const float *morph_value;                 /* [E][3] group: rate in [0]; bone: translation         */
    const float *morph_rotation;              /* [E][4] bone: rotation xyzw; NULL = identity          */
} mmdx_skeleton_desc;

enum {
    MMDX_SKELETON_PHYSICS_SEAM = 1u << 0      /* the skeleton will be solved in two steps with a physics reactor's
                                                 writes in between (mmdx_skeleton_solve_pre / _post): compiles the
                                                 ordered solver, which keeps per-bone state between the steps, for
                                                 rigs without IK / append bones too                              */
};

typedef struct mmdx_skeleton_info {
    uint32_t struct_size;
    uint32_t n_bones, n_pre_physics, n_post_physics;
    uint32_t max_chain;                       /* longest parent chain (parallel solver), else 0       */
    uint32_t solver;                          /* MMDX_SOLVER_*                                        */
    uint32_t n_ik_bones, n_ik_links, n_append_bones;
    uint32_t n_bone_morph_entries;            /* applications of a bone-morph entry (groups expanded)  */
    uint32_t n_solve_rounds;                  /* ordered solver: rounds the evaluation sequence was cut
                                                 into (independent bones share a round), else 0         */
    uint32_t n_ik_rounds_16_lanes;            /* ... of which rounds made of CCD-IK solves on plain chains: these
                                                 run with sixteen lanes per solve (ABI 3)                */
} mmdx_skeleton_info;

enum {
    MMDX_SOLVER_PARALLEL_FK = 0,  /* no IK / append: one thread per (instance, bone), bit-exact      */
    MMDX_SOLVER_SERIAL = 1        /* IK / append present: the reference's evaluation sequence; bones
                                     (and IK solves) that touch disjoint state run side by side, the
                                     rest in order; sin/cos/asin/acos/atan2 through the device's double
                                     libm like the reference's through the host's     */
};

/* Index validation happens here (MMDX_ERR_BAD_INDEX), so solve cannot read out of range.
 * Nested IK -- a link or target that is itself an IK bone -- is solved the way the reference's recursion does
 * (UpdateBoneTransform re-enters itself for links and target, L/motion/poser_impl.inl:196-206), up to 3 solves
 * deep.  MMDX_ERR_UNSUPPORTED: deeper nesting, or an IK bone that is (indirectly) part of its own solve (endless
 * recursion in the reference). */
MMDX_API mmdx_status mmdx_skeleton_create(const mmdx_skeleton_desc *desc, mmdx_skeleton_t *out_skeleton);
MMDX_API mmdx_status mmdx_skeleton_get_info(mmdx_skeleton_t skeleton, mmdx_skeleton_info *info);
/* out_palettes[i][b][16] from poses[i][b][MMDX_POSE_FLOATS].  flags: MMDX_POSES_ON_DEVICE |
 * MMDX_OUT_ON_DEVICE.  Runs on `model`'s stream when given, so the deform call that follows sees it. */
MMDX_API mmdx_status mmdx_skeleton_solve(mmdx_skeleton_t skeleton, mmdx_model_t model,
                                         uint32_t n_instances, const float *poses, uint32_t flags,
                                         float *out_palettes);
/* Bone tracks -> palettes in one call: out_palettes[i][b][16] at frames[i] -- bit for bit what mmdx_bone_motion_eval followed by
 * mmdx_skeleton_solve computes (MotionPlayer::SeekFrame + Pre/PostPhysicsPosing for every instance, main.cpp:1793-1810).  On a
 * skeleton that runs the parallel FK solver it is ONE launch and the [NI][NB][8] local poses never leave the chip (a workgroup per
 * instance keeps them in LDS); skeletons with append bones / IK and those of more than 2 048 bones take the two launches, the poses
 * in the motion's scratch buffer.  `motion` must have been bound to this skeleton's bones.  flags: MMDX_FRAMES_ON_DEVICE |
 * MMDX_OUT_ON_DEVICE; stream and device as for mmdx_bone_motion_eval; recordable into a graph. */
MMDX_API mmdx_status mmdx_skeleton_solve_motion(mmdx_skeleton_t skeleton, mmdx_bone_motion_t motion, mmdx_model_t model,
                                                uint32_t n_instances, const uint32_t *frames, uint32_t flags,
                                                float *out_palettes);
/* The same with bone morphs applied first: morph_weights[i][n_morphs] (or one shared row with
 * MMDX_WEIGHTS_SHARED; device pointer with MMDX_WEIGHTS_ON_DEVICE) are the raw per-frame morph rates, the
 * ones mmdx_deform_batched takes.  Bone-morph rotations go through SLerp, i.e. through the device's double
 * acos / sin like the reference's through the host's.  NULL weights = mmdx_skeleton_solve. */
/* ---- the physics seam ---------------------------------------------------------------------------------
 * The reference's frame is PrePhysicsPosing -> PhysicsReactor::React -> PostPhysicsPosing (main.cpp:1801-1810):
 * after the pre-physics bone list the reactor (Bullet, on the host) overwrites the skinning matrix of every bone one
 * of its bodies moved (PoserMotionState::Synchronize, mmd-bullet_impl.inl:34-40) and re-derives local_matrix_ of
 * the "strict" ones from it -- keeping the bone's own translation -- before it recomputes their skinning matrix
 * (Fix, :42-56); the post-physics bones then hang off those local matrices.  Two calls reproduce that:
 *   mmdx_skeleton_solve_pre   reset + bone morphs + the pre-physics list; out_palettes rows of the pre-physics bones
 *                             are written (what the reactor's kinematic bodies read), the others are unspecified;
 *   (the host steps its physics)
 *   mmdx_skeleton_solve_post  Synchronize for all listed bones, then Fix for the strict ones in list order, then the
 *                             post-physics list; out_palettes (the SAME array the pre step wrote) receives the
 *                             overridden rows and the post-physics bones' rows.
 * The skeleton must have been created with MMDX_SKELETON_PHYSICS_SEAM; n_instances must match between the two
 * calls.  Fix uses Matrix4f::Inverse's Gauss-Jordan elimination step for step (L/util/math_impl.inl:822-897):
 * bit-exact against libmmd like the rest of the solve. */
typedef struct mmdx_physics_overrides {
    uint32_t struct_size;
    uint32_t n_bones;                         /* K bones physics moved (the same set for every instance)         */
    const int32_t *bone;                      /* [K] host                                                        */
    const uint8_t *strict;                    /* [K] host, non-zero: Fix() applies; NULL = none                  */
    const float *skinning;                    /* [NI][K][16] the bodies' transforms as skinning matrices; host, or
                                                 device with MMDX_OVERRIDES_ON_DEVICE                            */
} mmdx_physics_overrides;
enum { MMDX_OVERRIDES_ON_DEVICE = 1u << 5 };
MMDX_API mmdx_status mmdx_skeleton_solve_pre(mmdx_skeleton_t skeleton, mmdx_model_t model, uint32_t n_instances,
                                             const float *poses, const float *morph_weights, uint32_t flags,
                                             float *out_palettes);
MMDX_API mmdx_status mmdx_skeleton_solve_post(mmdx_skeleton_t skeleton, mmdx_model_t model, uint32_t n_instances,
                                              const mmdx_physics_overrides *overrides /* may be NULL */,
                                              uint32_t flags, float *out_palettes);
MMDX_API mmdx_status mmdx_skeleton_solve_morphed(mmdx_skeleton_t skeleton, mmdx_model_t model,
                                                 uint32_t n_instances, const float *poses,
                                                 const float *morph_weights, uint32_t flags,
                                                 float *out_palettes);
MMDX_API void mmdx_skeleton_destroy(mmdx_skeleton_t skeleton);
/* Fills `desc` with pointers into `pmx` (valid until mmdx_pmx_destroy): rest positions, parents, transform
 * levels, flag words, append and IK tables exactly as the file states them (PmxReader,
 * L/reader/pmx_reader_impl.inl:192-264), ready for mmdx_skeleton_create(). */
MMDX_API mmdx_status mmdx_pmx_get_skeleton_desc(mmdx_pmx_t pmx, mmdx_skeleton_desc *desc);

#ifdef __cplusplus
}
#endif
#endif /* MMDX_H_INCLUDED */
