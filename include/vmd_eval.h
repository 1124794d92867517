/*
 * include/vmd_eval.h — the drop-in boundary: C ABI of the MI355X property evaluator.
 *
 * Mirrors the part of mdlib's md_script surface that VIAMD calls on its evaluation hot path
 * (SURVEY.md 8b).  Each entry point names the reference call site it replaces (paths relative to
 * /root/reference).  The full script *compiler* (md_script_ir_compile_from_source, src/main.cpp:878) is out
 * of scope: the IR here is a list of property descriptors filled by vmd_ir_add_*, or by the mini front-end
 * vmd_ir_compile_from_source for the script forms VIAMD ships and generates.
 *
 * All compute runs in hand-written HIP kernels (include/vmd_hip.h); there is no CPU fallback: every
 * function that needs the device returns false/NULL and logs when no HIP device is usable.
 */
#ifndef VMD_EVAL_H
#define VMD_EVAL_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data handed in by the host application ---------------------------------------------------- */

/* md_unitcell_t as VIAMD reads it (src/viamd.cpp:1837-1843: x,y,z,xy,xz,yz) + periodicity flags */
#define VMD_UNITCELL_PBC_X 1u
#define VMD_UNITCELL_PBC_Y 2u
#define VMD_UNITCELL_PBC_Z 4u
#define VMD_UNITCELL_PBC_ALL 7u
typedef struct vmd_unitcell_t {
    float x, y, z, xy, xz, yz;
    uint32_t flags;
} vmd_unitcell_t;

/* the slice of md_system_t the path reads: SoA coordinates of the displayed frame (src/main.cpp:642),
 * masses (md_atom_mass, src/viamd.cpp:2253), unit cell.  Coordinates may be NULL: the evaluator always
 * pulls frames through the trajectory (src/main.cpp:995-996). */
typedef struct vmd_system_t {
    size_t atom_count;
    const float* x;
    const float* y;
    const float* z;
    const float* mass;          /* atom_count entries */
    vmd_unitcell_t unitcell;
    /* md_system_t::bond (read by the evaluator while it runs, src/viamd.cpp:3088-3091; md_util_unwrap_vec4(..., &sys.bond, ...),
     * src/viamd.cpp:2257): atom index pairs, or NULL / 0.  With bonds an sdf() reference structure is made whole across the periodic
     * cell along its bond graph (breadth-first from its first atom); without, along its index order (SPEC S5, D-SDF-UNWRAP).  Read
     * once, by the first vmd_eval_frame_range of an eval. */
    const int32_t (*bonds)[2];
    size_t bond_count;
} vmd_system_t;

typedef struct vmd_frame_header_t {
    size_t num_atoms;
    int64_t index;
    double timestamp;
    vmd_unitcell_t unitcell;
} vmd_frame_header_t;

/* device-resident view of a trajectory: frame f has x at base + f*frame_stride, y at +row_stride,
 * z at +2*row_stride (floats).  cells: host array, one per frame. */
typedef struct vmd_device_view_t {
    const float* base;          /* device pointer */
    size_t frame_stride;
    size_t row_stride;
    const vmd_unitcell_t* cells;
    int device;
    size_t resident_beg, resident_end;   /* frames present behind `base` (a rank's shard); 0, 0 = all of them */
    uint64_t cells_version;              /* changes whenever the cells OR the coordinates behind the view are modified; 0 = unknown
                                            (the evaluator then re-reads the cells of every batch instead of keeping the boxes - and,
                                            for open axes, the bounding boxes - of an unchanged range on the device) */
} vmd_device_view_t;

/* host-resident view of a trajectory with the same SoA frame layout (e.g. a frame cache in pinned memory): lets the
 * evaluator DMA frames straight from it instead of copying them through load_frame first. */
typedef struct vmd_host_view_t {
    const float* base;          /* host pointer */
    size_t frame_stride;
    size_t row_stride;
    const vmd_unitcell_t* cells;
} vmd_host_view_t;

/* md_trajectory_i stand-in.  load_frame has the signature of md_trajectory_load_frame
 * (src/viamd.cpp:465-467, :1815-1817) and, like it, must be re-entrant: VIAMD's pool threads call it concurrently
 * (src/main.cpp:995-996), and the evaluator decodes the frames of one staged batch on several threads
 * (vmd_set_option("load_threads", n); 1 = strictly serial, 0 = default: an eighth of the host's hardware threads within
 * [8, 32]).  device_view is an extension: when non-NULL and successful the evaluator reads frames in place from HBM instead
 * of staging them through load_frame. */
/* A frame handed over still compressed, to be decoded on the device (extension; today: the XTC coordinate block).  The
 * decoder parameters are in host byte order; the `nbytes` bytes of the bit stream are written to the caller's buffer. */
#define VMD_RAW_CODEC_XTC 1u
/* plain 32-bit floats as the file stores them (TRR: big-endian nm, xyz interleaved; DCD: one block per component, either byte
 * order).  Only handed over through raw_mapped_view - load_raw fills `info` and refuses a payload copy: the copy engine takes the
 * frames out of the mapped file and a kernel swaps / scales / transposes them into the evaluator's layout (k_raw_f32). */
#define VMD_RAW_CODEC_F32 2u
#define VMD_RAW_F32_BIG_ENDIAN 1u
typedef struct vmd_raw_frame_t {
    uint32_t codec;             /* VMD_RAW_CODEC_* */
    float    precision;         /* XTC: grid steps per nm */
    int32_t  minint[3], maxint[3];
    int32_t  smallidx;
    uint32_t reserved;
    uint64_t nbytes;            /* XTC: bytes of the bit stream; F32: bytes from the frame's stream_offset to the end of its last float */
    /* VMD_RAW_CODEC_F32: component c of atom i is the float at stream_offset + f32_offset[c] + 4 * f32_stride * i */
    uint64_t f32_offset[3];
    uint32_t f32_stride;        /* floats from one atom to the next: 1 = a block per component, 3 = xyz interleaved */
    uint32_t f32_flags;         /* VMD_RAW_F32_* */
    float    f32_scale;         /* file unit -> Angstrom, applied as one fp32 multiply (1 = none) */
    uint32_t reserved2;
} vmd_raw_frame_t;

/* Compressed frames resident in HBM (extension): the bit streams of EVERY frame of the trajectory and their decoder records on
 * the device - 0.3 .. 0.5 of the float bytes for liquids, so 288 GB hold trajectories 2 - 3 times larger than as floats.  The
 * evaluator decompresses a batch straight from there (k_xtc_wave on the copy stream, under the pair kernel of the previous batch):
 * no host work and no PCIe traffic per evaluation.  vmd_rawtraj_* builds one from any trajectory that offers load_raw. */
typedef struct vmd_raw_device_view_t {
    const unsigned char* base;      /* device: bit streams */
    const void* info;               /* device: vmd_xtc_frame_t[num_frames] (include/vmd_hip.h), offsets relative to base */
    const vmd_unitcell_t* cells;    /* host, one per frame */
    uint32_t codec;                 /* VMD_RAW_CODEC_* */
    int device;
    /* optional (NULL = none): room for the decoder's checkpoints, so that only the FIRST decode of a frame has to walk its bit stream
     * from the start (vmd_hip.h: vmd_xtc_ck_t).  ck: device, num_frames x VMD_XTC_CK_MAX records; nck: device, one counter per frame;
     * ck_have: host, one byte per frame, set by the evaluator once the frame's checkpoints are valid */
    void*     ck;
    uint32_t* nck;
    uint8_t*  ck_have;
    /* optional (NULL / 0 = none): room for the decoder's group records (vmd_hip.h: vmd_hip_xtc_decode_wave_rec) - rec: device,
     * num_frames x rec_stride 16-bit entries (rec_stride >= num_atoms); nrec: device, one counter per frame; rec_failed: host, set
     * by the evaluator when a decode from records was rejected (it then walks the sections from their checkpoints again) */
    uint16_t* rec;
    uint32_t* nrec;
    size_t    rec_stride;
    bool*     rec_failed;
} vmd_raw_device_view_t;

/* A trajectory FILE mapped into the address space (read-only).  The evaluator pins the mapping (hipHostRegister, in windows) and the
 * copy engine reads the compressed frames of a batch straight out of the page cache: no host thread touches the bytes.  Streams sit
 * where the file has them (XDR: 4-byte aligned). */
typedef struct vmd_raw_mapped_view_t {
    const unsigned char* base;      /* host: the mapping, page aligned */
    size_t bytes;                   /* its length (the file size when it was mapped) */
    const uint64_t* stream_offset;  /* host, one per frame: where frame i's compressed bytes start, relative to base
                                     * (their length and decoder parameters: load_raw's info) */
    uint32_t codec;                 /* VMD_RAW_CODEC_* */
} vmd_raw_mapped_view_t;

typedef struct vmd_trajectory_i {
    void* inst;
    size_t (*num_frames)(void* inst);
    size_t (*num_atoms)(void* inst);
    bool (*load_frame)(void* inst, int64_t idx, vmd_frame_header_t* header, float* x, float* y, float* z);
    bool (*device_view)(void* inst, vmd_device_view_t* out);
    bool (*host_view)(void* inst, vmd_host_view_t* out);     /* extension, may be NULL */
    /* extension, may be NULL: frame `idx` as stored in the file.  dst == NULL: only fill `info` (nbytes = buffer size needed).
     * Returns false when this frame cannot be handed over raw (the evaluator then takes load_frame); re-entrant like
     * load_frame.  With vmd_set_option("xtc_device_decode", 1) the evaluator moves the compressed bytes over PCIe and
     * decompresses a whole batch on the GPU (k_xtc_decode). */
    bool (*load_raw)(void* inst, int64_t idx, vmd_frame_header_t* header, vmd_raw_frame_t* info, void* dst, size_t cap);
    /* extension, may be NULL: the whole trajectory compressed in HBM (see vmd_raw_device_view_t) */
    bool (*raw_device_view)(void* inst, vmd_raw_device_view_t* out);
    /* extension, may be NULL: the file behind load_raw, mapped (see vmd_raw_mapped_view_t); false = not mappable, use load_raw */
    bool (*raw_mapped_view)(void* inst, vmd_raw_mapped_view_t* out);
} vmd_trajectory_i;

/* where the evaluator currently is (a static string, process-wide, last writer wins): for crash handlers and hang reports */
const char* vmd_last_stage(void);

/* ---- IR: property descriptors (md_script_ir_t stand-in) ----------------------------------------- */

typedef struct vmd_script_ir_t vmd_script_ir_t;

/* md_script_property_flags_t (src/main.cpp:1317,1456,1485) */
typedef uint32_t vmd_property_flags_t;
#define VMD_PROPERTY_FLAG_NONE         0u
#define VMD_PROPERTY_FLAG_TEMPORAL     1u
#define VMD_PROPERTY_FLAG_DISTRIBUTION 2u
#define VMD_PROPERTY_FLAG_VOLUME       4u

#define VMD_RDF_NUM_BINS 1024   /* mdlib's distribution bin count */
#define VMD_VOLUME_DIM   128    /* mdlib's volume resolution, BASELINE config 4 "128^3" */

typedef enum vmd_distance_kind_t {
    VMD_DISTANCE_COM  = 0,      /* distance(a,b)      */
    VMD_DISTANCE_MIN  = 1,      /* distance_min(a,b)  */
    VMD_DISTANCE_MAX  = 2,      /* distance_max(a,b)  */
    VMD_DISTANCE_PAIR = 3       /* distance_pair(a,b) */
} vmd_distance_kind_t;

vmd_script_ir_t* vmd_ir_create(void);                                   /* md_script_ir_create, src/main.cpp:846 */
void             vmd_ir_free(vmd_script_ir_t* ir);                      /* md_script_ir_free,   src/main.cpp:968 */
/* `name = rdf(ref, target, {rmin,rmax});` — 0-based atom indices, copied.  src/main.cpp:528 */
bool vmd_ir_add_rdf(vmd_script_ir_t* ir, const char* name, const int32_t* ref, size_t nref,
                    const int32_t* target, size_t ntarget, float rmin, float rmax);
/* `name = sdf(structures, target, cutoff);` structures = K index lists of m atoms each.  src/main.cpp:528 */
bool vmd_ir_add_sdf(vmd_script_ir_t* ir, const char* name, const int32_t* structures, size_t K, size_t m,
                    const int32_t* target, size_t ntarget, float cutoff);
/* `name = distance*(a, b);`  src/main.cpp:2817-2858 */
bool vmd_ir_add_distance(vmd_script_ir_t* ir, const char* name, vmd_distance_kind_t kind,
                         const int32_t* a, size_t na, const int32_t* b, size_t nb);
/* `name = distance*(a, b) in <contexts>;` (src/main.cpp:2840-2858): a population of P contexts, one value (or pair block)
 * per context and frame -> dim[1] = P (x |a||b| for distance_pair).  a/b hold the absolute atom indices of all contexts
 * back to back; context c owns a[a_offsets[c] .. a_offsets[c+1]) and b[b_offsets[c] .. b_offsets[c+1]). */
bool vmd_ir_add_distance_population(vmd_script_ir_t* ir, const char* name, vmd_distance_kind_t kind, size_t P,
                                    const int32_t* a, const int32_t* a_offsets, const int32_t* b, const int32_t* b_offsets);
/* md_script_ir_compile_from_source stand-in (src/main.cpp:878) for the script subset of the hot path: statements
 * `s = <selection>;`, `r = rdf(sel, sel, rmax | rmin:rmax | {rmin, rmax});`, `v = sdf(structures, sel, cutoff);`,
 * `d = distance[_min|_max|_pair](sel, sel) [in <structures>];` with selections element('X'), type/name/label('X'),
 * resname("X"), residue(a:b) (1-based residue index), resid(a:b) (residue sequence number of the file), atom(a:b), a[:b] (1-based atom
 * indices, src/main.cpp:2817), all, water, protein,
 * and / or / not, parentheses, and sel[a:b] slicing an array of structures (viamd_amd/csrc/vmd_script.cpp).  Appends one
 * descriptor per property to `ir`; false + vmd_last_error() on a syntax or range error (the ir may then hold the properties
 * of the statements before the error).  The topology is the part of md_system_t selections resolve against. */
typedef struct vmd_topology_t {
    size_t num_atoms;
    const char* const* elements;     /* per atom, e.g. "O" */
    const char* const* names;        /* per atom (atom type / label); NULL = elements */
    const char* const* resnames;     /* per atom; NULL = "UNK" */
    const int32_t* residue_index;    /* per atom, 0-based; NULL = one residue */
    /* per atom: the residue sequence number the FILE carries (md_component_seq_id: PDB resSeq / GRO residue number), which is what
     * resid(a:b) selects by (src/main.cpp:2843-2848 emits `in resid(%i)` with the seq id next to `in residue(%i)` with the
     * 1-based residue index).  NULL = the host has none: resid() is then a compile error, never an alias of residue(). */
    const int32_t* residue_seq_id;
} vmd_topology_t;
bool     vmd_ir_compile_from_source(vmd_script_ir_t* ir, const char* source, const vmd_topology_t* topology);
/* The same, statement by statement: what the front-end understands is compiled, every other statement is REPORTED instead of failing the
 * script - VIAMD's own default script (src/main.cpp:528) carries `a1 = angle(2,1,3) in resname("ALA");` and
 * `{lin,plan,iso} = shape_weights(all);` next to its distance / rdf / sdf statements.  The report lists, per skipped statement, its
 * left-hand names ("a1", "lin,plan,iso"), its byte range in `source` (without the ';') and the reason; a statement that uses an identifier
 * of a skipped one is skipped with it.  vmd_script_report_fallback_source is `source` with the COMPILED property statements blanked out
 * (offsets unchanged, selections kept): the text the evaluator behind include/vmd_md_script_shim.h's fallback hooks compiles, so that no
 * property is evaluated twice.  Returns false only for NULL arguments or a malformed topology. */
typedef struct vmd_script_skipped_t {
    const char* names;       /* left-hand side as written, tuple members joined by ',' */
    size_t      beg, end;    /* [beg, end) in `source` */
    const char* reason;
} vmd_script_skipped_t;
typedef struct vmd_script_report_t vmd_script_report_t;
bool     vmd_ir_compile_from_source_partial(vmd_script_ir_t* ir, const char* source, const vmd_topology_t* topology, vmd_script_report_t** report);
size_t   vmd_script_report_skipped_count(const vmd_script_report_t* report);
const vmd_script_skipped_t* vmd_script_report_skipped(const vmd_script_report_t* report);
const char* vmd_script_report_fallback_source(const vmd_script_report_t* report);
void     vmd_script_report_free(vmd_script_report_t* report);
bool     vmd_ir_valid(const vmd_script_ir_t* ir);                       /* md_script_ir_valid, src/main.cpp:936 */
uint64_t vmd_ir_fingerprint(const vmd_script_ir_t* ir);                 /* md_script_ir_fingerprint, src/main.cpp:937 */
size_t   vmd_ir_property_count(const vmd_script_ir_t* ir);              /* md_script_ir_property_count, src/main.cpp:992,1277 */
const char* const* vmd_ir_property_names(const vmd_script_ir_t* ir);    /* md_script_ir_property_names, src/main.cpp:1278 */
vmd_property_flags_t vmd_ir_property_flags(const vmd_script_ir_t* ir, const char* name); /* src/main.cpp:1285 */
/* atom pairs ONE frame of the script asks for (rdf |ref| x |target|, sdf K x (|target| + m), distance |a| x |b| per context): the size a host
 * compares with a threshold before it sends a small script to the GPU (vmd_shim_set_min_work; VIAMD's default dataset, src/main.cpp:522-528) */
uint64_t vmd_ir_work_per_frame(const vmd_script_ir_t* ir);

/* ---- evaluation (md_script_eval_t stand-in) ------------------------------------------------------ */

typedef struct vmd_script_eval_t vmd_script_eval_t;

/* md_script_aggregate_t (src/main.cpp:1388-1449): per-frame population statistics of a temporal property */
typedef struct vmd_script_aggregate_t {
    size_t num_values;
    float* population_mean;
    float* population_var;
    float (*population_ext)[2];     /* vec2_t {min,max} */
} vmd_script_aggregate_t;

/* md_script_property_data_t — the fields VIAMD reads.  The struct address and the arrays
 * stay valid and fixed for the lifetime of the eval, across vmd_eval_clear_data (src/main.cpp:1286,1303).
 * Concurrent readers (VIAMD's GUI thread reads while pool threads are inside frame_range, src/main.cpp:1508-1524): the scalar
 * fields below change through relaxed atomic stores - a reader sees each of them whole, possibly old next to new; fingerprint
 * moves after the others.  The arrays are written by DMA and plain copies while a call runs: a reader may see a mix of old and
 * new elements (the reference's own contract), never a dangling pointer. */
typedef struct vmd_script_property_data_t {
    int32_t dim[4];                 /* [0] frames (temporal) | [2] bins (distribution) | [1..3] volume dims */
    float*  values;                 /* temporal: values[frame*dim[1]+i]; distribution: values[bin]; volume: x fastest.  READ THE FIELD WHEN YOU
                                     * READ THE DATA (VIAMD does: density_volume.cpp:279-283, src/main.cpp:5817): a volume's pointer is one of two
                                     * stable addresses - shared read-only zeros between clear_data and the evaluation's first view, the view's own
                                     * pinned pages otherwise (round 6: the 8.4 MB view is never zeroed) - changed by a relaxed atomic store */
    float*  weights;                /* distribution only */
    size_t  num_values;
    vmd_script_aggregate_t* aggregate;
    float   min_value, max_value;
    float   min_range[2], max_range[2];
    uint64_t fingerprint;
    /* extension: the exact integer accumulators behind values (distribution: dim[2], volume: dim[1]*dim[2]*dim[3]).
     * Distributions keep it current; for volumes (17 MB) call vmd_eval_refresh_counts before reading it. */
    const uint64_t* counts;
    const double*   weights64;
    /* md_script_property_data_t::unit[2] (x, y; src/main.cpp:1300-1301) as the strings VIAMD prints them into with md_unit_print
     * (src/main.cpp:1314-1315): "Å" (UTF-8) for a length, "" for none.  rdf: {"Å", ""}, sdf: {"", ""}, distance*: {"", "Å"}
     * (x of a temporal property is the frame axis; VIAMD labels it with the trajectory's time unit itself).  Static strings. */
    const char* unit_str[2];
} vmd_script_property_data_t;

/* md_script_eval_create(num_frames, ir, alloc), src/main.cpp:971 */
vmd_script_eval_t* vmd_eval_create(size_t num_frames, const vmd_script_ir_t* ir);
void     vmd_eval_free(vmd_script_eval_t* eval);                        /* md_script_eval_free, src/main.cpp:960 */
void     vmd_eval_clear_data(vmd_script_eval_t* eval);                  /* md_script_eval_clear_data, src/main.cpp:990 */
void     vmd_eval_interrupt(vmd_script_eval_t* eval);                   /* md_script_eval_interrupt, src/main.cpp:829,952,984 */
uint64_t vmd_eval_ir_fingerprint(const vmd_script_eval_t* eval);        /* md_script_eval_ir_fingerprint, src/main.cpp:987 */
/* the hot call — md_script_eval_frame_range, src/main.cpp:996,1032.  Re-entrant on one eval from many
 * threads with disjoint ranges; returns false on interrupt or error. */
bool vmd_eval_frame_range(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys,
                          vmd_trajectory_i* traj, uint32_t frame_beg, uint32_t frame_end);
/* md_script_eval_property_data, src/main.cpp:1286 */
const vmd_script_property_data_t* vmd_eval_property_data(const vmd_script_eval_t* eval, const char* name);
/* md_script_eval_frame_mask (src/main.cpp:1513): one byte per frame here, non-zero = evaluated */
const uint8_t* vmd_eval_frame_mask(const vmd_script_eval_t* eval);
/* the same mask as little-endian 64-bit words (VIAMD tests the bitfield bit by bit, src/main.cpp:194-210): bit (f & 63) of word f / 64 is
 * set when frame f is evaluated.  Writes min(cap, words) words and returns words = ceil(num_frames / 64); the shim turns them into an
 * md_bitfield_t through mdlib's own md_bitfield_clear / _set_bit (its storage layout is mdlib's business). */
size_t   vmd_eval_frame_mask_bits(const vmd_script_eval_t* eval, uint64_t* words, size_t cap);
size_t   vmd_eval_num_frames(const vmd_script_eval_t* eval);
size_t   vmd_eval_frames_done(const vmd_script_eval_t* eval);

/* md_script_vis_eval_payload(..., MD_SCRIPT_VISUALIZE_SDF) (density_volume.cpp:183-188, src/main.cpp:5751):
 * world->reference matrices of every reference structure of SDF property `name` at `frame`
 * (column-major mat4, as mat4_t), and the half extent.  matrices: K*16 floats. */
bool vmd_eval_sdf_matrices(vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys,
                           vmd_trajectory_i* traj, uint32_t frame, float* matrices, size_t* K_out, float* extent_out);

/* the rest of the vis payload VIAMD consumes next to the matrices (md_script_vis_eval_payload with MD_SCRIPT_VISUALIZE_ATOMS |
 * MD_SCRIPT_VISUALIZE_SDF: vis.sdf.structures, density_volume.cpp:263-269; export_cube writes the atoms of structure 0,
 * src/main.cpp:5793-5803): the K x m atom indices of the reference structures (eval-owned, valid for its lifetime) */
const int32_t* vmd_eval_sdf_structures(const vmd_script_eval_t* eval, const char* name, size_t* num_structures, size_t* atoms_per_structure);
typedef struct vmd_sdf_payload_t {
    size_t num_structures, atoms_per_structure;
    const int32_t* structures;      /* [num_structures][atoms_per_structure], eval-owned */
    const float*   matrices;        /* [num_structures][16] column-major world->reference of the requested frame; valid until the
                                       calling thread's next vmd_eval_sdf_payload */
    float extent;                   /* half edge of the volume in Angstrom (vis.sdf.extent) */
} vmd_sdf_payload_t;
bool vmd_eval_sdf_payload(vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys, vmd_trajectory_i* traj, uint32_t frame,
                          vmd_sdf_payload_t* out);

/* ---- export, viamd_amd/csrc/vmd_export.cpp: the files VIAMD writes from evaluated properties ---------------------
 * export_xvg / export_csv (src/main.cpp:5640-5716): columns[j][i], one label per column */
bool vmd_export_xvg(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows);
bool vmd_export_csv(const char* path, const float* const* columns, const char* const* labels, size_t num_columns, size_t num_rows);
/* the table the property-export window assembles for one property (src/main.cpp:5953-6040), labels included: temporal -> time
 * (frame_times, or the frame index when NULL; labelled "Frame", or "Time (<time_unit>)" when time_unit is a non-empty string) + the
 * values ("name", "name (<unit>)" when the property has a y unit, "name[i]" per population member); distribution -> num_bins
 * (0 = 128) display bins over [min_range[0], max_range[0]] labelled with the x unit + the downsampled histogram.  format: "xvg" or "csv" */
bool vmd_export_property_table(const char* path, vmd_script_eval_t* eval, const char* name, const char* format,
                               const double* frame_times, const char* time_unit, int num_bins);
/* export_cube (src/main.cpp:5718-5830): Gaussian cube file of volume property `name` in Bohr, x outermost / z innermost over the
 * x-fastest array, preceded by the atoms of reference structure 0 (coordinates of trajectory frame 0 through the
 * world->reference matrix of `frame` - VIAMD passes the displayed frame).  atomic_numbers: per atom, or NULL (written as 0) */
bool vmd_export_cube(const char* path, vmd_script_eval_t* eval, const char* name, const vmd_system_t* sys, vmd_trajectory_i* traj,
                     uint32_t frame, const uint8_t* atomic_numbers);

/* ---- multi-GPU merge: integer accumulators are exported / imported as plain device or host
 * buffers so the host runtime (torch.distributed = RCCL) can all-reduce them ------------------------- */
typedef struct vmd_accum_view_t {
    const char* name;
    vmd_property_flags_t flags;
    uint64_t* counts_dev;      /* device u64 accumulators (distribution bins or voxels), may be NULL */
    size_t    num_counts;
    double*   weights64;       /* host fp64, distribution only */
    size_t    num_weights;
    float*    temporal;        /* host, temporal rows */
    size_t    num_temporal;
    uint64_t  count_bound;     /* no element of counts_dev can exceed this after the merge of ALL frames (0 = unknown): a volume whose
                                  bound fits 32 bits is merged as u32, half the bytes on the links */
} vmd_accum_view_t;
size_t vmd_eval_accum_views(vmd_script_eval_t* eval, vmd_accum_view_t* out, size_t cap);
/* bring the host u64 mirror (`counts`) of a property up to date with the device accumulators */
bool   vmd_eval_refresh_counts(vmd_script_eval_t* eval, const char* name);
/* re-derive values/weights/aggregates from the accumulators (after an external reduce) */
bool   vmd_eval_finalize(vmd_script_eval_t* eval);
/* Deferred-settle mode (vmd_set_option("readahead_lone", 1); off by default).  mdlib's contract - and this library's default - is that
 * results are final when the last md_script_eval_frame_range call returns; since no call knows it is the last, a host that walks a range
 * frame by frame from ONE thread pays a full small evaluation per call (~0.25 ms).  With the option on, small calls are served by read-ahead
 * whoever makes them (evaluated ahead in regions, marked, committed by block) and the final settle - committing what was requested, the
 * ragged ends, the host views - runs on a helper thread once the eval has been quiet for readahead_lone_settle_us (300): results trail the
 * last call by that much, which a polling reader (VIAMD's GUI, src/main.cpp:1508-1524) does not notice.  vmd_eval_wait_settled performs
 * that settle at once on the calling thread (vmd_eval_finalize, vmd_eval_reduce and the exporters call it); `sys` and the trajectory of
 * the calls must stay valid until it has returned, or until interrupt / clear_data / free (all three drop a settle that is owed and wait for
 * one that is running).  A no-op for evals that are not in that mode. */
bool   vmd_eval_wait_settled(vmd_script_eval_t* eval);
/* the same choice per eval instead of per process: 1 = on, 0 = off, -1 = follow vmd_set_option("readahead_lone") (the default); read when an
 * evaluation makes its first small call, i.e. set it before the calls or before clear_data */
bool   vmd_eval_set_deferred_settle(vmd_script_eval_t* eval, int mode);
/* VIAMD's call pattern as a utility (src/main.cpp:993-997, src/task_system.cpp:73-81): num_threads threads (the caller is one of them) pull
 * ranges of `grain` frames off [frame_beg, frame_end) and call vmd_eval_frame_range on the one eval, each blocking until its frames are
 * evaluated.  bench.py times VIAMD's pattern with it; false + vmd_last_error() if a call failed, false + "" if it was interrupted. */
bool   vmd_eval_frame_range_pooled(vmd_script_eval_t* eval, const vmd_script_ir_t* ir, const vmd_system_t* sys, vmd_trajectory_i* traj,
                                   uint32_t frame_beg, uint32_t frame_end, int num_threads, uint32_t grain);
/* Deferred-settle mode: `fn(user)` is called after every settle the helper thread - or vmd_eval_wait_settled - has performed, on that
 * thread, with no lock of the eval held; clear_data / interrupt / free wait for a call that is running, none starts after they return.  A
 * host that caches scalar fields of the property records (include/vmd_md_script_shim.h re-publishes fingerprint / ranges / max_value to
 * VIAMD, which re-reads a property only when prop_data->fingerprint moves, src/main.cpp:1508-1509) refreshes its copies here.  fn must not
 * call clear_data / interrupt / free / wait_settled of the same eval.  Set before the evaluation's calls; NULL removes it. */
bool   vmd_eval_set_settled_callback(vmd_script_eval_t* eval, void (*fn)(void*), void* user);
/* A rank of a multi-GPU evaluation: do not materialise the float view of a VOLUME after every frame_range (8.4 MB over PCIe per call, for a
 * partial result nobody reads) - vmd_eval_finalize / vmd_eval_reduce derive it once, from the merged counts.  Distribution and temporal views
 * (a few KB) are kept current as ever.  Off by default: VIAMD reads `values` of a running evaluation (src/main.cpp:1508-1524). */
bool   vmd_eval_defer_volume_views(vmd_script_eval_t* eval, bool defer);
/* mark frames as evaluated elsewhere (after a mask all-reduce) */
void   vmd_eval_set_frame_mask(vmd_script_eval_t* eval, const uint8_t* mask, size_t n);

/* The merge itself, behind the ABI (viamd_amd/csrc/vmd_reduce.cpp).  VIAMD's evaluation is driven from C++
 * (src/main.cpp:993-1008: pool threads call md_script_eval_frame_range on disjoint ranges of one eval); across GPUs the same
 * happens one process per GPU, every rank on its block of frames, followed by ONE vmd_eval_reduce: the u64 accumulators are
 * summed in place on the device (8 KB per RDF, 16.8 MB per SDF volume), the host-side parts (fp64 weights, temporal rows,
 * frame mask) in one packed fp64 all-reduce, then the float views are re-derived (vmd_eval_finalize).  Afterwards every rank
 * holds the result of the whole trajectory, bit-identical in the integer parts for any rank count.
 * The collective is an interface so that hosts with their own transport can plug it in; vmd_comm_* is the RCCL one. */
typedef struct vmd_collective_i {
    void* inst;
    int  (*rank)(void* inst);
    int  (*size)(void* inst);
    /* in-place SUM over all ranks of `n` elements of DEVICE memory, enqueued on `stream` (hipStream_t) */
    bool (*allreduce_sum_u64)(void* inst, uint64_t* buf, size_t n, void* stream);
    bool (*allreduce_sum_f64)(void* inst, double* buf, size_t n, void* stream);
    /* optional, may be NULL.  group_begin / group_end bracket all all-reduces of one merge so that they leave as ONE collective
     * launch (ncclGroupStart / ncclGroupEnd); allreduce_sum_u32: in-place SUM of 32-bit device memory */
    bool (*group_begin)(void* inst);
    bool (*group_end)(void* inst);
    bool (*allreduce_sum_u32)(void* inst, uint32_t* buf, size_t n, void* stream);
} vmd_collective_i;
/* call on every rank after its last vmd_eval_frame_range has returned; `stream`: hipStream_t the collectives run on (NULL = the
 * default stream); synchronous on return */
bool   vmd_eval_reduce(vmd_script_eval_t* eval, const vmd_collective_i* coll, void* stream);
/* what the last vmd_eval_reduce of this eval did: bytes handed to the collective (per rank), all-reduce calls issued, whether they
 * were grouped into one launch, and the wall time of the call in milliseconds */
typedef struct vmd_reduce_stats_t {
    uint64_t bytes;
    uint32_t calls;
    uint32_t grouped;
    uint32_t volumes_as_u32;
    double   ms;
} vmd_reduce_stats_t;
void   vmd_eval_reduce_stats(const vmd_script_eval_t* eval, vmd_reduce_stats_t* out);

/* RCCL communicator (one per process = per GPU; xGMI inside a node).  librccl is loaded at run time by soname, so a host that
 * already carries an RCCL (e.g. PyTorch's) shares that copy.  Either let the library create the communicator - rank 0 makes the
 * id, the host program distributes its 128 bytes by whatever means it has (MPI, a file, a socket), every rank calls
 * vmd_comm_create on its own device - or wrap an ncclComm_t the host already owns (not destroyed by vmd_comm_destroy). */
#define VMD_COMM_ID_BYTES 128
typedef struct vmd_comm_t vmd_comm_t;
bool        vmd_comm_unique_id(uint8_t id[VMD_COMM_ID_BYTES]);                        /* ncclGetUniqueId */
vmd_comm_t* vmd_comm_create(int nranks, int rank, const uint8_t id[VMD_COMM_ID_BYTES]); /* ncclCommInitRank on the current device */
vmd_comm_t* vmd_comm_from_nccl(void* nccl_comm);                                      /* ncclComm_t owned by the caller */
void        vmd_comm_destroy(vmd_comm_t* comm);
const vmd_collective_i* vmd_comm_collective(vmd_comm_t* comm);
int         vmd_comm_rank(const vmd_comm_t* comm);
int         vmd_comm_size(const vmd_comm_t* comm);

/* ---- filtered evaluation (VIAMD: the "Eval Filt" task, src/main.cpp:1014-1039) ----------
 * VIAMD answers a timeline sub-range by re-running md_script_eval_frame_range over it on a second eval object
 * every time the range slider moves.  With frame blocks the full evaluation additionally keeps one partial
 * accumulator per block of `block_frames` frames in HBM (distribution: 8 KB, volume: 16 MB per block); a second
 * eval that names the first as its source then serves every whole block of a requested range with one u64 add
 * per bin / voxel and evaluates only the ragged frames at the two ends.  Results are bit-identical to a plain
 * evaluation of the same range (integer counts; fp64 weights to the last bits of a different summation order). */
/* keep block partials from now on (0 = off).  Call before the first frame_range or right after clear_data.
 * A block is stored when one frame_range call (or one merged group of concurrent calls) covers it entirely. */
bool   vmd_eval_set_block_frames(vmd_script_eval_t* eval, size_t block_frames);
/* let `eval` reuse the block partials of `source` (same IR fingerprint, same num_frames, same device); NULL detaches.
 * `source` must outlive the attachment and must not be cleared while `eval` evaluates. */
bool   vmd_eval_set_source(vmd_script_eval_t* eval, vmd_script_eval_t* source);
/* frames evaluated by kernels / frames served from block partials since the last clear_data */
void   vmd_eval_frame_stats(const vmd_script_eval_t* eval, size_t* frames_computed, size_t* frames_reused);
/* the two-level cell build of this eval's selections: how often a pencil bucket overflowed (each time the batch's pair passes were repeated
 * with wider buckets for THAT selection) and how many selections have given the buckets up for the single-level builds (three overflows) */
void   vmd_eval_cell_build_stats(const vmd_script_eval_t* eval, size_t* bucket_overflows, size_t* selections_off_buckets);
/* Read-ahead under VIAMD's call pattern (pool threads, a frame or a few per call: /root/reference/src/main.cpp:993-997,
 * src/task_system.cpp:73-81): what this eval did since it was created.  engaged = small concurrent calls were recognised and regions of
 * frame blocks evaluated ahead; slow_calls = calls that led or waited for a region (all others only marked their frames requested);
 * regions / region_frames = what was evaluated ahead; committed_blocks = block partials that joined the totals; direct_frames = frames a settle evaluated one by one (ragged ends of a
 * range); settles = calls that left alone and brought accumulators and views up to date.  Options: readahead (0 = off), readahead_frames,
 * readahead_growth, readahead_small, readahead_block, readahead_linger_us, readahead_company_us (vmd_set_option). */
typedef struct vmd_readahead_stats_t {
    uint32_t engaged, block_frames;
    uint64_t regions, region_frames, slow_calls, settles, direct_frames, committed_blocks;
} vmd_readahead_stats_t;
void   vmd_eval_readahead_stats(const vmd_script_eval_t* eval, vmd_readahead_stats_t* out);
/* frames whose coordinates were decompressed on the device (load_raw + k_xtc_wave) since the last clear_data, and how many of them
 * from decoder checkpoints (in sections, no walk from bit 0: the trajectory's frames had been decoded before - by any eval) */
size_t vmd_eval_frames_device_decoded(const vmd_script_eval_t* eval);
size_t vmd_eval_frames_section_decoded(const vmd_script_eval_t* eval);
/* ... and how many of them reached the device by DMA straight from the mapped file (raw_mapped_view), no host copy */
size_t vmd_eval_frames_mapped(const vmd_script_eval_t* eval);
/* Decoder checkpoints across processes.  The first device decode of a compressed (XTC) frame has to walk its bit stream from the
 * first bit and leaves the decoder state at up to 64 places of the frame; every later decode enters there (sections, no walk).  The
 * table lives with the trajectory for the process's lifetime; these two calls carry it over to the next one, the way mdlib keeps a
 * frame-offset cache file next to a trajectory (ref: ext/mdlib's xtc / trr loaders behind src/loader.cpp:147-150 write `.cache` files).
 * save: writes the table of `traj` (1 KB per frame) to `path` - false when nothing of the trajectory has been device-decoded yet.
 * load: installs a table for `traj` on `device`; returns the number of frames it covers, 0 when the file describes another
 * trajectory (frame or atom count), -1 on error.  A loaded table is a hint, never trusted: a frame uses its checkpoints only while
 * the signature of its bytes matches the stored one, and a section whose end state disagrees with the next checkpoint is rejected:
 * that batch is decoded by the host reader and its frames walk from bit 0 again the next time. */
bool   vmd_ckcache_save(const vmd_trajectory_i* traj, const char* path);
long   vmd_ckcache_load(const vmd_trajectory_i* traj, const char* path, int device);

/* ---- device-resident trajectories (pre-staged in HBM) ---------------------------------- */
typedef struct vmd_devtraj_t vmd_devtraj_t;
vmd_devtraj_t*    vmd_devtraj_create(size_t num_frames, size_t num_atoms);
/* one rank's shard of a frame-sharded trajectory: reports `num_frames_total` frames, keeps only
 * [frame_beg, frame_end) resident; every frame index of this API stays global */
vmd_devtraj_t*    vmd_devtraj_create_shard(size_t num_frames_total, size_t frame_beg, size_t frame_end, size_t num_atoms);
void              vmd_devtraj_free(vmd_devtraj_t* t);
vmd_trajectory_i* vmd_devtraj_interface(vmd_devtraj_t* t);
bool vmd_devtraj_upload_frame(vmd_devtraj_t* t, size_t frame, const vmd_unitcell_t* cell,
                              const float* x, const float* y, const float* z);
/* overwrite atoms [first_atom, first_atom+atom_count) of frames [frame_beg, frame_beg+frame_count): xyz is
 * host float[frame_count][3][atom_count] */
bool vmd_devtraj_upload_atoms(vmd_devtraj_t* t, size_t frame_beg, size_t frame_count, size_t first_atom, size_t atom_count,
                              const float* xyz);
/* seeded synthetic water box of SURVEY 8d (same integer RNG as oracle S9) for frames [beg,end) */
bool vmd_devtraj_synth(vmd_devtraj_t* t, uint64_t seed, float L, float sigma, uint32_t n_blob,
                       size_t frame_beg, size_t frame_end);
/* replace the unit cell of frames [beg,end) (coordinates stay as they are: they are wrapped on use) */
bool vmd_devtraj_set_cell(vmd_devtraj_t* t, size_t frame_beg, size_t frame_end, const vmd_unitcell_t* cell);
/* device address of the first RESIDENT frame */
float* vmd_devtraj_device_ptr(vmd_devtraj_t* t, size_t* frame_stride, size_t* row_stride);

/* DCD (CHARMM / NAMD) trajectory file as a vmd_trajectory_i — VIAMD attaches these through md_dcd_attach_from_file
 * (src/loader.cpp:151-152).  Random access, either byte order, unit-cell block -> {x,y,z,xy,xz,yz}; NULL + vmd_last_error
 * on failure.  load_frame is thread safe (pread). */
typedef struct vmd_dcdtraj_t vmd_dcdtraj_t;
vmd_dcdtraj_t*    vmd_dcdtraj_open(const char* path);
void              vmd_dcdtraj_close(vmd_dcdtraj_t* t);
vmd_trajectory_i* vmd_dcdtraj_interface(vmd_dcdtraj_t* t);

/* Text trajectories as a vmd_trajectory_i: multi-MODEL PDB (BASELINE configs[0] is one), XYZ / XMOL (incl. extended-XYZ Lattice="..."),
 * LAMMPS dump files - the remaining LoaderFlag_Trajectory types of VIAMD's loader table (src/loader.cpp:22-77; attached through mdlib,
 * :111-159).  The file is mapped, one pass indexes the frames (byte range, atom count, unit cell per frame), load_frame parses frame f into
 * the rows it is handed: re-entrant, so the evaluator decodes a staged batch on its load threads.  format: "pdb" | "xyz" | "xmol" | "arc" |
 * "lammpstrj", or NULL = by the file name's extension.  Numbers are converted like (float)strtod(text), without the locale.
 * NULL + vmd_last_error on failure (unknown type, no frames, a frame whose atom count differs from the first). */
typedef struct vmd_texttraj_t vmd_texttraj_t;
vmd_texttraj_t*   vmd_texttraj_open(const char* path, const char* format);
void              vmd_texttraj_close(vmd_texttraj_t* t);
vmd_trajectory_i* vmd_texttraj_interface(vmd_texttraj_t* t);
/* The SYSTEM of a PDB file (LoaderFlag_System, md_pdb_system_init_from_file, src/loader.cpp:113-128) as far as this path reads it: the
 * topology the script front-end resolves selections against (element, atom name, residue name, residue index, resSeq of the first model's
 * atoms), masses from the elements, the first model's coordinates ([3][num_atoms]: x row, y row, z row) and the CRYST1 cell.  With it a host
 * without mdlib runs BASELINE configs[0] from the file alone: vmd_textsys_open + vmd_texttraj_open + vmd_ir_compile_from_source. */
typedef struct vmd_textsys_t vmd_textsys_t;
vmd_textsys_t*        vmd_textsys_open(const char* path);
void                  vmd_textsys_close(vmd_textsys_t* s);
const vmd_topology_t* vmd_textsys_topology(const vmd_textsys_t* s);
const float*          vmd_textsys_mass(const vmd_textsys_t* s);
const float*          vmd_textsys_coords(const vmd_textsys_t* s, vmd_unitcell_t* cell);

/* GROMACS XTC (compressed) / TRR trajectory file as a vmd_trajectory_i - VIAMD attaches these through md_xtc_attach_from_file /
 * md_trr_attach_from_file (src/loader.cpp:147-150).  The file type is taken from the magic number; a frame-offset index is
 * built on open; nm -> Angstrom; box rows -> {x,y,z,xy,xz,yz}; TRR frames without positions are skipped.  load_frame is
 * thread safe (pread + thread-local scratch), so staged batches are decompressed on several host threads. */
typedef struct vmd_xdrtraj_t vmd_xdrtraj_t;
vmd_xdrtraj_t*    vmd_xdrtraj_open(const char* path);
void              vmd_xdrtraj_close(vmd_xdrtraj_t* t);
vmd_trajectory_i* vmd_xdrtraj_interface(vmd_xdrtraj_t* t);
int               vmd_xdrtraj_kind(const vmd_xdrtraj_t* t);                 /* 0 = XTC, 1 = TRR */
int64_t           vmd_xdrtraj_frame_step(const vmd_xdrtraj_t* t, size_t frame);   /* MD step number stored with the frame */
/* writer for the same two formats (test fixtures, `bench.py --traj xtc`): kind 0 = XTC at `precision` (1000 = 0.001 nm),
 * 1 = TRR single precision.  Coordinates and cell in Angstrom. */
typedef struct vmd_xdrwriter_t vmd_xdrwriter_t;
vmd_xdrwriter_t*  vmd_xdrwriter_open(const char* path, int kind, size_t num_atoms, float precision);
bool              vmd_xdrwriter_write_frame(vmd_xdrwriter_t* w, int64_t step, float time_ps, const vmd_unitcell_t* cell,
                                            const float* x, const float* y, const float* z);
bool              vmd_xdrwriter_close(vmd_xdrwriter_t* w);

/* A trajectory kept COMPRESSED in HBM: every frame of `src` (which must offer load_raw, i.e. an XTC file) is read once and uploaded
 * as stored; evaluations decompress their batches on the device straight from that copy (raw_device_view) - no host work, no PCIe
 * traffic per evaluation, 0.3 - 0.5 of the float footprint.  VIAMD's counterpart is the host-side frame cache of the loader
 * (src/loader.cpp:111-159).  `src` is borrowed and must outlive the object (single frames for vis payloads and streams the device
 * rejects are read through it). */
typedef struct vmd_rawtraj_t vmd_rawtraj_t;
vmd_rawtraj_t*    vmd_rawtraj_create(vmd_trajectory_i* src);
void              vmd_rawtraj_free(vmd_rawtraj_t* t);
vmd_trajectory_i* vmd_rawtraj_interface(vmd_rawtraj_t* t);
size_t            vmd_rawtraj_device_bytes(const vmd_rawtraj_t* t);         /* bytes of HBM the compressed frames occupy */

/* host-resident trajectory in pinned memory, float[F][3][npad] (the PCIe-inclusive path: frames cross the bus per batch) */
typedef struct vmd_hosttraj_t vmd_hosttraj_t;
vmd_hosttraj_t*   vmd_hosttraj_create(size_t num_frames, size_t num_atoms);
void              vmd_hosttraj_free(vmd_hosttraj_t* t);
vmd_trajectory_i* vmd_hosttraj_interface(vmd_hosttraj_t* t);
float*            vmd_hosttraj_frame_ptr(vmd_hosttraj_t* t, size_t frame, size_t* row_stride);   /* x row; y at +row_stride, z at +2*row_stride */
bool              vmd_hosttraj_set_cell(vmd_hosttraj_t* t, size_t frame, const vmd_unitcell_t* cell);
/* copy frames [frame_beg, frame_end) of a device trajectory into the host trajectory (same atom count) */
bool              vmd_hosttraj_copy_from_device(vmd_hosttraj_t* t, vmd_devtraj_t* src, size_t frame_beg, size_t frame_end);

/* ---- consumer post-processing VIAMD applies to the results (src/main.cpp:139-250), host side ------- */
void vmd_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights,
                              int num_src_bins);
void vmd_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                  int dim, const uint8_t* frame_mask, int num_frames, bool aggregate);
/* the same + y_range[2] = {y_min, y_max} of the histogram as VIAMD stores them (src/main.cpp:212-229); NULL = not wanted */
void vmd_compute_histogram_masked_y(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                    int dim, const uint8_t* frame_mask, int num_frames, bool aggregate, float* y_range);
/* compute_histogram (src/main.cpp:139-170) and scale_histogram (:252-261) */
void vmd_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values, int num_values,
                           float* bin_val_min, float* bin_val_max);
void vmd_scale_histogram(float* bins, const float* weights, int num_bins);

/* ---- runtime ---------------------------------------------------------------------------------------- */
int         vmd_device_count(void);                 /* 0 when no HIP device is usable */
bool        vmd_set_device(int device);
const char* vmd_last_error(void);                   /* thread-local message of the last failure */
void        vmd_clear_last_error(void);             /* a caller that handled a failure itself leaves no stale message behind */
/* md_log_register analogue (VIAMD turns mdlib's log messages into toasts, src/main.cpp:384-420): evaluator failures are
 * delivered to `fn` (from whichever thread hit them) instead of stderr; NULL restores stderr. */
#define VMD_LOG_INFO  1
#define VMD_LOG_ERROR 2
typedef void (*vmd_log_fn)(int level, const char* message, void* user);
void        vmd_log_register(vmd_log_fn fn, void* user);
void        vmd_log_message(int level, const char* message);      /* a layer above the ABI (the shim) reports through the same channel */
const char* vmd_version(void);
/* tuning knobs (kernel variant, frames per batch); returns previous value, -1 for unknown key */
int         vmd_set_option(const char* key, int value);
/* Evals are created and freed per script edit (src/main.cpp:960-972): the device blocks, pinned blocks, streams and events one gives up
 * are kept in a process-wide cache for the next (vmd_set_option("pool_mb", MB): bound on the cached device bytes, 0 = no cache).
 * vmd_pool_trim returns everything cached to the runtime (before another library needs the device memory, before exit). */
void        vmd_pool_trim(void);
void        vmd_pool_stats(size_t* cached_device_bytes, size_t* cached_pinned_bytes, size_t* cached_blocks);   /* any pointer may be NULL */
/* wall-clock ms of kernel `which` accumulated by hipEvents since the last reset (bench instrumentation) */
void        vmd_profile_reset(void);
double      vmd_profile_ms(const char* which, uint64_t* launches);
void        vmd_profile_enable(bool on);

#ifdef __cplusplus
}
#endif
#endif
