/*
 * include/vmd_hip.h — thin C-ABI layer over the hand-written gfx950 kernels (no host state, no torch types).
 *
 * Every function enqueues work on `stream` (a hipStream_t passed as void*) and returns 0 or a hipError_t.
 * Pointers are device pointers unless a comment says "host".  A *frame batch* is B consecutive frames:
 * frame b has x at xyz + b*frame_stride, y at +row_stride, z at +2*row_stride (floats); boxes[b*9+{0,1,2}]
 * are its cell edge lengths (SPEC S1), boxes[b*9+{3,4,5}] their reciprocals fl(1.0f/L) (SPEC S2) and boxes[b*9+{6,7,8}]
 * the tilt factors xy, xz, yz, all filled by the host.  pbc_flags: bits 0-2 periodic axes, bit 3 triclinic (SPEC S3t;
 * brute / sdf / distance kernels only — the pencil grid is orthorhombic).
 *
 * Reference functions replaced (the sources are in the empty submodule ext/mdlib, /root/reference/.gitmodules:10-12;
 * names from /root/reference/ext/ImGuiColorTextEdit/TextEditor.cpp:3318-3331 and SURVEY.md 8a):
 *   vmd_hip_cells_*   <- md_spatial_hash build            (a5)
 *   vmd_hip_rdf_*     <- rdf() pair loop + md_spatial_hash query (a4, a5)
 *   vmd_hip_sdf_*     <- sdf() alignment + density-volume accumulation (a6, a7, a9)
 *   vmd_hip_distance  <- distance / distance_min / distance_max / distance_pair (a8)
 *   vmd_hip_xtc_decode <- md_xtc frame decompression (f1; /root/reference/src/loader.cpp:147-148)
 */
#ifndef VMD_HIP_H
#define VMD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pencil grid of one batch: ny*nz pencils along x, each cut into nxf fine cells (SPEC S3 note) */
typedef struct vmd_grid_t {
    int32_t nxf, ny, nz;
    int32_t ncell;          /* nxf*ny*nz */
} vmd_grid_t;

/* K1: bin the `nsel` selected atoms of every frame of the batch into the grid and write them cell-sorted.
 *   sel        int32[nsel] atom indices (NULL = 0..nsel-1)
 *   cell_count u32[B][ncell+1]  scratch, zeroed by this call
 *   rank       u32[B][nsel]     scratch
 *   cell_start u32[B][ncell+1]  out: exclusive prefix of the cell populations
 *   sorted     f32[B][3][nsel_pad] out: wrapped coordinates in cell order (x row, y row, z row)
 *   aos        f32[B][nsel_pad][4] scratch or NULL: when given, the sort scatters one 16-byte record per atom and a
 *              coalesced repack pass produces `sorted` (fewer scattered L2 transactions) */
/* selections of <= 65536 atoms on grids with ncell + 1 <= 24576 are built by one block per frame entirely in LDS
 * (count -> scan -> scatter); larger ones take the three-kernel path with global atomics */
int vmd_hip_cells_fused_ok(vmd_grid_t grid, int nsel);
int vmd_hip_set_cells_fused(int on);   /* tuning / A-B switch, returns the previous value */
/* selections above 64k atoms: G blocks per frame, each with the cell table in LDS (no global atomics); 0 = not applicable */
int vmd_hip_cells_split_blocks(vmd_grid_t grid, int nsel);
int vmd_hip_set_cells_split(int on);   /* A-B switch, returns the previous value */
/* u32 words per frame the `rank` scratch of vmd_hip_cells_build must hold for this grid and selection */
size_t vmd_hip_cells_scratch_words(vmd_grid_t grid, int nsel);
/* bounding box of all atoms of each frame: out f32[B][6] = {min x,y,z, max x,y,z} (open axes: the grid spans this box) */
int vmd_hip_bbox(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, int B, int natoms, float* out);
/* pbc_flags: bits 0..2 periodic axes, bit 3 triclinic.  On an open axis boxes[b] holds the extent of the batch's bounding box
 * in the L slot, its inverse in the 1/L slot and its origin in slot 6 + axis (the tilt slots, unused without bit 3). */
int vmd_hip_cells_build(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                        const float* boxes, uint32_t pbc_flags, int B, const int32_t* sel, int nsel, int nsel_pad,
                        vmd_grid_t grid, uint32_t* cell_count, uint32_t* rank, uint32_t* cell_start, float* sorted,
                        float* aos);

/* K1, two-level build (the default whenever the grid has <= 4096 pencils): the frame is read once, every atom travels as one
 * 16-byte record {x, y, z, fine cell} through a per-pencil bucket, and one block per (frame, pencil) sorts its bucket in LDS and
 * writes its stretch of `sorted` / `cell_start` coalesced (k_cells_bin, k_cells_pen_scan, k_cells_pen_sort).
 *   pen_off    u32[npen + 1]: exclusive prefix of the bucket capacities in records (npen = ny*nz), chosen by the host from
 *              vmd_hip_cells_pencil_count of a few frames plus head room; every capacity <= vmd_hip_cells_pencil_cap_max()
 *   total_cap  pen_off[npen]; cap_max: the largest capacity
 *   pen_count  u32[B][npen] scratch (zeroed by the call), pen_start u32[B][npen + 1] scratch, bucket f32[B][total_cap][4] scratch
 *   overflow   u32[1]: set to 1 when an atom found its bucket full (never cleared here).  The sorted copy is then incomplete;
 *              vmd_hip_rdf_pencil / vmd_hip_axpy_u64 given the same flag do nothing, the caller enlarges the buckets and repeats */
/* the bit a bucket overflow of the NEXT vmd_hip_cells_build_pencil calls of this host thread ORs into *overflow (default 1): the evaluator
 * gives every selection its own, so that only the selection that overflowed gets wider buckets; returns the previous value */
uint32_t vmd_hip_set_cells_overflow_bit(uint32_t bit);
int vmd_hip_cells_pencil_ok(vmd_grid_t grid);
int vmd_hip_set_cells_pencil(int on);  /* A-B switch (0 = the single-level builds below), returns the previous value */
int vmd_hip_cells_pencil_cap_max(void);
/* atoms per pencil of S frames: counts u32[S][npen], zeroed by the call */
int vmd_hip_cells_pencil_count(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, const float* boxes,
                               uint32_t pbc_flags, int S, const int32_t* sel, int nsel, vmd_grid_t grid, uint32_t* counts);
int vmd_hip_cells_build_pencil(void* stream, const float* xyz, size_t frame_stride, size_t row_stride, const float* boxes,
                               uint32_t pbc_flags, int B, const int32_t* sel, int nsel, int nsel_pad, vmd_grid_t grid,
                               const uint32_t* pen_off, int total_cap, int cap_max, uint32_t* pen_count, uint32_t* pen_start,
                               float* bucket, uint32_t* overflow, uint32_t* cell_start, float* sorted);

/* K2: RDF pair histogram over a batch from cell-sorted selections (ref may equal tgt -> half shell).
 *   partial   u64[vmd_hip_rdf_partial_words()] scratch (per-wave rows + the work counter)
 *   counts    u64[nbins]  accumulated (+=) with device atomics
 *   variant   0 = wave queue (one compaction per candidate column; the default of the evaluator), 1 = inline hit path,
 *             2 = wave queue with pair entries (two columns share one compaction), 3 = 0 behind a bounding-box test of the j
 *             windows; 2 and 3 are measured A/B options (slower), all four give identical counts
 *   pbc_flags as for vmd_hip_cells_build (the same boxes and flags the selections were sorted with).  Triclinic: boxes
 *             carry tilt factors, cells live in the unsheared coordinates s_k * L_k (SPEC S3t) and the grid edge must be
 *             >= rmax measured perpendicular to the cell faces.  Open axes: no images, neighbours end at the bounding box. */
int vmd_hip_rdf_num_blocks(void);
int vmd_hip_set_rdf_pop(int mode);    /* how k_rdf_pencil drains its hit stack when r_min == 0: 0 = the 9-instruction pop, 1 (default) = margin folded into the constant, spare bin, stack read with ds_read_addtid_b32 (6); returns the previous value */
int vmd_hip_set_rdf_nsub(int n);       /* tuning knob: work items per pencil (1..64, 0 = automatic), returns the previous value */
int vmd_hip_set_rdf_nsub_pct(int pct);  /* tuning knob: the automatic number of work items per pencil as a percentage of the mean number of i-chunks per pencil */
int vmd_hip_set_rdf_shared_hist(int on); /* A-B switch: one LDS histogram per block instead of one per wave, returns the previous value */
int vmd_hip_set_cells_bin_lds(int on); /* A-B switch: level 1 of the two-level cell build orders a block's records by pencil in LDS and writes
                                        * them as coalesced runs (default on) instead of one scattered record per lane; returns the previous value */
int vmd_hip_set_cells_rec3(int on);   /* A-B switch: 12-byte bucket records {x, y, z} in the two-level cell build where the fine cell follows
                                        * from the wrapped x alone (x-periodic, non-triclinic cells); default on; returns the previous value */
void vmd_hip_set_pencil_reach(int ry, int rz); /* A-B switch: neighbour reach of the pencil walk in y / z (1 = pencils of cross-section >= rmax,
                                                 * 2 = split pencils >= rmax/2: 5 instead of 3 neighbours on that axis, x windows shrunk for the
                                                 * outer ones); process-wide, the grid passed to the cell build and to the walk must be cut to match */
int vmd_hip_set_rdf_blocks(int n);     /* tuning knob: persistent grid size (8..2048), returns the previous value */
size_t vmd_hip_rdf_partial_words(void);
int vmd_hip_rdf_pencil(void* stream, const float* sorted_ref, const uint32_t* cell_start_ref, int nref, int nref_pad,
                       const float* sorted_tgt, const uint32_t* cell_start_tgt, int ntgt, int ntgt_pad,
                       const float* boxes, int B, vmd_grid_t grid, float rmin, float rmax, int nbins,
                       int same_set, int variant, uint32_t pbc_flags, uint64_t* partial, uint64_t* counts,
                       const uint32_t* skip_flag /* device u32 or NULL: non-zero = do nothing (see vmd_hip_cells_build_pencil) */);

/* general RDF (any periodicity flags, any cutoff, no grid): O(nref*ntgt) per frame, SPEC S3 by comparison */
int vmd_hip_rdf_brute(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                      const float* boxes, uint32_t pbc_flags, int B,
                      const int32_t* ref, int nref, const int32_t* tgt, int ntgt,
                      float rmin, float rmax, int nbins, uint64_t* counts);

/* K3: per frame, per reference structure alignment (fp64, SPEC S5).
 *   structs  int32[K][m], mass f32[K][m], ref_pose f64[m][3] (COM-centred)
 *   R32 f32[B][K][9], c32 f32[B][K][3] out;  M64 f64[B][K][12] out (optional, may be NULL)
 *   group f32[B][4] out (optional): centre + radius of the set of structure COMs, the scatter's one-test pre-filter
 *   tree_order, tree_parent  int32[K][m] or both NULL: make every structure whole along its bond tree (atom order[t] hangs on atom
 *          parent[order[t]], local indices, parent < 0 = root) instead of along the index order; tree_pos f64[B*K][m][3] scratch */
int vmd_hip_sdf_align(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                      const float* boxes, uint32_t pbc_flags, int B,
                      const int32_t* structs, const float* mass, int K, int m, const double* ref_pose,
                      float* R32, float* c32, double* M64, float* group,
                      const int32_t* tree_order, const int32_t* tree_parent, double* tree_pos);
/* reference pose from one frame (structure 0): ref_pose f64[m][3] out; tree_order / tree_parent: int32[m] of structure 0 or NULL */
int vmd_hip_sdf_ref_pose(void* stream, const float* xyz, size_t row_stride, const float* box, uint32_t pbc_flags,
                         const int32_t* struct0, const float* mass0, int m, double* ref_pose,
                         const int32_t* tree_order, const int32_t* tree_parent);
/* K4: scatter target atoms of every frame into the dim^3 u64 volume (x fastest), SPEC S5.
 *   owner  int8[ntgt]: index of the structure target t belongs to, -1 if none (NULL: membership is searched in structs;
 *          an atom listed in several structures needs NULL) */
int vmd_hip_sdf_scatter(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                        const float* boxes, uint32_t pbc_flags, int B,
                        const int32_t* structs, int K, int m, const float* R32, const float* c32,
                        const int32_t* tgt, const int8_t* owner, int ntgt, float extent, int dim, uint64_t* volume,
                        const float* group /* from vmd_hip_sdf_align, or NULL */,
                        const uint8_t* atom_tag /* u8[row_stride] or NULL: dense-target path, one tag per ATOM: 255 = not a
                                                   target, 254 = target, k <= 253 = target that belongs to structure k */,
                        int tgt_first, int tgt_stride /* tgt_stride > 0: target t is atom tgt_first + t*tgt_stride (the index list
                                                         is an arithmetic progression: no index load in front of the gathers) */,
                        int unowned /* 1: no target is a member of any structure (owner all -1): skip the owner loads */);
int vmd_hip_set_sdf_rows(int n);       /* tuning knob: row-streaming scatter for progression targets with stride <= 4 (0 = off; 1, 2, 4 groups of 4 atoms per thread) */
int vmd_hip_set_sdf_ilp(int n);        /* tuning knob: target atoms per thread of the scatter (4 or 8), returns the previous value */
int vmd_hip_set_rdf_nsplit(int n);     /* pair launches with fewer work items than resident waves deal a chunk's neighbour pencils to n items each: -1 automatic (5 same-set / 9), 0 off */
int vmd_hip_set_sdf_wave(int on);      /* SDF scatter kernel: 0 = per-block compaction of the group test's survivors, 1 = per wave (no block barrier), 2 = the persistent streaming kernel (a wave walks tiles, next tile's gathers in flight) on 2 048 blocks, n >= 16 = on n blocks */

/* K5: distance family, one row per frame: out f32[B][P*per].  kind as vmd_distance_kind_t; P contexts (population);
 * context c uses a[aoff[c]..aoff[c+1]) and b[boff[c]..boff[c+1]); per = 1 (COM/MIN/MAX) or |a_c|*|b_c| (PAIR, equal
 * for all contexts).  mass_a/mass_b parallel to a/b (COM only). */
int vmd_hip_distance(void* stream, const float* xyz, size_t frame_stride, size_t row_stride,
                     const float* boxes, uint32_t pbc_flags, int B, int kind, int P, int per,
                     const int32_t* a, const float* mass_a, const int32_t* aoff,
                     const int32_t* b, const float* mass_b, const int32_t* boff, float* out);

/* dst[i] += mult * src[i] (u64): one pair pass feeding several histograms; does nothing when *skip_flag != 0 */
int vmd_hip_axpy_u64(void* stream, uint64_t* dst, const uint64_t* src, size_t n, uint64_t mult, const uint32_t* skip_flag);
/* dst[i] += src[i] (u64): merges a frame block's partial accumulator into the totals */
int vmd_hip_add_u64(void* stream, uint64_t* dst, const uint64_t* src, size_t n);
/* u64 counters -> f32 values (values[i] = fl((float)counts[i] * scale); scale = 1: the raw counts of SPEC S5) + max reduction into
 * max_out[0] (device f32) */
int vmd_hip_counts_to_float(void* stream, const uint64_t* counts, size_t n, float* values, float* max_out, float scale);
/* 1: k_sdf_scatter reads the frame with non-temporal loads; returns the previous value */
int vmd_hip_set_sdf_nt(int on);
/* candidate columns (one target atom against the 64 reference atoms of a chunk: 64 candidate lanes) k_rdf_pencil has walked on the
 * current device since the counter was last reset; synchronises the device */
uint64_t vmd_hip_rdf_columns(int reset);
/* counts[0] += value on the device (closed-interval RDF: the self pairs a half-shell pass never visits) */
int vmd_hip_bump_u64(void* stream, uint64_t* p, uint64_t value);
/* the selection the NEXT vmd_hip_cells_* calls of this host thread sort is periodic: atom(t) = first + (t / m) * period + off[t % m], 1 <= m <= 4
 * (the O of every water: m = 1, period 3) - the kernels compute it instead of reading sel[t].  m = 0: read the list (the default). */
void vmd_hip_set_cells_sel_pattern(int m, int first, int period, const int* off);
/* an empty kernel named k_marker_timed_region: a profiled command marks where its timed region begins (bench.py, scripts/pmc_traffic.py) */
int vmd_hip_marker(void* stream);
/* DECISION(D-RDF-OPEN) as a switch: 1 = hit iff r_min <= d <= r_max in the pair kernels launched from now on; returns the old value */
int vmd_hip_set_rdf_closed(int on);
int vmd_hip_set_rdf_raw(int on);        /* vmd_hip_rdf_brute: positions enter the pair computation unwrapped, minimum image by rounding (oracle/SPEC.md D-WRAP flipped); per host thread; returns the previous value */

/* XTC coordinate blocks decompressed on the device (the compressed bytes cross PCIe, not the floats): one
 * thread per frame walks its bit stream (frames are independent, a stream is strictly sequential).
 *   raw     u8: the bit streams, frame b at raw + info[b].offset (64-byte aligned, >= 32 readable bytes behind each stream)
 *   info    one record per frame, host byte order
 *   xyz     out, frame layout as above, Angstrom: fl(fl(int * fl(1/precision)) * 10) like the host reader (vmd_xdr.cpp)
 *   status  u32[B] out: 0 ok, 1 corrupt stream, 2 not supported on the device (a packed triple whose value needs more than 64 bits) */
typedef struct vmd_xtc_frame_t {
    float    precision;
    int32_t  minint[3], maxint[3];
    int32_t  smallidx;
    uint64_t offset, nbytes;
} vmd_xtc_frame_t;
/* Frames stored as plain floats (TRR, DCD), DMA'd as they lie in the file: swap / scale / transpose into the frame layout above.
 * raw + info[b].offset[c] + 4 * stride * i = component c of atom i of frame b (4-byte aligned); scale is one fp32 multiply, skipped
 * when it is 1 (the host readers do the same: vmd_xdr.cpp trr_load, vmd_dcd.cpp dcd_load_frame). */
typedef struct vmd_f32_frame_t {
    uint64_t offset[3];
    uint32_t stride, flags;     /* flags: bit 0 = big-endian */
    float    scale;
    uint32_t reserved;
} vmd_f32_frame_t;
int vmd_hip_raw_f32_decode(void* stream, const unsigned char* raw, const vmd_f32_frame_t* info, int B, int natoms,
                           float* xyz, size_t frame_stride, size_t row_stride);

int vmd_hip_xtc_decode(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                       float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status);
/* the same result in two passes: k_xtc_index (one thread per frame) follows only flags and field widths and drops a checkpoint
 * at the first atom-group boundary at or after every `chunk` atoms (>= 64), k_xtc_chunks decodes all chunks of all frames in
 * parallel (one thread per chunk).  scratch: vmd_hip_xtc_scratch_bytes(B, natoms, chunk) device bytes, 8-byte aligned. */
size_t vmd_hip_xtc_scratch_bytes(int B, int natoms, int chunk);
int vmd_hip_xtc_decode_chunked(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                               float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int chunk, void* scratch);
/* the same result with one WAVE per frame (k_xtc_wave): the wave walks the group boundaries of its stream speculatively (lane k
 * tests the flag bit of the k-th next group; the stream window lives in VGPRs, no LDS allocation) and decodes 64 groups at a time,
 * one per lane.  Streams of 2^27 bytes and more are reported as status 2. */
int vmd_hip_xtc_decode_wave(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                            float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status);
/* Checkpoints: the decoder state at a tile boundary of a frame's stream (bit position, atom index, smallidx | run << 8).  A first
 * pass (use = 0) decodes as vmd_hip_xtc_decode_wave does and writes up to VMD_XTC_CK_MAX of them per frame (ck[b][.], nck[b]); a
 * later pass over the SAME frames (use = 1) splits every frame into that many independent sections - no walk is repeated, a frame
 * occupies as many SIMDs as it has sections.  ck: device, B x VMD_XTC_CK_MAX records; nck: device, B counters. */
#define VMD_XTC_CK_MAX 64
typedef struct vmd_xtc_ck_t {
    uint32_t pos, atom, state, reserved;
} vmd_xtc_ck_t;
int vmd_hip_xtc_decode_wave_ck(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                               float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int use,
                               vmd_xtc_ck_t* ck, uint32_t* nck);
/* ... and with GROUP RECORDS next to the checkpoints: use = 0 also leaves one 16-bit record per decoded group (rec: B x rec_stride,
 * rec_stride >= natoms; nrec: B counters, 0 = no records for the frame): the bits the group spans (10), its smallidx step + 1 (2), the
 * atoms of its run (4).  use = 1 decodes from them (k_xtc_records): no walk, every tile of 64 groups is placed by three wave-wide
 * prefix sums and decoded independently; a record that does not describe the group found at its place fails the frame (status 1). */
int vmd_hip_xtc_decode_wave_rec(void* stream, const unsigned char* raw, const vmd_xtc_frame_t* info, int B, int natoms,
                                float* xyz, size_t frame_stride, size_t row_stride, uint32_t* status, int use,
                                vmd_xtc_ck_t* ck, uint32_t* nck, uint16_t* rec, uint32_t* nrec, size_t rec_stride);
/* waves that share one frame in k_xtc_wave (each walks the whole stream and decodes every n-th tile of 64 groups); 0 = automatic
 * (enough to put ~4 waves on every SIMD of the chip); returns the previous value */
int vmd_hip_set_xtc_waves(int n);

/* synthetic water box (oracle S9 twin): fills frames [frame0, frame0+B) of a batch laid out as above */
int vmd_hip_synth_frames(void* stream, float* xyz, size_t frame_stride, size_t row_stride, int B, uint32_t frame0,
                         uint64_t seed, uint32_t n_atoms, uint32_t n_blob, float L, float sigma);

#ifdef __cplusplus
}
#endif
#endif
