/*
 * oracle/vmd_oracle.h — CPU restatement of VIAMD/mdlib's per-frame RDF / SDF / distance evaluation.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/SPEC.md): the arithmetic of this path lives in
 * the git submodule ext/mdlib (/root/reference/.gitmodules:10-12) which is empty in this container, and the
 * reference ships no tests or golden vectors (/root/reference/TODO.md:7).  What is restated here follows
 *   - VIAMD's consumer code (layout + interpretation): /root/reference/src/main.cpp:139-250, :1353-1378,
 *     :1502-1529, :5774-5815; src/components/density_volume/density_volume.cpp:192-197,278-283;
 *     src/viamd.cpp:2240-2311 (alignment recipe);
 *   - the DECISION-tagged choices of oracle/SPEC.md for everything VIAMD does not pin.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this library.
 */
#ifndef VMD_ORACLE_H
#define VMD_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* unit cell as VIAMD reads it from md_unitcell_t (src/viamd.cpp:1837-1843) */
#define VO_CELL_PBC_X 1u
#define VO_CELL_PBC_Y 2u
#define VO_CELL_PBC_Z 4u
#define VO_CELL_PBC_ALL 7u
typedef struct vo_cell_t {
    float x, y, z;      /* edge lengths (orthorhombic) */
    float xy, xz, yz;   /* triclinic tilt factors: must be 0 (SPEC D-TRICLINIC) */
    uint32_t flags;     /* VO_CELL_PBC_* */
} vo_cell_t;

/* S2 */
float vo_wrap(float x, float L);

/* S3/S4: one frame, all ordered (ref,target) pairs, O(nref*ntgt).  counts += hits.  Returns #hits. */
uint64_t vo_rdf_frame_brute(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                            const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                            float rmin, float rmax, int nbins, uint64_t* counts);

/* same result through a uniform periodic cell grid (the md_spatial_hash stand-in).  Returns #hits, or
 * UINT64_MAX if the configuration cannot use the grid (caller falls back to brute). */
uint64_t vo_rdf_frame_cells(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                            const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                            float rmin, float rmax, int nbins, uint64_t* counts);

/* S4 normalisation of one frame, fp64, weights[b] += ... */
void vo_rdf_weights_frame(const vo_cell_t* cell, size_t nref, size_t ntgt, float rmin, float rmax, int nbins,
                          double* weights);

/* Multi-threaded driver mimicking VIAMD's "Eval Full" pool task (src/main.cpp:993-997,
 * src/task_system.cpp:73-81): frames handed out dynamically with grain 1 to nthreads workers, private
 * per-frame histograms merged into the shared accumulators at frame end.
 * traj layout: float[F][3][npad] (x row, y row, z row per frame), cells[F].  Returns total hits. */
uint64_t vo_rdf_run(const float* traj, const vo_cell_t* cells, size_t nframes, size_t npad,
                    const int32_t* ref_idx, size_t nref, const int32_t* tgt_idx, size_t ntgt,
                    float rmin, float rmax, int nbins, int nthreads, int use_cells,
                    uint64_t* counts, double* weights);

/* S5: alignment.  ref_pose: double[m*3] COM-centred reference pose, built by vo_sdf_ref_pose from frame 0.
 * M_out: double[K][12] row-major 3x4 world->reference matrices [R | -R*com] (optional).
 * R32/c32 out (optional): float[K][9], float[K][3] exactly as consumed by the scatter. */
void vo_sdf_ref_pose(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                     const int32_t* idx, const float* mass, size_t m, double* ref_pose);
void vo_sdf_frame_align(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                        const int32_t* struct_idx, const float* struct_mass, size_t K, size_t m,
                        const double* ref_pose, double* M_out, float* R32_out, float* c32_out);
/* the Jacobi eigen-solver used by the alignment (exposed for tests) */
void vo_jacobi4(double A[4][4], double V[4][4]);
/* S5 scatter of one frame for all K structures; vol is u64[dim^3], x fastest.  Returns #voxel hits. */
uint64_t vo_sdf_frame_scatter(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                              const int32_t* struct_idx, size_t K, size_t m,
                              const float* R32, const float* c32,
                              const int32_t* tgt_idx, size_t ntgt, float s, int dim, uint64_t* vol);

/* multi-threaded SDF driver over a trajectory float[F][3][npad] (threading as vo_rdf_run); returns #voxel hits */
uint64_t vo_sdf_run(const float* traj, const vo_cell_t* cells, size_t nframes, size_t npad,
                    const int32_t* struct_idx, const float* struct_mass, size_t K, size_t m,
                    const int32_t* tgt_idx, size_t ntgt, float s, int dim, int nthreads, uint64_t* vol);

/* S6 distance family, one frame */
void  vo_set_com(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                 const int32_t* idx, const float* mass, size_t n, float out[3]);
float vo_distance_com(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, const float* ma, size_t na, const int32_t* b, const float* mb, size_t nb);
float vo_distance_min(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, size_t na, const int32_t* b, size_t nb);
float vo_distance_max(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                      const int32_t* a, size_t na, const int32_t* b, size_t nb);
void  vo_distance_pair(const float* x, const float* y, const float* z, const vo_cell_t* cell,
                       const int32_t* a, size_t na, const int32_t* b, size_t nb, float* out);

/* D-SDF-UNWRAP as a switch: structures are made whole along a bond tree instead of along the index order.  vo_bond_tree builds the
 * tree of one structure (breadth-first from its first atom, see vmd_oracle.c); vo_set_unwrap_tree installs the tables ([K][m] each,
 * caller-owned, NULL removes them) for vo_sdf_ref_pose / vo_sdf_frame_align / vo_sdf_run. */
void vo_bond_tree(const int32_t* bonds, size_t nbonds, const int32_t* idx, size_t m, int32_t* order, int32_t* parent);
void vo_set_unwrap_tree(const int32_t* order, const int32_t* parent, size_t K, size_t m);

/* DECISION switches of SPEC.md ("rdf_closed", "sdf_include_self"): returns the previous value, -1 for an unknown key.  The other
 * two switches need no oracle code: "dist_geometric_com" = call with unit masses, "sdf_density" = the documented scaling of the
 * float view (SPEC S5). */
int vo_set_spec(const char* key, int value);

/* S8: restated from /root/reference/src/main.cpp:232-250, :139-170, :172-230 */
void vo_downsample_histogram(float* dst_bins, int num_dst_bins, const float* src_bins, const float* src_weights,
                             int num_src_bins);
void vo_compute_histogram(float* bins, int num_bins, float range_min, float range_max, const float* values,
                          int num_values, float* bin_val_min, float* bin_val_max);
/* mask: one byte per frame (non-zero = frame present) instead of md_bitfield_t */
void vo_compute_histogram_masked(float* bins, int num_bins, float range_min, float range_max, const float* values,
                                 int dim, const uint8_t* frame_mask, int num_frames, int aggregate);

/* S9 synthetic water-box trajectory (SURVEY 8d, C2/C3 family): atoms [n_blob, n_atoms) are O,H,H waters */
typedef struct vo_synth_t {
    uint64_t seed;
    uint32_t n_atoms;
    uint32_t n_blob;    /* leading atoms not generated here */
    float    L;         /* cubic box edge */
    float    sigma;     /* displacement scale: sigma_f = (float)(sigma*sqrt(f)) */
} vo_synth_t;
float vo_synth_uniform(uint64_t seed, uint32_t stream, uint32_t frame, uint32_t atom);
void  vo_synth_frame(const vo_synth_t* cfg, uint32_t frame, float* x, float* y, float* z);

#ifdef __cplusplus
}
#endif
#endif
