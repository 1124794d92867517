"""ctypes binding of the C ABI declared in include/vmd_eval.h and include/vmd_hip.h.

The product library is viamd_amd/libviamd_amd.so, built by hipcc for gfx950 (see viamd_amd/build.py).
There is no CPU fallback: if the library is missing, `default_lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIAMD_AMD_LIB selects another build of the same library (A/B builds, out-of-tree installs); never a CPU substitute.
LIB_PATH = os.environ.get("VIAMD_AMD_LIB") or os.path.join(_HERE, "libviamd_amd.so")

PBC_ALL = 7
FLAG_TEMPORAL, FLAG_DISTRIBUTION, FLAG_VOLUME = 1, 2, 4
DIST_COM, DIST_MIN, DIST_MAX, DIST_PAIR = 0, 1, 2, 3
RDF_NUM_BINS = 1024
VOLUME_DIM = 128

c_float_p = C.POINTER(C.c_float)
LOG_FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)      # vmd_log_fn
SETTLED_FN = C.CFUNCTYPE(None, C.c_void_p)                        # vmd_eval_set_settled_callback
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_uint64_p = C.POINTER(C.c_uint64)
c_double_p = C.POINTER(C.c_double)


class Unitcell(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float),
                ("xy", C.c_float), ("xz", C.c_float), ("yz", C.c_float),
                ("flags", C.c_uint32)]


class System(C.Structure):
    _fields_ = [("atom_count", C.c_size_t), ("x", c_float_p), ("y", c_float_p), ("z", c_float_p),
                ("mass", c_float_p), ("unitcell", Unitcell), ("bonds", C.POINTER(C.c_int32)), ("bond_count", C.c_size_t)]


class FrameHeader(C.Structure):
    _fields_ = [("num_atoms", C.c_size_t), ("index", C.c_int64), ("timestamp", C.c_double), ("unitcell", Unitcell)]


class DeviceView(C.Structure):
    _fields_ = [("base", c_float_p), ("frame_stride", C.c_size_t), ("row_stride", C.c_size_t),
                ("cells", C.POINTER(Unitcell)), ("device", C.c_int), ("resident_beg", C.c_size_t), ("resident_end", C.c_size_t), ("cells_version", C.c_uint64)]


NUM_FRAMES_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p)
NUM_ATOMS_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p)
LOAD_FRAME_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_int64, C.POINTER(FrameHeader), c_float_p, c_float_p, c_float_p)
DEVICE_VIEW_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.POINTER(DeviceView))


class HostView(C.Structure):
    _fields_ = [("base", c_float_p), ("frame_stride", C.c_size_t), ("row_stride", C.c_size_t), ("cells", C.POINTER(Unitcell))]


HOST_VIEW_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.POINTER(HostView))


class ScriptSkippedC(C.Structure):
    _fields_ = [("names", C.c_char_p), ("beg", C.c_size_t), ("end", C.c_size_t), ("reason", C.c_char_p)]


class TopologyC(C.Structure):
    _fields_ = [("num_atoms", C.c_size_t), ("elements", C.POINTER(C.c_char_p)), ("names", C.POINTER(C.c_char_p)),
                ("resnames", C.POINTER(C.c_char_p)), ("residue_index", c_int32_p), ("residue_seq_id", c_int32_p)]


class XtcFrame(C.Structure):             # vmd_xtc_frame_t (include/vmd_hip.h)
    _fields_ = [("precision", C.c_float), ("minint", C.c_int32 * 3), ("maxint", C.c_int32 * 3), ("smallidx", C.c_int32),
                ("offset", C.c_uint64), ("nbytes", C.c_uint64)]


class TrajectoryI(C.Structure):
    _fields_ = [("inst", C.c_void_p), ("num_frames", NUM_FRAMES_FN), ("num_atoms", NUM_ATOMS_FN),
                ("load_frame", LOAD_FRAME_FN), ("device_view", DEVICE_VIEW_FN), ("host_view", HOST_VIEW_FN),
                ("load_raw", C.c_void_p),          # native readers only (compressed frames for the device decoder); NULL here
                ("raw_device_view", C.c_void_p),   # vmd_rawtraj_* only (compressed frames resident in HBM); NULL here
                ("raw_mapped_view", C.c_void_p)]   # native XTC reader only (the file mapped for the copy engine); NULL here


class Aggregate(C.Structure):
    _fields_ = [("num_values", C.c_size_t), ("population_mean", c_float_p), ("population_var", c_float_p),
                ("population_ext", c_float_p)]


class PropertyData(C.Structure):
    _fields_ = [("dim", C.c_int32 * 4), ("values", c_float_p), ("weights", c_float_p), ("num_values", C.c_size_t),
                ("aggregate", C.POINTER(Aggregate)), ("min_value", C.c_float), ("max_value", C.c_float),
                ("min_range", C.c_float * 2), ("max_range", C.c_float * 2), ("fingerprint", C.c_uint64),
                ("counts", c_uint64_p), ("weights64", c_double_p), ("unit_str", C.c_char_p * 2)]


class AccumView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("flags", C.c_uint32), ("counts_dev", C.c_void_p), ("num_counts", C.c_size_t),
                ("weights64", c_double_p), ("num_weights", C.c_size_t), ("temporal", c_float_p), ("num_temporal", C.c_size_t),
                ("count_bound", C.c_uint64)]


ALLREDUCE_U64_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALLREDUCE_F64_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COMM_INT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


GROUP_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p)
ALLREDUCE_U32_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class CollectiveI(C.Structure):          # vmd_collective_i
    _fields_ = [("inst", C.c_void_p), ("rank", COMM_INT_FN), ("size", COMM_INT_FN),
                ("allreduce_sum_u64", ALLREDUCE_U64_FN), ("allreduce_sum_f64", ALLREDUCE_F64_FN),
                ("group_begin", GROUP_FN), ("group_end", GROUP_FN), ("allreduce_sum_u32", ALLREDUCE_U32_FN)]   # optional: may stay NULL


class ReduceStats(C.Structure):          # vmd_reduce_stats_t
    _fields_ = [("bytes", C.c_uint64), ("calls", C.c_uint32), ("grouped", C.c_uint32), ("volumes_as_u32", C.c_uint32), ("ms", C.c_double)]


class ReadAheadStats(C.Structure):      # vmd_readahead_stats_t
    _fields_ = [("engaged", C.c_uint32), ("block_frames", C.c_uint32), ("regions", C.c_uint64), ("region_frames", C.c_uint64),
                ("slow_calls", C.c_uint64), ("settles", C.c_uint64), ("direct_frames", C.c_uint64),
                ("committed_blocks", C.c_uint64)]


COMM_ID_BYTES = 128


class SdfPayload(C.Structure):          # vmd_sdf_payload_t
    _fields_ = [("num_structures", C.c_size_t), ("atoms_per_structure", C.c_size_t), ("structures", c_int32_p),
                ("matrices", c_float_p), ("extent", C.c_float)]


class Grid(C.Structure):
    _fields_ = [("nxf", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("ncell", C.c_int32)]


# every symbol include/vmd_eval.h and include/vmd_hip.h declare: (name, restype, argtypes)
_vp = C.c_void_p
SIGNATURES = [
    # vmd_eval.h
    ("vmd_ir_create", _vp, []),
    ("vmd_ir_free", None, [_vp]),
    ("vmd_ir_add_rdf", C.c_bool, [_vp, C.c_char_p, c_int32_p, C.c_size_t, c_int32_p, C.c_size_t, C.c_float, C.c_float]),
    ("vmd_ir_add_sdf", C.c_bool, [_vp, C.c_char_p, c_int32_p, C.c_size_t, C.c_size_t, c_int32_p, C.c_size_t, C.c_float]),
    ("vmd_ir_add_distance", C.c_bool, [_vp, C.c_char_p, C.c_int, c_int32_p, C.c_size_t, c_int32_p, C.c_size_t]),
    ("vmd_ir_add_distance_population", C.c_bool, [_vp, C.c_char_p, C.c_int, C.c_size_t, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ("vmd_ir_compile_from_source", C.c_bool, [_vp, C.c_char_p, C.POINTER(TopologyC)]),
    ("vmd_ir_compile_from_source_partial", C.c_bool, [_vp, C.c_char_p, C.POINTER(TopologyC), C.POINTER(_vp)]),
    ("vmd_script_report_skipped_count", C.c_size_t, [_vp]),
    ("vmd_script_report_skipped", C.POINTER(ScriptSkippedC), [_vp]),
    ("vmd_script_report_fallback_source", C.c_char_p, [_vp]),
    ("vmd_script_report_free", None, [_vp]),
    ("vmd_ir_valid", C.c_bool, [_vp]),
    ("vmd_ir_fingerprint", C.c_uint64, [_vp]),
    ("vmd_ir_property_count", C.c_size_t, [_vp]),
    ("vmd_ir_property_names", C.POINTER(C.c_char_p), [_vp]),
    ("vmd_ir_property_flags", C.c_uint32, [_vp, C.c_char_p]),
    ("vmd_ir_work_per_frame", C.c_uint64, [_vp]),
    ("vmd_eval_create", _vp, [C.c_size_t, _vp]),
    ("vmd_eval_free", None, [_vp]),
    ("vmd_eval_clear_data", None, [_vp]),
    ("vmd_eval_interrupt", None, [_vp]),
    ("vmd_eval_ir_fingerprint", C.c_uint64, [_vp]),
    ("vmd_eval_frame_range", C.c_bool, [_vp, _vp, C.POINTER(System), C.POINTER(TrajectoryI), C.c_uint32, C.c_uint32]),
    ("vmd_eval_property_data", C.POINTER(PropertyData), [_vp, C.c_char_p]),
    ("vmd_eval_frame_mask", c_uint8_p, [_vp]),
    ("vmd_eval_frame_mask_bits", C.c_size_t, [_vp, c_uint64_p, C.c_size_t]),
    ("vmd_eval_num_frames", C.c_size_t, [_vp]),
    ("vmd_eval_frames_done", C.c_size_t, [_vp]),
    ("vmd_eval_sdf_matrices", C.c_bool, [_vp, C.c_char_p, C.POINTER(System), C.POINTER(TrajectoryI), C.c_uint32, c_float_p,
                                         C.POINTER(C.c_size_t), c_float_p]),
    ("vmd_eval_accum_views", C.c_size_t, [_vp, C.POINTER(AccumView), C.c_size_t]),
    ("vmd_eval_refresh_counts", C.c_bool, [_vp, C.c_char_p]),
    ("vmd_eval_finalize", C.c_bool, [_vp]),
    ("vmd_eval_defer_volume_views", C.c_bool, [_vp, C.c_bool]),
    ("vmd_eval_wait_settled", C.c_bool, [_vp]),
    ("vmd_eval_frame_range_pooled", C.c_bool, [_vp, _vp, C.POINTER(System), C.POINTER(TrajectoryI), C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]),
    ("vmd_eval_set_deferred_settle", C.c_bool, [_vp, C.c_int]),
    ("vmd_eval_set_settled_callback", C.c_bool, [_vp, SETTLED_FN, C.c_void_p]),
    ("vmd_eval_set_frame_mask", None, [_vp, c_uint8_p, C.c_size_t]),
    ("vmd_eval_sdf_structures", c_int32_p, [_vp, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("vmd_eval_sdf_payload", C.c_bool, [_vp, C.c_char_p, C.POINTER(System), C.POINTER(TrajectoryI), C.c_uint32, C.POINTER(SdfPayload)]),
    ("vmd_export_xvg", C.c_bool, [C.c_char_p, C.POINTER(c_float_p), C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t]),
    ("vmd_export_csv", C.c_bool, [C.c_char_p, C.POINTER(c_float_p), C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t]),
    ("vmd_export_property_table", C.c_bool, [C.c_char_p, _vp, C.c_char_p, C.c_char_p, c_double_p, C.c_char_p, C.c_int]),
    ("vmd_export_cube", C.c_bool, [C.c_char_p, _vp, C.c_char_p, C.POINTER(System), C.POINTER(TrajectoryI), C.c_uint32, c_uint8_p]),
    ("vmd_eval_reduce", C.c_bool, [_vp, C.POINTER(CollectiveI), _vp]),
    ("vmd_eval_reduce_stats", None, [_vp, C.POINTER(ReduceStats)]),
    ("vmd_comm_unique_id", C.c_bool, [c_uint8_p]),
    ("vmd_comm_create", _vp, [C.c_int, C.c_int, c_uint8_p]),
    ("vmd_comm_from_nccl", _vp, [_vp]),
    ("vmd_comm_destroy", None, [_vp]),
    ("vmd_comm_collective", C.POINTER(CollectiveI), [_vp]),
    ("vmd_comm_rank", C.c_int, [_vp]),
    ("vmd_comm_size", C.c_int, [_vp]),
    ("vmd_eval_set_block_frames", C.c_bool, [_vp, C.c_size_t]),
    ("vmd_eval_set_source", C.c_bool, [_vp, _vp]),
    ("vmd_eval_frame_stats", None, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("vmd_eval_readahead_stats", None, [_vp, C.POINTER(ReadAheadStats)]),
    ("vmd_eval_cell_build_stats", None, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("vmd_eval_frames_device_decoded", C.c_size_t, [_vp]),
    ("vmd_eval_frames_section_decoded", C.c_size_t, [_vp]),
    ("vmd_eval_frames_mapped", C.c_size_t, [_vp]),
    ("vmd_pool_trim", None, []),
    ("vmd_pool_stats", None, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("vmd_ckcache_save", C.c_bool, [C.POINTER(TrajectoryI), C.c_char_p]),
    ("vmd_ckcache_load", C.c_long, [C.POINTER(TrajectoryI), C.c_char_p, C.c_int]),
    ("vmd_devtraj_create", _vp, [C.c_size_t, C.c_size_t]),
    ("vmd_devtraj_create_shard", _vp, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]),
    ("vmd_devtraj_free", None, [_vp]),
    ("vmd_devtraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_devtraj_upload_frame", C.c_bool, [_vp, C.c_size_t, C.POINTER(Unitcell), c_float_p, c_float_p, c_float_p]),
    ("vmd_devtraj_upload_atoms", C.c_bool, [_vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, c_float_p]),
    ("vmd_devtraj_set_cell", C.c_bool, [_vp, C.c_size_t, C.c_size_t, C.POINTER(Unitcell)]),
    ("vmd_devtraj_synth", C.c_bool, [_vp, C.c_uint64, C.c_float, C.c_float, C.c_uint32, C.c_size_t, C.c_size_t]),
    ("vmd_devtraj_device_ptr", _vp, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("vmd_rawtraj_create", _vp, [_vp]),
    ("vmd_rawtraj_free", None, [_vp]),
    ("vmd_rawtraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_rawtraj_device_bytes", C.c_size_t, [_vp]),
    ("vmd_hosttraj_create", _vp, [C.c_size_t, C.c_size_t]),
    ("vmd_hosttraj_free", None, [_vp]),
    ("vmd_hosttraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_hosttraj_frame_ptr", c_float_p, [_vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("vmd_dcdtraj_open", _vp, [C.c_char_p]),
    ("vmd_dcdtraj_close", None, [_vp]),
    ("vmd_dcdtraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_texttraj_open", _vp, [C.c_char_p, C.c_char_p]),
    ("vmd_texttraj_close", None, [_vp]),
    ("vmd_texttraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_textsys_open", _vp, [C.c_char_p]),
    ("vmd_textsys_close", None, [_vp]),
    ("vmd_textsys_topology", C.POINTER(TopologyC), [_vp]),
    ("vmd_textsys_mass", c_float_p, [_vp]),
    ("vmd_textsys_coords", c_float_p, [_vp, _vp]),
    ("vmd_xdrtraj_open", _vp, [C.c_char_p]),
    ("vmd_xdrtraj_close", None, [_vp]),
    ("vmd_xdrtraj_interface", C.POINTER(TrajectoryI), [_vp]),
    ("vmd_xdrtraj_kind", C.c_int, [_vp]),
    ("vmd_xdrtraj_frame_step", C.c_int64, [_vp, C.c_size_t]),
    ("vmd_xdrwriter_open", _vp, [C.c_char_p, C.c_int, C.c_size_t, C.c_float]),
    ("vmd_xdrwriter_write_frame", C.c_bool, [_vp, C.c_int64, C.c_float, C.POINTER(Unitcell), c_float_p, c_float_p, c_float_p]),
    ("vmd_xdrwriter_close", C.c_bool, [_vp]),
    ("vmd_hosttraj_set_cell", C.c_bool, [_vp, C.c_size_t, C.POINTER(Unitcell)]),
    ("vmd_hosttraj_copy_from_device", C.c_bool, [_vp, _vp, C.c_size_t, C.c_size_t]),
    ("vmd_downsample_histogram", None, [c_float_p, C.c_int, c_float_p, c_float_p, C.c_int]),
    ("vmd_compute_histogram_masked", None, [c_float_p, C.c_int, C.c_float, C.c_float, c_float_p, C.c_int, c_uint8_p, C.c_int, C.c_bool]),
    ("vmd_compute_histogram_masked_y", None, [c_float_p, C.c_int, C.c_float, C.c_float, c_float_p, C.c_int, c_uint8_p, C.c_int, C.c_bool, c_float_p]),
    ("vmd_compute_histogram", None, [c_float_p, C.c_int, C.c_float, C.c_float, c_float_p, C.c_int, c_float_p, c_float_p]),
    ("vmd_scale_histogram", None, [c_float_p, c_float_p, C.c_int]),
    ("vmd_device_count", C.c_int, []),
    ("vmd_set_device", C.c_bool, [C.c_int]),
    ("vmd_last_error", C.c_char_p, []),
    ("vmd_clear_last_error", None, []),
    ("vmd_last_stage", C.c_char_p, []),
    ("vmd_log_register", None, [LOG_FN, _vp]),
    ("vmd_log_message", None, [C.c_int, C.c_char_p]),
    ("vmd_version", C.c_char_p, []),
    ("vmd_set_option", C.c_int, [C.c_char_p, C.c_int]),
    ("vmd_profile_reset", None, []),
    ("vmd_profile_ms", C.c_double, [C.c_char_p, c_uint64_p]),
    ("vmd_profile_enable", None, [C.c_bool]),
    # vmd_hip.h
    ("vmd_hip_xtc_decode", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp]),
    ("vmd_hip_xtc_decode_wave", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp]),
    ("vmd_hip_set_xtc_waves", C.c_int, [C.c_int]),
    ("vmd_hip_xtc_decode_wave_ck", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_int, _vp, _vp]),
    ("vmd_hip_xtc_decode_wave_rec", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t]),
    ("vmd_hip_raw_f32_decode", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t]),
    ("vmd_hip_xtc_scratch_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("vmd_hip_xtc_decode_chunked", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_int, _vp]),
    ("vmd_hip_bbox", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _vp]),
    ("vmd_hip_cells_build", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, C.c_int, C.c_int, Grid, _vp, _vp, _vp, _vp, _vp]),
    ("vmd_hip_cells_fused_ok", C.c_int, [Grid, C.c_int]),
    ("vmd_hip_set_cells_fused", C.c_int, [C.c_int]),
    ("vmd_hip_rdf_num_blocks", C.c_int, []),
    ("vmd_hip_set_rdf_blocks", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_shared_hist", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_nsub_pct", C.c_int, [C.c_int]),
    ("vmd_hip_set_pencil_reach", None, [C.c_int, C.c_int]),
    ("vmd_hip_set_cells_rec3", C.c_int, [C.c_int]),
    ("vmd_hip_set_cells_bin_lds", C.c_int, [C.c_int]),
    ("vmd_hip_set_sdf_wave", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_nsplit", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_pop", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_nsub", C.c_int, [C.c_int]),
    ("vmd_hip_cells_split_blocks", C.c_int, [Grid, C.c_int]),
    ("vmd_hip_set_cells_split", C.c_int, [C.c_int]),
    ("vmd_hip_cells_scratch_words", C.c_size_t, [Grid, C.c_int]),
    ("vmd_hip_rdf_partial_words", C.c_size_t, []),
    ("vmd_hip_rdf_pencil", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, Grid,
                                     C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp, _vp, _vp]),
    ("vmd_hip_cells_pencil_ok", C.c_int, [Grid]),
    ("vmd_hip_set_cells_pencil", C.c_int, [C.c_int]),
    ("vmd_hip_cells_pencil_cap_max", C.c_int, []),
    ("vmd_hip_cells_pencil_count", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, C.c_int, Grid, _vp]),
    ("vmd_hip_cells_build_pencil", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, C.c_int, C.c_int, Grid,
                                             _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("vmd_hip_axpy_u64", C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint64, _vp]),
    ("vmd_hip_rdf_brute", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, C.c_int, _vp, C.c_int,
                                    C.c_float, C.c_float, C.c_int, _vp]),
    ("vmd_hip_sdf_align", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("vmd_hip_sdf_ref_pose", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    ("vmd_hip_sdf_scatter", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp,
                                      _vp, _vp, C.c_int, C.c_float, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    ("vmd_hip_set_sdf_ilp", C.c_int, [C.c_int]),
    ("vmd_hip_set_sdf_rows", C.c_int, [C.c_int]),
    ("vmd_hip_distance", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("vmd_hip_add_u64", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("vmd_hip_counts_to_float", C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_float]),
    ("vmd_hip_bump_u64", C.c_int, [_vp, _vp, C.c_uint64]),
    ("vmd_hip_rdf_columns", C.c_uint64, [C.c_int]),
    ("vmd_hip_marker", C.c_int, [C.c_void_p]),
    ("vmd_hip_set_cells_sel_pattern", None, [C.c_int, C.c_int, C.c_int, c_int32_p]),
    ("vmd_hip_set_sdf_nt", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_closed", C.c_int, [C.c_int]),
    ("vmd_hip_set_rdf_raw", C.c_int, [C.c_int]),
    ("vmd_hip_set_cells_overflow_bit", C.c_uint32, [C.c_uint32]),
    ("vmd_hip_synth_frames", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_float, C.c_float]),
]


class VmdLib:
    """A loaded libviamd_amd.so with typed entry points (attribute access -> ctypes function)."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the MI355X backend is a hipcc-built shared library and there is no CPU fallback. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (or viamd_amd/build.py).")
        self.path = path
        self.cdll = C.CDLL(path)
        missing = []
        for name, restype, argtypes in SIGNATURES:
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        if missing:
            raise ImportError(f"{path} lacks symbols declared in include/*.h: {missing}")

    def last_error(self):
        msg = self.vmd_last_error()
        return msg.decode() if msg else ""


_default = None


def default_lib():
    global _default
    if _default is None:
        _default = VmdLib(LIB_PATH)
    return _default
