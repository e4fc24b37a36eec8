"""ctypes binding of libntjoin_mx.so (C-ABI: include/ntjoin_mx.h).  No compute happens in Python."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MXG_LIB_DIR: another build of the same sources (the sanitizer builds of csrc/Makefile: lib_asan, lib_tsan)
LIB_PATH = os.path.join(os.environ.get("MXG_LIB_DIR") or os.path.join(_HERE, "lib"), "libntjoin_mx.so")

MXG_OK, MXG_EINVAL, MXG_EIO, MXG_ENOMEM, MXG_EDEVICE, MXG_ELIMIT = 0, -1, -2, -3, -4, -5
VARIANT_V2_SUM, VARIANT_V1_MIN = 0, 1
FLAG_DENSE_ONLY, FLAG_DROP_SEQ, FLAG_TIMING, FLAG_TIMING_FINE = 0x1, 0x2, 0x4, 0x8
MX_UNIQUE, MX_SHARED, MX_INALL = 0x1, 0x2, 0x4
ABI_VERSION = 2

SYMBOLS = [
    "mxg_abi_version", "mxg_create", "mxg_destroy", "mxg_last_error",
    "mxg_add_assembly_fasta", "mxg_add_assembly_fasta_shard", "mxg_add_assembly_fasta_split", "mxg_assembly_continues", "mxg_xchg_pack", "mxg_xchg_unpack_graph", "mxg_sketch_pack", "mxg_sketch_pack_parts", "mxg_sketch_dg_pack_slots", "mxg_part_packed_wait", "mxg_xchg_unpack_graph_parts", "mxg_sketch_finish", "mxg_shard_range", "mxg_assembly_shard",
    "mxg_add_assembly_buffers", "mxg_add_assembly_packed_device",
    "mxg_add_assembly_tsv", "mxg_add_assembly_bin", "mxg_write_sketch_bin", "mxg_add_assembly_minimizers", "mxg_num_assemblies", "mxg_assembly_name",
    "mxg_record_id", "mxg_record_length", "mxg_num_records", "mxg_assembly_weight",
    "mxg_sketch", "mxg_sketch_graph", "mxg_get_sketch", "mxg_get_sketch_device", "mxg_compute_strands", "mxg_set_sketch_device",
    "mxg_pack_sketch_device", "mxg_set_sketch_gathered", "mxg_set_sketch_gathered_strided", "mxg_write_tsv",
    "mxg_build_graph", "mxg_get_mx_flags", "mxg_get_graph", "mxg_find_paths", "mxg_path_segments", "mxg_mx_extremes", "mxg_dg_owner_counts", "mxg_dg_pack_items", "mxg_dg_set_items", "mxg_dg_vertices", "mxg_dg_item_results", "mxg_dg_msg_counts", "mxg_dg_pack_msgs", "mxg_dg_edges", "mxg_dg_pack_slots", "mxg_dg_owner_slots", "mxg_dg_slot_results", "mxg_dg_pack_msg_slots", "mxg_dg_edges_slots", "mxg_write_dot", "mxg_write_outputs", "mxg_dot_part_format", "mxg_dot_part_write",
    "mxg_py_repr_double", "mxg_py_repr_str", "mxg_get_stats", "mxg_reset_timers", "mxg_knobs",
    "mxg_synth_fill_packed_device", "mxg_synth_fill_packed_host", "mxg_synth_write_fasta",
    "mxg_plan_split", "mxg_add_assembly_packed_device_pieces", "mxg_dg_last_shared", "mxg_dg_set_ghosts",
]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("k", C.c_uint32), ("w", C.c_uint32), ("variant", C.c_uint32),
                ("device", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p),
                ("cand_per_window", C.c_uint32), ("host_threads", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class SketchView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("out_hash", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint32)),
                ("record", C.POINTER(C.c_uint32)), ("forward", C.POINTER(C.c_uint8)),
                ("n_records", C.c_uint64), ("record_first", C.POINTER(C.c_uint64))]


class SketchDView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("out_hash", C.c_void_p), ("pos", C.c_void_p), ("record", C.c_void_p),
                ("forward", C.c_void_p)]


class GraphView(C.Structure):
    _fields_ = [("n_assemblies", C.c_uint32), ("n_vertices", C.c_uint64),
                ("vertex_hash", C.POINTER(C.c_uint64)), ("vertex_pos", C.POINTER(C.c_uint32)),
                ("vertex_record", C.POINTER(C.c_uint32)), ("n_edges", C.c_uint64),
                ("edge_u", C.POINTER(C.c_uint32)), ("edge_v", C.POINTER(C.c_uint32)),
                ("edge_support", C.POINTER(C.c_uint32)), ("edge_weight", C.POINTER(C.c_double))]


class PathsView(C.Structure):
    _fields_ = [("n_components", C.c_uint64), ("n_paths", C.c_uint64), ("path_first", C.POINTER(C.c_uint64)), ("path_vertex", C.POINTER(C.c_uint32)),
                ("path_component", C.POINTER(C.c_uint32))]


class SegmentsView(C.Structure):
    _fields_ = [("n_segments", C.c_uint64), ("seg_path", C.POINTER(C.c_uint32)), ("seg_record", C.POINTER(C.c_uint32)),
                ("seg_first", C.POINTER(C.c_uint32)), ("seg_stat", C.POINTER(C.c_uint32))]


class SynthSeg(C.Structure):
    _fields_ = [("dst_base", C.c_uint64), ("src", C.c_uint64), ("len", C.c_uint64), ("rc", C.c_uint32),
                ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_assemblies", C.c_uint32), ("bases", C.c_uint64),
                ("kmers", C.c_uint64), ("minimizers", C.c_uint64), ("candidates", C.c_uint64),
                ("dense_kmers", C.c_uint64), ("unique", C.c_uint64), ("vertices", C.c_uint64),
                ("edges", C.c_uint64), ("ms_hash", C.c_double), ("ms_resolve", C.c_double),
                ("ms_graph", C.c_double), ("launches_hash", C.c_uint64), ("hash_kernel_bases", C.c_uint64),
                ("ms_reorder", C.c_double), ("ms_resolve_kernel", C.c_double), ("ms_emit", C.c_double),
                ("ms_join", C.c_double), ("ms_vertices", C.c_double), ("ms_edges", C.c_double),
                ("bs_filter_bases", C.c_uint64), ("graph_join", C.c_uint64),
                ("batches_redone", C.c_uint64), ("sync_assemblies", C.c_uint64), ("retried_assemblies", C.c_uint64), ("deferred_stretches", C.c_uint64),
                ("select_slices", C.c_uint64), ("slice_stretches", C.c_uint64)]


_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64; if this library were
    loaded first it would bind /opt/rocm's copy and a later `import torch` would initialise a second runtime that
    sees no GPU.  When torch is installed, map its bundled runtime first (without importing torch): the loader
    then resolves our DT_NEEDED libamdhip64.so.N to the copy already in the process, whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """Load libntjoin_mx.so.  There is no fallback: a missing/unloadable library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ntjoin_amd/csrc`). ntjoin_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, cp, u64, i32 = C.c_void_p, C.c_char_p, C.c_uint64, C.c_int
    L.mxg_abi_version.restype = i32
    L.mxg_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.mxg_destroy.argtypes = [vp]
    L.mxg_destroy.restype = None
    L.mxg_last_error.argtypes = [vp]
    L.mxg_last_error.restype = cp
    L.mxg_add_assembly_fasta.argtypes = [vp, cp, C.c_double, cp]
    L.mxg_add_assembly_fasta_shard.argtypes = [vp, cp, C.c_double, cp, C.c_uint32, C.c_uint32]
    L.mxg_add_assembly_fasta_split.argtypes = [vp, cp, C.c_double, cp, C.c_uint32, C.c_uint32]
    L.mxg_assembly_continues.argtypes = [vp, i32]
    L.mxg_xchg_pack.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.mxg_sketch_pack.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.mxg_sketch_finish.argtypes = [vp]
    L.mxg_xchg_unpack_graph.argtypes = [vp, vp, C.c_uint32, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.mxg_sketch_pack_parts.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.mxg_part_packed_wait.argtypes = [vp, C.c_int, vp]
    L.mxg_sketch_dg_pack_slots.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp]
    L.mxg_xchg_unpack_graph_parts.argtypes = [vp, C.POINTER(vp), C.c_uint32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mxg_shard_range.argtypes = [C.POINTER(u64), u64, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
    L.mxg_assembly_shard.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64)]
    L.mxg_add_assembly_buffers.argtypes = [vp, cp, C.c_double, vp, C.POINTER(u64), C.POINTER(cp), u64]
    L.mxg_add_assembly_packed_device.argtypes = [vp, cp, C.c_double, vp, C.POINTER(u64), C.POINTER(u64),
                                                 C.POINTER(cp), u64]
    L.mxg_plan_split.argtypes = [vp, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.mxg_add_assembly_packed_device_pieces.argtypes = [vp, cp, C.c_double, vp, vp, vp, vp, vp, vp, C.POINTER(cp), u64]
    L.mxg_add_assembly_tsv.argtypes = [vp, cp, C.c_double, cp]
    L.mxg_add_assembly_minimizers.argtypes = [vp, cp, C.c_double, vp, vp, vp, u64, C.POINTER(cp), u64]
    L.mxg_num_assemblies.argtypes = [vp]
    L.mxg_assembly_name.argtypes = [vp, i32]
    L.mxg_assembly_name.restype = cp
    L.mxg_record_id.argtypes = [vp, i32, u64]
    L.mxg_record_id.restype = cp
    L.mxg_record_length.argtypes = [vp, i32, u64]
    L.mxg_record_length.restype = u64
    L.mxg_num_records.argtypes = [vp, i32]
    L.mxg_num_records.restype = u64
    L.mxg_assembly_weight.argtypes = [vp, i32]
    L.mxg_assembly_weight.restype = C.c_double
    L.mxg_sketch.argtypes = [vp, i32]
    L.mxg_sketch_graph.argtypes = [vp]
    L.mxg_get_sketch.argtypes = [vp, i32, C.POINTER(SketchView)]
    L.mxg_get_sketch_device.argtypes = [vp, i32, C.POINTER(SketchDView)]
    L.mxg_set_sketch_device.argtypes = [vp, i32, vp, vp, vp, vp, u64]
    L.mxg_compute_strands.argtypes = [vp, i32]
    L.mxg_pack_sketch_device.argtypes = [vp, i32, vp, u64]
    L.mxg_set_sketch_gathered.argtypes = [vp, i32, vp, C.c_uint32, u64, C.POINTER(u64), C.POINTER(u64)]
    L.mxg_set_sketch_gathered_strided.argtypes = [vp, i32, vp, C.c_uint32, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.mxg_write_tsv.argtypes = [vp, i32, cp, i32, i32, i32]
    L.mxg_build_graph.argtypes = [vp]
    L.mxg_get_mx_flags.argtypes = [vp, i32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(u64)]
    L.mxg_get_graph.argtypes = [vp, C.POINTER(GraphView)]
    L.mxg_add_assembly_bin.argtypes = [vp, cp, C.c_double, cp]
    L.mxg_write_sketch_bin.argtypes = [vp, i32, cp]
    L.mxg_find_paths.argtypes = [vp, C.c_int64, C.POINTER(PathsView)]
    L.mxg_path_segments.argtypes = [vp, i32, C.POINTER(SegmentsView)]
    L.mxg_mx_extremes.argtypes = [vp, i32, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)),
                                  C.POINTER(C.c_uint64)]
    pu64 = C.POINTER(u64)
    L.mxg_dg_owner_counts.argtypes = [vp, C.c_uint32, pu64]
    L.mxg_dg_pack_items.argtypes = [vp, i32, C.c_uint32, C.c_uint32, pu64, vp]
    L.mxg_dg_set_items.argtypes = [vp, i32, vp, C.c_uint32, pu64, pu64]
    L.mxg_dg_vertices.argtypes = [vp, vp]
    L.mxg_dg_item_results.argtypes = [vp, i32, vp, C.c_uint32, pu64, pu64, vp]
    L.mxg_dg_msg_counts.argtypes = [vp, C.c_uint32, vp, vp, pu64]
    L.mxg_dg_pack_msgs.argtypes = [vp, i32, C.c_uint32, vp, pu64, vp]
    L.mxg_dg_edges.argtypes = [vp, vp, u64, pu64, pu64]
    L.mxg_dg_last_shared.argtypes = [vp, vp, vp]
    L.mxg_dg_set_ghosts.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    pu32 = C.POINTER(C.c_uint32)
    L.mxg_dg_pack_slots.argtypes = [vp, i32, C.c_uint32, C.c_uint32, C.c_uint32, pu32, vp]
    L.mxg_dg_owner_slots.argtypes = [vp, C.c_uint32, C.c_uint32, pu32, vp, vp]
    L.mxg_dg_slot_results.argtypes = [vp, C.c_uint32, C.c_uint32, pu32, vp, vp, vp]
    L.mxg_dg_pack_msg_slots.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.mxg_dg_edges_slots.argtypes = [vp, vp, C.c_uint32, C.c_uint32, pu64, pu64, pu32]
    L.mxg_write_dot.argtypes = [vp, cp]
    L.mxg_write_outputs.argtypes = [vp, cp, C.POINTER(cp), i32, i32, i32]
    L.mxg_dot_part_format.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.mxg_dot_part_write.argtypes = [vp, cp, u64, u64, i32, i32]
    L.mxg_py_repr_double.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    L.mxg_py_repr_double.restype = C.c_size_t
    L.mxg_py_repr_str.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.mxg_py_repr_str.restype = C.c_size_t
    L.mxg_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mxg_reset_timers.argtypes = [vp]
    L.mxg_knobs.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.mxg_knobs.restype = C.c_size_t
    L.mxg_synth_fill_packed_device.argtypes = [vp, u64, vp, u64, u64, u64, C.c_uint32, i32]
    L.mxg_synth_fill_packed_host.argtypes = [vp, u64, vp, u64, u64, u64, C.c_uint32, C.c_uint32]
    L.mxg_synth_write_fasta.argtypes = [cp, vp, vp, vp, u64, cp, C.c_uint32, C.c_uint32]
    for name in SYMBOLS:
        fn = getattr(L, name)  # raises AttributeError if the symbol is not exported
        if fn.restype is C.c_int and name not in ("mxg_abi_version",):
            fn.restype = i32
    if L.mxg_abi_version() != ABI_VERSION:
        raise RuntimeError("libntjoin_mx.so ABI version mismatch")
    _lib = L
    return L
