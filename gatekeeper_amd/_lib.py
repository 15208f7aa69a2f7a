"""ctypes binding of the C ABI declared in include/gkgpu.h.

The product library is gatekeeper_amd/libgkgpu.so (host engine + HIP kernels for gfx950), built in-tree by
`__graft_entry__.build()` / `make -C gatekeeper_amd/csrc`.  There is NO CPU fallback: if the library is missing, or
no MI355X is visible, loading / gk_engine_create fail loudly.

tests/ pass hostemu=True explicitly to load tests/native/libgkgpu_hostemu.so instead -- a test-only build in which the
kernels' per-row / per-review code (vm_core.hpp) is executed lane by lane on the CPU, so that the AOT compiler and
the flattener can be checked against the oracle in the GPU-less build container.  No environment variable selects it.
"""
from __future__ import annotations

import atexit
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

GK_OK = 0
GK_ERR_INVALID, GK_ERR_REGO, GK_ERR_UNSUPPORTED, GK_ERR_NOT_FOUND, GK_ERR_DEVICE, GK_ERR_REVIEW, GK_ERR_INTERNAL, GK_ERR_LIMIT = (
    -1, -2, -3, -4, -5, -6, -7, -8)
GK_REVIEW_ADMISSION_REQUEST, GK_REVIEW_OBJECT = 0, 1
GK_SRC_EMPTY, GK_SRC_ORIGINAL, GK_SRC_GENERATED, GK_SRC_ALL, GK_SRC_INVALID = 0, 1, 2, 3, 4
GK_TABLE_KEEP_DOCS = 1
GK_TABLE_RESIDENT = 2
GK_TABLE_KEEP_TEXT = 16
GK_TABLE_PRUNED = 32
GK_TABLE_PRE_MATCHED = 64
GK_TABLE_PROCESS_AUDIT = 4
GK_TABLE_PROCESS_WEBHOOK = 8
GK_REVIEW_EXCLUDED = 1
GK_EVAL_WANT_MATCH, GK_EVAL_NO_DOWNLOAD, GK_EVAL_WANT_LIST, GK_EVAL_ASYNC, GK_EVAL_COLLECT, GK_EVAL_TIME_EACH, GK_EVAL_KERNEL_ONLY = 1, 2, 4, 8, 16, 32, 64
GK_EVAL_DEVICE_ONLY = 128
GK_SWEEP_RESULT_TOTALS = 1

EXPORTS = [
    "gk_engine_create", "gk_engine_destroy", "gk_last_error", "gk_version", "gk_template_add", "gk_template_remove",
    "gk_constraint_add", "gk_constraint_remove", "gk_data_put", "gk_data_remove", "gk_excluder_replace", "gk_excluder_excluded", "gk_table_create", "gk_table_free",
    "gk_table_eval", "gk_eval_free", "gk_render", "gk_render_error", "gk_free", "gk_dump", "gk_table_topk", "gk_topk_free",
    "gk_table_totals", "gk_totals_free", "gk_table_get_stats", "gk_batcher_start", "gk_batcher_stop", "gk_query", "gk_query_ex", "gk_query_ex2",
    "gk_resident_sweep", "gk_sweep_free", "gk_resident_review", "gk_resident_review_ex",
    "gk_comm_unique_id", "gk_comm_init", "gk_comm_destroy", "gk_comm_info", "gk_table_sweep_sharded", "gk_shard_free",
    "gk_jit_quiesce", "gk_jit_cache_stats", "gk_jit_cache_dir", "gk_jit_cache_drop_memory", "gk_host_cpus", "gk_table_create_spool", "gk_spool_info_free", "gk_debug_set",
    # include/gksynth.h (bench / test plumbing)
    "gk_synth_batch_create", "gk_synth_batch_reviews", "gk_synth_batch_size", "gk_synth_batch_json_bytes", "gk_synth_batch_free", "gk_synth_query_storm", "gk_synth_query_storm_ex",
]


GK_OPT_NO_REFERENTIAL, GK_OPT_GATHER_STATS, GK_OPT_TRACE = 1, 2, 4
GK_QUERY_TRACE = 1
GK_QUERY_PRE_MATCHED = 2


class gk_opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("elem_cap", C.c_uint16 * 3), ("reserved", C.c_uint16), ("flags", C.c_uint32), ("reserved2", C.c_uint32),
                ("disabled_builtins", C.POINTER(C.c_char_p)), ("n_disabled_builtins", C.c_size_t)]


class gk_review_in(C.Structure):
    _fields_ = [("kind", C.c_int32), ("source", C.c_int32),
                ("json", C.c_char_p), ("json_len", C.c_size_t),
                ("namespace_json", C.c_char_p), ("namespace_len", C.c_size_t),
                ("ns_object_json", C.c_char_p), ("ns_object_len", C.c_size_t),
                ("operation", C.c_char_p)]


class gk_eval_out(C.Structure):
    _fields_ = [("n_reviews", C.c_uint32), ("n_constraints", C.c_uint32), ("n_tiles", C.c_uint32),
                ("constraint_ids", C.POINTER(C.c_uint32)),
                ("viol", C.POINTER(C.c_uint64)), ("err", C.POINTER(C.c_uint64)), ("match", C.POINTER(C.c_uint64)),
                ("too_big", C.POINTER(C.c_uint64)), ("counts", C.POINTER(C.c_uint32)), ("list", C.POINTER(C.c_uint32)),
                ("list_len", C.c_uint32), ("list_total", C.c_uint32), ("n_overflow", C.c_uint32),
                ("kernel_ms", C.c_float), ("fast_kernel_ms", C.c_float),
                ("algo_bytes", C.c_uint64), ("n_rows", C.c_uint64), ("n_launches", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("d_viol", C.c_void_p), ("d_err", C.c_void_p), ("d_counts", C.c_void_p), ("n_rows_read", C.c_uint64),
                ("algo_bytes_once", C.c_uint64), ("n_plan_groups", C.c_uint32), ("n_host_evaluated", C.c_uint32),
                ("host_evaluated", C.POINTER(C.c_uint32)), ("kernel_text_hash", C.c_uint64)]


class gk_topk_out(C.Structure):
    _fields_ = [("n_constraints", C.c_uint32), ("stride", C.c_uint32), ("constraint_ids", C.POINTER(C.c_uint32)),
                ("counts", C.POINTER(C.c_uint32)), ("reviews", C.POINTER(C.c_uint32)), ("overflow", C.POINTER(C.c_uint32))]


class gk_totals_out(C.Structure):
    _fields_ = [("n_constraints", C.c_uint32), ("constraint_ids", C.POINTER(C.c_uint32)), ("results", C.POINTER(C.c_uint64)),
                ("pairs", C.POINTER(C.c_uint64)), ("rendered_pairs", C.c_uint64)]


class gk_table_stats(C.Structure):
    _fields_ = [("n_reviews", C.c_uint64), ("n_rows", C.c_uint64), ("json_bytes", C.c_uint64), ("heap_bytes", C.c_uint64),
                ("device_bytes", C.c_uint64), ("flatten_s", C.c_double), ("upload_s", C.c_double), ("host_threads", C.c_uint32),
                ("reserved", C.c_uint32), ("fast_reviews", C.c_uint64), ("digest", C.c_uint64)]


class gk_sweep_out(C.Structure):
    _fields_ = [("n_objects", C.c_uint64), ("n_constraints", C.c_uint32), ("n_chunks", C.c_uint32), ("constraint_ids", C.POINTER(C.c_uint32)),
                ("pairs", C.POINTER(C.c_uint64)), ("results", C.POINTER(C.c_uint64)), ("flattened", C.c_uint64), ("beyond_limits", C.c_uint64),
                ("sync_s", C.c_double), ("eval_s", C.c_double)]


class gk_shard_out(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("n_constraints", C.c_uint32), ("stride_tiles", C.c_uint32), ("slot_bytes", C.c_uint64),
                ("constraint_ids", C.POINTER(C.c_uint32)), ("shard_reviews", C.POINTER(C.c_uint32)), ("totals", C.POINTER(C.c_int64)),
                ("gathered", C.POINTER(C.c_uint64)), ("d_gathered", C.c_void_p), ("kernel_ms", C.c_float), ("fast_kernel_ms", C.c_float),
                ("n_overflow", C.c_uint32), ("err_totals", C.POINTER(C.c_int64)), ("beyond_limits", C.c_int64), ("not_evaluated", C.c_int64),
                ("exchange_ms", C.c_float), ("exchange_overlapped", C.c_uint32), ("exchange_bytes_inbound", C.c_uint64)]


GK_SHARD_DOWNLOAD = 1
GK_SHARD_ENQUEUE = 2
GK_SHARD_COLLECT = 4
GK_COMM_ID_BYTES = 128
HE_ALLGATHER = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64)     # test-only (libgkgpu_hostemu.so gk_comm_init_host)
HE_ALLREDUCE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_longlong), C.c_uint64)


class gk_spool_info(C.Structure):
    _fields_ = [("n_files", C.c_uint64), ("n_reviews", C.c_uint64), ("n_unreadable", C.c_uint64), ("n_namespace_missing", C.c_uint64), ("bytes", C.c_uint64),
                ("names", C.POINTER(C.c_char_p)), ("n_folders_missing", C.c_uint64), ("statuses", C.POINTER(C.c_int32)), ("n_rejected", C.c_uint64),
                ("n_excluded", C.c_uint64)]


class gk_batch_opts(C.Structure):
    _fields_ = [("max_batch", C.c_uint32), ("window_us", C.c_uint32), ("workers", C.c_uint32), ("reserved", C.c_uint32)]


class gk_storm_out(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("errors", C.c_uint64), ("seconds", C.c_double), ("p50_us", C.c_double), ("p90_us", C.c_double),
                ("p99_us", C.c_double), ("max_us", C.c_double), ("mean_batch", C.c_double), ("mean_queue_us", C.c_double),
                ("mean_device_us", C.c_double), ("results", C.c_uint64)]


class gk_query_stats(C.Structure):
    _fields_ = [("batch_size", C.c_uint32), ("reserved", C.c_uint32), ("queue_us", C.c_double), ("device_us", C.c_double),
                ("total_us", C.c_double)]


class EngineLoadError(RuntimeError):
    pass


def library_path(hostemu: bool = False) -> str:
    if hostemu:
        return os.path.join(_ROOT, "tests", "native", "libgkgpu_hostemu.so")
    return os.path.join(_HERE, "libgkgpu.so")


_cache = {}


def load(hostemu: bool | None = None):
    """Load the C-ABI library (hostemu=True: the test-only emulation build, only ever passed by tests/)."""
    hostemu = bool(hostemu)
    if hostemu in _cache:
        return _cache[hostemu]
    path = library_path(hostemu)
    if not os.path.exists(path):
        raise EngineLoadError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the constraint-evaluation path)" % path)
    if not hostemu:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if it is going to be used in this
        # process (device memory / RCCL plumbing) it must be loaded BEFORE libgkgpu.so resolves the same soname.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    vp, cp, sz, u32 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32
    lib.gk_engine_create.argtypes = [C.POINTER(gk_opts), C.POINTER(vp)]
    lib.gk_engine_destroy.argtypes = [vp]
    lib.gk_engine_destroy.restype = None
    lib.gk_last_error.restype = cp
    lib.gk_version.restype = cp
    lib.gk_host_cpus.argtypes = []
    lib.gk_host_cpus.restype = C.c_uint32
    lib.gk_template_add.argtypes = [vp, cp, cp, C.POINTER(cp), sz]
    lib.gk_template_remove.argtypes = [vp, cp]
    lib.gk_constraint_add.argtypes = [vp, cp, sz, C.POINTER(u32)]
    lib.gk_constraint_remove.argtypes = [vp, cp, cp]
    lib.gk_data_put.argtypes = [vp, C.POINTER(cp), sz, cp, sz]
    lib.gk_data_remove.argtypes = [vp, C.POINTER(cp), sz]
    lib.gk_excluder_replace.argtypes = [vp, C.c_char_p, sz]
    lib.gk_excluder_excluded.argtypes = [vp, C.c_char_p, C.POINTER(gk_review_in), C.POINTER(C.c_int32)]
    lib.gk_table_create.argtypes = [vp, C.POINTER(gk_review_in), sz, u32, C.POINTER(C.c_int32), C.POINTER(vp)]
    lib.gk_table_free.argtypes = [vp]
    lib.gk_table_free.restype = None
    lib.gk_table_create_spool.argtypes = [vp, cp, cp, u32, u32, C.POINTER(C.POINTER(gk_spool_info)), C.POINTER(vp)]
    lib.gk_spool_info_free.argtypes = [C.POINTER(gk_spool_info)]
    lib.gk_spool_info_free.restype = None
    lib.gk_jit_quiesce.argtypes = []
    lib.gk_jit_quiesce.restype = None
    lib.gk_jit_cache_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gk_jit_cache_stats.restype = None
    lib.gk_jit_cache_dir.argtypes = []
    lib.gk_jit_cache_dir.restype = C.c_char_p
    lib.gk_jit_cache_drop_memory.argtypes = []
    lib.gk_jit_cache_drop_memory.restype = None
    lib.gk_table_eval.argtypes = [vp, vp, u32, C.POINTER(C.POINTER(gk_eval_out))]
    lib.gk_eval_free.argtypes = [C.POINTER(gk_eval_out)]
    lib.gk_eval_free.restype = None
    lib.gk_render.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
    lib.gk_render_error.argtypes = [vp, vp, u32, u32, C.POINTER(vp)]
    lib.gk_free.argtypes = [vp]
    lib.gk_free.restype = None
    lib.gk_dump.argtypes = [vp, C.POINTER(vp)]
    lib.gk_table_topk.argtypes = [vp, vp, u32, C.POINTER(C.POINTER(gk_topk_out))]
    lib.gk_topk_free.argtypes = [C.POINTER(gk_topk_out)]
    lib.gk_topk_free.restype = None
    lib.gk_comm_unique_id.argtypes = [C.c_char_p]
    lib.gk_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.gk_comm_destroy.argtypes = [vp]
    lib.gk_comm_destroy.restype = None
    lib.gk_comm_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.gk_table_sweep_sharded.argtypes = [vp, vp, u32, C.POINTER(C.POINTER(gk_shard_out))]
    lib.gk_shard_free.argtypes = [C.POINTER(gk_shard_out)]
    lib.gk_shard_free.restype = None
    if hostemu:
        lib.gk_comm_init_host.argtypes = [vp, C.c_int, C.c_int, HE_ALLGATHER, HE_ALLREDUCE, vp]
    lib.gk_resident_sweep.argtypes = [vp, u32, C.POINTER(C.POINTER(gk_sweep_out))]
    lib.gk_sweep_free.argtypes = [C.POINTER(gk_sweep_out)]
    lib.gk_sweep_free.restype = None
    lib.gk_resident_review.argtypes = [vp, C.POINTER(cp), sz, C.POINTER(vp)]
    lib.gk_batcher_start.argtypes = [vp, C.POINTER(gk_batch_opts)]
    lib.gk_batcher_stop.argtypes = [vp]
    lib.gk_batcher_stop.restype = None
    lib.gk_query.argtypes = [vp, C.POINTER(gk_review_in), C.POINTER(vp), C.POINTER(gk_query_stats)]
    lib.gk_query_ex.argtypes = [vp, C.POINTER(gk_review_in), u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(gk_query_stats)]
    lib.gk_query_ex2.argtypes = [vp, C.POINTER(gk_review_in), C.POINTER(u32), sz, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(gk_query_stats)]
    lib.gk_resident_review_ex.argtypes = [vp, C.POINTER(cp), sz, C.POINTER(u32), sz, u32, C.POINTER(vp)]
    lib.gk_debug_set.argtypes = [cp, C.c_int64]
    lib.gk_table_get_stats.argtypes = [vp, C.POINTER(gk_table_stats)]
    lib.gk_table_totals.argtypes = [vp, vp, C.POINTER(C.POINTER(gk_totals_out))]
    lib.gk_totals_free.argtypes = [C.POINTER(gk_totals_out)]
    lib.gk_totals_free.restype = None
    lib.gk_synth_batch_create.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(cp), sz, C.POINTER(vp)]
    lib.gk_synth_batch_reviews.argtypes = [vp]
    lib.gk_synth_batch_reviews.restype = C.POINTER(gk_review_in)
    lib.gk_synth_batch_size.argtypes = [vp]
    lib.gk_synth_batch_size.restype = sz
    lib.gk_synth_batch_json_bytes.argtypes = [vp]
    lib.gk_synth_batch_json_bytes.restype = C.c_uint64
    lib.gk_synth_batch_free.argtypes = [vp]
    lib.gk_synth_batch_free.restype = None
    lib.gk_synth_query_storm.argtypes = [vp, vp, u32, u32, C.POINTER(gk_storm_out)]
    lib.gk_synth_query_storm_ex.argtypes = [vp, vp, u32, u32, C.POINTER(u32), sz, u32, C.POINTER(gk_storm_out)]
    # a plan-specialised kernel may still be compiling in the background when the interpreter exits; exit() under a running
    # hiprtc compile crashes in the compiler's teardown, so the builds are joined first (Python's atexit runs before exit())
    atexit.register(lib.gk_jit_quiesce)
    _cache[hostemu] = lib
    return lib
