"""ctypes binding of libparametron_hip.so (the C ABI declared in include/parametron_hip.h).

This is the product path: there is NO CPU fallback.  If the shared library is missing or no
MI355X is visible, the calls raise — loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMT_LIB_PATH") or os.path.join(HERE, "lib", "libparametron_hip.so")   # override: A/B builds of the library

# Julia isbits layouts as numpy structured dtypes
LT = np.dtype([("coeff", "<f8"), ("var", "<i8")])
QT = np.dtype([("coeff", "<f8"), ("row", "<i8"), ("col", "<i8")])
VAT = np.dtype([("out", "<i8"), ("coeff", "<f8"), ("var", "<i8")])

PMT_OK, PMT_DIMENSION_MISMATCH, PMT_INVALID_ARGUMENT, PMT_HIP_ERROR, PMT_STATE_ERROR, PMT_OUT_OF_MEMORY = range(6)


class DimensionMismatch(Exception):
    """Julia's DimensionMismatch (src/functions.jl:780-781 and the other @boundscheck sites)."""


class ArgumentError(ValueError):
    """Julia's ArgumentError (src/lazyexpression.jl:175,185; src/model.jl:226,243,246)."""


class ErrorException(RuntimeError):
    """Julia's ErrorException — error(...) (src/model.jl:50,61,69,139)."""


class HipError(RuntimeError):
    pass


_vp, _i64, _f64, _ci, _u64, _sz = C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); every symbol include/parametron_hip.h declares
SIGNATURES = {
    "pmt_last_error": (C.c_char_p, []),
    "pmt_version": (_ci, []),
    "pmt_device_count": (_ci, []),
    "pmt_affine_assemble_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _vp, _vp]),
    "pmt_affine_pack_vector_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _i64, _vp, _vp, _vp]),
    "pmt_affine_pack_vector_background_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _i64, _vp, _vp, _vp]),
    "pmt_vars_addsub_f64": (_ci, [_vp, _i64, _vp, _ci, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pmt_affvec_combine_f64": (_ci, [_i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _ci, _vp, _vp, _i64, _vp, _vp]),
    "pmt_affvec_scale_f64": (_ci, [_i64, _i64, _vp, _vp, _vp, _f64, _vp, _vp, _vp]),
    "pmt_scale_vars_f64": (_ci, [_vp, _i64, _vp, _f64, _vp, _vp]),
    "pmt_scale_numbers_f64": (_ci, [_vp, _i64, _vp, _f64, _vp, _vp]),
    "pmt_transpose_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "pmt_quad_combine_f64": (_ci, [_vp, _i64, _vp, _i64, _ci, _vp, _vp]),
    "pmt_quad_scale_f64": (_ci, [_vp, _i64, _vp, _f64, _vp, _vp]),
    "pmt_copy_bytes": (_ci, [_vp, _vp, _sz, _vp]),
    "pmt_matvecmul_affs_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pmt_vecdot_numbers_vars_f64": (_ci, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "pmt_vecdot_numbers_affs_f64": (_ci, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "pmt_quad_expand_f64": (_ci, [_i64, _vp, _i64, _vp, _vp, _i64, _vp, _ci, _vp, _vp, _vp, _vp, _vp]),
    "pmt_quad_gram_workspace_bytes": (_sz, [_i64, _i64]),
    "pmt_quad_gram_constant_order": (_ci, [_i64, _i64, C.POINTER(_ci), C.POINTER(_ci), C.POINTER(_ci)]),
    "pmt_quad_gram_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmt_quad_gram_csc_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmt_quad_gram_csc_deliver_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _f64, _vp, _vp, _ci, _vp, _vp, _vp, _vp]),
    "pmt_quad_gram_deliver_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp]),
    "pmt_fetch_synchronize": (_ci, [_vp]),
    "pmt_set_host_delivery": (_ci, [_ci]),
    "pmt_get_host_delivery": (_ci, [_ci, C.POINTER(_ci), C.POINTER(_ci)]),
    "pmt_set_fault_injection": (_ci, [_ci]),
    "pmt_bilinear_f64": (_ci, [_vp, _i64, _i64, _i64, _vp, _vp, _ci, _vp, _vp, _vp]),
    "pmt_fill_uniform_matrix_f64": (_ci, [_vp, _i64, _i64, _i64, _u64, _f64, _vp]),
    "pmt_plan_upload_2d": (_ci, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "pmt_plan_fetch_2d": (_ci, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "pmt_vecdot_terms_f64": (_ci, [_i64, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "pmt_vecdot_affs_vars_f64": (_ci, [_i64, _vp, _i64, _vp, _vp, _ci, _vp, _vp, _vp, _vp]),
    "pmt_pack_scalar_affine_f64": (_ci, [_vp, _i64, _vp, _vp, _vp]),
    "pmt_pack_scalar_quadratic_f64": (_ci, [_vp, _i64, _vp, _vp, _vp]),
    "pmt_pack_vector_affine_f64": (_ci, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "pmt_canonical_order_affine": (_ci, [_i64, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pmt_canonical_order_quadratic": (_ci, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pmt_canonical_order_device": (_ci, [_vp, _i64, _ci, _vp, _vp, C.POINTER(_i64), _vp]),
    "pmt_canonical_init_terms": (_ci, [_vp, _ci, _vp, _vp, _i64, _vp, _vp]),
    "pmt_segment_sum_f64": (_ci, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp]),
    "pmt_prune_zero_workspace_bytes": (_sz, [_i64, _ci]),
    "pmt_prune_zero_f64": (_ci, [_vp, _i64, _ci, _f64, _vp, _vp, _vp, _sz, _vp]),
    "pmt_csc_order": (_ci, [_i64, _vp, _vp, _i64, _i64, _ci, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pmt_csc_values_f64": (_ci, [_vp, _i64, _i64, _vp, _vp, _i64, _f64, _vp, _vp, _vp]),
    "pmt_qp_bounds_f64": (_ci, [_vp, _i64, _ci, _f64, _f64, _vp, _vp, _vp]),
    "pmt_csc_values_gather_f64": (_ci, [_vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp]),
    "pmt_copy_2d_f64": (_ci, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "pmt_qp_bounds_rows_f64": (_ci, [_vp, _vp, _vp, _i64, _f64, _vp, _vp, _vp]),
    "pmt_sparse_rowmajor_order": (_ci, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmt_sparse_assemble_f64": (_ci, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "pmt_sparse_pack_vector_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "pmt_sparse_slab_ptr": (_ci, [_i64, _i64, _ci, _vp, _vp, _vp]),
    "pmt_sparse_pack_vector_slabs_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _vp, _i64, _vp, _vp]),
    "pmt_sparse_assemble_slabs_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _vp, _vp]),
    "pmt_sparse_pack_vector_slabs_u32_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _vp, _i64, _vp, _vp]),
    "pmt_sparse_assemble_slabs_u32_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _ci, _vp, _vp]),
    "pmt_sparse_blocks_width": (_ci, [_i64, _i64, _vp, _vp, _vp]),
    "pmt_sparse_blocks_build": (_ci, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _vp, _vp]),
    "pmt_sparse_pack_vector_blocks_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp, _i64, _vp, _ci, _vp, _vp, _vp]),
    "pmt_sparse_assemble_blocks_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _ci, _vp, _ci, _vp, _vp, _vp]),
    "pmt_batch_lsq_slab_doubles": (_i64, [_i64, _i64]),
    "pmt_batch_lsq_coeffs_f64": (_ci, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _vp, _i64, _vp]),
    "pmt_batch_expand_f64": (_ci, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pmt_comm_unique_id": (_ci, [_vp]),
    "pmt_comm_init_rank": (_ci, [_ci, _ci, _vp, _ci, C.POINTER(_vp)]),
    "pmt_comm_destroy": (_ci, [_vp]),
    "pmt_comm_rccl_calls": (_i64, [_vp]),
    "pmt_batch_num_chunks": (_i64, [_i64, _i64]),
    "pmt_batch_chunk_range": (_ci, [_i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "pmt_batch_gathered_offset": (_i64, [_ci, _i64, _i64, _i64]),
    "pmt_batch_shard": (_ci, [_i64, _ci, _ci, C.POINTER(_i64), C.POINTER(_i64)]),
    "pmt_batch_allgather_f64": (_ci, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "pmt_batch_step_f64": (_ci, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _ci, _ci, _vp, _vp, _i64, _i64, _vp]),
    "pmt_consts_f64": (_ci, [_vp, _i64, _ci, _vp, _vp]),
    "pmt_fill_uniform_f64": (_ci, [_vp, _i64, _u64, _f64, _vp]),
    "pmt_fill_uniform_offset_f64": (_ci, [_vp, _i64, _u64, _u64, _f64, _vp]),
    "pmt_profile_enable": (_ci, [_ci]),
    "pmt_profile_filter": (_ci, [C.c_char_p]),
    "pmt_profile_kernel_stamps": (_ci, [_vp, _i64]),
    "pmt_device_clock_khz": (_ci, [_ci, C.POINTER(_ci)]),
    "pmt_profile_report": (_i64, [C.c_char_p, _sz]),
    "pmt_plan_create": (_ci, [_ci, _vp, C.POINTER(_vp)]),
    "pmt_plan_destroy": (_ci, [_vp]),
    "pmt_plan_stream": (_vp, [_vp]),
    "pmt_plan_alloc": (_ci, [_vp, _sz, C.POINTER(_vp)]),
    "pmt_plan_bytes_allocated": (_sz, [_vp]),
    "pmt_host_alloc": (_ci, [_sz, C.POINTER(_vp)]),
    "pmt_host_free": (_ci, [_vp]),
    "pmt_plan_upload": (_ci, [_vp, _vp, _vp, _sz]),
    "pmt_plan_fetch": (_ci, [_vp, _vp, _vp, _sz]),
    "pmt_plan_zero": (_ci, [_vp, _vp, _sz]),
    "pmt_plan_synchronize": (_ci, [_vp]),
    "pmt_plan_check": (_ci, [_vp]),
    "pmt_model_create": (_ci, [_vp, C.POINTER(_vp)]),
    "pmt_model_destroy": (_ci, [_vp]),
    "pmt_model_add_mailbox": (_ci, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, C.POINTER(_ci)]),
    "pmt_model_add_seed": (_ci, [_vp, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(_ci)]),
    "pmt_model_set_host": (_ci, [_vp, _ci, _vp]),
    "pmt_model_add_constant": (_ci, [_vp, _vp, _vp]),
    "pmt_model_add_fetch": (_ci, [_vp, _vp, _vp, C.c_size_t]),
    "pmt_model_num_slots": (_ci, [_vp]),
    "pmt_model_update": (_ci, [_vp, _vp, _ci, _ci]),
    "pmt_model_wait": (_ci, [_vp]),
    "pmt_plan_record_fetch": (_ci, [_vp, _vp, _vp, _sz]),
    "pmt_plan_record_fetch_2d": (_ci, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "pmt_host_copy_2d": (_ci, [_vp, _sz, _vp, _sz, _sz, _sz, _ci]),
    "pmt_plan_fetch_synchronize": (_ci, [_vp]),
    "pmt_plan_stage_upload": (_ci, [_vp, _vp, _vp, _sz]),
    "pmt_plan_stage_upload_2d": (_ci, [_vp, _vp, _sz, _vp, _sz, _sz, _sz]),
    "pmt_plan_wait_staged": (_ci, [_vp]),
    "pmt_plan_commit_staged": (_ci, [_vp, _vp, _vp, _sz]),
    "pmt_plan_staging_consumed": (_ci, [_vp]),
    "pmt_plan_staged_synchronize": (_ci, [_vp]),
    "pmt_plan_stage_slot": (_ci, [_vp, _ci]),
    "pmt_plan_begin_record": (_ci, [_vp]),
    "pmt_plan_end_record": (_ci, [_vp]),
    "pmt_plan_commit_lane": (_ci, [_vp, _ci]),
    "pmt_plan_set_lane": (_ci, [_vp, _ci]),
    "pmt_plan_lane_stream": (_ci, [_vp, _ci, C.POINTER(_vp)]),
    "pmt_plan_recording_stream": (_vp, [_vp]),
    "pmt_plan_tape_length": (_i64, [_vp]),
    "pmt_plan_set_fusion": (_ci, [_vp, _ci]),
    "pmt_plan_fused": (_ci, [_vp, C.POINTER(_ci), C.POINTER(_ci), C.POINTER(_i64)]),
    "pmt_plan_fused_phases": (_ci, [_vp]),
    "pmt_plan_fused_workgroups": (_ci, [_vp]),
    "pmt_fill_uniform_dyn_f64": (_ci, [_vp, _i64, _i64, _i64, C.POINTER(_u64), _f64, _vp]),
    "pmt_plan_update": (_ci, [_vp]),
    "pmt_plan_instantiate_graph": (_ci, [_vp]),
}

_lib = None


def load():
    """Load the HIP library; raises if it has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ErrorException(
                "libparametron_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever copy is loaded
        # first serves the whole process, and torch cannot initialise on top of the system copy ("No HIP GPUs are
        # available").  Hosts that use torch for device memory / streams / torch.distributed must therefore have
        # torch loaded BEFORE this library; a torch-free host (the Julia binding) uses /opt/rocm's runtime.
        if os.environ.get("PMT_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def require_gpu():
    n = load().pmt_device_count()
    if n <= 0:
        raise ErrorException("no MI355X / HIP device visible: the Parametron hot path has no CPU fallback")
    return n


def check(rc):
    if rc == PMT_OK:
        return
    msg = load().pmt_last_error().decode("utf-8", "replace")
    if rc == PMT_DIMENSION_MISMATCH:
        raise DimensionMismatch(msg)
    if rc == PMT_INVALID_ARGUMENT:
        raise ArgumentError(msg)
    if rc == PMT_STATE_ERROR:
        raise ErrorException(msg)
    if rc == PMT_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise HipError(msg)


def call(name, *args):
    """Call a status-returning entry point and raise the reference's exception type on failure."""
    check(getattr(load(), name)(*args))
