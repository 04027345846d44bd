"""ctypes binding of libdfft_amd.so (include/dfft_c.h).  No fallback: a missing or stale
library is an ImportError / RuntimeError, never a silent CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFFT_AMD_LIBRARY names another build of the SAME library (the A/B build of `make -C csrc exp`: tools/kbench_exp, the
# fp32-twiddle proof of tests/parity_metric.py); there is still no fallback: what it names must exist and export every symbol
LIB_PATH = os.environ.get("DFFT_AMD_LIBRARY") or os.path.join(_HERE, "libdfft_amd.so")

ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                           C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                           C.c_int, C.c_int, C.c_void_p)
# dfft_sendrecv_list_fn: (user, nsend, speer, sptr, sbytes, nrecv, rpeer, rptr, rbytes, stream)
SENDRECV_LIST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                               C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)


class Config(C.Structure):
    _fields_ = [("cuda_aware", C.c_int), ("warmup_rounds", C.c_int), ("comm_method", C.c_int),
                ("send_method", C.c_int), ("comm_method2", C.c_int), ("send_method2", C.c_int)]


class PassDesc(C.Structure):
    """dfft_pass_desc (include/dfft_c.h): one axis pass as the kernels see it"""
    _fields_ = [("na", C.c_uint32), ("LB", C.c_uint32), ("nb", C.c_uint32), ("LA", C.c_uint32), ("T2shift", C.c_uint32),
                ("load_kind", C.c_int32), ("store_kind", C.c_int32), ("swap", C.c_int32), ("shift", C.c_int32),
                ("KS_in", C.c_uint64), ("KS_out", C.c_uint64), ("AS_in", C.c_uint64), ("AS_out", C.c_uint64),
                ("IA", C.c_uint64), ("IB", C.c_uint64), ("SK", C.c_uint64), ("SB", C.c_uint64),
                ("a_fastest", C.c_int32), ("xcd_swizzle", C.c_int32),
                ("in_off", C.c_uint64), ("out_off", C.c_uint64),
                ("lnseg", C.c_int32), ("snseg", C.c_int32),
                ("lstart", C.c_uint32 * 32), ("llen", C.c_uint32 * 32), ("lbase", C.c_uint64 * 32),
                ("sstart", C.c_uint32 * 32), ("slen", C.c_uint32 * 32), ("sbase", C.c_uint64 * 32)]


# every symbol include/dfft_c.h declares: (name, restype, argtypes)
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_psz = C.POINTER(C.c_size_t)
SYMBOLS = [
    ("dfft_comm_create_local", _i, [_i, C.POINTER(_vp)]),
    ("dfft_rccl_unique_id", _i, [_vp]),
    ("dfft_comm_create_rccl", _i, [_vp, _i, _i, C.POINTER(_vp)]),
    ("dfft_comm_create_callback", _i, [_i, _i, ALLTOALLV_FN, _vp, C.POINTER(_vp)]),
    ("dfft_comm_set_list_callback", _i, [_vp, SENDRECV_LIST_FN, _vp]),
    ("dfft_comm_info", _i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    ("dfft_comm_get_counter", _i, [_vp, C.c_char_p, C.POINTER(C.c_long)]),
    ("dfft_comm_set_option", _i, [_vp, C.c_char_p, C.c_long]),
    ("dfft_comm_alltoallv", _i, [_vp, _i, _vp, C.POINTER(_sz), C.POINTER(_sz), _vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i), _i, _i, _vp]),
    ("dfft_comm_sendrecv_list", _i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), _psz, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_vp), _psz, _i, _vp]),
    ("dfft_comm_destroy", _i, [_vp]),
    ("dfft_plan_create", _i, [C.POINTER(_vp), _i, _i, C.POINTER(Config), _vp, _i, _i]),
    ("dfft_plan_destroy", _i, [_vp]),
    ("dfft_init", _i, [_vp, _sz, _sz, _sz, _i, _i, _i, _i]),
    ("dfft_set_work_area", _i, [_vp, _vp, _vp]),
    ("dfft_set_pipeline_chunks", _i, [_vp, _i]),
    ("dfft_get_pipeline_chunks", _i, [_vp]),
    ("dfft_set_option", _i, [_vp, C.c_char_p, C.c_long]),
    ("dfft_get_option", C.c_long, [_vp, C.c_char_p]),
    ("dfft_set_stream", _i, [_vp, _vp]),
    ("dfft_exec_r2c", _i, [_vp, _vp, _vp]),
    ("dfft_exec_c2r", _i, [_vp, _vp, _vp]),
    ("dfft_exec_c2c", _i, [_vp, _vp, _vp, _i]),
    ("dfft_enqueue_c2c", _i, [_vp, _vp, _vp, _i]),
    ("dfft_exec_dim", _i, [_vp, _vp, _vp, _i, _i]),
    ("dfft_get_in_size", _i, [_vp, _psz]),
    ("dfft_get_in_start", _i, [_vp, _psz]),
    ("dfft_get_out_size", _i, [_vp, _psz]),
    ("dfft_get_out_start", _i, [_vp, _psz]),
    ("dfft_get_out_strides", _i, [_vp, _psz]),
    ("dfft_get_partition_dimensions", _i, [_vp, _i, _i, _psz, _psz, _sz, _psz]),
    ("dfft_domain_size", _sz, [_vp]),
    ("dfft_work_size_device", _sz, [_vp]),
    ("dfft_work_size_host", _sz, [_vp]),
    ("dfft_work_area_device", _vp, [_vp]),
    ("dfft_rank", _i, [_vp]),
    ("dfft_world_size", _i, [_vp]),
    ("dfft_get_exchange_tables", _i, [_vp, _i, _psz, _psz, _psz, _psz]),
    ("dfft_exchange", _i, [_vp, _i, _i, _vp, _vp]),
    ("dfft_get_pipeline_tables", _i, [_vp, _i, _i, _i, _psz, _psz, _psz, _psz]),
    ("dfft_tile_lines", _i, [_vp]),
    ("dfft_get_phase_times", _i, [_vp, C.POINTER(C.c_float), _i]),
    ("dfft_phase_name", C.c_char_p, [_i, _i]),
    ("dfft_enable_phase_timing", _i, [_vp, _i]),
    ("dfft_fft1d_batched", _i, [_i, _sz, _sz, _vp, _vp, _i, _vp]),
    ("dfft_fft1d_batched_ex", _i, [_i, _sz, _sz, _vp, _vp, _i, _vp, _i, _i]),
    ("dfft_get_pass_choices", _i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("dfft_debug_get_pass", _i, [_vp, C.c_char_p, _i, C.POINTER(PassDesc)]),
    ("dfft_debug_get_point_table", _i, [_vp, C.c_char_p, _i, _i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), _sz, _psz]),
    ("dfft_last_error", C.c_char_p, []),
    ("dfft_version", C.c_char_p, []),
    ("dfft_kernel_info", _i, [_i, _sz, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    ("dfft_axis_plan_info", _i, [_i, _sz, _i, C.POINTER(_sz)]),
    ("dfft_malloc", _i, [_sz, _sz, C.POINTER(_vp)]),
    ("dfft_free", _i, [_vp]),
    ("dfft_last_placement_info", _i, [C.c_char_p, _sz]),
    ("dfft_tune_variants", _i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    ("dfft_tune_placement", _i, [_vp, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_float), _i, C.POINTER(_i)]),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C distributedfft_amd/csrc` (hipcc, gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)       # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class DfftError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise DfftError(f"libdfft_amd error {rc}: {lib().dfft_last_error().decode()}")
