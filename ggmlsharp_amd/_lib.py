"""ctypes binding of libggml_hip.so -- the product: the C-ABI declared in include/ggml_hip.h and include/ggml_hip_ext.h.

Built in-tree (ggmlsharp_amd/lib/) by `make -C ggmlsharp_amd/csrc`.  Nothing here falls back to a CPU implementation: if the
library is missing, loading raises.  (The C++ stand-in for the reference's C# host and its Python face are TEST SUPPORT and live in
tests/support/ -- ggml_host.cpp, ggml_mirror.py; this package neither needs nor loads them.)
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("GGML_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libggml_hip.so")   # env: developer ablation builds

GGML_MAX_DIMS = 4
GGML_MAX_OPT = 4
GGML_MAX_NODES = 4096

# ggml_type (TypeDefinitions.cs:153-169)
F32, F16, Q4_0, Q4_1, Q4_2, Q4_3, Q5_0, Q5_1, Q8_0, Q8_1, I8, I16, I32 = range(13)
Q5_K = 113   # extension (include/ggml_hip_ext.h GGML_HIP_TYPE_Q5_K): upstream k-quant format, absent from the reference
Q4_K = 112   # ... and GGML_HIP_TYPE_Q4_K (r4): the same super-block without the fifth-bit bytes
Q6_K = 114   # ... and GGML_HIP_TYPE_Q6_K (r4): sixteen sub-blocks of 16 six-bit weights, resident in the planar Q4_2 form on int8 planes
TYPE_NAME = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q4_1: "q4_1", Q4_2: "q4_2", Q4_3: "q4_3", Q5_0: "q5_0",
             Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1", I8: "i8", I16: "i16", I32: "i32"}
BLCK_SIZE = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q4_2: 16, Q4_3: 16, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32,
             I8: 1, I16: 1, I32: 1, Q5_K: 256, Q4_K: 256, Q6_K: 256}
TYPE_SIZE = {F32: 4, F16: 2, Q4_0: 20, Q4_1: 24, Q4_2: 10, Q4_3: 12, Q5_0: 22, Q5_1: 24, Q8_0: 36, Q8_1: 44,
             I8: 1, I16: 2, I32: 4, Q5_K: 176, Q4_K: 144, Q6_K: 210}

GGML_OP_NONE, GGML_OP_ADD, GGML_OP_MUL_MAT, GGML_OP_CPY = 0, 2, 20, 22
GGML_OP_SILU = 17
GGML_OP_MUL, GGML_OP_RMS_NORM, GGML_OP_SCALE = 4, 19, 21
GGML_TASK_INIT, GGML_TASK_COMPUTE, GGML_TASK_FINALIZE = 0, 1, 2

OK, ERR_NO_DEVICE, ERR_TYPE, ERR_SHAPE, ERR_ARG, ERR_RUNTIME = 0, -1, -2, -3, -4, -5


class ggml_tensor(C.Structure):
    pass


ggml_tensor._fields_ = [
    ("type", C.c_int32), ("n_dims", C.c_int32),
    ("ne", C.c_int64 * GGML_MAX_DIMS), ("nb", C.c_uint64 * GGML_MAX_DIMS),
    ("op", C.c_int32), ("is_param", C.c_uint8), ("_pad0", C.c_uint8 * 3),
    ("grad", C.POINTER(ggml_tensor)), ("src0", C.POINTER(ggml_tensor)), ("src1", C.POINTER(ggml_tensor)),
    ("opt", C.c_int64 * GGML_MAX_OPT),
    ("n_tasks", C.c_int32), ("perf_runs", C.c_int32), ("perf_cycles", C.c_int64), ("perf_time_us", C.c_int64),
    ("data", C.c_void_p), ("padding", C.c_uint8 * 8),
]
assert C.sizeof(ggml_tensor) == 176


class ggml_compute_params(C.Structure):
    _fields_ = [("type", C.c_int32), ("ith", C.c_int32), ("nth", C.c_int32), ("wsize", C.c_size_t),
                ("wdata", C.c_void_p)]


class ggml_init_params(C.Structure):
    _fields_ = [("mem_size", C.c_uint64), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_uint8)]


class ggml_cgraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_leafs", C.c_int32), ("n_threads", C.c_int32), ("_pad0", C.c_int32),
                ("work_size", C.c_size_t), ("work", C.POINTER(ggml_tensor)),
                ("nodes", C.POINTER(ggml_tensor) * GGML_MAX_NODES),
                ("grads", C.POINTER(ggml_tensor) * GGML_MAX_NODES),
                ("leafs", C.POINTER(ggml_tensor) * GGML_MAX_NODES),
                ("perf_runs", C.c_int32), ("_pad1", C.c_int32), ("perf_cycles", C.c_int64),
                ("perf_time_us", C.c_int64)]


assert C.sizeof(ggml_cgraph) == 98360

class ggml_hip_mm_plan_t(C.Structure):
    """include/ggml_hip_ext.h: the plan of one product (csrc/plan.cpp)"""
    _fields_ = [("family", C.c_int32), ("image_kind", C.c_int32), ("form", C.c_int32), ("tree_id", C.c_uint32),
                ("arith", C.c_int32), ("ksplit", C.c_int32), ("kstyle", C.c_int32), ("kunit", C.c_int32),
                ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("waves", C.c_int32), ("tiles_per_wave", C.c_int32),
                ("workgroups", C.c_int64), ("flags", C.c_int32)]


# every symbol include/ggml_hip.h and include/ggml_hip_ext.h declare (exported by libggml_hip.so): name -> (restype, argtypes)
_P = C.c_void_p
_T = C.POINTER(ggml_tensor)
_PP = C.POINTER(C.c_void_p)
HIP_SYMBOLS = {
    "ggml_hip_blck_size": (C.c_int, [C.c_int]),
    "ggml_hip_type_size": (C.c_size_t, [C.c_int]),
    "ggml_hip_device_count": (C.c_int, []),
    "ggml_hip_init": (C.c_int, [C.c_int]),
    "ggml_hip_init_devices": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "ggml_hip_n_slots": (C.c_int, []),
    "ggml_hip_slot_device": (C.c_int, [C.c_int]),
    "ggml_hip_bind_thread": (C.c_int, [C.c_int]),
    "ggml_hip_register_host_pool": (C.c_int, [_P, C.c_size_t]),
    "ggml_hip_unregister_host_pool": (C.c_int, [_P]),
    "ggml_hip_invalidate_range": (None, [_P, C.c_size_t]),
    "ggml_hip_split_weight_upload": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(_P)]),
    "ggml_hip_split_weight_free": (None, [_P]),
    "ggml_hip_split_weight_rows": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ggml_hip_mul_mat_split_dev": (C.c_int, [_P, _PP, C.c_int64, C.c_int64, _PP, C.c_int64]),
    "ggml_hip_set_exchange": (C.c_int, [C.c_int]),
    "ggml_hip_sync_slots": (C.c_int, []),
    "ggml_hip_debug_rccl_selftest": (C.c_int, []),
    "ggml_hip_ipc_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P), _P]),
    "ggml_hip_ipc_open": (C.c_int, [_P, C.POINTER(_P)]),
    "ggml_hip_ipc_close": (C.c_int, [_P]),
    "ggml_hip_ipc_free": (C.c_int, [_P]),
    "ggml_hip_push_columns_dev": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _PP, C.c_int, C.c_int64, C.c_int64, _P]),
    "ggml_hip_mul_mat_push_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _PP, C.c_int, C.c_int, C.c_int64, C.c_int64, _P, C.c_size_t, _P]),
    "ggml_hip_mul_mat_push_fused": (C.c_int, [_P, C.c_int64, C.c_int]),
    "ggml_hip_slot_malloc": (_P, [C.c_int, C.c_size_t]),
    "ggml_hip_slot_free": (None, [C.c_int, _P]),
    "ggml_hip_slot_upload": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "ggml_hip_slot_download": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "ggml_hip_shutdown": (None, []),
    "ggml_hip_last_error": (C.c_char_p, []),
    "ggml_hip_arch": (C.c_char_p, []),
    "ggml_hip_compute_forward_mul_mat": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T]),
    "ggml_hip_invalidate": (None, [_P]),
    "ggml_hip_invalidate_all": (None, []),
    "ggml_hip_weight_upload": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, _P,
                                         C.POINTER(_P)]),
    "ggml_hip_weight_from_device": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, C.c_int64,
                                              _P, C.POINTER(_P)]),
    "ggml_hip_weight_download": (C.c_int, [_P, _P, _P]),
    "ggml_hip_weight_free": (None, [_P]),
    "ggml_hip_weight_rows": (C.c_int64, [_P]),
    "ggml_hip_weight_cols": (C.c_int64, [_P]),
    "ggml_hip_weight_type": (C.c_int, [_P]),
    "ggml_hip_mul_mat_work_size": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64]),
    "ggml_hip_mul_mat_multi_fused": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64]),
    "ggml_hip_mul_mat_multi_dev": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ggml_hip_mul_mat_multi_work_dev": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]),
    "ggml_hip_graph_begin": (C.c_int, []),
    "ggml_hip_graph_begin_keyed": (C.c_int, [C.c_uint64]),
    "ggml_hip_host_read": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ggml_hip_graph_outputs": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "ggml_hip_debug_scope_counters": (None, [C.POINTER(C.c_uint64)] * 4),
    "ggml_hip_graph_end": (C.c_int, []),
    "ggml_hip_debug_transfer_counters": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ggml_hip_debug_weight_cache_budget": (None, [C.c_size_t]),
    "ggml_hip_debug_weight_cache_stats": (None, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ggml_hip_act_image_kind": (C.c_int, [C.c_int, C.c_int64, C.c_int64]),
    "ggml_hip_mm_plan": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, _P]),
    "ggml_hip_debug_force_gemm": (None, [C.c_int]),
    "ggml_hip_quantize_act_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ggml_hip_mul_mat_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, _P, C.c_size_t, _P]),
    "ggml_hip_mul_mat_init_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_size_t, _P]),
    "ggml_hip_mul_mat_compute_dev": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_size_t, _P]),
    "ggml_hip_quantize_rows_dev": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, _P, _P]),
    "ggml_hip_dequantize_rows_dev": (C.c_int, [C.c_int, _P, C.c_int64, C.c_int64, _P, _P]),
    "ggml_hip_quantize_row": (C.c_int, [C.c_int, _P, _P, C.c_int]),
    "ggml_hip_dequantize_row": (C.c_int, [C.c_int, _P, _P, C.c_int]),
    "ggml_hip_vec_dot": (C.c_int, [C.c_int, C.c_int, _P, _P, _P]),
    "ggml_hip_compute_forward_cpy": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T]),
    "ggml_hip_compute_forward_add": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T]),
    "ggml_hip_compute_forward_mul": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T]),
    "ggml_hip_compute_forward_scale": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T]),
    "ggml_hip_compute_forward_rms_norm": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T]),
    "ggml_hip_compute_forward_silu": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T]),
    "ggml_hip_compute_forward_rms_norm_mul": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T, _T]),
    "ggml_hip_compute_forward_silu_mul": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T, _T]),
    "ggml_hip_compute_forward_mul_mat_add": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T, _T, _T]),
    "ggml_hip_compute_forward_mul_mat_scale": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T, _T, _T]),
    "ggml_hip_compute_forward_norm_mul_mat": (C.c_int, [C.POINTER(ggml_compute_params), _T, _T, _T, _T, _T, _T, _T, _T]),
    "ggml_hip_compute_forward_mul_mat_multi": (C.c_int, [C.POINTER(ggml_compute_params), C.c_int, C.POINTER(_T), _T, C.POINTER(_T), _T, _T, _T]),
    "ggml_hip_norm_mul_mat_dev": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, _P, C.c_size_t, C.c_int, _P,
                                             C.c_int64, _P, C.c_int64, C.c_float, _P]),
    "ggml_hip_norm_mul_mat_fused": (C.c_int, [_P, C.c_int64]),
    "ggml_hip_rms_norm_mul_rows_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, _P]),
    "ggml_hip_mul_mat_epilogue_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, _P, C.c_size_t, C.c_int, _P, C.c_int64, _P,
                                                 C.c_int64, C.c_float, _P]),
    "ggml_hip_mul_mat_epilogue_fused": (C.c_int, [_P, C.c_int64]),
    "ggml_hip_quantize_rows_src_dev": (C.c_int, [C.c_int, C.c_int, _P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "ggml_hip_add_q_f32_rows_dev": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int64, _P, _P]),
    "ggml_hip_relayout_gathered_dev": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P]),
}
SYMBOLS = HIP_SYMBOLS


def build(force=False):
    """Compile every HIP source for gfx950 into ggmlsharp_amd/lib/libggml_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC_DIR, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            if os.path.exists("/opt/rocm/bin/hipcc"):
                build()
            else:
                raise RuntimeError(f"{LIB_PATH} is missing and hipcc is not available; there is no CPU fallback")
        # torch ships its own libamdhip64 under the same SONAME as /opt/rocm's.  Whichever is loaded first serves
        # the whole process, and torch cannot see the GPU if the system one got in first -- so let torch go first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        hip = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)     # (global: a host library that binds ggml_hip_* by name resolves against it)
        for name, (res, args) in HIP_SYMBOLS.items():
            fn = getattr(hip, name)
            fn.restype = res
            fn.argtypes = args
        _lib = hip
    return _lib


class GgmlHipError(RuntimeError):
    def __init__(self, status, where):
        msg = lib().ggml_hip_last_error().decode(errors="replace")
        super().__init__(f"{where}: status {status}: {msg}")
        self.status = status


def check(status, where):
    if status != OK:
        raise GgmlHipError(status, where)
