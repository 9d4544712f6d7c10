"""TEST SUPPORT -- Python face of the host mirror libggml_hostmirror.so (tests/support/ggml.h + ggml_host.cpp: the C++ stand-in for the
reference's C# host): the same ggml_* names and argument meaning as GGMLSharp's public static API (Ggml.cs:1447, 2347-2395, 7137,
7648-7673, 3209), so a test program reads like the reference's Test0..Test3.  Not part of the product: the product package
(ggmlsharp_amd/) binds libggml_hip.so alone and loads without this file or the mirror library.  The mirror links the product through
its C-ABI as the C# host would; all device work is done by libggml_hip.so.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ggmlsharp_amd import _lib
from ggmlsharp_amd._lib import (F16, F32, I8, I16, I32, Q4_0, Q4_1, Q5_0, Q8_0, Q8_1, ggml_cgraph, ggml_compute_params,  # noqa: F401
                                ggml_init_params, ggml_tensor)

SUPPORT_DIR = os.path.dirname(os.path.abspath(__file__))
MIRROR_PATH = os.path.join(SUPPORT_DIR, "libggml_hostmirror.so")

_P = C.c_void_p
_T = C.POINTER(ggml_tensor)
# every symbol tests/support/ggml.h declares (exported by libggml_hostmirror.so): name -> (restype, argtypes)
MIRROR_SYMBOLS = {
    "ggml_init": (_P, [C.POINTER(ggml_init_params)]),
    "ggml_free": (None, [_P]),
    "ggml_used_mem": (C.c_size_t, [_P]),
    "ggml_new_tensor": (_T, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "ggml_new_tensor_1d": (_T, [_P, C.c_int, C.c_int64]),
    "ggml_new_tensor_2d": (_T, [_P, C.c_int, C.c_int64, C.c_int64]),
    "ggml_new_tensor_3d": (_T, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "ggml_new_tensor_4d": (_T, [_P, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "ggml_nelements": (C.c_int64, [_T]),
    "ggml_nrows": (C.c_int64, [_T]),
    "ggml_nbytes": (C.c_size_t, [_T]),
    "ggml_blck_size": (C.c_int, [C.c_int]),
    "ggml_type_size": (C.c_size_t, [C.c_int]),
    "ggml_is_quantized": (C.c_int, [C.c_int]),
    "ggml_is_contiguous": (C.c_int, [_T]),
    "ggml_can_mul_mat": (C.c_int, [_T, _T]),
    "ggml_set_f32": (_T, [_T, C.c_float]),
    "ggml_get_f32_1d": (C.c_float, [_T, C.c_int]),
    "ggml_set_f32_1d": (None, [_T, C.c_int, C.c_float]),
    "ggml_mul_mat": (_T, [_P, _T, _T]),
    "ggml_view_tensor": (_T, [_P, _T]),
    "ggml_dup_tensor": (_T, [_P, _T]),
    "ggml_cpy": (_T, [_P, _T, _T]),
    "ggml_add": (_T, [_P, _T, _T]),
    "ggml_mul": (_T, [_P, _T, _T]),
    "ggml_scale": (_T, [_P, _T, _T]),
    "ggml_rms_norm": (_T, [_P, _T]),
    "ggml_silu": (_T, [_P, _T]),
    "ggml_silu_inplace": (_T, [_P, _T]),
    "ggml_build_forward": (None, [C.POINTER(ggml_cgraph), _T]),
    "ggml_build_forward_expand": (None, [C.POINTER(ggml_cgraph), _T]),
    "ggml_graph_compute": (C.c_int, [_P, C.POINTER(ggml_cgraph)]),
}

_mirror = None


def mirror():
    """libggml_hostmirror.so, loaded behind the product library (whose ggml_hip_* symbols it binds, as the C# host would)."""
    global _mirror
    if _mirror is None:
        _lib.lib()                                   # RTLD_GLOBAL: the mirror resolves ggml_hip_* against it
        if not os.path.exists(MIRROR_PATH):
            subprocess.check_call(["make", "-C", SUPPORT_DIR], stdout=subprocess.DEVNULL)
        m = C.CDLL(MIRROR_PATH)
        for name, (res, args) in MIRROR_SYMBOLS.items():
            fn = getattr(m, name)
            fn.restype = res
            fn.argtypes = args
        _mirror = m
    return _mirror


def ggml_init(mem_size, mem_buffer=None, no_alloc=False):
    p = ggml_init_params(mem_size, mem_buffer, 1 if no_alloc else 0)
    ctx = mirror().ggml_init(C.byref(p))
    return ctx  # None when all 64 slots are taken (Ggml.cs:1529-1536)


def ggml_free(ctx):
    mirror().ggml_free(ctx)


def ggml_new_tensor_1d(ctx, type, ne0):
    return mirror().ggml_new_tensor_1d(ctx, type, ne0)


def ggml_new_tensor_2d(ctx, type, ne0, ne1):
    return mirror().ggml_new_tensor_2d(ctx, type, ne0, ne1)


def ggml_new_tensor_3d(ctx, type, ne0, ne1, ne2):
    return mirror().ggml_new_tensor_3d(ctx, type, ne0, ne1, ne2)


def ggml_new_tensor_4d(ctx, type, ne0, ne1, ne2, ne3):
    return mirror().ggml_new_tensor_4d(ctx, type, ne0, ne1, ne2, ne3)


def ggml_nelements(t):
    return mirror().ggml_nelements(t)


def ggml_nbytes(t):
    return mirror().ggml_nbytes(t)


def ggml_set_f32(t, v):
    return mirror().ggml_set_f32(t, v)


def ggml_get_f32_1d(t, i):
    return mirror().ggml_get_f32_1d(t, i)


def ggml_mul_mat(ctx, a, b):
    return mirror().ggml_mul_mat(ctx, a, b)


def ggml_cpy(ctx, a, b):
    return mirror().ggml_cpy(ctx, a, b)


def ggml_add(ctx, a, b):
    return mirror().ggml_add(ctx, a, b)


def ggml_mul(ctx, a, b):
    return mirror().ggml_mul(ctx, a, b)


def ggml_scale(ctx, a, b):
    return mirror().ggml_scale(ctx, a, b)


def ggml_rms_norm(ctx, a):
    return mirror().ggml_rms_norm(ctx, a)


def ggml_silu(ctx, a):
    return mirror().ggml_silu(ctx, a)


def ggml_silu_inplace(ctx, a):
    return mirror().ggml_silu_inplace(ctx, a)


def ggml_build_forward(tensor):
    g = ggml_cgraph()
    mirror().ggml_build_forward(C.byref(g), tensor)
    return g


def ggml_graph_compute(ctx, graph):
    rc = mirror().ggml_graph_compute(ctx, C.byref(graph))
    _lib.check(rc, "ggml_graph_compute")


def tensor_bytes(t):
    """numpy uint8 view of a tensor's data (the pool memory itself, no copy)."""
    n = ggml_nbytes(t)
    return np.ctypeslib.as_array((C.c_uint8 * n).from_address(t.contents.data))


def tensor_f32(t):
    tt = t.contents
    n = ggml_nelements(t)
    arr = np.ctypeslib.as_array((C.c_float * n).from_address(tt.data))
    return arr.reshape(tt.ne[3], tt.ne[2], tt.ne[1], tt.ne[0])
