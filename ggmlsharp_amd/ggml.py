"""Python face of the host mirror (TEST SUPPORT: tests/support/ggml.h): the same ggml_* names and argument meaning as
GGMLSharp's public static API (Ggml.cs:1447, 2347-2395, 7137, 7648-7673, 3209), so a test program reads
like the reference's Test0..Test3.  All work is done by libggml_hip.so; this file only converts arguments.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (F16, F32, I8, I16, I32, Q4_0, Q4_1, Q5_0, Q8_0, Q8_1, ggml_cgraph, ggml_compute_params,  # noqa: F401
                   ggml_init_params, ggml_tensor)


def ggml_init(mem_size, mem_buffer=None, no_alloc=False):
    p = ggml_init_params(mem_size, mem_buffer, 1 if no_alloc else 0)
    ctx = _lib.lib().ggml_init(C.byref(p))
    return ctx  # None when all 64 slots are taken (Ggml.cs:1529-1536)


def ggml_free(ctx):
    _lib.lib().ggml_free(ctx)


def ggml_new_tensor_1d(ctx, type, ne0):
    return _lib.lib().ggml_new_tensor_1d(ctx, type, ne0)


def ggml_new_tensor_2d(ctx, type, ne0, ne1):
    return _lib.lib().ggml_new_tensor_2d(ctx, type, ne0, ne1)


def ggml_new_tensor_3d(ctx, type, ne0, ne1, ne2):
    return _lib.lib().ggml_new_tensor_3d(ctx, type, ne0, ne1, ne2)


def ggml_new_tensor_4d(ctx, type, ne0, ne1, ne2, ne3):
    return _lib.lib().ggml_new_tensor_4d(ctx, type, ne0, ne1, ne2, ne3)


def ggml_nelements(t):
    return _lib.lib().ggml_nelements(t)


def ggml_nbytes(t):
    return _lib.lib().ggml_nbytes(t)


def ggml_set_f32(t, v):
    return _lib.lib().ggml_set_f32(t, v)


def ggml_get_f32_1d(t, i):
    return _lib.lib().ggml_get_f32_1d(t, i)


def ggml_mul_mat(ctx, a, b):
    return _lib.lib().ggml_mul_mat(ctx, a, b)


def ggml_cpy(ctx, a, b):
    return _lib.lib().ggml_cpy(ctx, a, b)


def ggml_add(ctx, a, b):
    return _lib.lib().ggml_add(ctx, a, b)


def ggml_mul(ctx, a, b):
    return _lib.lib().ggml_mul(ctx, a, b)


def ggml_scale(ctx, a, b):
    return _lib.lib().ggml_scale(ctx, a, b)


def ggml_rms_norm(ctx, a):
    return _lib.lib().ggml_rms_norm(ctx, a)


def ggml_silu(ctx, a):
    return _lib.lib().ggml_silu(ctx, a)


def ggml_silu_inplace(ctx, a):
    return _lib.lib().ggml_silu_inplace(ctx, a)


def ggml_build_forward(tensor):
    g = ggml_cgraph()
    _lib.lib().ggml_build_forward(C.byref(g), tensor)
    return g


def ggml_graph_compute(ctx, graph):
    rc = _lib.lib().ggml_graph_compute(ctx, C.byref(graph))
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
