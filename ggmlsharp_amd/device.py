"""Device-level helpers over the C-ABI: resident weights and the hot path on HBM-resident buffers.

torch is used for device memory and streams only (torch tensors own the buffers; the kernels are ours and run on
torch's current stream).  Nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BLCK_SIZE, TYPE_SIZE, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def init(device=None):
    if device is None:
        device = torch.cuda.current_device()
    check(lib().ggml_hip_init(int(device)), "ggml_hip_init")


class Weight:
    """One resident 2-D weight matrix [M rows, K] (a row shard when row_begin/row_end are given)."""

    def __init__(self, handle, type, M, K):
        self.handle, self.type, self.M, self.K = handle, type, M, K

    @staticmethod
    def row_bytes(type, K):
        return TYPE_SIZE[type] * (K // BLCK_SIZE[type])

    @classmethod
    def from_host(cls, type, rows, K, row_begin=0, row_end=None):
        """rows: numpy uint8/f32/f16 array holding M reference-format rows (contiguous)."""
        rows = np.ascontiguousarray(rows)
        rb = cls.row_bytes(type, K)
        M = rows.nbytes // rb
        if row_end is None:
            row_end = M
        h = C.c_void_p()
        check(lib().ggml_hip_weight_upload(type, rows.ctypes.data_as(C.c_void_p), K, M, rb, row_begin, row_end,
                                           _stream(), C.byref(h)), "ggml_hip_weight_upload")
        return cls(h, type, row_end - row_begin, K)

    @classmethod
    def from_device(cls, type, rows_t, K, row_begin=0, row_end=None):
        rb = cls.row_bytes(type, K)
        M = rows_t.numel() * rows_t.element_size() // rb
        if row_end is None:
            row_end = M
        h = C.c_void_p()
        check(lib().ggml_hip_weight_from_device(type, C.c_void_p(rows_t.data_ptr()), K, M, rb, row_begin, row_end,
                                                _stream(), C.byref(h)), "ggml_hip_weight_from_device")
        return cls(h, type, row_end - row_begin, K)

    def download(self):
        out = np.zeros(self.M * self.row_bytes(self.type, self.K), dtype=np.uint8)
        check(lib().ggml_hip_weight_download(self.handle, out.ctypes.data_as(C.c_void_p), _stream()),
              "ggml_hip_weight_download")
        return out

    def free(self):
        if self.handle:
            lib().ggml_hip_weight_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def work_size(type, K, N):
    return int(lib().ggml_hip_mul_mat_work_size(type, K, N))


def alloc_work(type, K, N, device=None):
    n = max(work_size(type, K, N), 16)
    return torch.empty(n, dtype=torch.uint8, device=device or "cuda")


def mul_mat(w, x, out=None, work=None):
    """dst[N, M] = mul_mat(w, x[N, K]) on the current stream; x f32 row-major on the device."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    N = x.shape[0]
    if out is None:
        out = torch.empty((N, w.M), dtype=torch.float32, device=x.device)
    if work is None:
        work = alloc_work(w.type, w.K, N, x.device)
    check(lib().ggml_hip_mul_mat_dev(w.handle, C.c_void_p(x.data_ptr()), N, x.stride(0), C.c_void_p(out.data_ptr()),
                                     out.stride(0), C.c_void_p(work.data_ptr()), work.numel(), _stream()),
          "ggml_hip_mul_mat_dev")
    return out


def mul_mat_init(w, x, work):
    check(lib().ggml_hip_mul_mat_init_dev(w.handle, C.c_void_p(x.data_ptr()), x.shape[0], x.stride(0),
                                          C.c_void_p(work.data_ptr()), work.numel(), _stream()),
          "ggml_hip_mul_mat_init_dev")


def mul_mat_compute(w, N, out, work):
    check(lib().ggml_hip_mul_mat_compute_dev(w.handle, N, C.c_void_p(out.data_ptr()), out.stride(0),
                                             C.c_void_p(work.data_ptr()), work.numel(), _stream()),
          "ggml_hip_mul_mat_compute_dev")


def quantize_rows(type, x):
    """x f32 [nrows, k] on the device -> uint8 [nrows, k/32*type_size] reference-format blocks."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    nrows, k = x.shape
    out = torch.empty((nrows, k // BLCK_SIZE[type] * TYPE_SIZE[type]), dtype=torch.uint8, device=x.device)
    check(lib().ggml_hip_quantize_rows_dev(type, C.c_void_p(x.data_ptr()), nrows, k, C.c_void_p(out.data_ptr()),
                                           _stream()), "ggml_hip_quantize_rows_dev")
    return out


def quantize_rows_from(type, x):
    """x f32 or f16 [nrows, k] on the device (row stride may exceed k) -> reference-format blocks."""
    assert x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float32, torch.float16)
    nrows, k = x.shape
    out = torch.empty((nrows, k // BLCK_SIZE[type] * TYPE_SIZE[type]), dtype=torch.uint8, device=x.device)
    check(lib().ggml_hip_quantize_rows_src_dev(type, 0 if x.dtype == torch.float32 else 1, C.c_void_p(x.data_ptr()),
                                               x.stride(0), nrows, k, C.c_void_p(out.data_ptr()), _stream()),
          "ggml_hip_quantize_rows_src_dev")
    return out


def add_q_f32_rows(type, blocks, x):
    """blocks uint8 [nrows, k/32*type_size], x f32 [nrows, k] -> quantize(dequantize(blocks) + x)."""
    assert blocks.is_cuda and x.is_cuda and blocks.is_contiguous() and x.is_contiguous() and x.dtype == torch.float32
    nrows, k = x.shape
    out = torch.empty_like(blocks)
    check(lib().ggml_hip_add_q_f32_rows_dev(type, C.c_void_p(blocks.data_ptr()), C.c_void_p(x.data_ptr()), nrows, k,
                                            C.c_void_p(out.data_ptr()), _stream()), "ggml_hip_add_q_f32_rows_dev")
    return out


def dequantize_rows(type, blocks, k):
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    nrows = blocks.numel() // (k // BLCK_SIZE[type] * TYPE_SIZE[type])
    out = torch.empty((nrows, k), dtype=torch.float32, device=blocks.device)
    check(lib().ggml_hip_dequantize_rows_dev(type, C.c_void_p(blocks.data_ptr()), nrows, k,
                                             C.c_void_p(out.data_ptr()), _stream()), "ggml_hip_dequantize_rows_dev")
    return out


def relayout_gathered(gathered, G, N, Ms, M, out=None):
    if out is None:
        out = torch.empty((N, M), dtype=torch.float32, device=gathered.device)
    check(lib().ggml_hip_relayout_gathered_dev(C.c_void_p(gathered.data_ptr()), G, N, Ms, C.c_void_p(out.data_ptr()),
                                               M, out.stride(0), _stream()), "ggml_hip_relayout_gathered_dev")
    return out
