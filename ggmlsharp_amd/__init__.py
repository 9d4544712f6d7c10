"""ggmlsharp_amd -- MI355X-native quantized mul_mat path behind GGMLSharp's ggml_* surface.

Layout:  csrc/   hand-written gfx950 kernels + the C-ABI (include/ggml_hip.h, include/ggml_hip_ext.h)
         _lib.py ctypes binding of libggml_hip.so (the product library alone; the host mirror is test support: tests/support/)
         device.py resident-weight / device-buffer helpers (torch is used for memory, streams and RCCL only)
         dist.py row-split across GPUs + all-gather
"""
from . import _lib  # noqa: F401
from ._lib import (F16, F32, Q4_0, Q4_1, Q5_0, Q8_0, Q8_1, BLCK_SIZE, TYPE_NAME, TYPE_SIZE, GgmlHipError, build,  # noqa: F401
                   lib)
