"""CPU tests of the boundary: the C-ABI library loads, exports every symbol the headers declare, and the host
mirror of the ggml API keeps the reference's layouts and pool arithmetic.  No compute call succeeds without a GPU:
the product has no CPU path, so on a GPU-less box every compute entry must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ggmlsharp_amd import _lib
import ggml_mirror as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return _lib.lib().ggml_hip_device_count() > 0


def _declared(header):
    path = os.path.join(ROOT, "include", header)
    if not os.path.exists(path):
        path = os.path.join(ROOT, "tests", "support", header)       # the host mirror's ggml.h (test support)
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(ggml_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.strip()})


def test_library_exports_every_declared_symbol():
    """libggml_hip.so (the product) exports exactly the ggml_hip_* names of include/ggml_hip.h -- no ggml_* name that
    could collide with a native ggml in the host process, no C++ symbol; the host mirror (test support, its own
    library, tests/support/) exports exactly what its ggml.h declares."""
    L = _lib.lib()
    core_decl, ext_decl = set(_declared("ggml_hip.h")), set(_declared("ggml_hip_ext.h"))
    # the drop-in core (lifecycle, pool, Seam 1 + invalidation + scope, Seam 2, resident weights, the two-phase
    # product) stays small and free of test hooks; everything else is the extension header's
    assert 30 <= len(core_decl) <= 40 and not any("debug" in n for n in core_decl), sorted(core_decl)
    ext_only = ext_decl - core_decl
    hip_decl = core_decl | ext_only
    mirror_decl = set(_declared("ggml.h")) - hip_decl
    assert len(hip_decl) >= 59 and len(mirror_decl) >= 30
    assert all(n.startswith("ggml_hip_") for n in hip_decl)
    assert set(_exported(_lib.LIB_PATH)) == hip_decl, set(_exported(_lib.LIB_PATH)) ^ hip_decl
    assert set(_exported(G.MIRROR_PATH)) == mirror_decl, set(_exported(G.MIRROR_PATH)) ^ mirror_decl
    for name in hip_decl:
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    for name in mirror_decl:
        assert hasattr(G.mirror(), name), f"{name} declared in tests/support/ggml.h but not exported"
    assert hip_decl == set(_lib.HIP_SYMBOLS), hip_decl ^ set(_lib.HIP_SYMBOLS)
    assert mirror_decl == set(G.MIRROR_SYMBOLS), mirror_decl ^ set(G.MIRROR_SYMBOLS)


def test_product_library_reads_no_environment_variable():
    """Developer A/B switches exist only in -DGGML_HIP_DEV builds: the product library does not import getenv."""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und


def test_struct_sizes_match_reference_layout():
    # TypeDefinitions.cs:65-99 (176 B), 102-121 (98 360 B); offsets from SURVEY.md 8(b)
    T = _lib.ggml_tensor
    assert C.sizeof(T) == 176
    assert (T.type.offset, T.n_dims.offset, T.ne.offset, T.nb.offset, T.op.offset, T.is_param.offset) == (0, 4, 8, 40, 72, 76)
    assert (T.grad.offset, T.src0.offset, T.src1.offset, T.opt.offset) == (80, 88, 96, 104)
    assert (T.n_tasks.offset, T.perf_runs.offset, T.perf_cycles.offset, T.perf_time_us.offset) == (136, 140, 144, 152)
    assert (T.data.offset, T.padding.offset) == (160, 168)
    assert C.sizeof(_lib.ggml_cgraph) == 98360


def test_type_tables():
    L, Mi = _lib.lib(), G.mirror()
    for t, s in _lib.TYPE_SIZE.items():
        assert L.ggml_hip_type_size(t) == s and L.ggml_hip_blck_size(t) == _lib.BLCK_SIZE[t]
        if t in (_lib.Q5_K, _lib.Q4_K, _lib.Q6_K):       # the k-quant extension is a device-level type only: the reference (and its mirror) cannot express it
            assert Mi.ggml_type_size(t) == 0 and Mi.ggml_blck_size(t) == 0
            continue
        assert Mi.ggml_type_size(t) == s and Mi.ggml_blck_size(t) == _lib.BLCK_SIZE[t]
    assert [Mi.ggml_is_quantized(t) for t in range(13)] == [0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0]


def test_test0_tensor_layout():
    """The reference's Test0 (Test0/Program.cs:18-38), same asserts."""
    ctx = G.ggml_init(128 * 1024 * 1024)
    assert ctx
    try:
        t1 = G.ggml_new_tensor_1d(ctx, G.F32, 10).contents
        t2 = G.ggml_new_tensor_2d(ctx, G.I16, 10, 20).contents
        t3 = G.ggml_new_tensor_3d(ctx, G.I32, 10, 20, 30).contents
        assert t1.n_dims == 1 and t1.ne[0] == 10 and t1.nb[1] == 10 * 4
        assert t2.n_dims == 2 and t2.ne[0] == 10 and t2.ne[1] == 20 and t2.nb[1] == 10 * 2 and t2.nb[2] == 10 * 20 * 2
        assert t3.n_dims == 3 and (t3.ne[0], t3.ne[1], t3.ne[2]) == (10, 20, 30)
        assert t3.nb[1] == 10 * 4 and t3.nb[2] == 10 * 20 * 4 and t3.nb[3] == 10 * 20 * 30 * 4
    finally:
        G.ggml_free(ctx)


def test_quantized_tensor_strides_and_pool_arithmetic():
    ctx = G.ggml_init(16 * 1024 * 1024)
    try:
        w = G.ggml_new_tensor_2d(ctx, G.Q4_0, 4096, 8)
        wc = w.contents
        assert wc.nb[0] == 20 and wc.nb[1] == 20 * 128 and wc.nb[2] == 20 * 128 * 8      # Ggml.cs:7856-7861
        assert G.ggml_nbytes(w) == 8 * 128 * 20
        assert wc.data % 16 == 0                                                          # GGML_MEM_ALIGN
        assert wc.data == C.addressof(wc) + 176                                           # data = result + 1, Ggml.cs:7839
        used0 = G.mirror().ggml_used_mem(ctx)
        assert used0 == 32 + 176 + 8 * 128 * 20                                           # object header + tensor + data
        q5 = G.ggml_new_tensor_2d(ctx, G.Q5_0, 64, 3).contents
        assert q5.nb[1] == 44 and C.addressof(q5) == C.addressof(wc) - 32 + used0 + 32    # next object follows
        # data size is rounded up to 16: 3 rows * 44 B = 132 -> 144
        assert G.mirror().ggml_used_mem(ctx) == used0 + 32 + 176 + 144
    finally:
        G.ggml_free(ctx)


def test_pool_exhaustion_returns_null_and_context_slots():
    ctx = G.ggml_init(1024)
    try:
        assert not G.ggml_new_tensor_1d(ctx, G.F32, 100000)   # Ggml.cs:7757-7763: prints and returns null
        assert G.ggml_new_tensor_1d(ctx, G.F32, 8)
    finally:
        G.ggml_free(ctx)
    ctxs = []
    try:
        for _ in range(64):
            c = G.ggml_init(256)
            assert c
            ctxs.append(c)
        assert not G.ggml_init(256)                            # all 64 slots used (Ggml.cs:1529-1536)
    finally:
        for c in ctxs:
            G.ggml_free(c)
    c = G.ggml_init(256)
    assert c
    G.ggml_free(c)


def test_mul_mat_node_construction_and_graph():
    ctx = G.ggml_init(8 * 1024 * 1024)
    try:
        a = G.ggml_new_tensor_3d(ctx, G.Q8_0, 64, 5, 2)
        b = G.ggml_new_tensor_3d(ctx, G.F32, 64, 7, 2)
        y = G.ggml_mul_mat(ctx, a, b)
        yc = y.contents
        assert yc.type == G.F32 and (yc.ne[0], yc.ne[1], yc.ne[2], yc.ne[3]) == (5, 7, 2, 1)   # Ggml.cs:8237
        assert yc.op == _lib.GGML_OP_MUL_MAT and C.addressof(yc.src0.contents) == C.addressof(a.contents)
        assert G.mirror().ggml_can_mul_mat(a, b) == 1
        assert not G.ggml_mul_mat(ctx, a, G.ggml_new_tensor_2d(ctx, G.F32, 64, 7))            # ne2 differs
        z = G.ggml_mul_mat(ctx, G.ggml_new_tensor_2d(ctx, G.F32, 5, 3), G.ggml_new_tensor_2d(ctx, G.F32, 5, 7))
        gf = G.ggml_build_forward(y)
        assert (gf.n_nodes, gf.n_leafs, gf.n_threads) == (1, 2, 4)                             # Ggml.cs:7659
        G.mirror().ggml_build_forward_expand(C.byref(gf), z)
        assert (gf.n_nodes, gf.n_leafs) == (2, 4)
        G.mirror().ggml_build_forward_expand(C.byref(gf), z)                                   # already visited
        assert (gf.n_nodes, gf.n_leafs) == (2, 4)
    finally:
        G.ggml_free(ctx)


def test_set_get_f32():
    ctx = G.ggml_init(1024 * 1024)
    try:
        t = G.ggml_new_tensor_2d(ctx, G.F32, 5, 3)
        G.ggml_set_f32(t, 2.5)
        assert all(G.ggml_get_f32_1d(t, i) == 2.5 for i in range(15))
        assert np.all(G.tensor_f32(t) == 2.5)
    finally:
        G.ggml_free(ctx)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: with no device the product path returns GGML_HIP_ERR_NO_DEVICE, it does not compute."""
    L = _lib.lib()
    assert L.ggml_hip_init(0) == _lib.ERR_NO_DEVICE
    assert b"no CPU path" in L.ggml_hip_last_error()
    ctx = G.ggml_init(1024 * 1024)
    try:
        a = G.ggml_new_tensor_2d(ctx, G.F32, 32, 4)
        b = G.ggml_new_tensor_2d(ctx, G.F32, 32, 2)
        G.ggml_set_f32(a, 1.0)
        G.ggml_set_f32(b, 1.0)
        y = G.ggml_mul_mat(ctx, a, b)
        G.ggml_set_f32(y, -1.0)
        gf = G.ggml_build_forward(y)
        with pytest.raises(_lib.GgmlHipError) as ei:
            G.ggml_graph_compute(ctx, gf)
        assert ei.value.status == _lib.ERR_NO_DEVICE
        assert G.ggml_get_f32_1d(y, 0) == -1.0   # untouched
        x = np.zeros(32, dtype=np.float32)
        out = np.zeros(20, dtype=np.uint8)
        assert L.ggml_hip_quantize_row(G.Q4_0, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 32) == _lib.ERR_NO_DEVICE
    finally:
        G.ggml_free(ctx)


def test_argument_validation_without_touching_a_device():
    L = _lib.lib()
    h = C.c_void_p()
    x = np.zeros(64, dtype=np.uint8)
    p = x.ctypes.data_as(C.c_void_p)
    assert L.ggml_hip_weight_upload(G.Q8_1, p, 32, 1, 44, 0, 1, None, C.byref(h)) == _lib.ERR_TYPE     # null slot (D8)
    assert L.ggml_hip_weight_upload(5, p, 32, 1, 12, 0, 1, None, C.byref(h)) == _lib.ERR_TYPE           # Q4_3
    assert L.ggml_hip_weight_upload(G.Q4_0, p, 48, 1, 30, 0, 1, None, C.byref(h)) == _lib.ERR_SHAPE     # K % 32
    assert L.ggml_hip_weight_upload(G.Q4_0, p, 64, 1, 20, 0, 1, None, C.byref(h)) == _lib.ERR_SHAPE     # nb01 < row
    assert L.ggml_hip_weight_upload(G.Q4_0, None, 64, 1, 40, 0, 1, None, C.byref(h)) == _lib.ERR_ARG
    assert L.ggml_hip_mul_mat_work_size(G.Q4_0, 4096, 500) == 128 * 4 * 512 * 16 + 2 * 128 * 512 * 4   # rows padded to 256
    assert L.ggml_hip_mul_mat_work_size(G.F32, 4096, 256) == 0                                          # Ggml.cs:3360-3364
    assert L.ggml_hip_mul_mat_work_size(G.F32, 4096, 300) == 4096 * 512 * 6                             # (above 256 rows: src1 as three bf16 pieces)


def test_reference_style_c_program_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    """include/*.h are valid C99 and the C-ABI links from plain C: tests/c/reference_style_program.c (the Test1-shaped
    program of INTEGRATION.md 3).  Without a GPU ggml_graph_compute must report NO_DEVICE -- no CPU fallback."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir, supdir = os.path.join(root, "ggmlsharp_amd", "lib"), os.path.join(root, "tests", "support")
    exe = str(tmp_path / "refprog")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"), "-I" + supdir,
                           os.path.join(root, "tests", "c", "reference_style_program.c"), "-L" + supdir, "-L" + libdir, "-lggml_hostmirror", "-lggml_hip", "-Wl,-rpath," + supdir,
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    import torch
    if not torch.cuda.is_available():
        assert out.stdout.strip() == "no-device"
    else:
        assert out.stdout.startswith("ok ")


def test_product_package_loads_without_the_test_support_mirror(tmp_path):
    """the product's Python binding may not depend on tests/support -- a fresh interpreter with the mirror
    library hidden imports ggmlsharp_amd, loads libggml_hip.so and resolves every declared symbol."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import ggmlsharp_amd as g; L = g.lib(); "
            "[getattr(L, n) for n in g._lib.SYMBOLS]; "
            "import ctypes; "
            "maps = open('/proc/self/maps').read(); assert 'libggml_hip.so' in maps and 'hostmirror' not in maps; "
            "assert not hasattr(g._lib, 'MIRROR_SYMBOLS') and not os.path.exists(os.path.join(os.path.dirname(g.__file__), 'ggml.py')); print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
