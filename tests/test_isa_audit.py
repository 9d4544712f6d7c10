"""Static check of the shipped ISA: no inline-asm statement may touch an MFMA's result registers before the matrix pipe has
written them (hipcc pads only instructions it generated itself).  Compiles the two kernels that mix
MFMAs with inline-asm VALU work to gfx950 assembly (device only, ~20 s each) and runs tools/mfma_hazard_audit.py on it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["gemm_qmx.hip", "gemm_q16.hip"])
def test_no_inline_asm_touches_an_mfma_result_early(src):
    import mfma_hazard_audit as A
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "isa_stats.sh"), src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "kernel<" in l]
    assert len(lines) >= 20, r.stdout[-2000:]                    # every instantiation was compiled and listed
    bad = A.audit(os.path.join(ROOT, "tools", "bin", src.replace(".hip", ".s")))
    assert not bad, "\n".join(f"{k} {op}: D touched after {s} slots (need {n}): {t}" for k, op, s, n, t, _ in bad)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,least", [("gemm_qmx.hip", 11), ("dense16.hip", 8)])
def test_relaxed_stage_drains_leave_only_loads_behind_the_last_dma_piece(src, least):
    """`s_waitcnt vmcnt(N)`, N > 0, written as inline asm = "the next stage's LDS-DMA pieces have landed": at least N vector-memory
    instructions must stand between the last `buffer_load ... lds` and the wait in the code hipcc emitted (tools/drain_audit.py)."""
    import drain_audit as D
    s_path = os.path.join(ROOT, "tools", "bin", src.replace(".hip", ".s"))
    if not os.path.exists(s_path) or os.path.getmtime(s_path) < os.path.getmtime(os.path.join(ROOT, "ggmlsharp_amd", "csrc", src)):
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "isa_stats.sh"), src], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    res = D.audit(s_path)
    assert len(res) >= least, res                                   # every relaxed drain was found
    short = [(k, n, c) for k, n, c in res if c is None or c < n]
    assert not short, short
