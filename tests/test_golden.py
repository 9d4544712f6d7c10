"""Committed fixtures (tests/golden/qpath_v1.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them through the C-ABI (GPU)."""
import os

import numpy as np
import pytest

import oracle_lib as O

_HERE = os.path.dirname(os.path.abspath(__file__))
G = dict(np.load(os.path.join(_HERE, "golden", "qpath_v1.npz")))
G.update(np.load(os.path.join(_HERE, "golden", "qpath_d7_v1.npz")))      # Q4_2 / Q5_1 (intent) on the same inputs
QT = (("q4_0", O.Q4_0), ("q4_1", O.Q4_1), ("q5_0", O.Q5_0), ("q8_0", O.Q8_0), ("q4_2", O.Q4_2), ("q5_1", O.Q5_1))


def test_oracle_reproduces_golden():
    x, K = G["x"], G["x"].shape[1]
    for name, t in QT + (("q8_1", O.Q8_1),):
        assert np.array_equal(O.quantize_row(t, x), G[f"quant_{name}"])
    for name, t in QT:
        assert np.array_equal(O.dequantize_row(t, G[f"quant_{name}"], K).view(np.uint32), G[f"dequant_{name}"].view(np.uint32))
        wq = O.quantize_row(t, G["w"])
        got = O.mul_mat(t, wq, G["a"], G["w"].shape[0], K, G["a"].shape[0], nth=3)[0, 0]
        assert np.array_equal(got.view(np.uint32), G[f"mulmat_{name}"].view(np.uint32))
    assert np.array_equal(O.mul_mat(O.F32, G["w"], G["a"], 12, K, 5)[0, 0], G["mulmat_f32"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    torch = pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    x, K = G["x"], G["x"].shape[1]
    for name, t in QT + (("q8_1", O.Q8_1),):
        got = device.quantize_rows(t, torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.array_equal(got, G[f"quant_{name}"]), name
    for name, t in QT:
        got = device.dequantize_rows(t, torch.from_numpy(G[f"quant_{name}"]).cuda(), K).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), G[f"dequant_{name}"].view(np.uint32)), name
        W = device.Weight.from_host(t, O.quantize_row(t, G["w"]), K)
        got = device.mul_mat(W, torch.from_numpy(G["a"]).cuda()).cpu().numpy()
        ref = G[f"mulmat_{name}"].astype(np.float64)
        O.assert_mul_mat_close(got, ref, K, name)
    for name, t, wraw in (("f32", O.F32, G["w"].view(np.uint8)), ("f16", O.F16, G["w"].astype(np.float16).view(np.uint8))):
        W = device.Weight.from_host(t, wraw, K)
        got = device.mul_mat(W, torch.from_numpy(G["a"]).cuda()).cpu().numpy()
        ref = G[f"mulmat_{name}"].astype(np.float64)
        O.assert_mul_mat_close(got, ref, K, name)
