#!/usr/bin/env python3
"""Regenerates tests/golden/qpath_v1.npz from the CPU oracle (oracle/ggml_oracle.c).

The reference (C#, net8.0) cannot be run in the build image and its own tests hold no vector for the quantized path
(SURVEY.md 8(c)), so these fixtures are NOT reference outputs: they freeze the oracle's outputs (which are pinned by
the hand-derived KATs and the independent numpy restatement in tests/test_oracle.py) so that later changes to either
the oracle or the HIP path are caught.  Data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    rng = np.random.default_rng(20240930)
    out = {}
    K = 128
    x = (rng.standard_normal((6, K)) * np.array([1, 1e-3, 40, 1, 1, 1])[:, None]).astype(np.float32)
    x[3, :32] = 0.0
    x[4, :64] = np.round(x[4, :64] * 4) / 2
    x[5, 32:64] = -np.abs(x[5, 32:64])
    out["x"] = x
    for name, t in (("q4_0", O.Q4_0), ("q4_1", O.Q4_1), ("q5_0", O.Q5_0), ("q8_0", O.Q8_0), ("q8_1", O.Q8_1)):
        q = O.quantize_row(t, x)
        out[f"quant_{name}"] = q
        if t != O.Q8_1:
            out[f"dequant_{name}"] = O.dequantize_row(t, q, K)
    M, N = 12, 5
    w = rng.standard_normal((M, K)).astype(np.float32)
    a = (rng.standard_normal((N, K)) * 2).astype(np.float32)
    out["w"], out["a"] = w, a
    for name, t in (("q4_0", O.Q4_0), ("q4_1", O.Q4_1), ("q5_0", O.Q5_0), ("q8_0", O.Q8_0)):
        wq = O.quantize_row(t, w)
        out[f"mulmat_{name}"] = O.mul_mat(t, wq, a, M, K, N)[0, 0]
    out["mulmat_f32"] = O.mul_mat(O.F32, w, a, M, K, N)[0, 0]
    out["mulmat_f16"] = O.mul_mat(O.F16, w.astype(np.float16).view(np.uint16), a, M, K, N)[0, 0]
    np.savez_compressed(os.path.join(HERE, "qpath_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "qpath_v1.npz"), {k: v.shape for k, v in out.items()})
    # Q4_2 / Q5_1 (built to the intent): same inputs, their own file so that v1 stays byte-stable
    d7 = {}
    for name, t in (("q4_2", O.Q4_2), ("q5_1", O.Q5_1)):
        q = O.quantize_row(t, x)
        d7[f"quant_{name}"] = q
        d7[f"dequant_{name}"] = O.dequantize_row(t, q, K)
        d7[f"mulmat_{name}"] = O.mul_mat(t, O.quantize_row(t, w), a, M, K, N)[0, 0]
    np.savez_compressed(os.path.join(HERE, "qpath_d7_v1.npz"), **d7)
    print("wrote", os.path.join(HERE, "qpath_d7_v1.npz"), {k: v.shape for k, v in d7.items()})


if __name__ == "__main__":
    main()
