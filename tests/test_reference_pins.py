"""Consumer of integration/PinEmitter's output.

The reference is C# (net8.0) and cannot be built or run in this image or on the GPU boxes (no .NET toolchain), so the oracle's
parity is UNPINNED by the reference.  integration/PinEmitter is a small console program that, on any machine
with the .NET 8 SDK, runs the UNTOUCHED reference and prints the inputs and outputs of the functions of this path that are
correct as written -- scalar quantize_row_q4_0_reference, scalar dequantize_row_q4_0 / _q4_1 / _q5_0, f32 mul_mat -- as
tests/golden/reference_pins_v1.json.  When that file is present these tests compare the oracle (and, with -m gpu, the HIP path)
with it bit for bit; until then they skip, and the consumer itself is exercised on a file of the same format written by the
oracle (so the day the real file arrives the only thing that can fail is parity)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = os.path.join(HERE, "golden", "reference_pins_v1.json")
FNS = {"quantize_row_q4_0_reference": ("q", O.Q4_0), "dequantize_row_q4_0": ("d", O.Q4_0), "dequantize_row_q4_1": ("d", O.Q4_1),
       "dequantize_row_q5_0": ("d", O.Q5_0)}


def _b(hexstr):
    return np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint8)


def check_pins(doc, quantize=O.quantize_row, dequantize=O.dequantize_row, mul_mat=None):
    """every case of a pins document against the given implementations; returns the number of cases checked"""
    assert doc["format"] == "ggmlsharp-reference-pins-v1"
    n = 0
    for c in doc["cases"]:
        fn = c["fn"]
        if fn in FNS:
            kind, t = FNS[fn]
            k = int(c["k"])
            if kind == "q":
                x = _b(c["input"]).view(np.float32).reshape(1, k)
                got = np.asarray(quantize(t, x)).reshape(-1)
            else:
                blk = _b(c["input"]).reshape(1, -1)
                got = np.asarray(dequantize(t, blk, k), dtype=np.float32).reshape(-1).view(np.uint8)
            assert np.array_equal(got, _b(c["output"])), f"{fn}: case {n} differs from the reference's bytes"
        elif fn == "mul_mat_f32":
            M, K, N = int(c["M"]), int(c["K"]), int(c["N"])
            w = _b(c["w"]).view(np.float32).reshape(M, K)
            x = _b(c["x"]).view(np.float32).reshape(N, K)
            ref = _b(c["output"]).view(np.float32).reshape(N, M)
            got = (mul_mat or (lambda w_, x_: O.mul_mat(O.F32, w_, x_, M, K, N)[0, 0]))(w, x)
            # ggml_vec_dot_f32 (Ggml.cs:2631-2640): f32 products summed in f64, rounded once -- the same value whatever the order
            # of the f64 additions up to 1 ulp of f32; bit-exact is expected, 2 ulp is the bar
            assert np.all(np.abs(np.asarray(got, np.float64) - ref) <= 2 * np.spacing(np.abs(ref)).astype(np.float64) + 1e-30), "mul_mat_f32"
        else:
            raise AssertionError(f"unknown pinned function {fn}")
        n += 1
    return n


def _oracle_written_document():
    """a pins document of the real format whose outputs come from the oracle (consumer self-test only -- pins nothing)"""
    rng = np.random.default_rng(5)
    cases = []
    k = 256
    for _ in range(3):
        x = rng.standard_normal((1, k)).astype(np.float32)
        cases.append({"fn": "quantize_row_q4_0_reference", "k": k, "input": x.tobytes().hex(), "output": O.quantize_row(O.Q4_0, x).tobytes().hex()})
    for name, (_, t) in FNS.items():
        if name.startswith("dequantize"):
            blk = O.quantize_row(t, rng.standard_normal((1, k)).astype(np.float32))
            cases.append({"fn": name, "k": k, "input": blk.tobytes().hex(), "output": O.dequantize_row(t, blk, k).astype(np.float32).tobytes().hex()})
    M, K, N = 8, 64, 5
    w, x = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((N, K)).astype(np.float32)
    cases.append({"fn": "mul_mat_f32", "M": M, "K": K, "N": N, "w": w.tobytes().hex(), "x": x.tobytes().hex(),
                  "output": O.mul_mat(O.F32, w, x, M, K, N)[0, 0].astype(np.float32).tobytes().hex()})
    return {"format": "ggmlsharp-reference-pins-v1", "runtime": "oracle (self-test)", "cases": cases}


def test_consumer_accepts_a_well_formed_document_and_rejects_a_wrong_byte():
    doc = _oracle_written_document()
    assert check_pins(json.loads(json.dumps(doc))) == len(doc["cases"])
    bad = json.loads(json.dumps(doc))
    out = bytearray(bytes.fromhex(bad["cases"][0]["output"]))
    out[5] ^= 0x10
    bad["cases"][0]["output"] = bytes(out).hex()
    with pytest.raises(AssertionError):
        check_pins(bad)


def test_emitter_and_binding_sources_are_in_the_tree():
    root = os.path.dirname(HERE)
    for f in ("integration/GgmlHip.cs", "integration/GGMLSharp.Hip.targets", "integration/Ggml.cs.hip.patch", "integration/PinEmitter/Program.cs",
              "integration/PinEmitter/PinEmitter.csproj"):
        assert os.path.isfile(os.path.join(root, f)), f
    cs = open(os.path.join(root, "integration", "GgmlHip.cs")).read()
    # every seam the patch calls is declared in the binding file
    for name in ("ggml_hip_compute_forward_mul_mat", "ggml_hip_register_host_pool", "ggml_hip_invalidate_range", "ggml_hip_graph_begin", "ggml_hip_graph_end"):
        assert name in cs, name
    prog = open(os.path.join(root, "integration", "PinEmitter", "Program.cs")).read()
    for fn in list(FNS) + ["mul_mat_f32", "ggmlsharp-reference-pins-v1"]:
        assert fn in prog, fn


@pytest.mark.skipif(not os.path.isfile(PINS), reason="tests/golden/reference_pins_v1.json absent: no .NET run of the reference exists yet (parity unpinned)")
def test_oracle_matches_the_reference_pins():
    assert check_pins(json.load(open(PINS))) > 0


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(PINS), reason="tests/golden/reference_pins_v1.json absent: no .NET run of the reference exists yet (parity unpinned)")
def test_hip_path_matches_the_reference_pins():
    torch = pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)

    def q(t, x):
        return device.quantize_rows(t, torch.from_numpy(np.ascontiguousarray(x)).cuda()).cpu().numpy()

    def d(t, blk, k):
        return device.dequantize_rows(t, torch.from_numpy(np.ascontiguousarray(blk)).cuda(), k).cpu().numpy()

    def mm(w, x):
        W = device.Weight.from_host(O.F32, w.view(np.uint8).reshape(w.shape[0], -1), w.shape[1])
        return device.mul_mat(W, torch.from_numpy(x).cuda()).cpu().numpy()
    doc = json.load(open(PINS))
    doc["cases"] = [c for c in doc["cases"] if c["fn"] != "mul_mat_f32"]        # (the dense f32 kernel sums in f32: ~1e-6, tested elsewhere)
    assert check_pins(doc, quantize=q, dequantize=d) > 0
