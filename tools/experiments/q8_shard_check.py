"""developer experiment: the shard / unsplit comparison of tests/test_gpu_fullsize.py::test_batched_decode_form_follows_M_only_in_its_geometry
for Q8_0, run after other shapes in the same process, with where and by how much the two differ"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device
device.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def make(t, M, K, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    w = torch.randn((M, K), generator=g, device="cuda"); x = torch.randn((N, K), generator=g, device="cuda") * 2
    return device.quantize_rows(t, w), x


for (t, M, K) in ((8, 8492, 2048), (8, 300, 4096), (8, 130, 2112), (8, 8492, 4160), (8, 8492, 4160), (8, 200, 11008)):
    rows, x = make(t, M, K, 11 + t + N)
    W = device.Weight.from_device(t, rows, K)
    full = device.mul_mat(W, x)
    again = device.mul_mat(W, x)
    print((t, M, K, N), "unsplit twice equal:", torch.equal(full, again), flush=True)
    for (r0, r1) in ((0, 2048), (M - 77, M), (0, min(M, 8192)), (31, 290)):
        if r1 > M:
            continue
        Ws = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
        a = device.mul_mat(Ws, x); b = full[:, r0:r1]
        d = (a != b)
        if d.any():
            nz = d.nonzero()
            print("   ", (r0, r1), f"differ in {int(d.sum())} of {a.numel()}: n {sorted(set(nz[:,0].tolist()))[:10]} m {sorted(set(nz[:,1].tolist()))[:6]}..{max(nz[:,1].tolist())} max |diff| {(a-b).abs().max().item():.3e}; shard again equal: {torch.equal(a, device.mul_mat(Ws, x))}; unsplit again equals shard: {torch.equal(device.mul_mat(W, x)[:, r0:r1], a)}")
        else:
            print("   ", (r0, r1), "equal")
        Ws.free()
    W.free()
