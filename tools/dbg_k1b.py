import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
from ggmlsharp_amd._lib import lib, check
device.init(0)
L = lib()
N, K = 2000, 2048
nbk, Npad = K // 32, 2048
nba = (nbk + 3) // 4 * 4
x = torch.randn((N, K), device="cuda") * 2
q8 = device.quantize_rows(8, x).view(N, nbk, 36)
ref_q = q8[:, :, 4:].contiguous().view(torch.int8).to(torch.int32)          # [N][nbk][32]
MAG = torch.tensor([0] * 12 + [1, 99, 99, 99, 2, 99, 3, 99, 4, 5, 6, 7, 8] + [99] * 7, device="cuda")


def decode(work):
    img = work[: nba * 48 * Npad].view(nba, 48 * Npad)
    p16 = img[:, : 32 * Npad].view(nba, 2, Npad, 16); p8 = img[:, 32 * Npad:].view(nba, 2, Npad, 8)
    frag = torch.cat([p16, p8], dim=-1)[:, :, :N, :].to(torch.int64)        # [nba][2][N][24] bytes
    bits = ((frag.unsqueeze(-1) >> torch.arange(8, device="cuda")) & 1).view(nba, 2, N, 192)
    code = (bits.view(nba, 2, N, 32, 6) * (1 << torch.arange(6, device="cuda"))).sum(-1)
    mag = MAG[code & 31]
    dig = torch.where((code & 32) != 0, -mag, mag)
    return (16 * dig[:, 0] + dig[:, 1]).permute(1, 0, 2)                     # [N][nba][32]


for it in range(int(sys.argv[1])):
    work = device.alloc_work(2, K, N)
    work.zero_()
    junk = torch.randn((1 + it % 5) * 300000, device="cuda").sum()
    check(L.ggml_hip_quantize_act_dev(C.c_void_p(x.data_ptr()), N, K, K, C.c_void_p(work.data_ptr()), work.numel(), 3, None), "q")
    torch.cuda.synchronize()
    q = decode(work)
    bad = (q[:, :nbk] != ref_q)
    if bad.any():
        idx = bad.any(dim=2).nonzero()
        rows = sorted(set(idx[:, 0].tolist())); blks = sorted(set(idx[:, 1].tolist()))
        print(f"iter {it}: {int(bad.sum())} wrong quants; rows {rows[:12]} ({len(rows)}), k-blocks {blks[:12]} ({len(blks)})", flush=True)
        r, b = int(idx[0, 0]), int(idx[0, 1])
        print("   got ", q[r, b].tolist()); print("   want", ref_q[r, b].tolist())
print("done")
