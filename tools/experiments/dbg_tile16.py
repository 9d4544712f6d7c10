import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ggmlsharp_amd import device
device.init(0)
t, K, N = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = torch.randn((16384, K), generator=g, device="cuda"); x = torch.randn((N, K), generator=g, device="cuda")
rows = device.quantize_rows(t, w)
W = device.Weight.from_device(t, rows, K); full = device.mul_mat(W, x)
Ws = device.Weight.from_device(t, rows, K, row_begin=0, row_end=4096); part = device.mul_mat(Ws, x)
eq = (part == full[:, :4096])
print("equal fraction", eq.float().mean().item())
print("by n:", [round(eq[n].float().mean().item(), 2) for n in range(N)])
print("by m % 32:", [round(eq[:, m::32].float().mean().item(), 2) for m in range(32)])
# is part a permutation of full?  compare sorted values of row 0
print("part[0,:8]", part[0, :8].tolist()); print("full[0,:8]", full[0, :8].tolist())
wd = device.dequantize_rows(t, rows[:64], K).double(); xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
ref = xq @ wd.T
print("ref [0,:8]", ref[0, :8].tolist())
# block-level reference: contributions of even / odd blocks
wb = wd.view(64, K // 32, 32); xb = xq.view(N, K // 32, 32)
ev = torch.einsum('nbk,mbk->nm', xb[:, 0::2], wb[:, 0::2]); od = torch.einsum('nbk,mbk->nm', xb[:, 1::2], wb[:, 1::2])
print("even-block part [0,:4]", ev[0, :4].tolist(), "odd", od[0, :4].tolist())
# candidate formulas for part[0, :4]
nb = K // 32
rb = rows[:64].view(64, nb, 20)
dw = rb[:, :, :4].contiguous().view(torch.float32).view(64, nb).double()
qs = rb[:, :, 4:].to(torch.int32)
wq = torch.stack([(qs & 15) - 8, (qs >> 4) - 8], dim=-1).view(64, nb, 32).double()      # element 2j = low nibble, 2j+1 = high
xr = device.quantize_rows(8, x.contiguous()).view(N, nb, 36)
da = xr[:, :, :4].contiguous().view(torch.float32).view(N, nb).double()
xa = xr[:, :, 4:].view(torch.int8).double()
sumi = torch.einsum('nbk,mbk->nmb', xa, wq)          # [N][64][nb]
ah = torch.floor((xa + 8) / 16); al = xa - 16 * ah
s_h = torch.einsum('nbk,mbk->nmb', ah, wq); s_l = torch.einsum('nbk,mbk->nmb', al, wq)
def show(name, v): print(f"{name:40s}", [round(float(a), 3) for a in v[0, :4]])
show("ref", (sumi * da[:, None, :] * dw[None]).sum(-1))
swd = dw.view(64, nb // 2, 2).flip(-1).reshape(64, nb); swa = da.view(N, nb // 2, 2).flip(-1).reshape(N, nb)
show("d of the pair swapped", (sumi * da[:, None, :] * swd[None]).sum(-1))
show("da of the pair swapped", (sumi * swa[:, None, :] * dw[None]).sum(-1))
show("both swapped", (sumi * swa[:, None, :] * swd[None]).sum(-1))
show("no x16 on ah", ((s_h + s_l) * da[:, None, :] * dw[None]).sum(-1))
show("x16 on al instead", ((s_h + 16 * s_l) * da[:, None, :] * dw[None]).sum(-1))
show("pair sums merged (both blocks in each)", ((sumi.view(N, 64, nb // 2, 2).sum(-1, keepdim=True).expand(-1, -1, -1, 2).reshape(N, 64, nb)) * da[:, None, :] * dw[None]).sum(-1))
sw_sumi = sumi.view(N, 64, nb // 2, 2).flip(-1).reshape(N, 64, nb)
show("sumi of the pair swapped", (sw_sumi * da[:, None, :] * dw[None]).sum(-1))
show("part", part.double())
if os.environ.get("K3S16_DEBUG"):
    which = int(os.environ["K3S16_DEBUG"]) - 1
    got = part[:, :64].double()
    want = sumi[:, :, which]
    print("debug block", which, "equal fraction", (got == want).double().mean().item())
    print("got ", got[:4, :6].tolist()); print("want", want[:4, :6].tolist())
    print("sum_h", s_h[:4, :6, which].tolist()); print("sum_l", s_l[:4, :6, which].tolist())
    other = sumi[:, :, 1 - which]
    print("other block", other[:2, :6].tolist())
    dbg = int(os.environ["K3S16_DEBUG"])
    if dbg == 3: print("d0 got", part[:2, :6].tolist(), "want", dw[:6, 0].tolist())
    if dbg == 4: print("d1 got", part[:2, :6].tolist(), "want", dw[:6, 1].tolist())
    if dbg == 5: print("da0 got", part[:6, :2].tolist(), "want", da[:6, 0].tolist(), " n=16..18", part[16:19, 0].tolist(), da[16:19, 0].tolist())
    if dbg == 6: print("da1 got", part[:6, :2].tolist(), "want", da[:6, 1].tolist())
