"""developer experiment: Q8_0 batched-decode sizes, weights cold: the stage-free int8 form (force 0) against the staged f16 form (force 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ggmlsharp_amd import device
from ggmlsharp_amd._lib import lib
device.init(0)
for force in (0, 2):
    lib().ggml_hip_debug_force_gemm(force)
    for (M, K, N) in ((4096, 4096, 8), (4096, 4096, 32), (4096, 4096, 64), (11008, 4096, 32), (32000, 4096, 32), (4096, 2048, 32), (4096, 11008, 32), (4096, 11008, 64)):
        r = bench.side_config(device, M, K, N, copies=16 if M * K <= 4096 * 11008 else 6, iters=40, qtype=8)
        print(f"force {force}: Q8_0 {M} x {K} x {N}: call {r['ms_per_step'] * 1e3:7.2f} us, compute kernel {(r.get('compute_kernel_ms') or 0) * 1e3:7.2f} us", flush=True)
