import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
from ggmlsharp_amd import device
device.init(0)
for L in (1, 2, 4, 8):
    r = bench.dropin_decode_layer(1, 100, layers=L)
    print(L, "layers:", r["us_per_graph"], "us per graph,", r["us_per_layer"], "us per layer, nodes", r["nodes"], r["named_scopes"], "hbm_frac", r["hbm_frac"], flush=True)
