import sys, os, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from gptqmodel_amd.utils.model import fuse_gate_up_interleaved, fuse_quant_linears
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
def raw(k, n):
    m = HipGptqLinear(bits=4, group_size=128, sym=True, desc_act=False, in_features=k, out_features=n, bias=False, register_buffers=False)
    m.qweight = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=gen)
    m.qzeros = torch.full((k // 128, n // 8), -2004318072, dtype=torch.int32, device=dev)
    m.scales = (torch.rand((k // 128, n), device=dev, generator=gen) * 0.01 + 0.005).half()
    m.g_idx = torch.arange(k, device=dev, dtype=torch.int32) // 128
    m.bias = None; m.qzero_format(format=2); m.eval()
    return m
mb = lambda: torch.cuda.memory_allocated() / 2**20
for mode in ("interleaved", "concat"):
    gc.collect(); torch.cuda.empty_cache()
    m0 = mb()
    g, u = raw(4096, 14336), raw(4096, 14336)
    m1 = mb()
    f = fuse_gate_up_interleaved(g, u) if mode == "interleaved" else fuse_quant_linears([g, u])
    m2 = mb()
    del g, u; gc.collect()
    m3 = mb()
    f.post_init(); torch.cuda.synchronize(); gc.collect()
    m4 = mb()
    print(f"{mode}: raw pair {m1-m0:.0f} MB, after fuse {m2-m0:.0f}, after del sources {m3-m0:.0f}, after post_init {m4-m0:.0f}")
    print("   buffers:", [(n, tuple(b.shape), str(b.dtype)) for n, b in f.named_buffers()])
    del f
