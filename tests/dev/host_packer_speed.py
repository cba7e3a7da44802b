"""CPU timing of the threaded C++ host packer (gptqhip_pack_gptq_host) next to the reference's own pack_block on the same layer
(the reference's tests/test_packing_speed.py idea).  Runs in the build container: needs /root/reference (or the
oracle/_ref snapshot) for the reference leg.  python tests/dev/host_packer_speed.py > profiles/r04_host_packer_speed.txt"""
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gptqmodel_amd import ops  # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    ref = load_reference()
    print(f"host cores: {os.cpu_count()}  torch threads: {torch.get_num_threads()}")
    for bits, k, n, gs in ((4, 4096, 4096, 128), (8, 4096, 4096, 128), (3, 4096, 4096, 128), (4, 14336, 4096, 128)):
        torch.manual_seed(bits + k)
        lin = nn.Linear(k, n, bias=False)
        g = k // gs
        scales = torch.rand(n, g) * 0.01 + 0.005
        zeros = torch.randint(0, 1 << bits, (n, g)).float()
        g_idx = torch.arange(k, dtype=torch.int32) // gs
        mod = ref.TorchLinear(bits=bits, group_size=gs, sym=False, desc_act=False, in_features=k, out_features=n, bias=False,
                              register_buffers=False)
        t_ref = best(lambda: mod.pack_block(lin, scales.clone(), zeros.clone(), g_idx.clone()), n=2)
        w, s, z = lin.weight.detach(), scales.T.contiguous(), zeros.T.contiguous().to(torch.int32)
        out = {}
        for threads in (1, 8, 32, 0):
            out[threads] = best(lambda: out.__setitem__("r", ops.pack_gptq_host(w, s, z, g_idx, bits, threads)))
        same = torch.equal(out["r"][0], mod.qweight) and torch.equal(out["r"][1], mod.qzeros)
        print(f"bits={bits} {k}x{n} g{gs}: reference pack_block {t_ref:8.1f} ms | gptqhip_pack_gptq_host "
              + "  ".join(f"{('all' if t == 0 else t)} thr {out[t]:7.1f} ms" for t in (1, 8, 32, 0)) + f" | bit-exact: {same}")


if __name__ == "__main__":
    main()
