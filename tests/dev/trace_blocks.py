"""dev: per-block timeline of the decode kernel: which CU ran which block, when.  Needs the trace hooks (GPTQHIP_ABLATE & 16:
gptqhip_dev_set_trace + wall_clock64 / HW_ID stamps) that lived in csrc/gptqhip_skinny.hip at commit 1701c9f ("Experiment (kept in
history): decode1_kernel"); profiles/r02_decode_block_trace.txt is its output."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, numpy as np
import gptqmodel_amd._lib as L
L.LIB_PATH = sys.argv[1]
import bench as B
from gptqmodel_amd import ops
lib = L.load()
lib.gptqhip_dev_set_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
for name, K, N in [("gate_up", 4096, 28672), ("down", 14336, 4096), ("o", 4096, 4096)]:
    NL = 6
    lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
    xin = (torch.randn(K, device=dev, generator=gen) * 0.5).to(dtype)
    outs = [torch.empty(N, dtype=dtype, device=dev) for _ in range(NL)]
    plain = [ops.make_decode_op(xin, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype) for l, o in zip(lins, outs)]
    nblk = 8192
    tr = torch.zeros((nblk, 4), dtype=torch.int64, device=dev)
    for d in plain: ops.launch_decode_op(d, dev)      # warm
    torch.cuda.synchronize()
    lib.gptqhip_dev_set_trace(tr.data_ptr())
    torch.cuda.synchronize()
    ops.launch_decode_op(plain[0], dev)                # cold weights? (they were touched above; 6 layers x 58 MB > L2, MALL 256 MB may hold) 
    torch.cuda.synchronize()
    lib.gptqhip_dev_set_trace(0)
    t = tr.cpu().numpy()
    used = t[:, 1] != 0
    t = t[used]
    n = len(t)
    t0 = t[:, 0] - t[:, 0].min(); t1 = t[:, 1] - t[:, 0].min()
    hw = t[:, 2]; xcc = t[:, 3] & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = xcc * 1000 + se * 100 + sh * 10 + cu
    print(f"== {name}: {n} blocks, kernel span {t1.max()/100:.2f} us (100 MHz ticks)")
    print("   distinct CUs", len(set(cuid.tolist())), " XCC counts", np.bincount(xcc.astype(int)).tolist())
    qs = [0, 10, 25, 50, 75, 90, 100]
    print("   start pct (us)", [round(float(np.percentile(t0, q)) / 100, 2) for q in qs])
    print("   end   pct (us)", [round(float(np.percentile(t1, q)) / 100, 2) for q in qs])
    print("   dur   pct (us)", [round(float(np.percentile(t1 - t0, q)) / 100, 2) for q in qs])
    # one CU's blocks
    ids = np.nonzero(used)[0]
    for target in sorted(set(cuid.tolist()))[:3]:
        sel = cuid == target
        print(f"   CU {target}: blocks", ids[sel].tolist()[:12], "start", (t0[sel] / 100).round(2).tolist()[:12], "end", (t1[sel] / 100).round(2).tolist()[:12])
    # block id vs start time correlation in chunks of 256 ids
    for lo in range(0, n, max(256, n // 8)):
        hi = min(n, lo + 256)
        print(f"   blocks {ids[lo]}..{ids[hi-1]}: start {t0[lo:hi].mean()/100:.2f} end {t1[lo:hi].mean()/100:.2f} CUs {len(set(cuid[lo:hi].tolist()))}")
    del lins, plain, outs
    torch.cuda.empty_cache()
