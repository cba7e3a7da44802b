import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gptqmodel_amd import ops
from helpers import synth_gptq, f32_to_torch
print(ops.device_info(0))
for (K,N,M) in [(4096,4096,1),(4096,14336,1),(14336,4096,1),(4096,28672,1),(4096,6144,1),(4096,4096,16),(4096,4096,64)]:
    qw,qz,sc,g = synth_gptq(1,4,K,N,128)
    qw=torch.from_numpy(qw).cuda(); qz=torch.from_numpy(qz).cuda(); sc=f32_to_torch(sc,"fp16","cuda")
    x=torch.randn(M,K,device="cuda",dtype=torch.float16)
    qw_t, meta = ops.repack_tiled(qw,qz,sc,None,128,4)
    for split in [0,1,2,4,8,16]:
        ops.set_tuning(split,0,0)
        try:
            for _ in range(5): ops.gemm(x,qw_t,meta,None,None,N,128,4,sc.dtype)
        except RuntimeError as e:
            print("err", e); continue
        torch.cuda.synchronize()
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): ops.gemm(x,qw_t,meta,None,None,N,128,4,sc.dtype)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e)*1000/50
        byts = K*N/2 + (K//128)*N*2.5 + M*(K+N)*2
        print(f"K={K} N={N} M={M} split={split}: {us:.2f} us  {byts/us/1e6:.2f} TB/s")
ops.set_tuning(0,0,0)
