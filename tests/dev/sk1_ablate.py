"""dev (GPU box): timing ablations of skinny1_kernel (tests/dev/sk1_ablate_build.sh): us per launch of the four Llama-3-8B op shapes under graph replay with
parts of the kernel switched off (results wrong by construction).  The persistent gate_up variant is switched off (GPTQHIP_NO_PERSIST=1) so that all
four shapes run skinny1_kernel.  argv: comma-separated masks (default 0,1,2,4,6,7,15)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    mask = os.environ.get("SK1_ABLATE", "0")
    if mask != "0":
        os.environ["GPTQHIP_LIB"] = os.path.join(HERE, "ablate", "libgptqhip_sk1abl%s.so" % mask)
    import bench
    from gptqmodel_amd import ops
    from gptqmodel_amd.utils.decode_chain import DecodeStep
    dtype, dev = torch.float16, torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    cfg = bench.LLAMA3_8B
    layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=16)
    step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
    stream = torch.cuda.Stream()
    per = []
    for j in range(4):
        sel = step.ops[j::4]
        def fn():
            for op in sel:
                ops.launch_decode_op(op, dev)
        ms, g = bench.time_graph(fn, stream, 30, 5)
        per.append(ms * 1e3 / len(sel))
    print("sk1 ablate=%s: " % mask + " ".join(f"{n} {u:.2f}" for n, u in zip(["qkv", "o", "gate_up", "down"], per)), flush=True)
else:
    for abl in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "4", "6", "7", "15", "0"]):
        env = dict(os.environ, SK1_ABLATE=str(abl), GPTQHIP_NO_PERSIST="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("sk1")) or r.stderr[-600:], flush=True)
