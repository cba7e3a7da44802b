"""dev (GPU box): timing ablations of decode_stream_kernel (instrumented library, tests/dev/stream_stamps_build.sh): us per launch of the
four Llama-3-8B op shapes under graph replay with parts of the loop switched off (results wrong by construction)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    os.environ["GPTQHIP_LIB"] = os.path.join(HERE, "ablate", "libgptqhip_abl%s.so" % os.environ.get("GPTQHIP_STREAM_ABLATE", "0"))
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
    print("ablate=%s waves=%s ring=%s: " % (os.environ.get("GPTQHIP_STREAM_ABLATE", "0"), os.environ.get("GPTQHIP_STREAM_WAVES", "auto"), os.environ.get("GPTQHIP_STREAM_RING", "auto"))
          + " ".join(f"{n} {u:.2f}" for n, u in zip(["qkv", "o", "gate_up", "down"], per)), flush=True)
else:
    for abl, waves, ring in [(int(a), "", "") for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "15", "31", "63", "16", "48"])]:
        env = dict(os.environ, GPTQHIP_STREAM_ABLATE=str(abl))
        if waves:
            env["GPTQHIP_STREAM_WAVES"] = waves
        if ring:
            env["GPTQHIP_STREAM_RING"] = ring
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("ablate")) or r.stderr[-600:], flush=True)
