"""Snapshot of the handful of REFERENCE files the import shim (oracle/ref_import.py) executes, into the git-ignored
`oracle/_ref/` -- so that the REAL reference modules (TorchLinear / AwqTorchLinear, gptqmodel/nn_modules/qlinear/torch.py:114,
torch_awq.py:20) can be timed as the CPU baseline on the GPU box, where /root/reference does not exist.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  `oracle/_ref/` is listed in .gitignore (reference sources never enter this repo's
history) but not in .gpurunignore, so it travels with a push exactly like the built libgptqhip.so.  `__graft_entry__.build()`
runs this whenever /root/reference is present; nothing under gptqmodel_amd/ ever imports it (tests/test_host_logic.py greps).

How the file list is found: import the reference through the shim from /root/reference and record every module whose file lives
under it -- no hand-kept list to go stale.  Run: `python -m oracle.make_ref_snapshot`.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SNAP = os.path.join(HERE, "_ref")
LIVE = "/root/reference"


def make_snapshot(verbose: bool = True) -> int:
    """Copies the executed reference files to oracle/_ref/ (returns the file count; 0 when /root/reference is not mounted)."""
    if not os.path.isdir(os.path.join(LIVE, "gptqmodel")):
        return 0
    import subprocess
    # a fresh interpreter: the recorded module set must not depend on what this process already imported
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        f"os.environ['GPTQ_REFERENCE_ROOT'] = {LIVE!r}\n"
        "from oracle.ref_import import load_reference\n"
        "ref = load_reference()\n"
        # exercise what the CPU baseline runs, so that lazily imported modules (e.g. the rotation hook imported inside forward())
        # are recorded too
        "import torch\n"
        "lin = ref.TorchLinear(bits=4, group_size=128, desc_act=False, sym=True, in_features=256, out_features=256, bias=False,\n"
        "                      pack_dtype=torch.int32, register_buffers=True)\n"
        "lin.optimize = lambda *a, **k: None\n"
        "lin = lin.eval(); lin.post_init()\n"
        "lin(torch.zeros(1, 256, dtype=torch.bfloat16)); lin.dequantize_weight()\n"
        "awq = ref.AwqTorchLinear(bits=4, group_size=128, desc_act=False, sym=False, in_features=256, out_features=256, bias=False,\n"
        "                         pack_dtype=torch.int32, register_buffers=True)\n"
        "awq.optimize = lambda *a, **k: None\n"
        "awq = awq.eval(); awq.post_init()\n"
        "awq(torch.zeros(1, 256, dtype=torch.bfloat16))\n"
        # ... and what tests/test_gpu_reference_dropin.py drives on the GPU box: QuantizeConfig -> make_quant -> gptqmodel_post_init
        "import torch.nn as nn\n"
        "from gptqmodel.quantization.config import QuantizeConfig\n"
        "from gptqmodel.quantization import FORMAT, METHOD\n"
        "from gptqmodel.utils.model import make_quant, gptqmodel_post_init\n"
        "for method, fmt, be in ((METHOD.GPTQ, FORMAT.GPTQ_V2, ref.BACKEND.GPTQ_TORCH), (METHOD.AWQ, FORMAT.GEMM, ref.BACKEND.AWQ_TORCH)):\n"
        "    q = QuantizeConfig(bits=4, group_size=128, desc_act=False, sym=False, method=method, format=fmt)\n"
        "    blk = nn.Sequential(nn.Linear(256, 256, bias=True, dtype=torch.float16))\n"
        "    make_quant(blk, q, quant_result={'0': {}}, backend=be, lm_head_name='lm_head', device=ref.DEVICE.CPU, from_quantized=True,\n"
        "               dtype=torch.float16)\n"
        "    blk[0].optimize = lambda *a, **k: None\n"
        "    gptqmodel_post_init(blk, use_act_order=False, quantize_config=q)\n"
        "    blk[0](torch.zeros(1, 256, dtype=torch.float16))\n"
        f"root = {LIVE!r} + os.sep\n"
        "for m in list(sys.modules.values()):\n"
        "    f = getattr(m, '__file__', None)\n"
        "    if f and os.path.abspath(f).startswith(root):\n"
        "        print(os.path.abspath(f))\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise RuntimeError("reference import failed:\n" + out.stderr[-2000:])
    files = {ln.strip() for ln in out.stdout.splitlines() if ln.strip().startswith(LIVE)}
    # The drop-in test on the GPU box (tests/test_gpu_reference_dropin.py) runs the reference's selector with a device visible: every
    # candidate kernel's validate_once() then lazily imports its own helper modules (gptqmodel/extension.py, utils/*), a set that cannot
    # be recorded in this GPU-less container.  So the rest of the package's Python files travel too (4 MB; the model zoo stays out).
    pkg = os.path.join(LIVE, "gptqmodel")
    for d, dirs, names in os.walk(pkg):
        dirs[:] = [x for x in dirs if x != "__pycache__" and os.path.join(d, x) != os.path.join(pkg, "models", "definitions")]
        files.update(os.path.join(d, n) for n in names if n.endswith(".py"))
    files = sorted(files)
    if os.path.isdir(SNAP):
        shutil.rmtree(SNAP)
    for f in files:
        rel = os.path.relpath(f, LIVE)
        dst = os.path.join(SNAP, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(f, dst)
    # the licence travels with the files
    for name in ("LICENSE", "LICENSE.txt"):
        if os.path.isfile(os.path.join(LIVE, name)):
            shutil.copyfile(os.path.join(LIVE, name), os.path.join(SNAP, name))
    with open(os.path.join(SNAP, "SNAPSHOT.txt"), "w") as fh:
        fh.write("Files of ModelCloud/GPTQModel executed by oracle/ref_import.load_reference(), copied by oracle/make_ref_snapshot.py\n"
                 "for timing the reference's own CPU path on the GPU box.  Not part of this repository's sources (git-ignored).\n")
        fh.write("\n".join(os.path.relpath(f, LIVE) for f in files) + "\n")
    if verbose:
        print(f"oracle/_ref: {len(files)} reference files")
    return len(files)


if __name__ == "__main__":
    n = make_snapshot()
    if n == 0:
        print("reference not mounted at /root/reference: nothing to snapshot")
