#!/usr/bin/env python
"""Apply the MI355X HIP backend overlay to a checkout of ModelCloud/GPTQModel.

    python integration/apply_overlay.py /path/to/GPTQModel            # patches the tree in place
    python integration/apply_overlay.py /path/to/GPTQModel --check    # verify only (exit 1 if not applied)

Three changes, all additive:
  1. gptqmodel/utils/backend.py  += BACKEND.GPTQ_HIP / AWQ_HIP / HIP (+ legacy-alias rows)     [utils/backend.patch]
  2. gptqmodel/nn_modules/qlinear/hip.py  (new file; found by the reference's own subclass-walk discovery)
  3. gptqmodel/utils/model.py  gptqmodel_post_init() first calls gptqmodel_amd.utils.hf_llama.auto_fuse(model), so that the model
     GPTQModel.load() returns decodes through the fused decode ops by itself (GPTQHIP_AUTO_FUSE=0 opts out)   [utils/model.patch]
The `gptqmodel_amd` package (this repo) must be importable in the same environment; it carries the kernels.
"""
import argparse
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OVERLAY = os.path.join(HERE, "gptqmodel_overlay")


def applied(tree: str) -> bool:
    b = os.path.join(tree, "gptqmodel", "utils", "backend.py")
    h = os.path.join(tree, "gptqmodel", "nn_modules", "qlinear", "hip.py")
    m = os.path.join(tree, "gptqmodel", "utils", "model.py")
    return (os.path.exists(h) and os.path.exists(b) and 'GPTQ_HIP = "gptq_hip"' in open(b).read()
            and os.path.exists(m) and "_gptqhip_auto_fuse" in open(m).read())


def apply(tree: str) -> None:
    if not os.path.isdir(os.path.join(tree, "gptqmodel", "nn_modules", "qlinear")):
        raise SystemExit(f"{tree} does not look like a GPTQModel checkout")
    backend = os.path.join(tree, "gptqmodel", "utils", "backend.py")
    if 'GPTQ_HIP = "gptq_hip"' not in open(backend).read():
        patch = os.path.join(OVERLAY, "utils", "backend.patch")
        res = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", patch], cwd=tree, capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit(f"backend.patch did not apply:\n{res.stdout}\n{res.stderr}")
    model_py = os.path.join(tree, "gptqmodel", "utils", "model.py")
    if "_gptqhip_auto_fuse" not in open(model_py).read():
        patch = os.path.join(OVERLAY, "utils", "model.patch")
        res = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", patch], cwd=tree, capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit(f"model.patch did not apply:\n{res.stdout}\n{res.stderr}")
    shutil.copyfile(os.path.join(OVERLAY, "nn_modules", "qlinear", "hip.py"),
                    os.path.join(tree, "gptqmodel", "nn_modules", "qlinear", "hip.py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tree")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    if a.check:
        sys.exit(0 if applied(a.tree) else 1)
    apply(a.tree)
    print(f"overlay applied to {a.tree}")


if __name__ == "__main__":
    main()
