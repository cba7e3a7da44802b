"""Dev tool: per-kernel register / scratch table of every translation unit (hipcc -Rpass-analysis=kernel-resource-usage; needs no
GPU).  Usage: python tests/dev/isa_audit.py [out.txt]   -- prints one line per kernel instantiation: VGPRs, AGPRs, scratch bytes
per lane, spilled VGPRs, occupancy, and a summary of the spilling ones."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CSRC = os.path.join(ROOT, "gptqmodel_amd", "csrc")
UNITS = ["gptqhip_skinny.hip", "gptqhip_tiled.hip", "gptqhip_tiled_f32.hip", "gptqhip_tiled8.hip", "gptqhip_aux.hip", "gptqhip_comm.hip"]


def audit(unit):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only", "-c",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, unit), "-o", os.path.join(td, "x.o")],
                           capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for row in rows:
        d = subprocess.run(["c++filt", row["name"]], capture_output=True, text=True).stdout.strip()
        row["demangled"] = re.sub(r"\(gptqhip::\w+\)$", "", d).replace("void gptqhip::", "")
    return unit, rows


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    with ThreadPoolExecutor(len(UNITS)) as ex:
        res = list(ex.map(audit, UNITS))
    print("# tests/dev/isa_audit.py: hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, every kernel instantiation", file=out)
    print("# skinny_kernel<BITS, ACT, SCL, MT, GPC, AM, D, GLUE, LB>; tiled kernels: see gptqhip_tiled_kernel.h", file=out)
    spill = []
    for unit, rows in res:
        print(f"## {unit}: {len(rows)} kernels", file=out)
        for r in rows:
            sp = int(r.get("VGPRs Spill", 0))
            sc = int(r.get("ScratchSize [bytes/lane]", 0))
            print(f"{r['demangled']:90s} vgpr {r.get('VGPRs'):>3s} agpr {r.get('AGPRs'):>3s} scratch {sc:4d} spill {sp:3d} occ {r.get('Occupancy [waves/SIMD]')}", file=out)
            if sp or sc:
                spill.append((unit, r["demangled"], sp, sc))
    print(f"## kernels with scratch / spills: {len(spill)}", file=out)
    for u, n, sp, sc in spill:
        print(f"SPILL {u} {n} spill {sp} scratch {sc}", file=out)


if __name__ == "__main__":
    main()
