#!/usr/bin/env python3
"""VGPR and spill counts of the kernels, from the device assembly (no GPU needed).
usage: python tools/vgpr_table.py [substring ...]     e.g.  python tools/vgpr_table.py 'k_shade<' k_intersect_closest
168 VGPRs is the most a kernel may use to keep three waves per SIMD, 128 for four."""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.gettempdir(), "enigma_dev.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-result",
                       "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-o", asm, "enigma_hip.hip"], cwd=os.path.join(root, "raytrace_amd", "csrc"))
rows = []
for b in open(asm).read().split("- .agpr_count")[1:]:
    n, v, s = re.search(r"\.name:\s+(\S+)", b), re.search(r"\.vgpr_count:\s+(\d+)", b), re.search(r"\.vgpr_spill_count:\s+(\d+)", b)
    if n and v:
        rows.append((n.group(1), int(v.group(1)), int(s.group(1)) if s else 0))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
for name, (_, vgpr, spill) in zip(names, rows):
    name = re.sub(r"\(.*", "", name).replace("void edev::", "")
    if not sys.argv[1:] or any(k in name for k in sys.argv[1:]):
        print(f"{vgpr:4d} vgpr {spill:3d} spilled  {name}")
