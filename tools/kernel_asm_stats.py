#!/usr/bin/env python3
"""Per-kernel register, spill and lane-copy counts from the device assembly (no GPU needed).
usage: python tools/kernel_asm_stats.py <csrc directory | assembly file> [substring ...]"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
made = not src.endswith(".s")
asm = os.path.join(tempfile.gettempdir(), "enigma_dev_%d.s" % os.getpid()) if made else src
if made:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-result",
                           "-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-o", asm, "enigma_hip.hip"], cwd=src)
text = open(asm).read()
if made:
    os.remove(asm)
meta = {}
for b in text.split("- .agpr_count")[1:]:
    n = re.search(r"\.name:\s+(\S+)", b)
    if n:
        meta[n.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count") if re.search(r"\.%s:\s+(\d+)" % k, b)}
bodies = re.split(r"\n(_Z\w+):[^\n]*\n", text)
for name, body in zip(bodies[1::2], bodies[2::2]):
    if name not in meta:
        continue
    body = body.split(".end_amdhsa_kernel")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void edev::", "")
    if sys.argv[2:] and not any(k in dem for k in sys.argv[2:]):
        continue
    m = meta[name]
    print("%-60s vgpr %3d spill %3d | sgpr %3d spill %3d | readlane %4d writelane %4d scratch %4d instr %6d" % (
        dem, m.get("vgpr_count", 0), m.get("vgpr_spill_count", 0), m.get("sgpr_count", 0), m.get("sgpr_spill_count", 0),
        len(re.findall(r"v_readlane", body)), len(re.findall(r"v_writelane", body)), len(re.findall(r"scratch_(load|store)", body)),
        len(re.findall(r"\n\t[a-z]", body))))
