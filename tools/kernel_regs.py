#!/usr/bin/env python
"""Register / scratch / LDS usage of every kernel in the built library (development): unbundles
libgadfly_hip.so in a temporary directory and reads the code objects' metadata notes.
python tools/kernel_regs.py [substring ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("GADFLY_SO") or os.path.join(ROOT, "gadfly_amd", "csrc", "libgadfly_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
pats = sys.argv[1:]
tmp = tempfile.mkdtemp()
try:
    shutil.copy(SO, os.path.join(tmp, "libgadfly_hip.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "libgadfly_hip.so"], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=tmp, check=True, capture_output=True,
                               text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)   # noqa: E731
            name = g("name").group(1)
            rows.append((name, int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)),
                         int(g("vgpr_spill_count").group(1)), int(g("private_segment_fixed_size").group(1)),
                         int(g("group_segment_fixed_size").group(1))))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for (n, v, s, sp, sc, lds), dem in zip(rows, names):
        dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", ""))
        if pats and not any(p in dem for p in pats):
            continue
        print(f"{dem[:70]:70s} vgpr {v:4d} sgpr {s:4d} spill {sp:4d} scratch {sc:6d} lds {lds:7d}")
finally:
    shutil.rmtree(tmp)
