#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a built library: tools/kernel_registers.py [vg_amd/libvgamd.so] [name filter]
(llvm-objdump --offloading unbundles the gfx950 code objects, llvm-readelf --notes holds the kernel descriptors' metadata)."""
import os, re, subprocess, sys, tempfile, shutil, glob
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "vg_amd", "libvgamd.so"))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
try:
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
    rows = []
    for co in sorted(glob.glob(tmp + "/lib.so.*gfx950")):
        cur = {}
        for line in subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout.splitlines():
            m = re.match(r"\s*-?\s*\.(agpr_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_count|vgpr_count|vgpr_spill_count|symbol|wavefront_size):\s+(\S+)", line)
            if not m: continue
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "wavefront_size":
                rows.append(cur); cur = {}
    for r in sorted(rows, key=lambda r: r.get("symbol", "")):
        name = subprocess.run(["c++filt", r.get("symbol", "?").replace(".kd", "")], capture_output=True, text=True).stdout.strip().split("(")[0]
        if flt and flt not in name: continue
        print("%-70s vgpr %3s agpr %3s spills %3s scratch %5s B  lds %6s B" % (name[-70:], r.get("vgpr_count"), r.get("agpr_count"), r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
finally:
    shutil.rmtree(tmp)
