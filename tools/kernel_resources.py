#!/usr/bin/env python3
"""Dev tool: register / scratch / LDS usage of every kernel in the built gfx950 objects (from the code objects' metadata notes).
usage: kernel_resources.py [substring]   -- prints name, VGPRs, AGPRs, SGPRs, spills, scratch bytes, static LDS bytes"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
want = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for obj in sorted(glob.glob(os.path.join(ROOT, "qpmpc_amd/lib/obj/*.o"))):
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "k.co"), os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True)
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co):
            continue
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void mpcqp::", "")
        if want in name:
            rows.append((os.path.basename(obj).split(".")[0], name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"),
                         g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'unit':16s} {'kernel':70s} vgpr agpr sgpr vspill sspill scratch lds")
for r in rows:
    print(f"{r[0]:16s} {r[1][:70]:70s} {r[2]:>4s} {r[3]:>4s} {r[4]:>4s} {r[5]:>6s} {r[6]:>6s} {r[7]:>7s} {r[8]:>6s}")
