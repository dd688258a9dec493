"""Kernel resource usage (VGPR / SGPR / spills / LDS / scratch / kernarg bytes) from a device code object's notes.
usage: llvm-objcopy --dump-section .hip_fatbin=fat.bin X.o; clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950
       --input=fat.bin --output=g.co --unbundle; llvm-readelf --notes g.co > notes.txt; python scratch/kres.py notes.txt [filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for e in re.split(r"\n\s+- \.agpr_count", t)[1:]:
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", e).group(1))
    name = re.search(r"\.name:\s+(\S+)", e).group(1)
    n = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n).replace("void ali::", "")
    if flt in n:
        print(f"{n[:72]:72s} vgpr {g('vgpr_count'):4d} sgpr {g('sgpr_count'):4d} spill {g('vgpr_spill_count'):3d} "
              f"lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size'):4d} kernarg {g('kernarg_segment_size')}")
