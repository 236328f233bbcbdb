#!/usr/bin/env python3
"""Registers / spills / LDS of every kernel in the built objects (code-object metadata):
    python tools/kernel_resources.py [substring ...]   # e.g. fast_kernelILi32ELi64
Reads outeffhop_amd/lib/liboeh_hip.so (make -C outeffhop_amd/csrc), or the build named by OEH_LIB (an experiment build: make ... alt NAME=...)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    pats = sys.argv[1:]
    tmp = tempfile.mkdtemp()
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(os.environ.get("OEH_LIB") or os.path.join(ROOT, "outeffhop_amd", "lib", "liboeh_hip.so"), lib)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], check=True, capture_output=True, cwd=tmp)  # writes lib.so.N.hipv4-...gfx950
        for co in sorted(glob.glob(lib + ".*gfx950")):
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                if pats and not any(p in name for p in pats):
                    continue
                g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))  # noqa: E731
                print(f"{name[:100]:100s} vgpr {g('vgpr_count'):3d} spill {g('vgpr_spill_count'):3d} sgpr {g('sgpr_count'):3d} "
                      f"lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size')}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
