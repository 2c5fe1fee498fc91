"""Registers, LDS and SCRATCH of the kernels in a built object (stonkgs_amd/csrc/*.o): the .hip_fatbin section is unbundled
with the ROCm LLVM tools and the code object's metadata notes are read. A kernel whose hand-laid-out register file spills
still produces right answers - only slower (round 4: a few more scalars alive across the written-out K loop put 29-78
registers of every 256-wide gemm_a4 instance into scratch; FFN-up 144 -> 175 us) - so tests/test_host_cpu.py asserts on this.
  python tools/kernel_resources.py stonkgs_amd/csrc/gemm_a4.o"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(obj):
    """[{name, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill}] for every kernel of the gfx950 code object in `obj`"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    keys = {".name": "name", ".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds",
            ".private_segment_fixed_size": "scratch", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill"}
    for line in notes.splitlines():
        m = re.match(r"^\s*-?\s*(\.\w+):\s+(\S+)\s*$", line)
        if not m or m.group(1) not in keys:
            continue
        k, v = keys[m.group(1)], m.group(2)
        if k == "name":
            cur = {"name": v}
            out.append(cur)
        elif cur is not None:
            cur[k] = int(v)
    return [k for k in out if "scratch" in k]


if __name__ == "__main__":
    for k in kernel_resources(sys.argv[1]):
        print(f"{k['name'][:90]:90s} vgpr {k.get('vgpr', 0):3d} agpr {k.get('agpr', 0):3d} sgpr {k.get('sgpr', 0):3d} lds {k.get('lds', 0):6d} "
              f"scratch {k['scratch']:4d} spills v{k.get('vgpr_spill', 0)} s{k.get('sgpr_spill', 0)}")
