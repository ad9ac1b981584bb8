#!/usr/bin/env python3
"""Development tool: VGPRs, SGPRs, scratch and LDS of the kernels in a built library, read from the code objects embedded in it
(the notes the compiler writes: .vgpr_count is what decides the waves per SIMD, not the amdgpu_waves_per_eu attribute).

    python tools/kernel_resources.py [substring of the kernel name] [library]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "aocl-compression_amd", "libaocl_compression_gpu.so")
    d = open(lib, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", d)][1:]         # the first ELF header is the library itself
    with tempfile.TemporaryDirectory() as tmp:
        for n, i in enumerate(starts):
            path = os.path.join(tmp, "co%d.elf" % n)
            open(path, "wb").write(d[i:i + (starts[n + 1] - i if n + 1 < len(starts) else len(d) - i)])
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                nm = re.search(r"\.name:\s+(\S+)", blk)
                if not nm or want not in nm.group(1):
                    continue
                f = dict(re.findall(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", blk))
                print("%-90s vgpr %3s sgpr %3s scratch %5s B lds %6s B" % (nm.group(1), f.get("vgpr_count"), f.get("sgpr_count"),
                      f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
