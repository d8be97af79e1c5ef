"""Register / LDS / spill figures of every kernel in a built library, read from its gfx950 code objects (no GPU needed).
   python tools/kernel_regs.py [--test] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(lib):
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "copy.so")], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, "b%d.bin" % i)
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "co%d.o" % i)
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + co], check=True, capture_output=True)
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            block = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count" and line.lstrip().startswith("-"):
                    block = {}
                    out.append(block)
                block[k] = int(v) if v.isdigit() else v
    return [b for b in out if "name" in b]


if __name__ == "__main__":
    test = "--test" in sys.argv
    flt = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = os.path.join(ROOT, "4dflownet_amd", "lib4dflow_hip_test.so" if test else "lib4dflow_hip.so")
    for b in kernels(lib):
        name = subprocess.run(["c++filt", b["name"]], capture_output=True, text=True).stdout.strip()
        if flt and not any(f in name for f in flt):
            continue
        print("%-70s vgpr %3d agpr %3d sgpr %3d lds %6d spill %d scratch %d" % (name[:70], b.get("vgpr_count", -1), b.get("agpr_count", -1),
              b.get("sgpr_count", -1), b.get("group_segment_fixed_size", 0), b.get("vgpr_spill_count", 0), b.get("private_segment_fixed_size", 0)))
