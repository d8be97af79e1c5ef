"""Disassembly of one kernel of a built library (gfx950 code object), to a file or stdout.   python tools/kernel_isa.py <name part> [--test]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
def main():
    test = "--test" in sys.argv
    part = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    lib = os.path.join(ROOT, "4dflownet_amd", "lib4dflow_hip_test.so" if test else "lib4dflow_hip.so")
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "c.so")], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            b = os.path.join(d, "b%d" % i); open(b, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, "co%d.o" % i)
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + b, "--output=" + co], check=True, capture_output=True)
            asm = subprocess.run([LLVM + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <([^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.S | re.M):
                if part in m.group(1):
                    print("== " + m.group(1)); print(m.group(2))
main()
