"""Per-kernel ISA comparison of csrc/*.hip between a git revision and the working tree (gfx950 disassembly, labels and comments
normalised).  Used to show that a source-level refactor — e.g. moving a kernel body into a textual include so that another kernel
can share it — left every already-validated kernel byte-identical.

    python tools/check_isa_identical.py 9c728bf gemv.hip llm_ops.hip
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("videollm-online_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only"]


def kernels(asm_path):
    txt = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+|\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
        body = re.sub(r"\.LBB\d+_\d+", ".LBB", m.group(2))
        body = re.sub(r";.*", "", body)
        out[m.group(1)] = re.sub(r"[ \t]+\n", "\n", body)
    return out


def compile_tree(tree, src, out):
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(tree, CSRC, src), "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    rev, files = sys.argv[1], sys.argv[2:] or ["gemv.hip", "llm_ops.hip"]
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        subprocess.run(f"git -C {ROOT} archive {rev} {CSRC} include | tar -x -C {old}", shell=True, check=True)
        for f in files:
            a, b = os.path.join(tmp, f + ".old.s"), os.path.join(tmp, f + ".new.s")
            compile_tree(old, f, a)
            compile_tree(ROOT, f, b)
            ka, kb = kernels(a), kernels(b)
            diff = sorted(k for k in ka if k in kb and ka[k] != kb[k])
            gone = sorted(k for k in ka if k not in kb)
            print(f"{f}: {len(ka)} kernels at {rev}, {len(kb)} now; changed {len(diff)}, removed {len(gone)}, new {len(kb) - len(ka) + len(gone)}")
            for k in diff + gone:
                print("   ", k)
            bad += len(diff) + len(gone)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
