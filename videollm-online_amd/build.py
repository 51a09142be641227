"""Build libvlo.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot; there is no
JIT cache and no CPU fallback."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvlo.so")
SOURCES = ["gemv.hip", "prefill.hip", "llm_ops.hip", "layer.hip", "vit.hip", "engine.hip", "tp.hip"]
HEADERS = ["common.cuh", "gemv.h", "prefill.h", "llm_ops.h", "vit.h", "engine.h", "tp_p2p.cuh", "layer.h", "gemv_body.inc", "gemv_head.inc", "attn_body.inc", "rmsnorm_body.inc", os.path.join("..", "..", "include", "vlo.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine has no non-HIP build")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into videollm-online_amd/libvlo.so."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
