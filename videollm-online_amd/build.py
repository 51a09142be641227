"""Build libvlo.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot; there is no
JIT cache and no CPU fallback."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvlo.so")
SOURCES = ["gemv.hip", "prefill.hip", "llm_ops.hip", "vit.hip", "ingest.hip", "engine.hip", "tp.hip"]
HEADERS = ["common.cuh", "glds_asm.cuh", "vit_gemm.inc", "vit_attn.inc", "vit_ln.inc", "vit_tall.inc", "gemv.h", "prefill.h", "llm_ops.h", "vit.h", "engine.h", "tp_p2p.cuh", "gemv_body.inc", "gemv_head.inc", "gemv_epi.inc", "attn_body.inc", "attn_prefill_body.inc", "attn_prefill_pp_body.inc", "rmsnorm_body.inc", os.path.join("..", "..", "include", "vlo.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine has no non-HIP build")


def source_hash():
    """sha256 over every source and header of the library (names + contents, fixed order) and the compile flags; None when
    the sources are not next to the package (an installed binary)."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for s in SOURCES + HEADERS:
        path = os.path.join(CSRC, s)
        if not os.path.exists(path):
            return None
        h.update(os.path.basename(s).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def library_build_id(path=LIB):
    """The id stamped into a built library (`vlo_build_id()`), read from the file without loading it."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    tag = b"VLO_BUILD_ID="
    i = blob.find(tag)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(tag):j].decode(errors="replace")


def _stale():
    return library_build_id() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into videollm-online_amd/libvlo.so."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    build_id = source_hash()
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if s == "engine.hip":
            cmd.insert(1, f'-DVLO_BUILD_ID="{build_id}"')
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
