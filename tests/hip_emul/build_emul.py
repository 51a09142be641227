"""Builds tests/hip_emul/_build/libvlo_emul.so: the engine's SOURCES (csrc/{gemv,prefill,llm_ops,vit,ingest,engine,tp}.hip)
compiled as host C++ against the HIP-on-threads shim (hip_emul.h); csrc/vit.hip as well.  Test infrastructure only — the product is libvlo.so."""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "videollm-online_amd", "csrc")
OUT = os.path.join(HERE, "_build")
SOURCES = ["gemv.hip", "prefill.hip", "llm_ops.hip", "vit.hip", "ingest.hip", "engine.hip", "tp.hip"]


def clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def build(force=False, sanitize=None):
    """sanitize: None, or a -fsanitize= list such as "address,undefined" / "thread" (VLO_EMUL_SANITIZE): the kernels' index
    arithmetic and their use of shared memory then run under the sanitizer (LD_PRELOAD its runtime for python)."""
    cc = clang()
    if cc is None:
        return None
    sanitize = sanitize or os.environ.get("VLO_EMUL_SANITIZE") or None
    global OUT
    if sanitize:
        OUT = os.path.join(HERE, "_build", "san_" + sanitize.replace(",", "_"))
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libvlo_emul.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, f) for f in ("hip_emul.h", "emul_stubs.cpp", "build_emul.py")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    objs, procs = [], []
    flags = ["-x", "c++", "-std=c++17", "-O1", "-g0", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-unknown-attributes",
             "-I", HERE, "-I", CSRC]
    san = ["-fsanitize=" + sanitize, "-fno-omit-frame-pointer", "-g"] if sanitize else []
    flags = [f for f in flags if not (san and f == "-g0")] + san
    def patch(src):
        # dynamic shared memory: `extern __shared__ T name[]` refers to an array the harness defines (emul_stubs.cpp)
        src = re.sub(r"extern\s+__shared__", "extern", src)
        # GPU assembly (explicit s_waitcnt around direct-to-LDS loads / release sequences): the emulated loads are synchronous
        # vmcnt waits retire emulated direct-to-LDS loads (hip_emul.h: VLO_EMUL_GLDS=late lands them only there); other counters: nothing to wait for
        src = re.sub(r'asm volatile\("s_waitcnt vmcnt\(0\)[^"]*"\s*:::\s*"memory"\);', "emul_vmcnt(0);", src)
        src = re.sub(r'asm volatile\("s_waitcnt vmcnt\(%0\)"\s*::\s*"n"\(([^;]*)\)\s*:\s*"memory"\);', r"emul_vmcnt(\1);", src)   # counted form
        src = re.sub(r'asm volatile\("s_waitcnt[^"]*"\s*:::\s*"memory"\);', ";", src)
        src = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(%0\)"\s*::\s*"n"\([^;]*\)\s*:\s*"memory"\);', ";", src)
        # csrc/gemv_engine.inc::eng_glds16 — one direct-to-LDS piece issued in inline asm (invisible to hipcc's vmcnt bookkeeping): lane l moves 16
        # bytes from its own gsrc to lds_dst + 16 l
        src = re.sub(r'asm volatile\("s_mov_b32 %0, m0[^;]*;', "emul_glds(gsrc, (char *)lds_dst + (threadIdx.x & 63) * 16, 16); (void)keep; (void)dst;", src)
        # csrc/vit_tall.inc::glds16_saddr — the same with an SGPR base + per-lane byte offset
        src = re.sub(r'asm volatile\("s_mov_b32 m0, %2[^;]*;', "emul_glds((const char *)sbase + voff, (char *)lds_dst + (threadIdx.x & 63) * 16, 16); (void)dst;", src)
        # csrc/common.cuh::mfma_fp8_k128_acc — the fp8 MFMA accumulating in place through inline asm, and the software wait states around it
        src = re.sub(r'asm volatile\("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0"\s*:\s*"\+v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);',
                     r"\1 = emul_mfma_fp8_16x16x128(\2, \3, \1);", src)
        src = re.sub(r'asm volatile\("s_nop[^"]*"(?:\s*:::\s*"memory")?\);', ";", src)
        src = re.sub(r'asm volatile\(""\s*:::\s*"memory"\);', ";", src)                  # compiler-only memory barrier
        src = re.sub(r'asm(?: volatile)?\(""\s*:\s*"\+v"\([\w\[\]]+\)\);', ";", src)      # optimisation barrier on a VGPR value
        src = re.sub(r'asm\("s_nop 7\\n\\ts_nop 3\\n\\tv_max3_f32[^;]*;', "r = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(a4, a5), fmaxf(a6, a7)));", src)
        src = re.sub(r'asm\("v_max_f32 %0, %1, %2"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);', r"\1 = fmaxf(\2, \3);", src)
        return src

    # textually included kernel bodies are patched too: the copies in OUT shadow the originals (quote includes resolve next to
    # the including file first)
    for inc in os.listdir(CSRC):
        if inc.endswith((".inc", ".cuh")):                  # (.cuh: common.cuh carries the inline-asm direct-to-LDS helper)
            with open(os.path.join(OUT, inc), "w") as f:
                f.write(f'#line 1 "{os.path.join(CSRC, inc)}"\n' + patch(open(os.path.join(CSRC, inc)).read()))
    for s in SOURCES:
        src = patch(open(os.path.join(CSRC, s)).read())
        patched = os.path.join(OUT, s.replace(".hip", "_emul.cpp"))
        with open(patched, "w") as f:
            f.write(f'#line 1 "{os.path.join(CSRC, s)}"\n' + src)
        obj = patched.replace(".cpp", ".o")
        objs.append(obj)
        procs.append((s, subprocess.Popen([cc] + flags + ["-c", patched, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    stub_o = os.path.join(OUT, "emul_stubs.o")
    objs.append(stub_o)
    procs.append(("emul_stubs.cpp", subprocess.Popen([cc] + flags + ["-c", os.path.join(HERE, "emul_stubs.cpp"), "-o", stub_o],
                                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"emulated build failed on {s}:\n{out[-6000:]}")
    # -Bsymbolic: the library binds its internal references to ITSELF even when libvlo.so (same symbol names, RTLD_GLOBAL)
    # is already loaded in the process
    r = subprocess.run([cc, "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic"] + san + objs + ["-ldl", "-lrt", "-o", lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"emulated link failed:\n{r.stdout[-4000:]}")
    return lib


def build_rccl_shim():
    """tests/hip_emul/rccl_shim.cpp -> _build/librccl_shim.so: the shared-memory stand-in for librccl (VLO_RCCL_LIBRARY)."""
    cc = clang() or shutil.which("g++")
    if cc is None:
        return None
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    src, lib = os.path.join(HERE, "rccl_shim.cpp"), os.path.join(HERE, "_build", "librccl_shim.so")
    if not os.path.exists(lib) or os.path.getmtime(src) > os.path.getmtime(lib):
        r = subprocess.run([cc, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", src, "-lrt", "-o", lib],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"rccl shim build failed:\n{r.stdout[-3000:]}")
    return lib


if __name__ == "__main__":
    print(build(force=True))
