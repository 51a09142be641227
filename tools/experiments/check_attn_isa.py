"""ISA check of the live-step attention kernels (csrc/llm_ops.hip: attn_cols_kernel, attn_chunk_kernel): inside the key loop of every
instantiation there must be
  * no `s_waitcnt vmcnt(0)` (every wait of the loop is COUNTED: the next block's K / V^T loads stay in flight while a block is multiplied),
  * no vector load of the page table (`global_load_dword ` — page ids are scalar loads),
  * no scratch traffic (a spill reload is a vmcnt(0) of its own) and no waterfall loop (`s_cbranch_execnz`: a buffer resource that hipcc
    believes divergent),
and the kernel must actually prefetch (buffer_load_dwordx4 / dwordx2 inside the loop).  Round 4's loop failed the first two: a dependent
vector load of kv.page_table and two full drains per 32-key block (lesson 0).

    python tools/check_attn_isa.py            # compiles llm_ops.hip to gfx950 assembly (hipcc -S, ~5 s) and prints one line per kernel
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "videollm-online_amd", "csrc", "llm_ops.hip")      # (point it at a tree that has the experiment wired in)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only"]
KERNELS = ("attn_cols_kernel", "attn_chunk_kernel")


def disassemble(path=None):
    out = path or os.path.join(tempfile.mkdtemp(prefix="vlo_isa_"), "llm_ops.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [SRC, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def key_loops(asm):
    """{mangled kernel name: [lines of its outermost loop ... last back edge]} for the attention kernels"""
    res = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        if not any(k in name for k in KERNELS) or "_v1_" in name:
            continue
        head = next((i for i, l in enumerate(body) if "Loop Header: Depth=1" in l), None)
        assert head is not None, f"{name}: no loop found"
        label = body[head].split(":")[0].strip()
        last = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", l))
        res[name] = body[head:last + 1]
    return res


def check(asm):
    report, ok = [], True
    for name, loop in key_loops(asm).items():
        txt = "\n".join(loop)
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", txt)]
        bad = dict(vmcnt0=waits.count(0), vector_page_loads=len(re.findall(r"global_load_dword ", txt)), scratch=len(re.findall(r"scratch_(load|store)", txt)),
                   waterfall=len(re.findall(r"s_cbranch_execnz", txt)))
        loads = len(re.findall(r"buffer_load_dwordx[24]", txt))
        good = not any(bad.values()) and loads > 0 and waits
        ok &= bool(good)
        short = re.sub(r"^_Z\d+(\w+?)ILi(\d+)ELi(\d+)E.*", r"\1<\2,\3>", name)
        report.append(f"{'ok ' if good else 'BAD'} {short}: {len(loop)} lines in the key loop, {loads} K/V^T loads, vmcnt waits min {min(waits) if waits else '-'} max {max(waits) if waits else '-'}"
                      + ("" if good else f"  {bad}"))
    return ok, report


if __name__ == "__main__":
    ok, report = check(disassemble(sys.argv[1] if len(sys.argv) > 1 else None))
    print("\n".join(report))
    sys.exit(0 if ok else 1)
