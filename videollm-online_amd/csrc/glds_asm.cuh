// Direct-to-LDS piece loads issued in inline asm (users: llm_ops.hip::attn_prefill_kernel, vit_attn.inc::vit_attn_tiles_kernel).  Its own header so that
// translation units that only need common.cuh's helpers (the peer-to-peer exchange kernels and their host-compiled test harness) never see the statement.
#pragma once
#include "common.cuh"

// One 1-KiB direct-to-LDS piece issued in inline asm: lane l moves 16 bytes from its own gsrc to lds_dst + 16 l (lds_dst wave-uniform).
// Invisible to hipcc's s_waitcnt bookkeeping ON PURPOSE (hipcc drains builtin direct-to-LDS loads before every LDS read it cannot prove
// disjoint): the caller retires it with its own `s_waitcnt vmcnt(N)` + barrier.  The statement saves / restores M0 (cdna guide 5.7).
VLO_DEV void glds16_untracked(const void *gsrc, void *lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
