// prefill.h — host-visible interface of prefill.hip (block-of-64-tokens projections for long teacher-forced inputs)
#pragma once
#include <hip/hip_runtime.h>

#include "gemv.h"

#define VLO_BLOCK_TOKENS 64      // token rows one weight pass of the block path covers (4 MFMA token tiles)

struct Gemm64Plan { int NW, KF, KC; };
int gemm64_plan(int K, Gemm64Plan *p, bool even_kf = false);     // even_kf: the fp8 image (two fragments per 16-byte register)
// y[m][n] = sum_k x[m][k] W[n][k] for up to 64 rows of x against the SAME packed weight image the 16-row GEMV streams
// (gemv.hip; bf16, or fp8 e4m3 + scales when a.wq); epilogues EPI_BF16 / EPI_SWIGLU / EPI_RESID / EPI_ROPE with the rounding points of the GEMV path.
// x is a PACKED-64 matrix (llm_ops.h::vlo_pack64_elem); EPI_SWIGLU also writes its output packed-64 (it feeds the down
// projection).  Uses of GemvArgs: Wp, x, K, NT, N_valid, n_rows (<= 64), out_bf16 (+ ldo), h + ldo, cos/sin/kv/layer/num_heads/pos0.
hipError_t gemm64_launch(GemvArgs a, const Gemm64Plan &p, int epi, hipStream_t st);
