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

// ---- long inputs (teacher-forced evaluation, first step of a long prompt): blocks of up to VLO_PREFILL_TOKENS tokens ------------------
// The projections run as real GEMMs (MFMA-bound: hundreds of tokens per weight byte instead of 64): the ViT's ping-pong kernel
// (vit_gemm.inc: 256 x 256 tiles, direct-to-LDS double buffering, persistent XCD-aware tile walk) instantiated for bf16 operands with
// the weight operand read from the SAME packed fragment image the GEMV streams (a per-lane source address of the direct-to-LDS load)
// and the Llama epilogues with the GEMV path's rounding points.  fp8 engines: each projection's e4m3 image is expanded (exactly) to bf16 into
// a per-session scratch right before its GEMM (expand_fp8_image_launch), its per-channel scales multiply the sums in the epilogue.
#define VLO_PREFILL_TOKENS 4096   // rows of the prefill workspace; x must stay readable 256 rows past M (the kernel reads whole tiles)
#define VLO_PREFILL_MIN 256       // shorter inputs take the 64-token block path
enum { LLM_GEMM_BF16 = 0, LLM_GEMM_SWIGLU = 1, LLM_GEMM_RESID = 2, LLM_GEMM_F32 = 3 };
// X bf16 [M][K] row-major; Wp = packed image of W [N][K]; N % 256 == 0, K % 128 == 0.
//   LLM_GEMM_BF16:   out bf16 [M][ldo] = bf16(X W^T)
//   LLM_GEMM_SWIGLU: Wp = the gate/up image (N = 2 I): out bf16 [M][ldo = I] = bf16(silu(bf16 g) * bf16 u)
//   LLM_GEMM_RESID:  out = the residual stream bf16 [M][N]: out = bf16(out + bf16(X W^T))
//   LLM_GEMM_F32:    out = FLOAT [M][ldo] = X W^T (raw sums): a tensor-parallel rank's partial o-proj / down-proj, all-reduced by the caller
// wscale: null (bf16 image) or the fp32 per-output-channel scales of an fp8 engine, packed row order; Wp is then the bf16 EXPANSION of the
// fp8 image (expand_fp8_image_launch below), the scales multiply the fp32 sums in the epilogue as they do in the GEMV
hipError_t llm_gemm_launch(const unsigned short *X, const void *Wp, int M, int N, int K, void *out, int ldo, int kind, hipStream_t st,
                           const float *wscale = nullptr);
// fp8 e4m3 image Wp8[tile][kf2][lane] (gemv.hip) -> bf16 image Wp[tile][kf][lane], exact (every e4m3 value is a bf16 value); NT tiles of K
hipError_t expand_fp8_image_launch(const void *Wp8, void *Wp_bf16, int NT, int K, hipStream_t st);
// ---- native fp8 MFMA for the prefill GEMMs of an fp8 engine (vlo_config.prefill_act_dtype = 1; BASELINE.json configs[4] "fp8 MFMA weights") ----
// W8A8: the X operand of a projection is quantised per ROW to OCP e4m3 — scale[m] = max|X[m]| * (1 / 448) (1 for a zero row), code =
// e4m3_rne(clamp(X[m][k] / scale[m], -448, 448)), the weights' rule — and the GEMM multiplies e4m3 by e4m3 on v_mfma_f32_16x16x128_f8f6f4
// (fp32 accumulation, 2 x the bf16 MFMA rate), W read from the fp8 GEMV image AS STORED (no expansion pass, no bf16 scratch);
// out[m][n] = (sum_k xq wq) * wscale[n] * scale[m], then the kind's epilogue with its usual rounding points.
// Xq layout: [M][K] bytes, row-major, the k's of every 64-k group in the order the image holds them: byte p of group g is
// k = 64 g + (p % 16 / 8) * 32 + (p / 16) * 8 + p % 8 (vlo_fp8_row_pos below is the inverse).  Xq and scale must stay readable 256 rows past M.
// N % 256 == 0, K % 256 == 0 (llm_gemm_fp8_ok); kinds as llm_gemm_launch.
static inline int vlo_fp8_row_pos(int k) { const int g = k >> 6, r = k & 63; return g * 64 + ((r & 31) >> 3) * 16 + (r >> 5) * 8 + (r & 7); }
hipError_t quantize_rows_fp8_launch(const unsigned short *X, int M, int K, void *Xq, float *scale, hipStream_t st);
bool llm_gemm_fp8_ok(int N, int K);
hipError_t llm_gemm_fp8_launch(const void *Xq, const float *xscale, const void *Wp8, const float *wscale, int M, int N, int K, void *out, int ldo,
                               int kind, hipStream_t st);
// qkv bf16 [M][(nh + 2 nkv) hd] (projection outputs) -> RoPE (HF rounding points) -> q bf16 [M][nh hd], K / V^T appended to the paged pool
// at positions pos0 .. pos0 + M - 1
hipError_t rope_kv_append_launch(const unsigned short *qkv, int M, int num_heads, const unsigned short *cos_tab, const unsigned short *sin_tab,
                                 KvGeom kv, int layer, long long pos0, unsigned short *q_out, hipStream_t st);
