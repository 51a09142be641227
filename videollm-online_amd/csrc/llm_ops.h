// llm_ops.h — host-visible launchers of llm_ops.hip (Llama step kernels other than the GEMVs)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VLO_PAGE_TOKENS 256      // tokens per KV page (multiple of 32)
#define VLO_MAX_SPLITS 64        // max KV splits of the attention kernel

// Paged KV pool geometry.  One page id addresses, for every layer, a K block
// [kvh][PAGE][hd] and a V^T block [kvh][hd][PAGE] (V is stored transposed so the
// P.V MFMA can take it as an A operand straight from HBM).
struct KvGeom {
    unsigned short *k_pool;      // [layer][page][kvh][PAGE][hd]   bf16
    unsigned short *vt_pool;     // [layer][page][kvh][hd][PAGE]   bf16
    const int *page_table;       // logical page -> physical page (device)
    int64_t layer_stride;        // elements between layers  (= pool_pages * page_elems)
    int64_t page_elems;          // kvh * PAGE * hd
    int num_kv_heads, head_dim;
};

// "packed-64" activation layout of the 64-token block path (prefill.hip): the B-operand fragments of
// v_mfma_f32_16x16x32_bf16 stored contiguously, so one fragment load is one coalesced 1 KiB read instead of 16 half-used
// cache lines of a row-major matrix.  Element (row < 64, k) lives at
//   ((k / 32 * 4 + row / 16) * 64 + (k % 32 / 8) * 16 + row % 16) * 8 + k % 8
__host__ __device__ inline size_t vlo_pack64_elem(int row, int k) {
    return ((size_t)((k >> 5) * 4 + (row >> 4)) * 64 + (size_t)(((k & 31) >> 3) * 16 + (row & 15))) * 8 + (k & 7);
}

// h[m] (+)= bf16(sum_s partial[s][m]) ; x[m] = RMSNorm(h[m]) * w     (rows m < n).  ldx == 0: x is written packed-64.
hipError_t add_rmsnorm_launch(unsigned short *h, const float *partial, int ksplit, int partial_ld,
                              const unsigned short *w, unsigned short *x, int H, int ldx, float eps, int n,
                              hipStream_t st);

// copy `rows` embedding rows into the residual stream h and write their sums of squares to sq_out[0..rows)

// chunk attention (n <= 16 queries at positions pos0..pos0+n-1 against keys [0, pos0+n); the block path passes n <= 64
// = up to four 16-query sub-chunks in one launch)
// pack_row0 >= 0: `out` is a packed-64 matrix and query i goes to row pack_row0 + i (block path); -1: row-major [n][nh*hd]
// part_cap: capacity of part_o / part_ml in (16-query sub-chunk x split) partial states of [nh][16][hd] / [nh][16][2] floats: the
// session's buffers hold VLO_MAX_SPLITS (n <= 64); the prefill path brings VLO_PREFILL_TOKENS / 16 and runs a whole block of new
// tokens as ONE launch (grid.z = its 16-query sub-chunks, one split each)
hipError_t attention_launch(const unsigned short *q, KvGeom kv, int layer, int num_heads, int64_t pos0, int n,
                            float *part_o, float *part_ml, unsigned short *out, hipStream_t st, int pack_row0 = -1, int part_cap = VLO_MAX_SPLITS);

// flash-style attention for a block of n new tokens at positions pos0 .. pos0 + n - 1 whose keys are already appended (prefill path):
// out bf16 [n][nh * hd] row-major.  hipErrorNotSupported for GQA shapes it is not instantiated for (the caller uses attention_launch).
bool attention_prefill_supported(int head_dim, int gqa_group);      // is attn_prefill_kernel instantiated for this shape (else attention_prefill_launch returns hipErrorNotSupported)
hipError_t attention_prefill_launch(const unsigned short *q, KvGeom kv, int layer, int num_heads, int64_t pos0, int n, unsigned short *out,
                                    hipStream_t st);

// launch geometry of the chunk attention for n new tokens at cache length pos0 (what attention_launch computes first)
struct AttnGeom { int G, KS, hpw, nhg, nz, chunk, nsplit, nct; float scale; size_t lds_bytes; };
hipError_t attention_geometry(const KvGeom &kv, int num_heads, int64_t pos0, int n, AttnGeom *g, int part_cap = VLO_MAX_SPLITS);

hipError_t embed_gather_launch(const unsigned short *table, const int64_t *ids, int k, int H, int64_t vocab,
                               unsigned short *out, hipStream_t st);

// samplers on bf16 logits [V]
#define VLO_SAMPLE_SCRATCH_FLOATS 512     // >= 6 * SAMPLE_BLOCKS + 1
// `scratch`: VLO_SAMPLE_SCRATCH_FLOATS floats of device memory owned by the session
hipError_t greedy_sample_launch(const unsigned short *logits, int V, int64_t *tok_out, int eos, int force_mode,
                                float *scratch, hipStream_t st);
hipError_t stream_sample_launch(const unsigned short *logits, int V, float threshold, int interval_id,
                                int64_t *tok_out, float *p_interval_out, float *scratch, hipStream_t st);

#define VLO_STEP_IDS_MAX 32
struct StepIds { int64_t v[VLO_STEP_IDS_MAX]; };      // token ids handed to step_input_kernel by value (kernel arguments)
hipError_t step_input_launch(const unsigned short *table, const StepIds &ids, int k, const unsigned short *frame_rows, int rows, int H,
                             int64_t vocab, unsigned short *out, hipStream_t st);
hipError_t copy_rows_launch(const unsigned short *src, unsigned short *dst, int rows, int H, hipStream_t st);
hipError_t read_kv_launch(KvGeom kv, int layer, int which, int kv_head, int64_t t0, int64_t t1, unsigned short *dst,
                          hipStream_t st);

// ---- teacher-forced evaluation helpers (models/modeling_live.py:29-42, 44-168, 170-171)
// out[i] = ids[i] == v_id ? frame_rows[rank of i among placeholder positions] : table[clamp(ids[i])];
// *count_out (device int) = number of placeholder positions; src_idx_scratch: k device ints
hipError_t joint_embed_launch(const unsigned short *table, const int64_t *ids, int k, int64_t v_id, const unsigned short *frame_rows,
                              int n_frame_rows, int H, int64_t vocab, int *src_idx_scratch, int *count_out, unsigned short *out,
                              hipStream_t st);
// copy `pages` whole KV pages (all layers, K and V^T) from the physical pages src_pt[i] to dst_pt[i] (device int arrays)
hipError_t kv_copy_pages_launch(KvGeom kv, const int *src_pt, const int *dst_pt, int pages, int layers, hipStream_t st);
// per-row logit statistics, see llm_ops.hip
hipError_t logit_rows_launch(const unsigned short *logits, int n, int V, int64_t ld, const int64_t *labels, int interval_id, float *lse,
                             int64_t *amax, float *label_logit, float *p_interval, int64_t *p_amax, hipStream_t st);
