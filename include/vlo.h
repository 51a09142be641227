/*
 * vlo.h — C ABI of the MI355X-native streaming video-LLM engine (libvlo.so).
 *
 * Drop-in boundary for the ONE hot path of showlab/videollm-online
 * (BASELINE.json north_star): per-frame SigLIP ViT encode ->
 * connector -> Llama streaming step over a growing KV cache -> samplers.
 * The reference has no FFI; its boundary is the duck-typed Python surface that
 * demo/inference.py uses on `self.model`.  Each entry point below names the
 * reference call it replaces (paths relative to the reference tree; HF: paths
 * relative to site-packages/transformers).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / HIP types
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream).
 *   - every function returning int: 0 = ok, < 0 = VLO_E_*; message via
 *     vlo_last_error() (thread-local).  No exceptions cross the ABI.
 *   - *_dev pointers are device pointers on the engine's GPU; inputs are
 *     borrowed for the (stream-ordered) duration of the call, outputs are
 *     written into caller-provided device buffers (the reference's
 *     `inplace_output_ids` convention, models/modeling_live.py:173-182).
 *   - weights are copied and re-packed into engine-owned HBM at load time;
 *     the caller may free its copies afterwards.
 *   - a session is NOT thread-safe; an engine may host many sessions.
 */
#ifndef VLO_H
#define VLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLO_ABI_VERSION 3

enum {
    VLO_OK = 0,
    VLO_E_INVALID = -1,     /* bad argument / shape */
    VLO_E_HIP = -2,         /* HIP runtime error (message has the hipError string) */
    VLO_E_NOMEM = -3,       /* KV pool or device memory exhausted */
    VLO_E_STATE = -4,       /* call out of order (e.g. step before finalize) */
    VLO_E_MISSING = -5,     /* weight not loaded */
    VLO_E_UNSUPPORTED = -6  /* shape the kernels do not cover */
};

enum { VLO_DT_F32 = 0, VLO_DT_BF16 = 1, VLO_DT_F16 = 2, VLO_DT_FP8_E4M3 = 3 /* OCP e4m3fn, one byte per element */ };

typedef struct vlo_engine vlo_engine;
typedef struct vlo_session vlo_session;

/* Mirrors the fields LiveInfer reads from model.config (demo/inference.py:19-32;
 * models/configuration_live.py:4-21) plus the HF LlamaConfig / SiglipVisionConfig
 * dimensions the arithmetic needs. */
typedef struct vlo_config {
    int32_t abi_version;          /* = VLO_ABI_VERSION */
    /* Llama */
    int32_t hidden_size;
    int32_t intermediate_size;
    int32_t num_layers;
    int32_t num_heads;
    int32_t num_kv_heads;
    int32_t vocab_size;
    float   rope_theta;
    float   rms_eps;
    /* connector / frame tokens */
    int32_t vision_hidden_size;   /* LiveConfigMixin.vision_hidden_size */
    int32_t frame_num_tokens;     /* 1 (CLS) + pool_h*pool_w */
    /* SigLIP vision tower (has_vit = 0: LLM only, vlo_visual_embed unavailable).  Shapes: hidden % 64 == 0 and <= 2048, head dim 64
     * (SigLIP-L/16-384, what the reference accepts: models/vision_live.py:56-60) or 68..80 in steps of 4 (SigLIP-so400m/14-384: 72),
     * any MLP width % 4 == 0, any patch size; image_size / patch_size rounds down like the strided conv (HF modeling_siglip.py:175-186) */
    int32_t has_vit;
    int32_t vit_hidden_size;
    int32_t vit_intermediate_size;
    int32_t vit_num_layers;
    int32_t vit_num_heads;
    int32_t vit_image_size;       /* frame_resolution */
    int32_t vit_patch_size;
    float   vit_ln_eps;
    int32_t pool_h, pool_w;       /* frame_token_pooled */
    /* KV pool: total tokens the paged pool can hold across all sessions */
    int64_t kv_pool_tokens;
    /* tensor parallel: this engine holds rank tp_rank's shard of a tp_size-way group (0/1 = none); see vlo_tp_* */
    int32_t tp_rank, tp_size;
    /* storage of the streamed Llama projections (q/k/v/o, gate/up/down, lm_head): 0 = bf16; 1 = fp8 e4m3 with one fp32 scale
     * per output channel (BASELINE.json configs[4] "fp8 MFMA weights").  With 1, vlo_engine_load_weight takes those matrices as
     * VLO_DT_FP8_E4M3 [N][K] plus "<name>_scale" f32 [N] (W ~= q * scale[n]); activations, KV and accumulation are unchanged. */
    int32_t weight_dtype;
    /* X operands of the LONG-INPUT projections (>= 256 new tokens: teacher-forced evaluation, a long first prompt) on an engine with
     * weight_dtype = 1: 0 = bf16 (the e4m3 image is expanded to bf16 per GEMM, bf16 MFMA: the arithmetic of the live step); 1 = every X row
     * quantised to e4m3 with one fp32 scale (max|row| / 448) and multiplied e4m3 x e4m3 on v_mfma_f32_16x16x128_f8f6f4 (W8A8; twice the MFMA
     * rate; own parity band: the oracle run with the same activation rule).  The live step (<= 16 rows) and the 64-token block path keep
     * bf16 activations either way. */
    int32_t prefill_act_dtype;
} vlo_config;

/* ---- engine lifetime: replaces build_model_and_tokenizer(...)[0] + model.to('cuda')
 *      (demo/inference.py:15-16; models/modeling_live.py:184-222) ------------------ */
int  vlo_engine_create(const vlo_config *cfg, int device, vlo_engine **out);
/* name = HF state-dict key ("model.layers.0.self_attn.q_proj.weight", "lm_head.weight",
 * "connector.0.bias", "vision.encoder.layers.0.mlp.fc1.weight", ... ); optional
 * "rope.inv_freq" [head_dim/2] f32 overrides the engine's own powf() table.
 * data may be a host or a device pointer.  */
int  vlo_engine_load_weight(vlo_engine *e, const char *name, const void *data, int dtype,
                            const int64_t *shape, int ndim);
int  vlo_engine_finalize(vlo_engine *e);              /* verifies completeness, builds tables */
void vlo_engine_destroy(vlo_engine *e);               /* destroy the engine's sessions FIRST: they borrow its KV pool */
int64_t vlo_engine_weight_bytes(const vlo_engine *e); /* packed bytes resident in HBM */

/* ---- session = the KV handle (`past_key_values`, HF:cache_utils.py DynamicCache;
 *      demo/inference.py:61,69-70,84-91) -------------------------------------------- */
int     vlo_session_create(vlo_engine *e, int64_t max_tokens_hint, vlo_session **out);
int     vlo_session_reset(vlo_session *s);            /* LiveInfer.reset(): past_key_values=None */
int64_t vlo_session_len(const vlo_session *s);        /* DynamicCache.get_seq_length() */
void    vlo_session_destroy(vlo_session *s);

/* ---- model.visual_embed(frames) (models/modeling_live.py:21-27 ->
 *      models/vision_live.py:10-30 -> HF SiglipVisionModel -> connector
 *      models/live_llama/modeling_live_llama.py:18-22).
 *      frames_dev: uint8 [B,3,R,R] NCHW;  out_dev: bf16 [B*frame_num_tokens, hidden_size].
 *      The encode workspace belongs to the ENGINE: encodes of one engine issued on different streams must be ordered by the caller
 *      (LiveInfer issues all of them on its one encode stream).  From 4 frames up the batch runs as two parallel half-batch branches
 *      (the second on an internal stream, joined back before the call's work on `stream` ends). */
int vlo_visual_embed(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, void *stream);
/* ---- vision tokens only: `vision_encode(model, frames)` as the offline feature extraction uses it
 *      (data/utils.py:86-104 distributed_encode -> models/vision_live.py:10-30).
 *      frames_dev: uint8 [B,3,R,R];  out_dev: bf16 [B, frame_num_tokens, vision_hidden_size] (CLS + pooled, PRE-connector;
 *      the reference stores them with `.to(torch.bfloat16)`, data/utils.py:101) */
int vlo_vision_tokens(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, void *stream);
/* ---- frame preparation in front of `load_video`: what the reference does with an external ffmpeg
 *      (data/utils.py:51-66 `ffmpeg_once`: scale the longer side to R keeping the aspect ratio — the other side a multiple of
 *      2 —, -sws_flags bicubic, pad to R x R with black; demo/cli.py:13-22) followed by read_video(..., 'TCHW')
 *      (demo/inference.py:112).  src_dev: decoded uint8 RGB frames, layout 0 = [T,H,W,3] (decoder output), 1 = [T,3,H,W];
 *      resolution: R, or 0 for the vision tower's image size; cubic_a: Keys parameter of the antialiased bicubic (-0.6 =
 *      libswscale's default B=0,C=0.6; -0.5 = PIL / torch antialias); out_dev: uint8 [T,3,R,R].
 *      Thread-safe and stream-safe: the engine's one fp32 scratch is handed from call to call through an event (a call's kernels
 *      start after the previous call's kernels, whichever streams and host threads the two calls came from); src_dev / out_dev
 *      must stay valid until `stream` has run the call. */
int vlo_frame_ingest(vlo_engine *e, const uint8_t *src_dev, int T, int H, int W, int layout, int resolution, float cubic_a,
                     uint8_t *out_dev, void *stream);
/* host-only: scaled size (ow, oh) and pad offset (x0, y0) of the call above for W x H frames (any pointer may be NULL) */
int vlo_frame_ingest_geometry(int W, int H, int R, int *ow, int *oh, int *x0, int *y0);

/* connector only (frames already encoded: the `hasattr(self,'vision_encode')` false branch,
 * models/modeling_live.py:22-26).  feats_dev: bf16 [rows, vision_hidden_size] */
int vlo_connector(vlo_engine *e, const void *feats_dev, int rows, void *out_dev, void *stream);

/* ---- model.get_input_embeddings()(ids) (demo/inference.py:46,66).
 *      ids_dev: int64 [k]; out_dev: bf16 [k, hidden_size] */
int vlo_embed(vlo_engine *e, const int64_t *ids_dev, int k, void *out_dev, void *stream);

/* ---- the step input of a frame step, `torch.cat([embed(last_ids), frame_embeds])` (demo/inference.py:61-68), in one launch
 *      into a caller-owned staging buffer.  ids_host: int64 [k] in HOST memory (the reference holds them as Python ints and
 *      uploads them with torch.tensor(..., device='cuda'); here they ride in the kernel arguments);
 *      frame_rows_dev: bf16 [rows, hidden_size] (connector output of this frame) or NULL when rows == 0;
 *      out_dev: bf16 [k + rows, hidden_size]. */
int vlo_step_input(vlo_engine *e, const int64_t *ids_host, int k, const void *frame_rows_dev, int rows, void *out_dev, void *stream);

/* ---- model(inputs_embeds=..., use_cache=True, past_key_values=...) (demo/inference.py:69;
 *      models/live_llama/modeling_live_llama.py:24-53 -> HF LlamaForCausalLM.forward).
 *      embeds_dev: bf16 [n, hidden_size].  Appends n tokens to the session's KV.
 *      last_logits_dev: NULL or bf16 [vocab] receiving logits[:, -1] (the only row any
 *      caller reads: demo/inference.py:76, models/modeling_live.py:177).
 *      all_logits_dev: NULL or bf16 [n, vocab] (full HF output, for parity tests). */
int vlo_llm_step(vlo_session *s, const void *embeds_dev, int n, void *last_logits_dev,
                 void *all_logits_dev, void *stream);

/* ---- streaming sampler (demo/inference.py:76-81) on the session's last-row logits:
 *      softmax (model dtype), zero p[interval] if below threshold, argmax.
 *      tok_dev: int64 [1]; p_interval_dev: NULL or float [1] (p before zeroing). */
int vlo_stream_sample(vlo_session *s, float threshold, int interval_id, int64_t *tok_dev,
                      float *p_interval_dev, void *stream);

/* ---- fast_greedy_generate (models/modeling_live.py:173-182): up to max_new decode steps,
 *      token i written to out_ids_dev[i] (int64), stops after writing eos.
 *      force_len > 0: scheduled mode for throughput runs — argmax still computed each step,
 *      but exactly force_len tokens are produced and the last is eos.
 *      *n_written (host) receives i+1.  Synchronises the stream (the reference syncs once
 *      per token at :179).  After a response that ended with eos the session holds NO logits
 *      (vlo_stream_sample then returns VLO_E_STATE until the next vlo_llm_step), whichever
 *      internal path produced the eos. */
int vlo_greedy_generate(vlo_session *s, const void *embeds_dev, int m, int eos_token_id,
                        int64_t *out_ids_dev, int max_new, int force_len, int *n_written, void *stream);

/* ---- introspection for tests / bench ------------------------------------------------ */
/* copy the session's K or V for (layer, kv_head) tokens [t0,t1) to dst_dev bf16 [t1-t0, head_dim] */
int vlo_session_read_kv(vlo_session *s, int layer, int which /*0=K,1=V*/, int kv_head, int64_t t0, int64_t t1,
                        void *dst_dev, void *stream);
/* algorithmic bytes of a step with n new tokens at cache length Lc */
double vlo_step_algorithmic_bytes(const vlo_engine *e, int64_t Lc, int n);
/* raw skinny-GEMM entry used by unit tests: y[n,N] (f32) = x[n,K](bf16) @ W[N,K]^T (bf16), n<=16.
 * W_dev is an ordinary row-major device tensor; packs on every call (tests only). */
int vlo_test_gemv(const void *x_dev, const void *W_dev, float *y_dev, int n, int N, int K, void *stream);
/* the same through the fp8 e4m3 weight image: Wq_dev fp8 [N][K], scale_dev f32 [N]; y = (x @ Wq^T) * scale */
int vlo_test_gemv_fp8(const void *x_dev, const void *Wq_dev, const float *scale_dev, float *y_dev, int n, int N, int K, void *stream);
/* the long-input W8A8 GEMM of an engine with prefill_act_dtype = 1 (csrc/prefill.h): x bf16 [M][K] quantised per row to e4m3, multiplied with
 * Wq fp8 [N][K] on the native fp8 MFMA; y f32 [M][N] = (xq @ Wq^T) * scale[n] * xscale[m].  N, K multiples of 256.  Optional outputs: xq_dev
 * [M][K] e4m3 codes in the kernel's row order (prefill.h::vlo_fp8_row_pos), xscale_dev f32 [M]; iters > 0: *avg_us = the GEMM alone, timed. */
int vlo_test_gemm_fp8(const void *x_dev, const void *Wq_dev, const float *scale_dev, float *y_dev, void *xq_dev, float *xscale_dev, int M, int N, int K,
                      int iters, double *avg_us, void *stream);

/* ---- teacher-forced evaluation: the arithmetic of LiveMixin.joint_embed / stream_evaluate /
 *      trim_past_key_values (models/modeling_live.py:29-42, 44-168, 170-171).  The per-turn bookkeeping over these
 *      per-row numbers stays on the host (videollm-online_amd/modeling_live.py::LiveModel.stream_evaluate).
 *
 * joint_embed (:29-42): out[i] = ids[i] == v_placeholder_id ? frame_rows[r_i] : embed_tokens[min(ids[i], vocab-1)], r_i =
 *      number of placeholder positions before i.  ids_dev int64 [k]; frame_rows_dev bf16 [n_frame_rows, hidden] (the
 *      output of vlo_visual_embed / vlo_connector); out_dev bf16 [k, hidden].  Synchronises the stream; fails with
 *      VLO_E_INVALID when the number of placeholder positions != n_frame_rows (the reference's masked assignment raises). */
int vlo_joint_embed(vlo_engine *e, const int64_t *ids_dev, int k, int64_t v_placeholder_id, const void *frame_rows_dev,
                    int n_frame_rows, void *out_dev, void *stream);
/* per-row statistics of logits_dev bf16 [n, vocab] (e.g. vlo_llm_step's all_logits_dev): lse (fp32 log-sum-exp; cross
 *      entropy :95 = lse - label_logit), argmax (:97), label_logit = logits[r][labels[r]] (labels_dev int64 [n] or NULL;
 *      0 for labels outside the vocabulary), p_interval = bf16-rounded softmax probability of interval_id (:107-110),
 *      p_argmax = argmax of the bf16-rounded softmax row (:112, :144).  All outputs are device arrays of n elements. */
int vlo_logit_rows(vlo_engine *e, const void *logits_dev, int n, const int64_t *labels_dev, int interval_id, float *lse_dev,
                   int64_t *argmax_dev, float *label_logit_dev, float *p_interval_dev, int64_t *p_argmax_dev, void *stream);
/* trim_past_key_values(past, 0, n_tokens) (:170-171) as the reference uses it (:118, :134): a NEW session holding a copy
 *      of the first n_tokens positions; the source keeps its full length.  Copies whole KV pages on `stream`. */
int vlo_session_fork(vlo_session *src, int64_t n_tokens, vlo_session **out, void *stream);
/* in-place variant: forget every position >= n_tokens and return the freed pages to the pool (device-synchronising). */
int vlo_session_crop(vlo_session *s, int64_t n_tokens);

/* ---- tensor parallelism (north_star; new capability — the reference has none) -------------------
 * Engines created with vlo_config.tp_size = T > 1 / tp_rank = r hold rank r's shard (load_weight still takes the FULL
 * tensors and slices them).  They are stepped through a group:
 *   - one process per GPU (torchrun): n_local = 1, `rccl_unique_id` = the 128 bytes produced by vlo_tp_unique_id() on
 *     rank 0 and broadcast by the host; exchanges are RCCL all-reduce / all-gather on the caller's stream;
 *   - single process: n_local = T engines on ONE device (logical ranks; validates the sharding without a multi-GPU box),
 *     `rccl_unique_id` = NULL; exchanges are device kernels.
 * The tp_* calls mirror vlo_session_* / vlo_llm_step / vlo_stream_sample / vlo_greedy_generate; embeddings and the
 * vision tower are replicated (use vlo_embed / vlo_visual_embed on any local engine). */
typedef struct vlo_tp_group vlo_tp_group;
typedef struct vlo_tp_session vlo_tp_session;
int  vlo_tp_unique_id(void *out128);
/* one-rank RCCL round trip (communicator, fp32 all-reduce, byte all-gather) on `device`: checks the run-time binding of
 * librccl that the one-process-per-GPU mode relies on, on a box with a single GPU */
int  vlo_tp_selftest(int device);
int  vlo_tp_group_create(vlo_engine **engines, int n_local, const void *rccl_unique_id, vlo_tp_group **out);
void vlo_tp_group_destroy(vlo_tp_group *g);
int  vlo_tp_session_create(vlo_tp_group *g, int64_t max_tokens_hint, vlo_tp_session **out);
int  vlo_tp_session_reset(vlo_tp_session *t);
int64_t vlo_tp_session_len(const vlo_tp_session *t);
void vlo_tp_session_destroy(vlo_tp_session *t);
/* trim_past_key_values(past, 0, n) (models/modeling_live.py:170-171) for a tensor-parallel KV handle: every local rank's shard is
 * forked / cropped alike (vlo_session_fork / vlo_session_crop per shard); one process per GPU: every rank makes the same call */
int  vlo_tp_session_fork(vlo_tp_session *src, int64_t n_tokens, vlo_tp_session **out, void *stream);
int  vlo_tp_session_crop(vlo_tp_session *t, int64_t n_tokens);
int  vlo_tp_llm_step(vlo_tp_session *t, const void *embeds_dev, int n, void *last_logits_dev, void *all_logits_dev, void *stream);
int  vlo_tp_stream_sample(vlo_tp_session *t, float threshold, int interval_id, int64_t *tok_dev, float *p_interval_dev, void *stream);
int  vlo_tp_greedy_generate(vlo_tp_session *t, const void *embeds_dev, int m, int eos_token_id, int64_t *out_ids_dev, int max_new,
                            int force_len, int *n_written, void *stream);

/* One-shot peer-to-peer all-reduce over xGMI, fused with the residual add + RMSNorm that consumes it (opt-in replacement
 * of the two RCCL all-reduces per layer and of the logits all-gather; SURVEY.md §8e "xGMI mapping").  Every rank owns a
 * mailbox in its HBM that all peers map; an exchange is one posted write per peer and one hop of latency instead of a
 * ring's 2(T-1).  No RCCL communicator is needed: vlo_tp_group_create may then be given rccl_unique_id = NULL.
 *   one process per GPU: every rank calls vlo_tp_p2p_export (64-byte hipIpc handle of its mailbox), the host gathers
 *       the T handles in rank order (e.g. torch.distributed.all_gather_object) and every rank calls
 *       vlo_tp_p2p_enable(g, handles) with handles = [T][64 bytes];
 *   single process (T logical ranks on one device): vlo_tp_p2p_enable(g, NULL).
 * All ranks must issue the same sequence of steps (they already do: the step is lock-step).  A rank that waits longer
 * than VLO_TP_P2P_TIMEOUT_MS (default 2000) for a peer raises a sticky error: the stream drains, the next
 * vlo_tp_llm_step fails with VLO_E_HIP, vlo_tp_p2p_status reports timed_out = 1. */
int  vlo_tp_p2p_export(vlo_tp_group *g, void *out_handle64);
int  vlo_tp_p2p_enable(vlo_tp_group *g, const void *handles);
int  vlo_tp_p2p_status(vlo_tp_group *g, int *enabled, int *timed_out, int *uncached_mailbox);
/* micro-benchmark of ONE tensor-parallel exchange (all-reduce of [m][hidden] fp32 partial sums + residual add + RMSNorm, what a
 * decoder layer issues twice), `iters` back to back on `stream`, average microseconds; RCCL or peer-to-peer, whichever the group
 * uses.  Every rank of the group makes the same call.  The session's residual stream is scratch afterwards (reset it). */
int  vlo_tp_bench_exchange(vlo_tp_session *t, int m, int iters, double *avg_us, void *stream);
/* what RCCL reports for the group's communicator (ncclCommCount / ncclCommUserRank): 0 / -1 when the group has none (logical
 * ranks, peer-to-peer exchange).  The RCCL library is dlopen'ed; VLO_RCCL_LIBRARY names a specific build. */
int  vlo_tp_comm_info(vlo_tp_group *g, int *nranks, int *rank);
/* byte all-gather over the group's RCCL communicator (one process per GPU): recv = rank 0's bytes | rank 1's | ...  Used for the
 * frame-parallel vision tower under TP: rank r encodes frames r, r + T, ... of a pending batch (models/modeling_live.py:21-27 per
 * frame), the gather gives every rank every frame's [frame_num_tokens, hidden] embedding — BASELINE.json north_star's "broadcasting
 * the 10-token frame embedding each step" (81 920 B per frame for Llama-3-8B). */
int  vlo_tp_allgather(vlo_tp_group *g, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *stream);
/* host-side mailbox geometry of the exchange above (no GPU needed; unit tests): for a group of T ranks, hidden size H,
 * vocabulary shard Vl, the seq-th exchange of a region (seq counts from 0 per region) and the tag `epoch` of the previous
 * exchange, out6 = {first granule of the reduce slot, granules between two sources of a reduce slot, first granule of the
 * gather slot, granules between two sources of a gather slot, mailbox size in granules, the next epoch} */
int  vlo_debug_p2p_layout(int T, int H, int Vl, unsigned seq, unsigned epoch, int64_t *out6);

/* live kernel timing for bench.py's roofline: when enabled, every `stride`-th launch of the dominant
 * kernel (the gate/up weight-streaming GEMV, gemv16_kernel<KF,SWIGLU>) is bracketed by HIP events on the
 * stream it is launched on.  vlo_profile_read synchronises those events and returns the number of timed
 * launches, their total milliseconds and the algorithmic bytes of ONE launch. */
int vlo_profile_enable(vlo_engine *e, int stride);     /* stride <= 0 disables and clears */
int vlo_profile_read(vlo_engine *e, int64_t *launches, double *total_ms, double *bytes_per_launch);
/* average elapsed time of an EMPTY event bracket (two hipEventRecord back to back) on `stream`, in microseconds:
 * the fixed cost every bracket above includes on top of the kernel's own duration */
int vlo_profile_calibrate(vlo_engine *e, void *stream, double *empty_bracket_us);

/* host-side planner of the weight-streaming GEMV (no GPU needed): for a reduction length K returns
 * out4 = {waves per block, fragments per wave per chunk, K chunks per wave, K slices across blocks}; < 0 if K is not covered */
int vlo_debug_gemv_plan(int K, int allow_ksplit, int *out4);
/* block path (csrc/prefill.hip): (NW, KF, KC) chosen for a K; element offset of (row < 64, k) in the packed-64 layout */
int vlo_debug_gemm64_plan(int K, int *out3);
int64_t vlo_debug_pack64_elem(int row, int k);

/* micro-benchmark of the weight-streaming GEMV on synthetic data (tools/bench_gemv.py): `nbuf` distinct
 * packed weight images are cycled so the 256 MiB Infinity Cache cannot serve re-reads. */
int vlo_bench_gemv(int N, int K, int n_rows, int epi, int iters, int nbuf, double *avg_us);

/* debugging aid for tests: copy an internal per-session activation buffer of the LAST chunk to dst_dev.
 * which: 0 = q [16][nh*hd] bf16, 1 = attention output [16][nh*hd] bf16, 2 = residual stream h [16][H] bf16,
 *        3 = MLP activation [16][I] bf16, 4 = normed x [16][H] bf16 */
int vlo_debug_read(vlo_session *s, int which, void *dst_dev, int64_t bytes, void *stream);

const char *vlo_last_error(void);
int vlo_abi_version(void);
/* "VLO_BUILD_ID=<sha256 of every source and header the library was compiled from>" (videollm-online_amd/build.py stamps it;
 * _C.py refuses a library whose id differs from the sources lying next to it) */
const char *vlo_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* VLO_H */
