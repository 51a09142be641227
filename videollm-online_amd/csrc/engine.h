// engine.h — internal structures behind the opaque vlo_engine / vlo_session handles
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vlo.h"
#include "gemv.h"
#include "prefill.h"

struct RawTensor {
    void *ptr = nullptr;
    int dtype = 0;
    std::vector<int64_t> shape;
};

struct PackedLinear {
    void *Wp = nullptr;
    int N = 0, K = 0, NT = 0;
    int NT_gemm = 0;          // tiles the image is ALLOCATED (and zero-filled) for: NT rounded up to 16 where a GEMM reads whole 256-column tiles of a
                              // width that is not one (a TP rank's lm_head shard: 16 032 -> 16 128 columns); 0 = NT
    int wq = 0;               // 1: Wp is the fp8 e4m3 image (gemv.hip) and wscale holds the per-output-channel scales
    float *wscale = nullptr;  // fp32 [NT * 16], packed row order
    GemvPlan plan{};
    Gemm64Plan plan64{};      // the 64-token block path over the same packed image (prefill.hip)
};

struct LayerWeights {
    PackedLinear qkv, o, gate_up, down;
    void *ln_in = nullptr, *ln_post = nullptr;
};

struct VitState;

// Prefill workspaces (~275 MB per set at the 8B shape, + the fp8 scratches, + 64 MiB of partials under TP) are pooled per engine: a destroyed
// session hands its set back, the next session that needs one takes it.  The pool keeps at most VLO_PREFILL_POOL_MAX idle sets — more are freed
// on release — so the resident cost is bounded by (live sessions that ran a prefill) + 2 sets.  A pooled set's spare rows hold whatever the
// last block left there: finite or not, they are computed and dropped (no output row depends on another row's X).
#define VLO_PREFILL_POOL_MAX 2
struct PrefillWs {
    unsigned short *ph = nullptr, *px = nullptr, *pqkv = nullptr, *pq = nullptr, *pact = nullptr;
    void *wexp = nullptr;
    size_t wexp_bytes = 0;
    float *partial = nullptr;
    void *xq = nullptr;
};

struct vlo_engine {
    vlo_config cfg{};
    int device = 0;
    int head_dim = 0;
    // tensor-parallel shard of this engine (== the global dims when tp_size == 1): q heads, kv heads, MLP columns,
    // vocabulary rows owned by this rank (HF tp plan: q/k/v/gate/up column-wise, o/down row-wise, lm_head column-wise)
    int tp_rank = 0, tp_size = 1;
    int nh_l = 0, nkv_l = 0, I_l = 0, V_l = 0;
    bool finalized = false;
    bool has_connector = false;
    std::map<std::string, RawTensor> raw;      // row-major weights staged on the device until finalize
    std::vector<void *> owned;                 // device allocations freed at destroy
    int64_t weight_bytes = 0;

    std::vector<LayerWeights> layers;
    PackedLinear lm_head, conn0, conn2;
    void *norm_w = nullptr, *embed = nullptr, *conn0_b = nullptr, *conn2_b = nullptr;
    void *conn_x = nullptr, *conn_mid = nullptr, *conn_out = nullptr;   // connector scratch: 2 slots of 32 rows each (two encode branches may run concurrently)
    void *cos_tab = nullptr, *sin_tab = nullptr;
    int64_t max_positions = 0;

    // paged KV pool
    void *k_pool = nullptr, *vt_pool = nullptr;
    int pool_pages = 0;
    int64_t page_elems = 0, layer_stride = 0;
    std::vector<int> free_pages;
    std::mutex pool_mu;


    VitState *vit = nullptr;
    void *ingest = nullptr;                      // ingest.hip: cached tap tables + scratch of vlo_frame_ingest

    // prefill-path workspaces (~275 MB at the 8B shape, + the fp8 expansion scratch) are POOLED per engine: a session takes a set at its first
    // prefill block and hands it back when it is destroyed — stream_evaluate forks a session per turn, serving holds many sessions; each set is
    // allocated (hipMalloc + a blocking memset) once and owned by exactly one session at a time (sessions step on streams of their own).
    // Guarded by pool_mu; freed with the engine.
    std::vector<PrefillWs> prefill_free;

    // live timing of the dominant kernel (vlo_profile_*)
    int prof_stride = 0;
    int64_t prof_seen = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;   // pool
    size_t prof_used = 0;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
};

struct vlo_session {
    vlo_engine *e = nullptr;
    int64_t len = 0;
    bool has_logits = false;
    std::vector<int> pages;
    std::vector<void *> owned;
    unsigned short *h = nullptr, *x = nullptr, *act = nullptr, *attn = nullptr, *q = nullptr, *emb1 = nullptr;
    float *part_o = nullptr, *part_ml = nullptr, *partial = nullptr;
    float *sq[2] = {nullptr, nullptr};          // row sum-of-squares partials handed from EPI_RESID to XSRC_NORM
    unsigned short *logits = nullptr, *last_logits = nullptr;
    unsigned short *logits_local = nullptr;      // TP: this rank's vocabulary shard [16][V_l]
    float *partial_o = nullptr;                  // TP: o_proj partial sums [16][H] awaiting the all-reduce
    int64_t *tok = nullptr;
    float *sample_scratch = nullptr;
    int64_t *host_tok = nullptr;                 // pinned, 8 slots: tokens read back by the greedy loop (double-buffered)
    hipEvent_t tok_ev[2] = {nullptr, nullptr};   // "token i is on the host" (created on first use)
    int *page_table = nullptr, *host_pt = nullptr;
    // workspaces of the 64-token block path (allocated on first use): residual stream, normed rows, q, attention out, MLP act
    unsigned short *bh = nullptr, *bx = nullptr, *bq = nullptr, *battn = nullptr, *bact = nullptr;
    // workspaces of the prefill path (blocks of up to VLO_PREFILL_TOKENS tokens as real GEMMs, prefill.h; allocated on first use):
    // residual stream, normed rows / attention output (the GEMMs' X operand: 256 spare rows), qkv projection, q after RoPE, MLP act
    unsigned short *ph = nullptr, *px = nullptr, *pqkv = nullptr, *pq = nullptr, *pact = nullptr;
    float *ppart_o = nullptr, *ppart_ml = nullptr;   // attention partials of a whole prefill block (VLO_PREFILL_TOKENS / 16 sub-chunk states): only for the fallback kernel
    // prefill path of an fp8 engine: bf16 expansion of ONE projection's image (the largest), rewritten before each GEMM.  Per SESSION: sessions
    // step on streams of their own, a scratch shared through the engine would be overwritten under another session's GEMM
    void *pf_wexp = nullptr;
    size_t pf_wexp_bytes = 0;
    // prefill_act_dtype = 1: the e4m3 codes of ONE projection's X operand ([VLO_PREFILL_TOKENS + 256][widest layer K] bytes) followed by its
    // row scales ([VLO_PREFILL_TOKENS + 256] floats), rewritten before each GEMM (prefill.h::quantize_rows_fp8_launch)
    void *pxq = nullptr;
    unsigned short *plogits = nullptr;           // TP prefill, every row's logits: this rank's PADDED vocabulary shard of one row chunk, bf16 [VLO_TP_LOGIT_ROWS][NT_gemm * 16]
    float *ppartial = nullptr;                   // TP prefill: this rank's o-proj / down-proj partial sums fp32 [VLO_PREFILL_TOKENS][H] awaiting the all-reduce
};

int dev_alloc(void **p, size_t bytes);
// helpers shared with tp.hip
struct KvGeom;
int ensure_pages(vlo_session *s, int64_t new_len, hipStream_t st);
GemvArgs gemv_args(const PackedLinear &pl, const unsigned short *x, int ldx, int n_rows);
KvGeom kv_geom(const vlo_session *s);
int ensure_prefill_ws(vlo_session *s);                        // prefill-path workspaces of a session (sized for the engine's shard)
bool prefill_ok(const vlo_engine *e);                         // do this engine's (shard) shapes take the prefill GEMMs
// layer_proj: a decoder-layer projection (takes the native fp8 MFMA when the engine asks for it); false = the lm_head (bf16 activations always)
int prefill_gemm(vlo_session *s, const unsigned short *X, const PackedLinear &pl, int m, int N, int K, void *out, int ldo, int kind, hipStream_t st,
                 bool layer_proj = true);
void ingest_create(vlo_engine *e);
void ingest_destroy(vlo_engine *e);
int connector_run(vlo_engine *e, int slot, const void *feats_dev, int rows, void *out_dev, hipStream_t st);   // slot 0 / 1: scratch set
int session_fork_shard(vlo_session *src, int64_t n_tokens, vlo_session **out, void *stream);   // one KV shard (engine.hip)
int session_crop_shard(vlo_session *s, int64_t n_tokens);
int vlo_fail(int code, const std::string &msg);   // sets the thread-local error string, returns code
