// tp.hip — tensor-parallel Llama step (north_star: shard the attention / MLP projections across the GPUs of a
// node, all-reduce over xGMI).  New capability: the reference has no TP and no collectives.
//
// Sharding (HF's plan, HF:models/llama/configuration_llama.py:49-57): q/k/v and gate/up column-wise (a rank owns
// nh/T query heads, nkv/T kv heads and I/T MLP columns, and ONLY its kv heads' pages of the KV cache), o_proj and
// down_proj row-wise (fp32 partial sums of [n, H]), lm_head column-wise (a vocabulary shard).  Per decoder layer
// there are exactly two exchanges — all-reduce(sum) of the o_proj and of the down_proj partials, 16*H*4 B = 256 KiB
// fp32 at most (n*H*4 when the projection is not K-split) — plus one all-gather of the last-row logits shard per step.
// RMSNorm, residual stream, embeddings, ViT + connector and the samplers are replicated.
//
// A "group" owns this process's local ranks: all T of them in single-process mode (several logical ranks driven in
// lock-step by one host thread; on ONE device this is how the sharding arithmetic is validated without a multi-GPU
// box — exchanges are a sum / copy kernel), or exactly one in the one-process-per-GPU mode (exchanges are RCCL
// ncclAllReduce / ncclAllGather on the step's stream, communicator bootstrapped from a unique id that the host
// broadcasts with torch.distributed).  librccl is dlopen'ed so libvlo.so has no link-time dependency on it.
//
// Second exchange implementation, opt-in (vlo_tp_p2p_*): a ONE-SHOT peer-to-peer all-reduce over xGMI fused with the
// residual add + RMSNorm that consumes it.  SURVEY.md §8e: the messages are 16-180 KB, so a ring all-reduce (14 hops,
// per-link bound) pays latency 14 times for a wire time of ~1 us; here every rank writes its fp32 partial row straight
// into a mailbox in EVERY peer's HBM (7 posted writes over 7 distinct links) and then sums the T mailbox rows in rank
// order — one hop, and bit-identical sums on every rank.  See "p2p exchange" below.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/vlo.h"
#include "common.cuh"
#include "engine.h"
#include "gemv.h"
#include "llm_ops.h"
#include "prefill.h"
#include "tp_p2p.cuh"

#define TP_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return vlo_fail(VLO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------
struct NcclUid { char internal[128]; };
typedef int (*fn_ncclGetUniqueId)(NcclUid *);
typedef int (*fn_ncclCommInitRank)(void **, int, NcclUid, int);
typedef int (*fn_ncclAllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_ncclAllGather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_ncclCommDestroy)(void *);
typedef int (*fn_ncclCommCount)(void *, int *);
typedef int (*fn_ncclCommUserRank)(void *, int *);
typedef const char *(*fn_ncclGetErrorString)(int);
enum { kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0 };

struct Rccl {
    void *dl = nullptr;
    fn_ncclGetUniqueId GetUniqueId = nullptr;
    fn_ncclCommInitRank CommInitRank = nullptr;
    fn_ncclAllReduce AllReduce = nullptr;
    fn_ncclAllGather AllGather = nullptr;
    fn_ncclCommDestroy CommDestroy = nullptr;
    fn_ncclCommCount CommCount = nullptr;
    fn_ncclCommUserRank CommUserRank = nullptr;
    fn_ncclGetErrorString GetErrorString = nullptr;
};
static Rccl g_rccl;

static std::string rccl_err(const char *what, int code) {
    std::string m = std::string(what) + " failed (rccl code " + std::to_string(code) + ")";
    if (g_rccl.GetErrorString) m += std::string(": ") + g_rccl.GetErrorString(code);
    return m;
}

static int rccl_load() {
    if (g_rccl.dl) return VLO_OK;
    // VLO_RCCL_LIBRARY: a specific RCCL build (any library that exports the same entry points)
    const char *names[] = {getenv("VLO_RCCL_LIBRARY"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *dl = nullptr;
    for (const char *n : names) {
        if (!n || !*n) continue;
        dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (dl) break;
    }
    if (!dl) return vlo_fail(VLO_E_UNSUPPORTED, std::string("cannot dlopen librccl: ") + dlerror());
    g_rccl.GetUniqueId = (fn_ncclGetUniqueId)dlsym(dl, "ncclGetUniqueId");
    g_rccl.CommInitRank = (fn_ncclCommInitRank)dlsym(dl, "ncclCommInitRank");
    g_rccl.AllReduce = (fn_ncclAllReduce)dlsym(dl, "ncclAllReduce");
    g_rccl.AllGather = (fn_ncclAllGather)dlsym(dl, "ncclAllGather");
    g_rccl.CommDestroy = (fn_ncclCommDestroy)dlsym(dl, "ncclCommDestroy");
    g_rccl.GetErrorString = (fn_ncclGetErrorString)dlsym(dl, "ncclGetErrorString");
    g_rccl.CommCount = (fn_ncclCommCount)dlsym(dl, "ncclCommCount");
    g_rccl.CommUserRank = (fn_ncclCommUserRank)dlsym(dl, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return vlo_fail(VLO_E_UNSUPPORTED, "librccl lacks an expected symbol");
    g_rccl.dl = dl;
    return VLO_OK;
}

// ---- group / session ----------------------------------------------------------------------------------------
struct P2PState {
    bool allocated = false, enabled = false, uncached = false;
    int H = 0, Vh = 0;                     // granules per reduce row / per gather row (V_l / 2)
    size_t gat_base = 0, granules = 0;     // first granule of the gather region; mailbox size in granules
    unsigned long long *mbox[8] = {};      // the local ranks' own mailboxes (device memory of their GPU), by local index
    unsigned long long *peer[8][8] = {};   // [local index][global rank] -> that rank's mailbox as addressable from here
    std::vector<void *> ipc_opened;        // mappings returned by hipIpcOpenMemHandle
    unsigned epoch = 0;                    // tag of the latest exchange (never 0), advanced in lock-step by every rank
    unsigned n_reduce = 0, n_gather = 0;   // exchanges issued per mailbox region: slot = count & 1
    unsigned *err_dev[8] = {};             // per local rank: set by a spin that timed out (later kernels stop waiting)
    unsigned *err_host = nullptr;          // pinned, device-visible: the host reads it without synchronising
    long long timeout_ticks = 200000000;   // s_memrealtime ticks (100 MHz): 2 s
    bool fused = true;                     // one kernel publishes AND collects (one process per GPU only)
    bool direct = true;                    // o / down GEMVs with ksplit == 1 publish their sums from the epilogue (EPI_PARTIAL_MBOX); VLO_TP_P2P_DIRECT=0: tests' A/B
};

struct vlo_tp_group {
    std::vector<vlo_engine *> eng;     // local ranks, consecutive tp_rank values
    int tp_size = 1;
    void *comm = nullptr;              // RCCL communicator (one-process-per-GPU mode)
    P2PState p2p;
};
struct vlo_tp_session {
    vlo_tp_group *g = nullptr;
    std::vector<vlo_session *> ss;     // one KV shard per local rank
    unsigned short *gather_tmp = nullptr;   // [T][16][V_l] receive buffer of the logits all-gather
    unsigned short *gather_big = nullptr;   // RCCL, every row's logits of a long input: [T][VLO_TP_LOGIT_ROWS][padded V_l] (allocated on first use)
};

struct PtrList { float *p[8]; int n; };
__global__ void tp_sum_kernel(PtrList L, size_t count) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < L.n; ++r) s += L.p[r][i];
        for (int r = 0; r < L.n; ++r) L.p[r][i] = s;
    }
}

int vlo_tp_unique_id(void *out128) {
    if (!out128) return vlo_fail(VLO_E_INVALID, "null unique id buffer");
    int rc = rccl_load();
    if (rc) return rc;
    NcclUid id;
    if (g_rccl.GetUniqueId(&id) != 0) return vlo_fail(VLO_E_HIP, "ncclGetUniqueId failed");
    memcpy(out128, &id, sizeof(id));
    return VLO_OK;
}

// One-rank RCCL round trip on `device`: communicator from a fresh unique id, an fp32 sum all-reduce and a byte all-gather
// (the two collectives tp_chunk issues), results checked on the host.  Exercises the dlopen'ed entry points, their
// argument layout and the datatype / op enums on a box with a single GPU.
int vlo_tp_selftest(int device) {
    int rc = rccl_load();
    if (rc) return rc;
    TP_TRY(hipSetDevice(device));
    NcclUid id;
    if (g_rccl.GetUniqueId(&id) != 0) return vlo_fail(VLO_E_HIP, "ncclGetUniqueId failed");
    void *comm = nullptr;
    if (g_rccl.CommInitRank(&comm, 1, id, 0) != 0 || !comm) return vlo_fail(VLO_E_HIP, "ncclCommInitRank(nranks=1) failed");
    const int N = 4096;
    std::vector<float> h(N), back(N);
    for (int i = 0; i < N; ++i) h[i] = 0.25f * (float)(i % 97) - 3.f;
    float *d = nullptr;
    unsigned char *gsrc = nullptr, *gdst = nullptr;
    hipStream_t st = nullptr;
    int out = VLO_OK;
    auto cleanup = [&]() {
        if (st) hipStreamDestroy(st);
        if (d) hipFree(d);
        if (gsrc) hipFree(gsrc);
        if (gdst) hipFree(gdst);
        g_rccl.CommDestroy(comm);
    };
#define ST_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cleanup();                                                                            \
            return vlo_fail(VLO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
        }                                                                                         \
    } while (0)
    ST_TRY(hipStreamCreate(&st));
    ST_TRY(hipMalloc((void **)&d, N * sizeof(float)));
    ST_TRY(hipMalloc((void **)&gsrc, N));
    ST_TRY(hipMalloc((void **)&gdst, N));
    ST_TRY(hipMemcpyAsync(d, h.data(), N * sizeof(float), hipMemcpyHostToDevice, st));
    ST_TRY(hipMemcpyAsync(gsrc, h.data(), N, hipMemcpyHostToDevice, st));
    ST_TRY(hipMemsetAsync(gdst, 0, N, st));
    if (g_rccl.AllReduce(d, d, N, kNcclFloat32, kNcclSum, comm, st) != 0) out = vlo_fail(VLO_E_HIP, "ncclAllReduce failed");
    if (!out && g_rccl.AllGather(gsrc, gdst, N, kNcclInt8, comm, st) != 0) out = vlo_fail(VLO_E_HIP, "ncclAllGather failed");
    if (!out) {
        std::vector<unsigned char> gb(N);
        ST_TRY(hipMemcpyAsync(back.data(), d, N * sizeof(float), hipMemcpyDeviceToHost, st));
        ST_TRY(hipMemcpyAsync(gb.data(), gdst, N, hipMemcpyDeviceToHost, st));
        ST_TRY(hipStreamSynchronize(st));
        if (memcmp(back.data(), h.data(), N * sizeof(float)) != 0) out = vlo_fail(VLO_E_HIP, "one-rank all-reduce changed the data");
        else if (memcmp(gb.data(), h.data(), N) != 0) out = vlo_fail(VLO_E_HIP, "one-rank all-gather did not copy the data");
    }
#undef ST_TRY
    cleanup();
    return out;
}

int vlo_tp_group_create(vlo_engine **engines, int n_local, const void *rccl_unique_id, vlo_tp_group **out) {
    if (!engines || n_local <= 0 || !out) return vlo_fail(VLO_E_INVALID, "bad tp_group_create arguments");
    const int T = engines[0]->tp_size;
    for (int i = 0; i < n_local; ++i) {
        if (!engines[i] || !engines[i]->finalized) return vlo_fail(VLO_E_STATE, "tp engines must be finalized");
        if (engines[i]->tp_size != T || engines[i]->tp_rank != engines[0]->tp_rank + i)
            return vlo_fail(VLO_E_INVALID, "local engines must carry consecutive tp_rank values of one tp_size");
    }
    if (n_local != T && n_local != 1) return vlo_fail(VLO_E_INVALID, "a process drives either all ranks or exactly one");
    if (n_local == T)
        for (int i = 1; i < T; ++i)
            if (engines[i]->device != engines[0]->device)
                return vlo_fail(VLO_E_UNSUPPORTED, "single-process groups exchange through device kernels and must share one device; "
                                                   "use one process per GPU (RCCL) across devices");
    vlo_tp_group *g = new vlo_tp_group();
    g->eng.assign(engines, engines + n_local);
    g->tp_size = T;
    if (n_local == 1 && T > 1 && rccl_unique_id) {     // no unique id: the p2p exchange must be enabled before the first step
        int rc = rccl_load();
        if (rc) { delete g; return rc; }
        NcclUid id;
        memcpy(&id, rccl_unique_id, sizeof(id));
        int nrc = -1;
        if (hipSetDevice(engines[0]->device) != hipSuccess || (nrc = g_rccl.CommInitRank(&g->comm, T, id, engines[0]->tp_rank)) != 0) {
            delete g;
            return vlo_fail(VLO_E_HIP, rccl_err("ncclCommInitRank", nrc));
        }
    }
    *out = g;
    return VLO_OK;
}

int vlo_tp_comm_info(vlo_tp_group *g, int *nranks, int *rank) {
    if (!g) return vlo_fail(VLO_E_INVALID, "null group");
    int n = 0, r = -1;
    if (g->comm && g_rccl.CommCount && g_rccl.CommUserRank) {      // what RCCL itself reports for the communicator
        if (g_rccl.CommCount(g->comm, &n) != 0 || g_rccl.CommUserRank(g->comm, &r) != 0) return vlo_fail(VLO_E_HIP, "ncclCommCount / ncclCommUserRank failed");
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return VLO_OK;
}

// frame-parallel vision tower under tensor parallelism (north_star: "broadcasting the 10-token frame embedding each step"): every
// rank encodes its share of the pending frames, one all-gather hands every rank all the [frame_num_tokens, H] embeddings
// (81 920 B per frame for the 8B model).  One process per GPU over RCCL only; other groups keep the replicated tower.
int vlo_tp_allgather(vlo_tp_group *g, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *stream) {
    if (!g || !send_dev || !recv_dev || bytes_per_rank <= 0) return vlo_fail(VLO_E_INVALID, "bad tp_allgather arguments");
    if (!g->comm) return vlo_fail(VLO_E_STATE, "tp_allgather needs the RCCL communicator of a one-process-per-GPU group");
    TP_TRY(hipSetDevice(g->eng[0]->device));
    const int nrc = g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, kNcclInt8, g->comm, (hipStream_t)stream);
    if (nrc != 0) return vlo_fail(VLO_E_HIP, rccl_err("ncclAllGather", nrc));
    return VLO_OK;
}

static void p2p_release(vlo_tp_group *g);
void vlo_tp_group_destroy(vlo_tp_group *g) {
    if (!g) return;
    if (g->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(g->comm);
    p2p_release(g);
    delete g;
}

int vlo_tp_session_create(vlo_tp_group *g, int64_t max_tokens_hint, vlo_tp_session **out) {
    if (!g || !out) return vlo_fail(VLO_E_INVALID, "bad tp_session_create arguments");
    vlo_tp_session *t = new vlo_tp_session();
    t->g = g;
    for (vlo_engine *e : g->eng) {
        vlo_session *s = nullptr;
        int rc = vlo_session_create(e, max_tokens_hint, &s);
        if (rc) { vlo_tp_session_destroy(t); return rc; }
        t->ss.push_back(s);
    }
    vlo_engine *e0 = g->eng[0];
    if (dev_alloc((void **)&t->gather_tmp, (size_t)g->tp_size * 16 * e0->V_l * 2)) { vlo_tp_session_destroy(t); return VLO_E_HIP; }
    *out = t;
    return VLO_OK;
}
void vlo_tp_session_destroy(vlo_tp_session *t) {
    if (!t) return;
    for (vlo_session *s : t->ss) vlo_session_destroy(s);
    if (t->gather_tmp) hipFree(t->gather_tmp);
    if (t->gather_big) hipFree(t->gather_big);
    delete t;
}
int vlo_tp_session_reset(vlo_tp_session *t) {
    if (!t) return vlo_fail(VLO_E_INVALID, "null tp session");
    for (vlo_session *s : t->ss) vlo_session_reset(s);
    return VLO_OK;
}
int64_t vlo_tp_session_len(const vlo_tp_session *t) { return t && !t->ss.empty() ? t->ss[0]->len : -1; }

// trim_past_key_values(past, 0, n) (models/modeling_live.py:170-171) under tensor parallelism: every local rank's KV shard is forked /
// cropped the same way (each shard holds its own kv heads of the same positions; no exchange is involved).  One process per GPU:
// every rank makes the same call, as for every other vlo_tp_* entry point.
int vlo_tp_session_fork(vlo_tp_session *src, int64_t n_tokens, vlo_tp_session **out, void *stream) {
    if (!src || !out || n_tokens < 0 || n_tokens > vlo_tp_session_len(src)) return vlo_fail(VLO_E_INVALID, "bad tp_session_fork arguments");
    vlo_tp_session *t = new vlo_tp_session();
    t->g = src->g;
    for (vlo_session *s : src->ss) {
        vlo_session *d = nullptr;
        const hipError_t de = hipSetDevice(s->e->device);          // (not TP_TRY: every error path below destroys the half-built session and its shards)
        if (de != hipSuccess) { vlo_tp_session_destroy(t); return vlo_fail(VLO_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(de)); }
        const int rc = session_fork_shard(s, n_tokens, &d, stream);
        if (rc) { vlo_tp_session_destroy(t); return rc; }
        t->ss.push_back(d);
    }
    vlo_engine *e0 = src->g->eng[0];
    if (dev_alloc((void **)&t->gather_tmp, (size_t)src->g->tp_size * 16 * e0->V_l * 2)) { vlo_tp_session_destroy(t); return VLO_E_HIP; }
    *out = t;
    return VLO_OK;
}
int vlo_tp_session_crop(vlo_tp_session *t, int64_t n_tokens) {
    if (!t || n_tokens < 0 || n_tokens > vlo_tp_session_len(t)) return vlo_fail(VLO_E_INVALID, "bad tp_session_crop arguments");
    for (vlo_session *s : t->ss) {
        const int rc = session_crop_shard(s, n_tokens);
        if (rc) return rc;
    }
    return VLO_OK;
}

// ---- p2p exchange ------------------------------------------------------------------------------------------------
// Every rank owns a MAILBOX in its own HBM, mapped by every peer (hipIpc across processes, plain pointers inside one):
//   reduce region  [2 slots][T sources][16 rows][H]        8-byte granules {tag = epoch, fp32 partial sum}
//   gather region  [2 slots][T sources][16 rows][V_l / 2]  8-byte granules {tag = epoch, two bf16 logits}
// A granule is written by ONE naturally aligned 8-byte system-scope store and read by one 8-byte system-scope load, so
// the data carries its own "ready" flag: no fence, no store ordering and no separate flag are relied upon
// (cdna_hip_programming.md §6 Guideline 16, form R2 — applied here across GPUs: the mailbox is uncached / fine-grained
// memory, remote writes land in the owner's HBM, the owner polls with sc0 sc1 loads that no cache level serves).
// An exchange = every rank PUBLISHES its [m][H] row sums into slot (epoch & 1) of all T mailboxes (its own included),
// then COLLECTS the T source rows of its own mailbox in rank order 0..T-1 — the same fp32 sum on every rank — and goes
// straight on with `h += bf16(sum); x = RMSNorm(h) * w` (what add_rmsnorm_kernel does after an RCCL all-reduce).
// Round 6: when the o / down GEMV's plan has ONE K slice per block (every TP = 8 shard shape) the PUBLISH step is the GEMV's own epilogue
// (gemv.h EPI_PARTIAL_MBOX: the reduced fp32 sums leave the block as granules, tag and slot fixed on the host before the launch: p2p_begin_reduce) and
// the exchange kernel runs collect-only — no partial matrix in HBM, no publish pass.  VLO_TP_P2P_DIRECT=0 keeps the round-3 form (tests' A/B).
// Two slots per region suffice: a rank can only publish the region's exchange k+2 (the slot of k) after collecting k+1,
// which needs every peer's k+1 granules, which a peer stores (stream order) after it finished collecting k.  Tags are
// compared for equality with the epoch, a host-side counter that all ranks advance in lock-step, so stale slots never match.  Every spin is bounded
// (s_memrealtime); a timeout raises a sticky error word and later kernels stop waiting, so the stream always drains.
// kernels: tp_p2p.cuh (tp_xchg_norm_kernel, tp_gather_kernel)

// mailbox layout, in granules (shared by the launch code and vlo_debug_p2p_layout)
// `seq` = how many exchanges of THAT region were issued before this one: consecutive exchanges of a region alternate slots
// whatever the other region does in between (the two-slot argument above is per region); the TAG is the group-wide epoch.
static inline size_t p2p_reduce_slot_off(int T, int H, unsigned seq) { return (size_t)(seq & 1u) * T * 16 * H; }
static inline size_t p2p_gather_base(int T, int H) { return (size_t)2 * T * 16 * H; }
static inline size_t p2p_gather_slot_off(int T, int H, int Vh, unsigned seq) { return p2p_gather_base(T, H) + (size_t)(seq & 1u) * T * 16 * Vh; }
static inline size_t p2p_total_granules(int T, int H, int Vh) { return p2p_gather_base(T, H) + (size_t)2 * T * 16 * Vh; }
static inline unsigned p2p_next_epoch(unsigned epoch) { return epoch + 1u == 0u ? 1u : epoch + 1u; }   // the tag is never 0

int vlo_debug_p2p_layout(int T, int H, int Vl, unsigned seq, unsigned epoch, int64_t *out6) {
    if (!out6 || T < 2 || T > 8 || H <= 0 || (H & 7) || Vl <= 0 || (Vl & 1)) return vlo_fail(VLO_E_INVALID, "bad debug_p2p_layout arguments");
    const int Vh = Vl / 2;
    out6[0] = (int64_t)p2p_reduce_slot_off(T, H, seq);         // first granule of the reduce slot of the seq-th reduce exchange
    out6[1] = (int64_t)16 * H;                                 // granules between two sources in a reduce slot (row stride = H)
    out6[2] = (int64_t)p2p_gather_slot_off(T, H, Vh, seq);     // first granule of the gather slot of the seq-th gather exchange
    out6[3] = (int64_t)16 * Vh;                                // granules between two sources in a gather slot (row stride = Vl / 2)
    out6[4] = (int64_t)p2p_total_granules(T, H, Vh);           // mailbox size
    out6[5] = (int64_t)p2p_next_epoch(epoch);
    return VLO_OK;
}

static void p2p_release(vlo_tp_group *g) {
    P2PState &P = g->p2p;
    for (void *m : P.ipc_opened) hipIpcCloseMemHandle(m);
    P.ipc_opened.clear();
    for (size_t i = 0; i < g->eng.size() && i < 8; ++i) {
        hipSetDevice(g->eng[i]->device);
        if (P.mbox[i]) hipFree(P.mbox[i]);
        if (P.err_dev[i]) hipFree(P.err_dev[i]);
        P.mbox[i] = nullptr;
        P.err_dev[i] = nullptr;
    }
    if (P.err_host) hipHostFree(P.err_host);
    P.err_host = nullptr;
    P.allocated = P.enabled = false;
}

// mailboxes of the local ranks: uncached (first choice) or fine-grained device memory — no cache level may serve a poll
// with stale data while a peer GPU writes the line — zeroed so that no tag matches.  `first_kind` = 0: try uncached, then
// fine-grained; 1: fine-grained only (the export path retries with it when hipIpc refuses an uncached allocation).
static int p2p_allocate(vlo_tp_group *g, int first_kind = 0) {
    P2PState &P = g->p2p;
    if (P.allocated) return VLO_OK;
    const vlo_engine *e0 = g->eng[0];
    const int T = g->tp_size;
    if (T < 2 || T > 8) return vlo_fail(VLO_E_INVALID, "the p2p exchange needs 2 <= tp_size <= 8");
    if ((e0->cfg.hidden_size & 7) || (e0->V_l & 1)) return vlo_fail(VLO_E_UNSUPPORTED, "p2p exchange: hidden_size % 8 or odd vocabulary shard");
    P.H = e0->cfg.hidden_size;
    P.Vh = e0->V_l / 2;
    P.gat_base = p2p_gather_base(T, P.H);
    P.granules = p2p_total_granules(T, P.H, P.Vh);
    if (const char *v = getenv("VLO_TP_P2P_TIMEOUT_MS")) P.timeout_ticks = (long long)atoll(v) * 100000ll;
    if (!P.err_host) {
        if (hipHostMalloc((void **)&P.err_host, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess)
            return vlo_fail(VLO_E_HIP, "p2p exchange: hipHostMalloc of the error word failed");
        *P.err_host = 0u;
    }
    P.uncached = first_kind == 0;
    for (size_t i = 0; i < g->eng.size(); ++i) {
        TP_TRY(hipSetDevice(g->eng[i]->device));
        void *m = nullptr;
        const size_t bytes = P.granules * 8;
        if (!P.uncached || hipExtMallocWithFlags(&m, bytes, hipDeviceMallocUncached) != hipSuccess) {
            if (P.uncached) (void)hipGetLastError();
            P.uncached = false;
            m = nullptr;
            if (hipExtMallocWithFlags(&m, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                p2p_release(g);
                return vlo_fail(VLO_E_NOMEM, "p2p exchange: cannot allocate an uncached / fine-grained mailbox");
            }
        }
        P.mbox[i] = (unsigned long long *)m;
        hipError_t he = hipMemset(m, 0, bytes);
        if (he == hipSuccess) he = hipMalloc((void **)&P.err_dev[i], 64);
        if (he == hipSuccess) he = hipMemset(P.err_dev[i], 0, 64);
        if (he == hipSuccess) he = hipDeviceSynchronize();
        if (he != hipSuccess) {
            p2p_release(g);
            return vlo_fail(VLO_E_HIP, std::string("p2p exchange: mailbox setup: ") + hipGetErrorString(he));
        }
    }
    P.allocated = true;
    return VLO_OK;
}

int vlo_tp_p2p_export(vlo_tp_group *g, void *out_handle64) {
    if (!g || !out_handle64) return vlo_fail(VLO_E_INVALID, "bad tp_p2p_export arguments");
    if (g->eng.size() != 1) return vlo_fail(VLO_E_STATE, "tp_p2p_export is for one-process-per-GPU groups (single-process groups need no handles)");
    if (g->p2p.enabled) return vlo_fail(VLO_E_STATE, "the p2p exchange is already enabled");
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    for (int kind = 0; kind < 2; ++kind) {          // uncached first; fine-grained if hipIpc refuses that allocation
        int rc = p2p_allocate(g, kind);
        if (rc) return rc;
        TP_TRY(hipSetDevice(g->eng[0]->device));
        const hipError_t he = hipIpcGetMemHandle(&h, g->p2p.mbox[0]);
        if (he == hipSuccess) {
            memcpy(out_handle64, &h, sizeof(h));
            return VLO_OK;
        }
        (void)hipGetLastError();
        if (kind == 1 || !g->p2p.uncached) return vlo_fail(VLO_E_HIP, std::string("hipIpcGetMemHandle(mailbox): ") + hipGetErrorString(he));
        p2p_release(g);
    }
    return vlo_fail(VLO_E_HIP, "hipIpcGetMemHandle(mailbox) failed");
}

int vlo_tp_p2p_enable(vlo_tp_group *g, const void *handles) {
    if (!g) return vlo_fail(VLO_E_INVALID, "null tp group");
    P2PState &P = g->p2p;
    if (P.enabled) return VLO_OK;
    const int T = g->tp_size, R = (int)g->eng.size();
    int rc = p2p_allocate(g);
    if (rc) return rc;
    if (R == T) {                                   // single process: every mailbox is a local allocation
        for (int i = 0; i < R; ++i)
            for (int p = 0; p < T; ++p) P.peer[i][p] = P.mbox[p];
        P.fused = false;                            // the logical ranks share one stream: publish all, then collect all
    } else {
        if (!handles) return vlo_fail(VLO_E_INVALID, "one-process-per-GPU groups need the T mailbox handles (vlo_tp_p2p_export of every rank, in rank order)");
        const int me = g->eng[0]->tp_rank;
        TP_TRY(hipSetDevice(g->eng[0]->device));
        int ndev = 0;
        TP_TRY(hipGetDeviceCount(&ndev));
        for (int d = 0; d < ndev; ++d) {            // peers' HBM is written over xGMI: peer access to every visible GPU
            if (d == g->eng[0]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->eng[0]->device, d) == hipSuccess && can) {
                const hipError_t he = hipDeviceEnablePeerAccess(d, 0);
                if (he != hipSuccess) (void)hipGetLastError();      // already enabled is fine
            }
        }
        for (int p = 0; p < T; ++p) {
            if (p == me) { P.peer[0][p] = P.mbox[0]; continue; }
            hipIpcMemHandle_t h;
            memcpy(&h, (const char *)handles + (size_t)p * sizeof(h), sizeof(h));
            void *m = nullptr;
            const hipError_t he = hipIpcOpenMemHandle(&m, h, hipIpcMemLazyEnablePeerAccess);
            if (he != hipSuccess) {
                (void)hipGetLastError();
                for (void *q : P.ipc_opened) hipIpcCloseMemHandle(q);
                P.ipc_opened.clear();
                return vlo_fail(VLO_E_HIP, "hipIpcOpenMemHandle(mailbox of rank " + std::to_string(p) + "): " + hipGetErrorString(he));
            }
            P.ipc_opened.push_back(m);
            P.peer[0][p] = (unsigned long long *)m;
        }
    }
    P.enabled = true;
    P.direct = !(getenv("VLO_TP_P2P_DIRECT") && atoi(getenv("VLO_TP_P2P_DIRECT")) == 0);
    return VLO_OK;
}

int vlo_tp_p2p_status(vlo_tp_group *g, int *enabled, int *timed_out, int *uncached) {
    if (!g) return vlo_fail(VLO_E_INVALID, "null tp group");
    if (enabled) *enabled = g->p2p.enabled ? 1 : 0;
    if (timed_out) *timed_out = (g->p2p.err_host && *(volatile unsigned *)g->p2p.err_host) ? 1 : 0;
    if (uncached) *uncached = g->p2p.uncached ? 1 : 0;
    return VLO_OK;
}

static P2PPeers p2p_peers(const P2PState &P, int local_idx, int T) {
    P2PPeers pp{};
    for (int p = 0; p < T; ++p) pp.mbox[p] = P.peer[local_idx][p];
    return pp;
}

// ---- exchanges -------------------------------------------------------------------------------------------------
static int tp_allreduce(vlo_tp_session *t, float *(vlo_session::*buf), size_t count, hipStream_t st) {
    vlo_tp_group *g = t->g;
    if (g->tp_size == 1) return VLO_OK;
    if (g->comm) {
        float *b = t->ss[0]->*buf;
        const int nrc = g_rccl.AllReduce(b, b, count, kNcclFloat32, kNcclSum, g->comm, st);
        if (nrc != 0) return vlo_fail(VLO_E_HIP, rccl_err("ncclAllReduce", nrc));
        return VLO_OK;
    }
    PtrList L;
    L.n = (int)t->ss.size();
    for (int r = 0; r < L.n; ++r) L.p[r] = t->ss[r]->*buf;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(tp_sum_kernel, dim3(blocks), dim3(256), 0, st, L, count);
    TP_TRY(hipGetLastError());
    return VLO_OK;
}

// all-reduce of every local rank's partial sums `buf` [ks][16][H] + `h += bf16(sum); x = RMSNorm(h) * w` on every local rank.
// RCCL / sum-kernel: the slabs are all-reduced in place, add_rmsnorm_kernel combines them.  p2p: see "p2p exchange".
// One exchange of the reduce region = a tag and a slot, fixed on the host BEFORE the kernels that publish into it are launched
struct XchgId { unsigned epoch; size_t slot_off; };
static XchgId p2p_begin_reduce(vlo_tp_group *g) {
    P2PState &P = g->p2p;
    P.epoch = p2p_next_epoch(P.epoch);              // the tag is never 0 (zeroed mailboxes must not match)
    return XchgId{P.epoch, p2p_reduce_slot_off(g->tp_size, g->eng[0]->cfg.hidden_size, P.n_reduce++)};
}
// may the o / down GEMV of this step publish straight from its epilogue?  (one K slice per block: the sums a block holds are final)
static bool p2p_direct(const vlo_tp_group *g, const GemvPlan &plan) { return g->p2p.enabled && g->p2p.direct && plan.ksplit == 1; }
static void p2p_publish_args(const vlo_tp_group *g, int local_idx, const XchgId &x, GemvArgs *a) {
    const int T = g->tp_size, H = g->eng[0]->cfg.hidden_size;
    for (int p = 0; p < 8; ++p) a->mbox[p] = p < T ? g->p2p.peer[local_idx][p] : nullptr;
    a->mbox_off = x.slot_off + (size_t)g->eng[local_idx]->tp_rank * 16 * H;
    a->mbox_epoch = x.epoch;
    a->mbox_T = T;
    a->ldo = H;
}

// `published`: the partial sums are already in the mailboxes (the GEMVs' epilogues wrote them under *published): collect only
static int tp_reduce_norm(vlo_tp_session *t, float *(vlo_session::*buf), int ks, int m, const void *(*norm_w)(const vlo_engine *, int), int layer,
                          hipStream_t st, const XchgId *published = nullptr) {
    vlo_tp_group *g = t->g;
    const int R = (int)t->ss.size();
    const vlo_config &c = g->eng[0]->cfg;
    const int H = c.hidden_size;
    P2PState &P = g->p2p;
    if (!P.enabled) {
        int rc = tp_allreduce(t, buf, ks == 1 ? (size_t)m * H : (size_t)ks * 16 * H, st);
        if (rc) return rc;
        for (int r = 0; r < R; ++r) {
            vlo_session *s = t->ss[r];
            TP_TRY(add_rmsnorm_launch(s->h, s->*buf, ks, H, (const unsigned short *)norm_w(s->e, layer), s->x, H, H, c.rms_eps, m, st));
        }
        return VLO_OK;
    }
    const XchgId x = published ? *published : p2p_begin_reduce(g);
    XchgArgs a{};
    a.ks = ks; a.ld = H; a.T = g->tp_size;
    a.slot_off = x.slot_off;
    a.epoch = x.epoch; a.H = H; a.ldx = H; a.eps = c.rms_eps;
    a.err_host = P.err_host; a.timeout_ticks = P.timeout_ticks;
    const int passes = (P.fused || published) ? 1 : 2;
    for (int pass = 0; pass < passes; ++pass)
        for (int r = 0; r < R; ++r) {
            vlo_session *s = t->ss[r];
            a.partial = s->*buf; a.peers = p2p_peers(P, r, g->tp_size); a.me = s->e->tp_rank;
            a.mode = published ? 2 : (P.fused ? 3 : (pass == 0 ? 1 : 2));
            a.h = s->h; a.w = (const unsigned short *)norm_w(s->e, layer); a.x = s->x; a.err_dev = P.err_dev[r];
            hipLaunchKernelGGL(tp_xchg_norm_kernel, dim3(m), dim3(XCHG_THREADS), 0, st, a);
            TP_TRY(hipGetLastError());
        }
    return VLO_OK;
}
static const void *norm_in(const vlo_engine *e, int l) { return e->layers[l].ln_in; }
static const void *norm_post(const vlo_engine *e, int l) { return e->layers[l].ln_post; }
static const void *norm_final(const vlo_engine *e, int) { return e->norm_w; }

// logits_local [nr][V_l] of every rank -> logits [nr][V] on every local rank
static int tp_gather_logits(vlo_tp_session *t, int nr, hipStream_t st) {
    vlo_tp_group *g = t->g;
    const int T = g->tp_size, V = g->eng[0]->cfg.vocab_size, Vl = g->eng[0]->V_l;
    if (g->p2p.enabled) {
        P2PState &P = g->p2p;
        P.epoch = p2p_next_epoch(P.epoch);
        GatherArgs a{};
        a.T = T; a.Vl = Vl; a.V = V; a.epoch = P.epoch;
        a.slot_off = p2p_gather_slot_off(T, P.H, P.Vh, P.n_gather++);
        a.err_host = P.err_host; a.timeout_ticks = P.timeout_ticks;
        const int R = (int)t->ss.size();
        const int passes = P.fused ? 1 : 2;
        // fused mode spins on the peers' granules inside the publishing kernel: every block must be resident
        int blocks = (T * P.Vh + 255) / 256;
        const int cap = 256 / nr > 0 ? 256 / nr : 1;
        if (blocks > cap) blocks = cap;
        for (int pass = 0; pass < passes; ++pass)
            for (int r = 0; r < R; ++r) {
                vlo_session *s = t->ss[r];
                a.local = s->logits_local; a.out = s->logits; a.peers = p2p_peers(P, r, T); a.me = s->e->tp_rank;
                a.mode = P.fused ? 3 : (pass == 0 ? 1 : 2);
                a.err_dev = P.err_dev[r];
                hipLaunchKernelGGL(tp_gather_kernel, dim3(blocks, nr), dim3(256), 0, st, a);
                TP_TRY(hipGetLastError());
            }
        return VLO_OK;
    }
    if (g->comm) {
        vlo_session *s = t->ss[0];
        if (g_rccl.AllGather(s->logits_local, t->gather_tmp, (size_t)nr * Vl * 2, kNcclInt8, g->comm, st) != 0)
            return vlo_fail(VLO_E_HIP, "ncclAllGather failed");
        for (int r = 0; r < T; ++r)       // [T][nr][Vl] -> [nr][V]
            TP_TRY(hipMemcpy2DAsync(s->logits + (size_t)r * Vl, (size_t)V * 2, t->gather_tmp + (size_t)r * nr * Vl, (size_t)Vl * 2,
                                    (size_t)Vl * 2, nr, hipMemcpyDeviceToDevice, st));
        return VLO_OK;
    }
    for (size_t d = 0; d < t->ss.size(); ++d)
        for (int r = 0; r < T; ++r)
            TP_TRY(hipMemcpy2DAsync(t->ss[d]->logits + (size_t)r * Vl, (size_t)V * 2, t->ss[r]->logits_local, (size_t)Vl * 2,
                                    (size_t)Vl * 2, nr, hipMemcpyDeviceToDevice, st));
    return VLO_OK;
}

// ---- one chunk of m <= 16 new tokens on every local rank, lock-step ------------------------------------------------
static int tp_chunk(vlo_tp_session *t, const unsigned short *src, int m, bool want_last, bool want_all, hipStream_t st) {
    vlo_tp_group *g = t->g;
    const int R = (int)t->ss.size();
    const vlo_config &c = g->eng[0]->cfg;
    const int H = c.hidden_size, hd = g->eng[0]->head_dim;
    int rc;
    for (int r = 0; r < R; ++r) {
        vlo_session *s = t->ss[r];
        if ((rc = ensure_pages(s, s->len + m, st))) return rc;
        TP_TRY(copy_rows_launch(src, s->h, m, H, st));
    }
    int ks_d = 1;
    bool pub_o = false, pub_d = false;          // this exchange's sums were published by the GEMV epilogues (p2p_direct)
    XchgId xo{}, xd{};
    for (int l = 0; l < c.num_layers; ++l) {
        int ks_o = 1;
        for (int r = 0; r < R; ++r) {           // attention half: everything up to the o_proj partial sums
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            const LayerWeights &L = e->layers[l];
            const KvGeom kv = kv_geom(s);
            if (l == 0)                         // later layers: x comes out of the reduce + norm that closed the previous layer
                TP_TRY(add_rmsnorm_launch(s->h, nullptr, 0, H, (const unsigned short *)L.ln_in, s->x, H, H, c.rms_eps, m, st));
            GemvArgs a = gemv_args(L.qkv, s->x, H, m);
            a.out_bf16 = s->q; a.cos_tab = (const unsigned short *)e->cos_tab; a.sin_tab = (const unsigned short *)e->sin_tab;
            a.kv = kv; a.layer = l; a.num_heads = e->nh_l; a.pos0 = s->len;
            TP_TRY(gemv_launch(a, L.qkv.plan, XSRC_PLAIN, EPI_ROPE, st));
            TP_TRY(attention_launch(s->q, kv, l, e->nh_l, s->len, m, s->part_o, s->part_ml, s->attn, st));
            GemvArgs o = gemv_args(L.o, s->attn, e->nh_l * hd, m);
            if (r == 0 && (pub_o = p2p_direct(g, L.o.plan))) xo = p2p_begin_reduce(g);
            if (pub_o) {                        // the sums leave the epilogue as granules in every rank's mailbox
                p2p_publish_args(g, r, xo, &o);
                TP_TRY(gemv_launch(o, L.o.plan, XSRC_PLAIN, EPI_PARTIAL_MBOX, st));
            } else {
                o.out_f32 = s->partial_o; o.ldo = H;
                TP_TRY(gemv_launch(o, L.o.plan, XSRC_PLAIN, EPI_PARTIAL_F32, st));
            }
            ks_o = L.o.plan.ksplit;
        }
        // exchange 1: h += all-reduce(o_proj partials); x = post-attention RMSNorm(h)
        if ((rc = tp_reduce_norm(t, &vlo_session::partial_o, ks_o, m, norm_post, l, st, pub_o ? &xo : nullptr))) return rc;
        for (int r = 0; r < R; ++r) {           // MLP half
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            const LayerWeights &L = e->layers[l];
            GemvArgs a = gemv_args(L.gate_up, s->x, H, m);
            a.out_bf16 = s->act; a.ldo = e->I_l;
            TP_TRY(gemv_launch(a, L.gate_up.plan, XSRC_PLAIN, EPI_SWIGLU, st));
            GemvArgs d = gemv_args(L.down, s->act, e->I_l, m);
            // (the last layer's sums are only exchanged when logits are wanted: nothing is published that nobody collects)
            const bool exchanged = l + 1 < c.num_layers || want_last || want_all;
            if (r == 0 && (pub_d = exchanged && p2p_direct(g, L.down.plan))) xd = p2p_begin_reduce(g);
            if (pub_d) {
                p2p_publish_args(g, r, xd, &d);
                TP_TRY(gemv_launch(d, L.down.plan, XSRC_PLAIN, EPI_PARTIAL_MBOX, st));
            } else {
                d.out_f32 = s->partial; d.ldo = H;
                TP_TRY(gemv_launch(d, L.down.plan, XSRC_PLAIN, EPI_PARTIAL_F32, st));
            }
            ks_d = L.down.plan.ksplit;
        }
        // exchange 2: h += all-reduce(down_proj partials); x = the next layer's input RMSNorm(h).  After the last layer it
        // is the final norm, and only when logits are wanted (h is not read again otherwise).
        if (l + 1 < c.num_layers && (rc = tp_reduce_norm(t, &vlo_session::partial, ks_d, m, norm_in, l + 1, st, pub_d ? &xd : nullptr))) return rc;
    }
    if (want_last || want_all) {
        const int r0 = want_all ? 0 : m - 1, nr = want_all ? m : 1;
        if ((rc = tp_reduce_norm(t, &vlo_session::partial, ks_d, m, norm_final, 0, st, pub_d ? &xd : nullptr))) return rc;
        for (int r = 0; r < R; ++r) {
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            GemvArgs a = gemv_args(e->lm_head, s->x + (size_t)r0 * H, H, nr);
            a.out_bf16 = s->logits_local; a.ldo = e->V_l;
            TP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
        }
        if ((rc = tp_gather_logits(t, nr, st))) return rc;
        for (int r = 0; r < R; ++r) {
            t->ss[r]->last_logits = t->ss[r]->logits + (size_t)(nr - 1) * c.vocab_size;
            t->ss[r]->has_logits = true;
        }
    }
    for (int r = 0; r < R; ++r) t->ss[r]->len += m;
    return VLO_OK;
}

// ---- one block of VLO_PREFILL_MIN <= m <= VLO_PREFILL_TOKENS new tokens on every local rank, lock-step: the projections as GEMMs ----------
// engine.hip::run_prefill under HF's tensor-parallel plan: column-sharded q|k|v and gate|up (this rank's heads / MLP columns), RoPE + KV append and
// the flash-style prefill attention on the rank's own kv heads, row-sharded o / down as fp32 PARTIAL matrices [m][H] (EP_LLM_F32) that are
// all-reduced in fp32 and then folded into the residual stream by the row kernel (h += bf16(sum); x = RMSNorm(h) w) — the rounding points of the
// 16-row TP step and of the one-GPU prefill.  The exchange here is bandwidth-sized (m x H x 4 bytes: 64 MiB at 4 096 tokens of the 8B model), so
// it goes through RCCL (one process per GPU) or the sum kernel (all ranks in this process); the latency-sized p2p mailboxes are not used for it.
#define VLO_TP_LOGIT_ROWS 1024      // rows of a long input whose logits one lm_head GEMM + one gather handle (33 MB per rank at the 8B shape, T = 8)
static bool tp_prefill_ok(const vlo_tp_session *t) {
    const vlo_tp_group *g = t->g;
    if (!(g->comm || (int)g->eng.size() == g->tp_size)) return false;       // mailbox-only groups keep the 16-row step
    // tp_prefill has no fallback for its attention launch (the one-GPU run_prefill has: attention_launch over pooled partials), so a shard shape the
    // flash kernel is not built for — hd 64 MHA, G = 16, odd groups — keeps the 16-row step for long inputs
    for (const vlo_engine *e : g->eng)
        if (!prefill_ok(e) || e->nkv_l <= 0 || e->nh_l % e->nkv_l || !attention_prefill_supported(e->head_dim, e->nh_l / e->nkv_l)) return false;
    return true;
}
static int tp_prefill_exchange(vlo_tp_session *t, int m, const void *(*norm_w)(const vlo_engine *, int), int layer, hipStream_t st) {
    vlo_tp_group *g = t->g;
    const vlo_config &c = g->eng[0]->cfg;
    const int H = c.hidden_size;
    const size_t count = (size_t)m * H;
    if (g->comm) {
        float *b = t->ss[0]->ppartial;
        const int nrc = g_rccl.AllReduce(b, b, count, kNcclFloat32, kNcclSum, g->comm, st);
        if (nrc != 0) return vlo_fail(VLO_E_HIP, rccl_err("ncclAllReduce", nrc));
    } else {
        PtrList L;
        L.n = (int)t->ss.size();
        for (int r = 0; r < L.n; ++r) L.p[r] = t->ss[r]->ppartial;
        hipLaunchKernelGGL(tp_sum_kernel, dim3(1024), dim3(256), 0, st, L, count);
        TP_TRY(hipGetLastError());
    }
    for (vlo_session *s : t->ss)
        TP_TRY(add_rmsnorm_launch(s->ph, s->ppartial, 1, H, (const unsigned short *)norm_w(s->e, layer), s->px, H, H, c.rms_eps, m, st));
    return VLO_OK;
}
static int tp_prefill(vlo_tp_session *t, const unsigned short *src, int m, bool want_last, unsigned short *all_logits, hipStream_t st) {
    vlo_tp_group *g = t->g;
    const int R = (int)t->ss.size();
    const vlo_config &c = g->eng[0]->cfg;
    const int H = c.hidden_size, V = c.vocab_size, hd = g->eng[0]->head_dim;
    int rc;
    for (int r = 0; r < R; ++r) {
        vlo_session *s = t->ss[r];
        if ((rc = ensure_prefill_ws(s))) return rc;
        if (!s->ppartial) {
            void *p = nullptr;
            if ((rc = dev_alloc(&p, (size_t)VLO_PREFILL_TOKENS * H * 4))) return rc;
            s->ppartial = (float *)p;                 // part of the session's pooled prefill set from here on (engine.h)
        }
        if ((rc = ensure_pages(s, s->len + m, st))) return rc;
        TP_TRY(copy_rows_launch(src, s->ph, m, H, st));
    }
    for (int l = 0; l < c.num_layers; ++l) {
        for (int r = 0; r < R; ++r) {           // attention half
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            const LayerWeights &L = e->layers[l];
            const KvGeom kv = kv_geom(s);
            const int qd = e->nh_l * hd, Nqkv = qd + 2 * e->nkv_l * hd;
            if (l == 0) TP_TRY(add_rmsnorm_launch(s->ph, nullptr, 0, H, (const unsigned short *)L.ln_in, s->px, H, H, c.rms_eps, m, st));
            if ((rc = prefill_gemm(s, s->px, L.qkv, m, Nqkv, H, s->pqkv, Nqkv, LLM_GEMM_BF16, st))) return rc;
            TP_TRY(rope_kv_append_launch(s->pqkv, m, e->nh_l, (const unsigned short *)e->cos_tab, (const unsigned short *)e->sin_tab, kv, l, s->len, s->pq, st));
            TP_TRY(attention_prefill_launch(s->pq, kv, l, e->nh_l, s->len, m, s->px, st));       // (tp_prefill_ok asked attention_prefill_supported for this shard's head dim / GQA group)
            if ((rc = prefill_gemm(s, s->px, L.o, m, H, qd, s->ppartial, H, LLM_GEMM_F32, st))) return rc;
        }
        if ((rc = tp_prefill_exchange(t, m, norm_post, l, st))) return rc;      // exchange 1: h += sum(o partials); x = post-attention norm
        for (int r = 0; r < R; ++r) {           // MLP half
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            const LayerWeights &L = e->layers[l];
            if ((rc = prefill_gemm(s, s->px, L.gate_up, m, 2 * e->I_l, H, s->pact, e->I_l, LLM_GEMM_SWIGLU, st))) return rc;
            if ((rc = prefill_gemm(s, s->pact, L.down, m, H, e->I_l, s->ppartial, H, LLM_GEMM_F32, st))) return rc;
        }
        // exchange 2: h += sum(down partials); x = the next layer's input norm — after the last layer the final norm, and only when logits are wanted
        if (l + 1 < c.num_layers) {
            if ((rc = tp_prefill_exchange(t, m, norm_in, l + 1, st))) return rc;
        } else if (want_last || all_logits) {
            if ((rc = tp_prefill_exchange(t, m, norm_final, 0, st))) return rc;
        }
    }
    if (all_logits) {
        // every row's logits: each rank's vocabulary shard as ONE GEMM per chunk of VLO_TP_LOGIT_ROWS rows over its lm_head image — padded at load to whole
        // 256-column tiles (engine.hip::make_linear: Llama-3 at T = 8 holds 16 032 columns in 16 128) — into a padded [rows][V_l'] matrix, whose real
        // columns are then laid side by side into the caller's [m][V] matrix: straight copies between the local ranks, or one ncclAllGather per chunk.
        // (Rounds 4 - 5 sent these rows 16 at a time through the GEMV: 256 passes over the vocabulary shard for a 4096-token block.)
        const int Vl = g->eng[0]->V_l, Vp = g->eng[0]->lm_head.NT_gemm * 16;
        for (int r0 = 0; r0 < m; r0 += VLO_TP_LOGIT_ROWS) {
            const int nr = std::min(VLO_TP_LOGIT_ROWS, m - r0);
            for (int r = 0; r < R; ++r) {
                vlo_session *s = t->ss[r];
                if (!s->plogits) {
                    void *p = nullptr;
                    if ((rc = dev_alloc(&p, (size_t)VLO_TP_LOGIT_ROWS * Vp * 2))) return rc;
                    s->owned.push_back(p);
                    s->plogits = (unsigned short *)p;
                }
                if ((rc = prefill_gemm(s, s->px + (size_t)r0 * H, s->e->lm_head, nr, Vp, H, s->plogits, Vp, LLM_GEMM_BF16, st, false))) return rc;
            }
            if (g->comm) {
                if (!t->gather_big && (rc = dev_alloc((void **)&t->gather_big, (size_t)g->tp_size * VLO_TP_LOGIT_ROWS * Vp * 2))) return rc;
                if (g_rccl.AllGather(t->ss[0]->plogits, t->gather_big, (size_t)nr * Vp * 2, kNcclInt8, g->comm, st) != 0)
                    return vlo_fail(VLO_E_HIP, "ncclAllGather failed");
                for (int r = 0; r < g->tp_size; ++r)      // [T][nr][Vp] -> columns [r Vl, (r + 1) Vl) of rows r0 .. r0 + nr
                    TP_TRY(hipMemcpy2DAsync(all_logits + (size_t)r0 * V + (size_t)r * Vl, (size_t)V * 2, t->gather_big + (size_t)r * nr * Vp, (size_t)Vp * 2,
                                            (size_t)Vl * 2, nr, hipMemcpyDeviceToDevice, st));
            } else {
                for (int r = 0; r < R; ++r)
                    TP_TRY(hipMemcpy2DAsync(all_logits + (size_t)r0 * V + (size_t)t->ss[r]->e->tp_rank * Vl, (size_t)V * 2, t->ss[r]->plogits, (size_t)Vp * 2,
                                            (size_t)Vl * 2, nr, hipMemcpyDeviceToDevice, st));
            }
        }
        for (int r = 0; r < R; ++r) {               // the last row is what the samplers read on every local rank
            vlo_session *s = t->ss[r];
            TP_TRY(hipMemcpyAsync(s->logits, all_logits + (size_t)(m - 1) * V, (size_t)V * 2, hipMemcpyDeviceToDevice, st));
            s->last_logits = s->logits;
            s->has_logits = true;
        }
    } else if (want_last) {
        // the last row only: the vocabulary shards through the GEMV + the logits all-gather of the 16-row step
        for (int r = 0; r < R; ++r) {
            vlo_session *s = t->ss[r];
            vlo_engine *e = s->e;
            GemvArgs a = gemv_args(e->lm_head, s->px + (size_t)(m - 1) * H, H, 1);
            a.out_bf16 = s->logits_local; a.ldo = e->V_l;
            TP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
        }
        if ((rc = tp_gather_logits(t, 1, st))) return rc;
        for (int r = 0; r < R; ++r) {
            t->ss[r]->last_logits = t->ss[r]->logits;
            t->ss[r]->has_logits = true;
        }
    }
    for (int r = 0; r < R; ++r) t->ss[r]->len += m;
    return VLO_OK;
}

int vlo_tp_llm_step(vlo_tp_session *t, const void *embeds_dev, int n, void *last_logits_dev, void *all_logits_dev, void *stream) {
    if (!t || !embeds_dev || n <= 0) return vlo_fail(VLO_E_INVALID, "bad tp_llm_step arguments");
    vlo_engine *e0 = t->g->eng[0];
    {
        const vlo_tp_group *g = t->g;
        if (g->tp_size > 1 && g->eng.size() == 1 && !g->comm && !g->p2p.enabled)
            return vlo_fail(VLO_E_STATE, "one-process-per-GPU group without an exchange: pass the RCCL unique id to vlo_tp_group_create or call vlo_tp_p2p_enable");
        if (g->p2p.enabled && *(volatile unsigned *)g->p2p.err_host)
            return vlo_fail(VLO_E_HIP, "p2p exchange timed out waiting for a peer rank (ranks out of step, or a peer's writes are not visible here); "
                                       "results since then are invalid");
    }
    TP_TRY(hipSetDevice(e0->device));
    hipStream_t st = (hipStream_t)stream;
    const int H = e0->cfg.hidden_size, V = e0->cfg.vocab_size;
    int rc;
    static const bool block_path = getenv("VLO_BLOCK_PATH") ? atoi(getenv("VLO_BLOCK_PATH")) != 0 : true;     // (0: 16-row steps everywhere, as vlo_llm_step)
    const bool prefill = block_path && tp_prefill_ok(t);
    for (int c0 = 0; c0 < n;) {
        const int left = n - c0;
        if (prefill && left >= VLO_PREFILL_MIN) {
            // long inputs (teacher-forced evaluation, a long first prompt): blocks of up to VLO_PREFILL_TOKENS tokens, projections as GEMMs
            const int m = std::min(VLO_PREFILL_TOKENS, left);
            if ((rc = tp_prefill(t, (const unsigned short *)embeds_dev + (size_t)c0 * H, m, c0 + m == n,
                                 all_logits_dev ? (unsigned short *)all_logits_dev + (size_t)c0 * V : nullptr, st))) return rc;
            c0 += m;
            continue;
        }
        const int m = std::min(16, left);
        const bool last = (c0 + m == n);
        if ((rc = tp_chunk(t, (const unsigned short *)embeds_dev + (size_t)c0 * H, m, last, all_logits_dev != nullptr, st))) return rc;
        if (all_logits_dev)
            TP_TRY(hipMemcpyAsync((unsigned short *)all_logits_dev + (size_t)c0 * V, t->ss[0]->logits, (size_t)m * V * 2,
                                  hipMemcpyDeviceToDevice, st));
        c0 += m;
    }
    if (last_logits_dev) TP_TRY(hipMemcpyAsync(last_logits_dev, t->ss[0]->last_logits, (size_t)V * 2, hipMemcpyDeviceToDevice, st));
    return VLO_OK;
}

// `iters` back-to-back exchanges of m rows (all-reduce of the [m][H] fp32 partial sums + residual add + RMSNorm, exactly what a
// decoder layer issues twice) on the session's own buffers, timed with HIP events on `stream`: the latency of ONE exchange in
// microseconds, RCCL or peer-to-peer whichever the group uses.  Every rank of the group must make the same call (lock-step,
// like a step).  The session's residual stream is scratch afterwards: reset the session before stepping it again.
int vlo_tp_bench_exchange(vlo_tp_session *t, int m, int iters, double *avg_us, void *stream) {
    if (!t || m <= 0 || m > 16 || iters <= 0 || !avg_us) return vlo_fail(VLO_E_INVALID, "bad tp_bench_exchange arguments");
    vlo_tp_group *g = t->g;
    vlo_engine *e0 = g->eng[0];
    TP_TRY(hipSetDevice(e0->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t H = e0->cfg.hidden_size;
    for (vlo_session *s : t->ss) {            // finite inputs: partial sums of zeros, a residual stream of zeros
        TP_TRY(hipMemsetAsync(s->partial, 0, (size_t)16 * H * 4, st));
        TP_TRY(hipMemsetAsync(s->h, 0, (size_t)16 * H * 2, st));
    }
    hipEvent_t e0v = nullptr, e1v = nullptr;
    TP_TRY(hipEventCreate(&e0v));
    if (hipEventCreate(&e1v) != hipSuccess) { hipEventDestroy(e0v); return vlo_fail(VLO_E_HIP, "hipEventCreate failed"); }
    int rc = VLO_OK;
    for (int i = 0; i < 3 && !rc; ++i) rc = tp_reduce_norm(t, &vlo_session::partial, 1, m, norm_final, 0, st);   // warm-up
    hipError_t he = rc ? hipSuccess : hipEventRecord(e0v, st);
    for (int i = 0; i < iters && !rc && he == hipSuccess; ++i) rc = tp_reduce_norm(t, &vlo_session::partial, 1, m, norm_final, 0, st);
    if (!rc && he == hipSuccess) he = hipEventRecord(e1v, st);
    if (!rc && he == hipSuccess) he = hipEventSynchronize(e1v);
    float ms = 0.f;
    if (!rc && he == hipSuccess) he = hipEventElapsedTime(&ms, e0v, e1v);
    hipEventDestroy(e0v);
    hipEventDestroy(e1v);
    if (rc) return rc;
    if (he != hipSuccess) return vlo_fail(VLO_E_HIP, std::string("tp_bench_exchange: ") + hipGetErrorString(he));
    if (g->p2p.enabled && *(volatile unsigned *)g->p2p.err_host) return vlo_fail(VLO_E_HIP, "p2p exchange timed out during tp_bench_exchange");
    *avg_us = (double)ms * 1e3 / iters;
    return VLO_OK;
}

int vlo_tp_stream_sample(vlo_tp_session *t, float threshold, int interval_id, int64_t *tok_dev, float *p_interval_dev, void *stream) {
    if (!t || !tok_dev) return vlo_fail(VLO_E_INVALID, "bad tp_stream_sample arguments");
    vlo_session *s = t->ss[0];
    if (!s->has_logits) return vlo_fail(VLO_E_STATE, "no logits: call vlo_tp_llm_step first");
    TP_TRY(hipSetDevice(s->e->device));
    TP_TRY(stream_sample_launch(s->last_logits, s->e->cfg.vocab_size, threshold, interval_id, tok_dev, p_interval_dev, s->sample_scratch, (hipStream_t)stream));
    return VLO_OK;
}

int vlo_tp_greedy_generate(vlo_tp_session *t, const void *embeds_dev, int m, int eos_token_id, int64_t *out_ids_dev, int max_new,
                           int force_len, int *n_written, void *stream) {
    if (!t || !embeds_dev || !out_ids_dev || m <= 0 || max_new <= 0) return vlo_fail(VLO_E_INVALID, "bad tp_greedy_generate arguments");
    vlo_session *s = t->ss[0];
    vlo_engine *e = s->e;
    TP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    const int V = e->cfg.vocab_size;
    if (force_len > max_new) force_len = max_new;
    int rc = vlo_tp_llm_step(t, embeds_dev, m, nullptr, nullptr, stream);
    if (rc) return rc;
    const bool forced = force_len > 0;
    int i = 0;
    for (;; ++i) {
        int mode = 0;
        if (forced) mode = (i == force_len - 1) ? 2 : 1;
        TP_TRY(greedy_sample_launch(s->last_logits, V, out_ids_dev + i, eos_token_id, mode, s->sample_scratch, st));
        const bool last = (i == max_new - 1) || (forced && i == force_len - 1);
        if (!forced) {          // every rank reads the same token (the logits are gathered on every rank)
            TP_TRY(hipMemcpyAsync(s->host_tok, out_ids_dev + i, 8, hipMemcpyDeviceToHost, st));
            TP_TRY(hipStreamSynchronize(st));
            if (*s->host_tok == eos_token_id) {                                  // same post-EOS state as vlo_greedy_generate, on EVERY local shard
                for (vlo_session *sh : t->ss) sh->has_logits = false;
                break;
            }
        }
        if (last) break;
        TP_TRY(embed_gather_launch((const unsigned short *)e->embed, out_ids_dev + i, 1, e->cfg.hidden_size, V, s->emb1, st));
        if ((rc = vlo_tp_llm_step(t, s->emb1, 1, nullptr, nullptr, stream))) return rc;
    }
    TP_TRY(hipStreamSynchronize(st));
    if (n_written) *n_written = i + 1;
    return VLO_OK;
}
