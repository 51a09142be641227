// engine.hip — host side of libvlo.so: the C ABI declared in include/vlo.h.
//
// Owns: packed weights in HBM, the paged KV pool, per-session workspaces; sequences the
// gfx950 kernels of one Llama streaming step (gemv.hip, llm_ops.hip) and of the SigLIP
// encode (vit.hip) on caller-provided HIP streams.  No torch types, no CPU fallback:
// every entry point either runs the HIP path or returns an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vlo.h"
#include "common.cuh"
#include "engine.h"
#include "gemv.h"
#include "llm_ops.h"
#include "vit.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
int vlo_fail(int code, const std::string &msg) { return fail(code, msg); }
const char *vlo_last_error(void) { return g_err.c_str(); }
int vlo_abi_version(void) { return VLO_ABI_VERSION; }
#ifndef VLO_BUILD_ID
#define VLO_BUILD_ID "unstamped"
#endif
const char *vlo_build_id(void) { return "VLO_BUILD_ID=" VLO_BUILD_ID; }

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return fail(VLO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                                       std::to_string(__LINE__));                                             \
    } while (0)

// ------------------------------------------------------------------------------------
// small device utilities
// ------------------------------------------------------------------------------------
__global__ void convert_kernel(const void *src, int sdt, void *dst, int ddt, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v;
        if (sdt == VLO_DT_F32) v = ((const float *)src)[i];
        else if (sdt == VLO_DT_BF16) v = bf2f(((const bf16_t *)src)[i]);
        else v = h2f(((const f16_t *)src)[i]);
        if (ddt == VLO_DT_F32) ((float *)dst)[i] = v;
        else if (ddt == VLO_DT_BF16) ((bf16_t *)dst)[i] = f2bf(v);
        else ((f16_t *)dst)[i] = f2h(v);
    }
}
__global__ void sum_partials_kernel(const float *P, int ksplit, int ld, float *y, int n, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * N) return;
    const int m = i / N, c = i % N;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += P[((size_t)k * 16 + m) * ld + c];
    y[i] = s;
}

static size_t dt_size(int dt) { return dt == VLO_DT_F32 ? 4 : (dt == VLO_DT_FP8_E4M3 ? 1 : 2); }

int dev_alloc(void **p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    return VLO_OK;
}

// ------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------
static bool name_is(const std::string &n, const char *suffix) {
    const size_t l = strlen(suffix);
    return n.size() >= l && n.compare(n.size() - l, l, suffix) == 0;
}

int vlo_engine_create(const vlo_config *cfg, int device, vlo_engine **out) {
    if (!cfg || !out) return fail(VLO_E_INVALID, "null argument");
    if (cfg->abi_version != VLO_ABI_VERSION) return fail(VLO_E_INVALID, "vlo_config.abi_version mismatch");
    if (cfg->hidden_size <= 0 || cfg->num_heads <= 0 || cfg->num_kv_heads <= 0 || cfg->hidden_size % cfg->num_heads ||
        cfg->num_heads % cfg->num_kv_heads || cfg->num_layers <= 0 || cfg->vocab_size <= 0 || cfg->intermediate_size <= 0)
        return fail(VLO_E_INVALID, "bad Llama dimensions");
    const int hd = cfg->hidden_size / cfg->num_heads;
    if (hd != 64 && hd != 128) return fail(VLO_E_UNSUPPORTED, "head_dim must be 64 or 128");
    if (cfg->weight_dtype != 0 && cfg->weight_dtype != 1) return fail(VLO_E_INVALID, "weight_dtype must be 0 (bf16) or 1 (fp8 e4m3)");
    if (cfg->prefill_act_dtype != 0 && !(cfg->prefill_act_dtype == 1 && cfg->weight_dtype == 1))
        return fail(VLO_E_INVALID, "prefill_act_dtype must be 0 (bf16), or 1 (fp8 e4m3 per-row-scaled, native fp8 MFMA) on an engine with weight_dtype = 1");
    if ((cfg->hidden_size & 31) || (cfg->intermediate_size & 31) || (cfg->vocab_size & 3))
        return fail(VLO_E_UNSUPPORTED, "hidden/intermediate must be multiples of 32, vocab of 4");
    const int T = cfg->tp_size > 1 ? cfg->tp_size : 1;
    if (T > 1) {
        if (cfg->tp_rank < 0 || cfg->tp_rank >= T) return fail(VLO_E_INVALID, "tp_rank out of range");
        if (cfg->num_kv_heads % T || cfg->num_heads % T || cfg->intermediate_size % (T * 32) || cfg->vocab_size % (T * 16))
            return fail(VLO_E_UNSUPPORTED, "tp_size must divide kv heads, heads, intermediate_size/32 and vocab_size/16");
        if (T > 8) return fail(VLO_E_UNSUPPORTED, "tp_size > 8");
    }
    HIP_TRY(hipSetDevice(device));
    vlo_engine *e = new vlo_engine();
    e->cfg = *cfg;
    e->device = device;
    e->head_dim = hd;
    e->tp_size = T;
    e->tp_rank = T > 1 ? cfg->tp_rank : 0;
    e->nh_l = cfg->num_heads / T;
    e->nkv_l = cfg->num_kv_heads / T;
    e->I_l = cfg->intermediate_size / T;
    e->V_l = cfg->vocab_size / T;
    if (e->cfg.kv_pool_tokens <= 0) e->cfg.kv_pool_tokens = 16384;
    ingest_create(e);
    *out = e;
    return VLO_OK;
}

void vlo_engine_destroy(vlo_engine *e) {
    if (!e) return;
    hipSetDevice(e->device);
    for (auto &kv : e->raw) hipFree(kv.second.ptr);
    for (void *p : e->owned) hipFree(p);
    for (const PrefillWs &w : e->prefill_free)
        for (void *p : {(void *)w.ph, (void *)w.px, (void *)w.pqkv, (void *)w.pq, (void *)w.pact, w.wexp, (void *)w.partial, w.xq}) if (p) hipFree(p);
    for (auto &pr : e->prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    vit_destroy(e);
    ingest_destroy(e);
    delete e;
}

int64_t vlo_engine_weight_bytes(const vlo_engine *e) { return e ? e->weight_bytes : 0; }

int vlo_engine_load_weight(vlo_engine *e, const char *name, const void *data, int dtype, const int64_t *shape, int ndim) {
    if (!e || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(VLO_E_INVALID, "bad load_weight arguments");
    if (e->finalized) return fail(VLO_E_STATE, "engine already finalized");
    HIP_TRY(hipSetDevice(e->device));
    const std::string n(name);
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    // storage dtype inside the engine: LLM + connector bf16; ViT matmul weights f16, the rest of the ViT f32
    int ddt = VLO_DT_BF16;
    if (n == "rope.inv_freq") ddt = VLO_DT_F32;
    if (dtype == VLO_DT_FP8_E4M3 || name_is(n, "_scale")) {
        // fp8 storage of the streamed projections (vlo_config.weight_dtype = 1): quantised by the caller, kept as given
        const bool streamed = n.rfind("vision.", 0) != 0 && n.rfind("connector.", 0) != 0 &&
                              (name_is(n, "proj.weight") || n == "lm_head.weight" || name_is(n, "proj.weight_scale") || n == "lm_head.weight_scale");
        if (e->cfg.weight_dtype != 1 || !streamed)
            return fail(VLO_E_INVALID, n + ": fp8 weights / scales are accepted for the Llama projections of an engine created with weight_dtype = 1");
        if (name_is(n, "_scale") ? dtype != VLO_DT_F32 : dtype != VLO_DT_FP8_E4M3) return fail(VLO_E_INVALID, n + ": expected fp8 e4m3 data and f32 scales");
        ddt = dtype;
    }
    if (n.rfind("vision.", 0) == 0) {
        const bool is_mat = name_is(n, "proj.weight") || name_is(n, "fc1.weight") || name_is(n, "fc2.weight") ||
                            name_is(n, "patch_embedding.weight") || name_is(n, "in_proj_weight");
        ddt = is_mat ? VLO_DT_F16 : VLO_DT_F32;
    }
    void *stage = nullptr, *dst = nullptr;
    HIP_TRY(hipMalloc(&dst, numel * dt_size(ddt)));
    if (dtype == ddt) {
        HIP_TRY(hipMemcpy(dst, data, numel * dt_size(ddt), hipMemcpyDefault));
    } else {
        HIP_TRY(hipMalloc(&stage, numel * dt_size(dtype)));
        HIP_TRY(hipMemcpy(stage, data, numel * dt_size(dtype), hipMemcpyDefault));
        int blocks = (int)((numel + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(convert_kernel, dim3(blocks), dim3(256), 0, 0, stage, dtype, dst, ddt, numel);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(stage));
    }
    auto it = e->raw.find(n);
    if (it != e->raw.end()) hipFree(it->second.ptr);
    RawTensor t;
    t.ptr = dst;
    t.dtype = ddt;
    t.shape.assign(shape, shape + ndim);
    e->raw[n] = t;
    return VLO_OK;
}

static int take(vlo_engine *e, const std::string &name, std::vector<int64_t> shape, RawTensor *out) {
    auto it = e->raw.find(name);
    if (it == e->raw.end()) return fail(VLO_E_MISSING, "missing weight: " + name);
    if (!shape.empty() && it->second.shape != shape) {
        std::string s = "bad shape for " + name + ": got [";
        for (auto d : it->second.shape) s += std::to_string(d) + ",";
        s += "]";
        return fail(VLO_E_INVALID, s);
    }
    *out = it->second;
    return VLO_OK;
}

// pack rows [row0, row0+N) x columns [col0, col0+K) of the full bf16 linear `name` [Nfull][Kfull] into dst tiles
// (see gemv.hip); the slice is this rank's tensor-parallel shard (the whole matrix when tp_size == 1)
static int pack_into(vlo_engine *e, const std::string &name, int Nfull, int Kfull, int row0, int N, int col0, int K, PackedLinear &pl,
                     int tile_stride, int tile_offset, int half = -1) {
    RawTensor t;
    int rc = take(e, name, {Nfull, Kfull}, &t);
    if (rc) return rc;
    const int NT = half < 0 ? (N + 15) / 16 : (N + 7) / 8;
    if (pl.wq) {
        RawTensor sc;
        if (t.dtype != VLO_DT_FP8_E4M3) return fail(VLO_E_INVALID, name + ": this engine streams fp8 weights (weight_dtype = 1): load it as VLO_DT_FP8_E4M3 with its _scale");
        if ((rc = take(e, name + "_scale", {Nfull}, &sc))) return rc;
        const uint8_t *src = (const uint8_t *)t.ptr + (size_t)row0 * Kfull + col0;
        HIP_TRY(pack_weight_fp8_launch(src, (const float *)sc.ptr + row0, pl.Wp, pl.wscale, N, K, Kfull, NT, tile_stride, tile_offset, half, 0));
        return VLO_OK;
    }
    if (t.dtype != VLO_DT_BF16) return fail(VLO_E_INVALID, name + ": expected a bf16 matrix");
    const unsigned short *src = (const unsigned short *)t.ptr + (size_t)row0 * Kfull + col0;
    HIP_TRY(pack_weight_launch(src, pl.Wp, N, K, Kfull, NT, tile_stride, tile_offset, half, 0));
    return VLO_OK;
}

static int make_linear(vlo_engine *e, PackedLinear *pl, int N, int K, bool allow_ksplit, bool fp8 = false, bool pad_for_gemm = false) {
    pl->N = N;
    pl->K = K;
    pl->NT = (N + 15) / 16;
    pl->NT_gemm = pad_for_gemm ? (pl->NT + 15) & ~15 : pl->NT;
    pl->wq = fp8 ? 1 : 0;
    if (gemv_plan(K, allow_ksplit, &pl->plan)) return fail(VLO_E_UNSUPPORTED, "no GEMV plan for K=" + std::to_string(K));
    if (fp8 && ((pl->plan.KF & 1) || !(pl->plan.NW == 8 || (pl->plan.NW == 4 && pl->plan.KF == 14)) || (K & 63)))      // (4 x 14: K = 1792, gemv.hip)
        return fail(VLO_E_UNSUPPORTED, "fp8 weight image needs an even fragment count per wave (K=" + std::to_string(K) + ")");
    if (gemm64_plan(K, &pl->plan64, fp8))
        return fail(VLO_E_UNSUPPORTED, "no block-GEMM plan for K=" + std::to_string(K));
    const size_t bytes = (size_t)pl->NT_gemm * 16 * K * (fp8 ? 1 : 2);
    int rc = dev_alloc(&pl->Wp, bytes);
    if (rc) return rc;
    e->owned.push_back(pl->Wp);
    e->weight_bytes += (int64_t)bytes;
    if (pl->NT_gemm > pl->NT)                    // the pad tiles: zero weights (and zero scales below) = zero logits in columns nobody reads
        HIP_TRY(hipMemset((char *)pl->Wp + (size_t)pl->NT * 16 * K * (fp8 ? 1 : 2), 0, (size_t)(pl->NT_gemm - pl->NT) * 16 * K * (fp8 ? 1 : 2)));
    if (fp8) {
        if ((rc = dev_alloc((void **)&pl->wscale, (size_t)pl->NT_gemm * 16 * 4))) return rc;
        e->owned.push_back(pl->wscale);
        HIP_TRY(hipMemset(pl->wscale, 0, (size_t)pl->NT_gemm * 16 * 4));
        e->weight_bytes += (int64_t)pl->NT_gemm * 16 * 4;
    }
    return VLO_OK;
}

static int take_vec(vlo_engine *e, const std::string &name, int64_t n, void **out) {
    RawTensor t;
    int rc = take(e, name, {n}, &t);
    if (rc) return rc;
    *out = t.ptr;
    e->owned.push_back(t.ptr);
    e->raw.erase(name);
    e->weight_bytes += n * (int64_t)dt_size(t.dtype);
    return VLO_OK;
}

static void drop_raw(vlo_engine *e, const std::string &name) {
    auto it = e->raw.find(name);
    if (it != e->raw.end()) {
        hipFree(it->second.ptr);
        e->raw.erase(it);
    }
}

int vlo_engine_finalize(vlo_engine *e) {
    if (!e) return fail(VLO_E_INVALID, "null engine");
    if (e->finalized) return VLO_OK;
    HIP_TRY(hipSetDevice(e->device));
    const vlo_config &c = e->cfg;
    const int H = c.hidden_size, hd = e->head_dim;
    const int r = e->tp_rank;
    const int Ifull = c.intermediate_size, I = e->I_l;                 // this rank's MLP columns
    const int NqF = c.num_heads * hd, NkvF = c.num_kv_heads * hd;       // full projection widths
    const int Nq = e->nh_l * hd, Nkv = e->nkv_l * hd, Nqkv = Nq + 2 * Nkv;   // this rank's heads
    int rc;
    const bool f8 = c.weight_dtype == 1;
    e->layers.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        LayerWeights &L = e->layers[l];
        const std::string p = "model.layers." + std::to_string(l) + ".";
        if ((rc = make_linear(e, &L.qkv, Nqkv, H, false, f8))) return rc;
        if ((rc = pack_into(e, p + "self_attn.q_proj.weight", NqF, H, r * Nq, Nq, 0, H, L.qkv, 1, 0))) return rc;
        if ((rc = pack_into(e, p + "self_attn.k_proj.weight", NkvF, H, r * Nkv, Nkv, 0, H, L.qkv, 1, Nq / 16))) return rc;
        if ((rc = pack_into(e, p + "self_attn.v_proj.weight", NkvF, H, r * Nkv, Nkv, 0, H, L.qkv, 1, (Nq + Nkv) / 16))) return rc;
        if ((rc = make_linear(e, &L.o, H, Nq, e->tp_size > 1, f8))) return rc;          // TP: fp32 partial sums, all-reduced
        if ((rc = pack_into(e, p + "self_attn.o_proj.weight", H, NqF, 0, H, r * Nq, Nq, L.o, 1, 0))) return rc;
        if ((rc = make_linear(e, &L.gate_up, 2 * I, H, false, f8))) return rc;       // SwiGLU epilogue needs whole K
        if (I % 16) return fail(VLO_E_UNSUPPORTED, "intermediate_size (per rank) must be a multiple of 16");
        // gate and up share every tile (8 + 8 rows): I/8 single tiles, SwiGLU inside one tile
        if ((rc = pack_into(e, p + "mlp.gate_proj.weight", Ifull, H, r * I, I, 0, H, L.gate_up, 1, 0, 0))) return rc;
        if ((rc = pack_into(e, p + "mlp.up_proj.weight", Ifull, H, r * I, I, 0, H, L.gate_up, 1, 0, 1))) return rc;
        if ((rc = make_linear(e, &L.down, H, I, true, f8))) return rc;
        if ((rc = pack_into(e, p + "mlp.down_proj.weight", H, Ifull, 0, H, r * I, I, L.down, 1, 0))) return rc;
        if ((rc = take_vec(e, p + "input_layernorm.weight", H, &L.ln_in))) return rc;
        if ((rc = take_vec(e, p + "post_attention_layernorm.weight", H, &L.ln_post))) return rc;
        HIP_TRY(hipDeviceSynchronize());
        for (const char *sfx : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"}) {
            drop_raw(e, p + sfx);
            drop_raw(e, p + sfx + "_scale");
        }
    }
    if ((rc = take_vec(e, "model.norm.weight", H, &e->norm_w))) return rc;
    // a tensor-parallel rank's vocabulary shard is padded to whole 256-column GEMM tiles (Llama-3 at T = 8: 16 032 -> 16 128), so that every row's
    // logits of a long input take the GEMM path like every other projection (tp.hip::tp_prefill)
    if ((rc = make_linear(e, &e->lm_head, e->V_l, H, false, f8, e->tp_size > 1))) return rc;
    if ((rc = pack_into(e, "lm_head.weight", c.vocab_size, H, r * e->V_l, e->V_l, 0, H, e->lm_head, 1, 0))) return rc;
    {   // embedding table stays row-major (gather), replicated on every rank
        RawTensor t;
        if ((rc = take(e, "model.embed_tokens.weight", {c.vocab_size, H}, &t))) return rc;
        e->embed = t.ptr;
        e->owned.push_back(t.ptr);
        e->raw.erase("model.embed_tokens.weight");
    }
    // connector (optional: an LLM-only engine may omit it); replicated
    if (e->raw.count("connector.0.weight")) {
        const int Hv = c.vision_hidden_size;
        if ((rc = make_linear(e, &e->conn0, H, Hv, false))) return rc;
        if ((rc = pack_into(e, "connector.0.weight", H, Hv, 0, H, 0, Hv, e->conn0, 1, 0))) return rc;
        if ((rc = make_linear(e, &e->conn2, H, H, false))) return rc;
        if ((rc = pack_into(e, "connector.2.weight", H, H, 0, H, 0, H, e->conn2, 1, 0))) return rc;
        if ((rc = take_vec(e, "connector.0.bias", H, &e->conn0_b))) return rc;
        if ((rc = take_vec(e, "connector.2.bias", H, &e->conn2_b))) return rc;
        e->has_connector = true;
    }
    HIP_TRY(hipDeviceSynchronize());
    drop_raw(e, "lm_head.weight");
    drop_raw(e, "lm_head.weight_scale");
    drop_raw(e, "connector.0.weight");
    drop_raw(e, "connector.2.weight");

    // RoPE tables: cos/sin of pos * inv_freq in fp32, cast to bf16 (HF:modeling_llama.py:113-127)
    {
        const int half = hd / 2;
        std::vector<float> inv(half);
        if (e->raw.count("rope.inv_freq")) {
            RawTensor t;
            if ((rc = take(e, "rope.inv_freq", {half}, &t))) return rc;
            HIP_TRY(hipMemcpy(inv.data(), t.ptr, half * sizeof(float), hipMemcpyDeviceToHost));
            drop_raw(e, "rope.inv_freq");
        } else {
            for (int i = 0; i < half; ++i) inv[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)hd);
        }
        e->pool_pages = (int)((c.kv_pool_tokens + VLO_PAGE_TOKENS - 1) / VLO_PAGE_TOKENS);
        e->max_positions = (int64_t)e->pool_pages * VLO_PAGE_TOKENS;
        std::vector<unsigned short> ct((size_t)e->max_positions * half), st((size_t)e->max_positions * half);
        auto tobf = [](float f) {
            unsigned u;
            memcpy(&u, &f, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (unsigned short)(u >> 16);
        };
        for (int64_t p = 0; p < e->max_positions; ++p)
            for (int i = 0; i < half; ++i) {
                const float ang = inv[i] * (float)p;
                ct[p * half + i] = tobf(cosf(ang));
                st[p * half + i] = tobf(sinf(ang));
            }
        if ((rc = dev_alloc(&e->cos_tab, ct.size() * 2))) return rc;
        if ((rc = dev_alloc(&e->sin_tab, st.size() * 2))) return rc;
        e->owned.push_back(e->cos_tab);
        e->owned.push_back(e->sin_tab);
        HIP_TRY(hipMemcpy(e->cos_tab, ct.data(), ct.size() * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->sin_tab, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    }
    // KV pool
    {
        e->page_elems = (int64_t)e->nkv_l * VLO_PAGE_TOKENS * hd;
        e->layer_stride = e->page_elems * e->pool_pages;
        const size_t bytes = (size_t)e->layer_stride * c.num_layers * 2;
        if ((rc = dev_alloc(&e->k_pool, bytes))) return rc;
        if ((rc = dev_alloc(&e->vt_pool, bytes))) return rc;
        e->owned.push_back(e->k_pool);
        e->owned.push_back(e->vt_pool);
        HIP_TRY(hipMemset(e->k_pool, 0, bytes));
        HIP_TRY(hipMemset(e->vt_pool, 0, bytes));
        e->free_pages.clear();
        for (int p = e->pool_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    }
    if (c.has_vit) {
        if (!e->has_connector) return fail(VLO_E_MISSING, "vision tower needs the connector weights");
        if ((rc = vit_finalize(e))) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    e->finalized = true;
    return VLO_OK;
}

double vlo_step_algorithmic_bytes(const vlo_engine *e, int64_t Lc, int n) {
    if (!e) return 0;
    const vlo_config &c = e->cfg;
    const double H = c.hidden_size, I = c.intermediate_size, hd = e->head_dim;
    const double per_layer = (H * (c.num_heads * hd) * 2 + H * (c.num_kv_heads * hd) * 2 + 3.0 * H * I) * 2.0;
    double W = per_layer * c.num_layers + (double)c.vocab_size * H * 2.0;
    if (c.weight_dtype == 1) {      // fp8 image: one byte per weight + one fp32 scale per output channel
        const double rows_per_layer = (c.num_heads + 2.0 * c.num_kv_heads) * hd + H + 2.0 * I + H;
        W = W / 2.0 + (rows_per_layer * c.num_layers + c.vocab_size) * 4.0;
    }
    const double kv = 2.0 * c.num_layers * c.num_kv_heads * hd * 2.0;
    return W + kv * (double)(Lc + n) + kv * n + 2.0 * n * H * 2.0;
}

// ------------------------------------------------------------------------------------
// sessions
// ------------------------------------------------------------------------------------
int vlo_session_create(vlo_engine *e, int64_t max_tokens_hint, vlo_session **out) {
    if (!e || !out) return fail(VLO_E_INVALID, "null argument");
    if (!e->finalized) return fail(VLO_E_STATE, "engine not finalized");
    HIP_TRY(hipSetDevice(e->device));
    (void)max_tokens_hint;
    const vlo_config &c = e->cfg;
    const int H = c.hidden_size, I = e->I_l, hd = e->head_dim, nh = e->nh_l, nkv = e->nkv_l;
    const int Nqkv = (nh + 2 * nkv) * hd;
    vlo_session *s = new vlo_session();
    s->e = e;
    int rc = 0;
    auto A = [&](void **p, size_t bytes) {
        if (rc) return;
        rc = dev_alloc(p, bytes);
        if (!rc) {
            s->owned.push_back(*p);
            hipMemset(*p, 0, bytes);
        }
    };
    A((void **)&s->h, (size_t)32 * H * 2);
    A((void **)&s->x, (size_t)32 * H * 2);
    A((void **)&s->act, (size_t)32 * I * 2);
    A((void **)&s->attn, (size_t)32 * nh * hd * 2);
    A((void **)&s->q, (size_t)16 * nh * hd * 2);
    (void)Nqkv;
    int ksmax = 1;
    for (auto &L : e->layers) ksmax = std::max(ksmax, L.down.plan.ksplit);
    A((void **)&s->partial, (size_t)ksmax * 16 * H * 4);
    A((void **)&s->sq[0], (size_t)(512 + 16) * 16 * 4);      // row sum-of-squares partials: [gemv grid.x <= 512][16] (+ row offset slack)
    A((void **)&s->sq[1], (size_t)(512 + 16) * 16 * 4);
    A((void **)&s->part_o, (size_t)VLO_MAX_SPLITS * nh * 16 * hd * 4);
    A((void **)&s->part_ml, (size_t)VLO_MAX_SPLITS * nh * 16 * 2 * 4);
    A((void **)&s->logits, (size_t)16 * c.vocab_size * 2);
    if (e->tp_size > 1) {
        int ks_o = 1;
        for (auto &L : e->layers) ks_o = std::max(ks_o, L.o.plan.ksplit);
        A((void **)&s->logits_local, (size_t)16 * e->V_l * 2);
        A((void **)&s->partial_o, (size_t)ks_o * 16 * H * 4);
    }
    A((void **)&s->tok, 64);
    A((void **)&s->sample_scratch, VLO_SAMPLE_SCRATCH_FLOATS * 4);
    A((void **)&s->emb1, (size_t)32 * H * 2);
    A((void **)&s->page_table, (size_t)e->pool_pages * 4);
    if (rc) {
        vlo_session_destroy(s);
        return rc;
    }
    if (hipHostMalloc((void **)&s->host_tok, 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&s->host_pt, (size_t)e->pool_pages * 4, hipHostMallocDefault) != hipSuccess) {
        vlo_session_destroy(s);
        return fail(VLO_E_HIP, "hipHostMalloc failed");
    }
    s->last_logits = s->logits;
    HIP_TRY(hipDeviceSynchronize());
    *out = s;
    return VLO_OK;
}

int vlo_session_reset(vlo_session *s) {
    if (!s) return fail(VLO_E_INVALID, "null session");
    if (!s->pages.empty()) {
        // pages go back to the shared pool and the pinned page-table mirror will be rewritten from slot 0: queued kernels of
        // this session that still write K/V into those pages, and a queued page-table upload, must have drained first
        HIP_TRY(hipSetDevice(s->e->device));
        HIP_TRY(hipDeviceSynchronize());
    }
    std::lock_guard<std::mutex> g(s->e->pool_mu);
    for (int p : s->pages) s->e->free_pages.push_back(p);
    s->pages.clear();
    s->len = 0;
    s->has_logits = false;
    return VLO_OK;
}
int64_t vlo_session_len(const vlo_session *s) { return s ? s->len : -1; }

static void release_prefill_ws(vlo_session *s);      // (defined with the prefill path below)
void vlo_session_destroy(vlo_session *s) {
    if (!s) return;
    hipSetDevice(s->e->device);
    hipDeviceSynchronize();
    release_prefill_ws(s);
    vlo_session_reset(s);
    for (void *p : s->owned) hipFree(p);
    if (s->host_tok) hipHostFree(s->host_tok);
    for (hipEvent_t ev : s->tok_ev) if (ev) hipEventDestroy(ev);
    if (s->host_pt) hipHostFree(s->host_pt);
    delete s;
}

int ensure_pages(vlo_session *s, int64_t new_len, hipStream_t st) {
    vlo_engine *e = s->e;
    if (new_len > e->max_positions) return fail(VLO_E_NOMEM, "sequence exceeds kv_pool_tokens");
    const int need = (int)((new_len + VLO_PAGE_TOKENS - 1) / VLO_PAGE_TOKENS);
    const int have = (int)s->pages.size();
    if (need <= have) return VLO_OK;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        if ((int)e->free_pages.size() < need - have) return fail(VLO_E_NOMEM, "KV pool exhausted");
        for (int i = have; i < need; ++i) {
            const int p = e->free_pages.back();
            e->free_pages.pop_back();
            s->pages.push_back(p);
            s->host_pt[i] = p;      // pinned mirror is append-only => safe source for the async copy
        }
    }
    HIP_TRY(hipMemcpyAsync(s->page_table + have, s->host_pt + have, (size_t)(need - have) * 4, hipMemcpyHostToDevice, st));
    return VLO_OK;
}

KvGeom kv_geom(const vlo_session *s) {
    const vlo_engine *e = s->e;
    KvGeom g;
    g.k_pool = (unsigned short *)e->k_pool;
    g.vt_pool = (unsigned short *)e->vt_pool;
    g.page_table = s->page_table;
    g.layer_stride = e->layer_stride;
    g.page_elems = e->page_elems;
    g.num_kv_heads = e->nkv_l;
    g.head_dim = e->head_dim;
    return g;
}

// ---- live timing of the dominant kernel -------------------------------------------------
static void prof_flush(vlo_engine *e) {
    for (size_t i = 0; i < e->prof_used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(e->prof_events[i].second) == hipSuccess &&
            hipEventElapsedTime(&ms, e->prof_events[i].first, e->prof_events[i].second) == hipSuccess) {
            e->prof_ms += ms;
            e->prof_launches++;
        }
    }
    e->prof_used = 0;
}
static void prof_acquire(vlo_engine *e, hipEvent_t *a, hipEvent_t *b) {
    if (e->prof_used == e->prof_events.size()) {
        if (e->prof_events.size() >= 4096) {
            prof_flush(e);
        } else {
            hipEvent_t x, y;
            if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return;
            e->prof_events.push_back({x, y});
        }
    }
    *a = e->prof_events[e->prof_used].first;
    *b = e->prof_events[e->prof_used].second;
    e->prof_used++;
}
int vlo_profile_enable(vlo_engine *e, int stride) {
    if (!e) return fail(VLO_E_INVALID, "null engine");
    prof_flush(e);
    e->prof_stride = stride > 0 ? stride : 0;
    e->prof_seen = 0;
    e->prof_launches = 0;
    e->prof_ms = 0.0;
    return VLO_OK;
}
int vlo_profile_read(vlo_engine *e, int64_t *launches, double *total_ms, double *bytes_per_launch) {
    if (!e || !e->finalized) return fail(VLO_E_INVALID, "bad profile_read arguments");
    HIP_TRY(hipSetDevice(e->device));
    prof_flush(e);
    if (launches) *launches = e->prof_launches;
    if (total_ms) *total_ms = e->prof_ms;
    if (bytes_per_launch) {
        // gate+up weights [2I][H] bf16 streamed once + h [n<=16][H] in + act [n][I] out (n = 11 nominal)
        const double H = e->cfg.hidden_size, I = e->cfg.intermediate_size;
        const double wbytes = e->cfg.weight_dtype == 1 ? 2.0 * I * H + 2.0 * I * 4.0 : 2.0 * I * H * 2.0;   // fp8: codes + fp32 scales
        *bytes_per_launch = wbytes + 11.0 * H * 2.0 + 11.0 * I * 2.0;
    }
    return VLO_OK;
}

int vlo_profile_calibrate(vlo_engine *e, void *stream, double *empty_bracket_us) {
    if (!e || !empty_bracket_us) return fail(VLO_E_INVALID, "bad profile_calibrate arguments");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    const int N = 64;
    hipEvent_t a[N], b[N];
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < N; ++i) {
        HIP_TRY(hipEventCreate(&a[i]));
        HIP_TRY(hipEventCreate(&b[i]));
    }
    for (int i = 0; i < N; ++i) {
        HIP_TRY(hipEventRecord(a[i], st));
        HIP_TRY(hipEventRecord(b[i], st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    double tot = 0.0;
    for (int i = 0; i < N; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, a[i], b[i]));
        tot += ms;
        hipEventDestroy(a[i]);
        hipEventDestroy(b[i]);
    }
    *empty_bracket_us = tot * 1e3 / N;
    return VLO_OK;
}

GemvArgs gemv_args(const PackedLinear &pl, const unsigned short *x, int ldx, int n_rows) {
    GemvArgs a{};
    a.Wp = pl.Wp;
    a.wq = pl.wq;
    a.wscale = pl.wscale;
    a.x = x;
    a.K = pl.K;
    a.ldx = ldx;
    a.NT = pl.NT;
    a.N_valid = pl.N;
    a.n_rows = n_rows;
    return a;
}

// one chunk of m <= 16 new tokens whose embeddings are at `src`.  Per decoder layer (7 launches):
//   add_rmsnorm   [+ down-proj split-K combine + residual of the previous layer]      -> x
//   qkv GEMV      [RoPE + paged KV append in the epilogue]                             -> q, K, V^T
//   attention, combine                                                                 -> attn
//   o GEMV        [residual add + row sum-of-squares partials in the epilogue]         -> h, sq
//   gate/up GEMV  [post-attention RMSNorm fused into the operand load | SwiGLU]        -> act
//   down GEMV     [K split over blocks, fp32 partials]                                 -> partial
static int run_chunk(vlo_session *s, const unsigned short *src, int m, bool want_last, bool want_all, hipStream_t st) {
    vlo_engine *e = s->e;
    const vlo_config &c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, hd = e->head_dim, nh = c.num_heads;
    int rc;
    if ((rc = ensure_pages(s, s->len + m, st))) return rc;
    const KvGeom kv = kv_geom(s);
    HIP_TRY(copy_rows_launch(src, s->h, m, H, st));
    const float *prev = nullptr;
    int prev_ks = 0;
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerWeights &L = e->layers[l];
        HIP_TRY(add_rmsnorm_launch(s->h, prev, prev_ks, H, (const unsigned short *)L.ln_in, s->x, H, H, c.rms_eps, m, st));
        {   // qkv
            GemvArgs a = gemv_args(L.qkv, s->x, H, m);
            a.out_bf16 = s->q; a.cos_tab = (const unsigned short *)e->cos_tab; a.sin_tab = (const unsigned short *)e->sin_tab;
            a.kv = kv; a.layer = l; a.num_heads = nh; a.pos0 = s->len;
            HIP_TRY(gemv_launch(a, L.qkv.plan, XSRC_PLAIN, EPI_ROPE, st));
        }
        HIP_TRY(attention_launch(s->q, kv, l, nh, s->len, m, s->part_o, s->part_ml, s->attn, st));
        int sq_parts;
        {   // o_proj + residual
            GemvArgs a = gemv_args(L.o, s->attn, nh * hd, m);
            a.h = s->h; a.ldo = H; a.sq_out = s->sq[0];
            sq_parts = gemv_grid_x(a, L.o.plan, EPI_RESID);
            HIP_TRY(gemv_launch(a, L.o.plan, XSRC_PLAIN, EPI_RESID, st));
        }
        {   // gate/up + SwiGLU
            hipEvent_t ev0 = nullptr, ev1 = nullptr;
            if (e->prof_stride > 0 && (e->prof_seen++ % e->prof_stride) == 0) prof_acquire(e, &ev0, &ev1);
            GemvArgs a = gemv_args(L.gate_up, s->h, H, m);
            a.norm_w = (const unsigned short *)L.ln_post; a.sq_in = s->sq[0]; a.sq_in_parts = sq_parts; a.eps = c.rms_eps;
            a.out_bf16 = s->act; a.ldo = I;
            if (ev0) hipEventRecord(ev0, st);
            HIP_TRY(gemv_launch(a, L.gate_up.plan, XSRC_NORM, EPI_SWIGLU, st));
            if (ev1) hipEventRecord(ev1, st);
        }
        {   // down_proj: fp32 K-slice partials, combined by the next add_rmsnorm
            GemvArgs a = gemv_args(L.down, s->act, I, m);
            a.out_f32 = s->partial; a.ldo = H;
            HIP_TRY(gemv_launch(a, L.down.plan, XSRC_PLAIN, EPI_PARTIAL_F32, st));
            prev = s->partial;
            prev_ks = L.down.plan.ksplit;
        }
    }
    if (want_last || want_all) {
        HIP_TRY(add_rmsnorm_launch(s->h, prev, prev_ks, H, (const unsigned short *)e->norm_w, s->x, H, H, c.rms_eps, m, st));
        const int r0 = want_all ? 0 : m - 1, nr = want_all ? m : 1;        // only the rows that are read
        GemvArgs a = gemv_args(e->lm_head, s->x + (size_t)r0 * H, H, nr);
        a.out_bf16 = s->logits; a.ldo = c.vocab_size;
        HIP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
        s->last_logits = s->logits + (size_t)(nr - 1) * c.vocab_size;
        s->has_logits = true;
    }
    s->len += m;
    return VLO_OK;
}

// ---- block path: up to 64 new tokens per weight pass (prefill.hip) ---------------------------------------------------
static int ensure_block_ws(vlo_session *s) {
    if (s->bact) return VLO_OK;                 // the LAST buffer allocated below: set only when all of them exist
    vlo_engine *e = s->e;
    const size_t H = e->cfg.hidden_size, I = e->I_l, qd = (size_t)e->nh_l * e->head_dim, R = VLO_BLOCK_TOKENS;
    struct { unsigned short **p; size_t elems; } want[] = {{&s->bh, R * H}, {&s->bx, R * H}, {&s->bq, R * qd}, {&s->battn, R * qd}, {&s->bact, R * I}};
    HIP_TRY(hipSetDevice(e->device));
    for (auto &w : want) {
        if (*w.p) continue;                     // kept from an earlier, partially failed attempt
        void *p = nullptr;
        int rc = dev_alloc(&p, w.elems * 2);
        if (rc) return rc;
        s->owned.push_back(p);
        *w.p = (unsigned short *)p;
    }
    return VLO_OK;
}

// one block of 16 < m <= 64 new tokens.  Per decoder layer: RMSNorm rows -> qkv GEMM [RoPE + KV append] -> attention per
// 16-query sub-chunk (+ combine) -> o GEMM [residual add] -> RMSNorm rows -> gate/up GEMM [SwiGLU] -> down GEMM [residual
// add].  Same rounding points as run_chunk (projection outputs and residual sums are bf16, accumulation fp32); only the
// fp32 summation order inside a dot product differs (K is split over the waves of a block differently).
static int run_block(vlo_session *s, const unsigned short *src, int m, bool want_last, unsigned short *all_logits, hipStream_t st) {
    vlo_engine *e = s->e;
    const vlo_config &c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, hd = e->head_dim, nh = c.num_heads, V = c.vocab_size;
    int rc;
    if ((rc = ensure_block_ws(s))) return rc;
    if ((rc = ensure_pages(s, s->len + m, st))) return rc;
    const KvGeom kv = kv_geom(s);
    HIP_TRY(copy_rows_launch(src, s->bh, m, H, st));
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerWeights &L = e->layers[l];
        HIP_TRY(add_rmsnorm_launch(s->bh, nullptr, 0, H, (const unsigned short *)L.ln_in, s->bx, H, 0, c.rms_eps, m, st));   // packed-64
        {   // qkv
            GemvArgs a = gemv_args(L.qkv, s->bx, H, m);
            a.out_bf16 = s->bq; a.cos_tab = (const unsigned short *)e->cos_tab; a.sin_tab = (const unsigned short *)e->sin_tab;
            a.kv = kv; a.layer = l; a.num_heads = nh; a.pos0 = s->len;
            HIP_TRY(gemm64_launch(a, L.qkv.plan64, EPI_ROPE, st));
        }
        // one launch for the (up to four) 16-query sub-chunks; the keys of the whole block are already appended
        HIP_TRY(attention_launch(s->bq, kv, l, nh, s->len, m, s->part_o, s->part_ml, s->battn, st, 0));
        {   // o_proj + residual
            GemvArgs a = gemv_args(L.o, s->battn, nh * hd, m);
            a.h = s->bh; a.ldo = H;
            HIP_TRY(gemm64_launch(a, L.o.plan64, EPI_RESID, st));
        }
        HIP_TRY(add_rmsnorm_launch(s->bh, nullptr, 0, H, (const unsigned short *)L.ln_post, s->bx, H, 0, c.rms_eps, m, st));
        {   // gate/up + SwiGLU
            GemvArgs a = gemv_args(L.gate_up, s->bx, H, m);
            a.out_bf16 = s->bact; a.ldo = I;
            HIP_TRY(gemm64_launch(a, L.gate_up.plan64, EPI_SWIGLU, st));
        }
        {   // down_proj + residual
            GemvArgs a = gemv_args(L.down, s->bact, I, m);
            a.h = s->bh; a.ldo = H;
            HIP_TRY(gemm64_launch(a, L.down.plan64, EPI_RESID, st));
        }
    }
    if (want_last || all_logits) {
        HIP_TRY(add_rmsnorm_launch(s->bh, nullptr, 0, H, (const unsigned short *)e->norm_w, s->bx, H, all_logits ? 0 : H, c.rms_eps, m, st));
        if (all_logits) {                       // every row, straight into the caller's matrix
            GemvArgs a = gemv_args(e->lm_head, s->bx, H, m);
            a.out_bf16 = all_logits; a.ldo = V;
            HIP_TRY(gemm64_launch(a, e->lm_head.plan64, EPI_BF16, st));
            HIP_TRY(hipMemcpyAsync(s->logits, all_logits + (size_t)(m - 1) * V, (size_t)V * 2, hipMemcpyDeviceToDevice, st));
        } else {                                // only the row that is read
            GemvArgs a = gemv_args(e->lm_head, s->bx + (size_t)(m - 1) * H, H, 1);
            a.out_bf16 = s->logits; a.ldo = V;
            HIP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
        }
        s->last_logits = s->logits;
        s->has_logits = true;
    }
    s->len += m;
    return VLO_OK;
}

// ---- prefill path: up to VLO_PREFILL_TOKENS new tokens per weight pass, the projections as MFMA-bound GEMMs (prefill.h) ------------------
int ensure_prefill_ws(vlo_session *s) {
    if (s->pact) return VLO_OK;                 // the LAST buffer taken below: set only when all of them exist
    vlo_engine *e = s->e;
    {
        std::lock_guard<std::mutex> g(e->pool_mu);
        if (!e->prefill_free.empty()) {         // a set another session handed back (engine.h: pooled per engine)
            const PrefillWs w = e->prefill_free.back();
            e->prefill_free.pop_back();
            s->ph = w.ph; s->px = w.px; s->pqkv = w.pqkv; s->pq = w.pq;
            s->pf_wexp = w.wexp; s->pf_wexp_bytes = w.wexp_bytes; s->ppartial = w.partial; s->pxq = w.xq;
            s->pact = w.pact;
            return VLO_OK;
        }
    }
    const size_t H = e->cfg.hidden_size, I = e->I_l, qd = (size_t)e->nh_l * e->head_dim, kvd = (size_t)e->nkv_l * e->head_dim;
    const size_t R = VLO_PREFILL_TOKENS, RX = R + 256;        // X operands: the GEMM reads whole 256-row tiles
    struct { unsigned short **p; size_t elems; } want[] = {{&s->ph, R * H}, {&s->px, RX * std::max(H, qd)}, {&s->pqkv, R * (qd + 2 * kvd)},
                                                            {&s->pq, R * qd}, {&s->pact, RX * I}};
    HIP_TRY(hipSetDevice(e->device));
    for (auto &w : want) {
        if (*w.p) continue;
        void *p = nullptr;
        int rc = dev_alloc(&p, w.elems * 2);
        if (rc) return rc;                      // (a partly built set is freed with the session: release_prefill_ws)
        HIP_TRY(hipMemset(p, 0, w.elems * 2));  // the spare rows are read (and dropped) by the GEMMs: keep them finite
        *w.p = (unsigned short *)p;
    }
    return VLO_OK;
}

// a destroyed session's prefill set goes back to its engine's pool (the device is idle: vlo_session_destroy synchronises first)
static void release_prefill_ws(vlo_session *s) {
    if (!s->ph && !s->px && !s->pqkv && !s->pq && !s->pact && !s->pf_wexp && !s->ppartial && !s->pxq) return;
    vlo_engine *e = s->e;
    if (!s->pact) {                             // an allocation failed half way: nothing reusable
        for (void *p : {(void *)s->ph, (void *)s->px, (void *)s->pqkv, (void *)s->pq, s->pf_wexp, (void *)s->ppartial, s->pxq}) if (p) hipFree(p);
    } else {
        PrefillWs w;
        w.ph = s->ph; w.px = s->px; w.pqkv = s->pqkv; w.pq = s->pq; w.pact = s->pact;
        w.wexp = s->pf_wexp; w.wexp_bytes = s->pf_wexp_bytes; w.partial = s->ppartial; w.xq = s->pxq;
        std::unique_lock<std::mutex> g(e->pool_mu);
        if (e->prefill_free.size() < VLO_PREFILL_POOL_MAX) {
            e->prefill_free.push_back(w);
        } else {                                // a burst of long-prompt sessions does not pin its peak for the engine's lifetime (engine.h)
            g.unlock();
            for (void *p : {(void *)w.ph, (void *)w.px, (void *)w.pqkv, (void *)w.pq, (void *)w.pact, w.wexp, (void *)w.partial, w.xq}) if (p) hipFree(p);
        }
    }
    s->ph = s->px = s->pqkv = s->pq = s->pact = nullptr;
    s->pf_wexp = nullptr; s->pf_wexp_bytes = 0; s->ppartial = nullptr; s->pxq = nullptr;
}

// partial states for the fallback attention kernel (shapes attn_prefill_kernel is not instantiated for): 67 MB at the 8B shape,
// so only sessions that take the fallback pay for them
static int ensure_prefill_partials(vlo_session *s) {
    if (s->ppart_ml) return VLO_OK;
    vlo_engine *e = s->e;
    void *po = nullptr, *pm = nullptr;
    int rc = dev_alloc(&po, (size_t)(VLO_PREFILL_TOKENS / 16) * e->nh_l * 16 * e->head_dim * 4);
    if (rc) return rc;
    s->owned.push_back(po);
    if ((rc = dev_alloc(&pm, (size_t)(VLO_PREFILL_TOKENS / 16) * e->nh_l * 16 * 2 * 4))) return rc;
    s->owned.push_back(pm);
    s->ppart_o = (float *)po; s->ppart_ml = (float *)pm;
    return VLO_OK;
}

// One projection of a prefill block: out = X W^T through the GEMM of prefill.h (kind = LLM_GEMM_*).  fp8 engines: the projection's e4m3 image is
// expanded to bf16 (exactly) into the session's scratch right before its GEMM — 3 bytes of extra traffic per weight against hundreds of tokens of
// MFMA work per weight — and its per-channel scales go to the GEMM's epilogue.  Shared with tp.hip (a rank's shard of a tensor-parallel prefill).
int prefill_gemm(vlo_session *s, const unsigned short *X, const PackedLinear &pl, int m, int N, int K, void *out, int ldo, int kind, hipStream_t st,
                 bool layer_proj) {
    vlo_engine *e = s->e;
    if (!pl.wq) { HIP_TRY(llm_gemm_launch(X, pl.Wp, m, N, K, out, ldo, kind, st)); return VLO_OK; }
    if (layer_proj && e->cfg.prefill_act_dtype == 1 && llm_gemm_fp8_ok(N, K)) {
        // native fp8 MFMA (prefill.h): X quantised per row into the session's code scratch (codes, then the scales), W = the fp8 image as stored.
        // Rows past m keep whatever the scratch held: they are computed and dropped, no output row depends on another row's X.
        const size_t RX = VLO_PREFILL_TOKENS + 256;
        const LayerWeights &L0 = e->layers[0];
        const size_t kmax = (size_t)std::max({L0.qkv.K, L0.o.K, L0.gate_up.K, L0.down.K});
        if ((size_t)K > kmax) return fail(VLO_E_STATE, "prefill_gemm: a layer projection wider than the layer's widest K");
        if (!s->pxq) {
            void *p = nullptr;
            int rc2 = dev_alloc(&p, RX * kmax + RX * sizeof(float));
            if (rc2) return rc2;
            HIP_TRY(hipMemset(p, 0, RX * kmax + RX * sizeof(float)));
            HIP_TRY(hipDeviceSynchronize());
            s->pxq = p;
        }
        float *xs = reinterpret_cast<float *>((char *)s->pxq + RX * kmax);
        HIP_TRY(quantize_rows_fp8_launch(X, m, K, s->pxq, xs, st));
        HIP_TRY(llm_gemm_fp8_launch(s->pxq, xs, pl.Wp, pl.wscale, m, N, K, out, ldo, kind, st));
        return VLO_OK;
    }
    // sized for the largest projection of a layer at the first use (not grown projection by projection); the lm_head image may grow it once more
    const LayerWeights &L0 = e->layers[0];
    const int NTg = std::max(pl.NT, N / 16);                  // (N may be the padded width of a TP lm_head shard: pl.NT_gemm tiles)
    const size_t img = (size_t)NTg * 16 * K * 2;
    auto bytes = [](const PackedLinear &q) { return (size_t)q.NT * 16 * q.K * 2; };
    const size_t need = std::max({img, bytes(L0.qkv), bytes(L0.o), bytes(L0.gate_up), bytes(L0.down)});
    if (s->pf_wexp_bytes < need) {
        HIP_TRY(hipStreamSynchronize(st));                       // (grows at most twice per pooled set: the largest layer projection, then the lm_head image)
        void *p = nullptr;
        int rc2 = dev_alloc(&p, need);
        if (rc2) return rc2;
        if (s->pf_wexp) hipFree(s->pf_wexp);                     // the superseded scratch: nothing reads it after the synchronise above
        s->pf_wexp = p; s->pf_wexp_bytes = need;
    }
    HIP_TRY(expand_fp8_image_launch(pl.Wp, s->pf_wexp, NTg, K, st));
    HIP_TRY(llm_gemm_launch(X, s->pf_wexp, m, N, K, out, ldo, kind, st, pl.wscale));
    return VLO_OK;
}

// one block of VLO_PREFILL_MIN <= m <= VLO_PREFILL_TOKENS new tokens.  Per decoder layer: RMSNorm rows -> qkv GEMM -> RoPE + KV append ->
// attention in one launch (the block path's kernel, grid.z = 16-query sub-chunks: the whole block's keys are already appended) ->
// o GEMM [residual add] -> RMSNorm rows -> gate/up GEMM [SwiGLU] -> down GEMM [residual add].  Rounding points as run_chunk / run_block
// (projection outputs, RoPE products and sums, residual sums in bf16; accumulation fp32); only the summation order over K differs.
static int run_prefill(vlo_session *s, const unsigned short *src, int m, bool want_last, unsigned short *all_logits, hipStream_t st) {
    vlo_engine *e = s->e;
    const vlo_config &c = e->cfg;
    const int H = c.hidden_size, I = c.intermediate_size, hd = e->head_dim, nh = c.num_heads, nkv = c.num_kv_heads, V = c.vocab_size;
    const int qd = nh * hd, Nqkv = qd + 2 * nkv * hd;
    int rc;
    if ((rc = ensure_prefill_ws(s))) return rc;
    if ((rc = ensure_pages(s, s->len + m, st))) return rc;
    const KvGeom kv = kv_geom(s);
    HIP_TRY(copy_rows_launch(src, s->ph, m, H, st));
    auto gemm = [&](const unsigned short *X, const PackedLinear &pl, int N, int K, unsigned short *out, int ldo, int kind) -> int {
        return prefill_gemm(s, X, pl, m, N, K, out, ldo, kind, st);
    };
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerWeights &L = e->layers[l];
        HIP_TRY(add_rmsnorm_launch(s->ph, nullptr, 0, H, (const unsigned short *)L.ln_in, s->px, H, H, c.rms_eps, m, st));
        if ((rc = gemm(s->px, L.qkv, Nqkv, H, s->pqkv, Nqkv, LLM_GEMM_BF16))) return rc;
        HIP_TRY(rope_kv_append_launch(s->pqkv, m, nh, (const unsigned short *)e->cos_tab, (const unsigned short *)e->sin_tab, kv, l, s->len, s->pq, st));
        // the whole block's keys are appended: ONE attention launch, grid.z = the block's 16-query sub-chunks (causal mask per sub-chunk,
        // one split each); the output lands in px (the o-proj's X operand)
        hipError_t ae = attention_prefill_launch(s->pq, kv, l, nh, s->len, m, s->px, st);
        if (ae == hipErrorNotSupported) {
            if ((rc = ensure_prefill_partials(s))) return rc;
            ae = attention_launch(s->pq, kv, l, nh, s->len, m, s->ppart_o, s->ppart_ml, s->px, st, -1, VLO_PREFILL_TOKENS / 16);
        }
        HIP_TRY(ae);
        if ((rc = gemm(s->px, L.o, H, qd, s->ph, H, LLM_GEMM_RESID))) return rc;
        HIP_TRY(add_rmsnorm_launch(s->ph, nullptr, 0, H, (const unsigned short *)L.ln_post, s->px, H, H, c.rms_eps, m, st));
        if ((rc = gemm(s->px, L.gate_up, 2 * I, H, s->pact, I, LLM_GEMM_SWIGLU))) return rc;
        if ((rc = gemm(s->pact, L.down, H, I, s->ph, H, LLM_GEMM_RESID))) return rc;
    }
    if (want_last || all_logits) {
        HIP_TRY(add_rmsnorm_launch(s->ph, nullptr, 0, H, (const unsigned short *)e->norm_w, s->px, H, H, c.rms_eps, m, st));
        if (all_logits) {                       // every row, straight into the caller's matrix
            if (V % 256 == 0) {
                if ((rc = prefill_gemm(s, s->px, e->lm_head, m, V, H, all_logits, V, LLM_GEMM_BF16, st, false))) return rc;   // logits: bf16 activations always
            } else {                            // vocabularies that are not whole 256-column tiles: 16 rows at a time through the GEMV
                for (int r0 = 0; r0 < m; r0 += 16) {
                    GemvArgs a = gemv_args(e->lm_head, s->px + (size_t)r0 * H, H, std::min(16, m - r0));
                    a.out_bf16 = all_logits + (size_t)r0 * V; a.ldo = V;
                    HIP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
                }
            }
            HIP_TRY(hipMemcpyAsync(s->logits, all_logits + (size_t)(m - 1) * V, (size_t)V * 2, hipMemcpyDeviceToDevice, st));
        } else {                                // only the row that is read
            GemvArgs a = gemv_args(e->lm_head, s->px + (size_t)(m - 1) * H, H, 1);
            a.out_bf16 = s->logits; a.ldo = V;
            HIP_TRY(gemv_launch(a, e->lm_head.plan, XSRC_PLAIN, EPI_BF16, st));
        }
        s->last_logits = s->logits;
        s->has_logits = true;
    }
    s->len += m;
    return VLO_OK;
}

// shapes the prefill GEMMs take (bf16 image, or the fp8 image expanded per GEMM): every projection width a multiple of 256, every K of 128 —
// of THIS rank's shard under tensor parallelism (tp.hip::tp_prefill: column-sharded q|k|v and gate|up, row-sharded o and down)
bool prefill_ok(const vlo_engine *e) {
    static const bool on = getenv("VLO_PREFILL") ? atoi(getenv("VLO_PREFILL")) != 0 : true;
    const vlo_config &c = e->cfg;
    const int hd = e->head_dim, qd = e->nh_l * hd, Nqkv = qd + 2 * e->nkv_l * hd;
    return on && !(Nqkv & 255) && !(c.hidden_size & 255) && !((2 * e->I_l) & 255) && !(e->I_l & 127) && !(qd & 127) && (hd == 64 || hd == 128);
}

int vlo_llm_step(vlo_session *s, const void *embeds_dev, int n, void *last_logits_dev, void *all_logits_dev, void *stream) {
    if (!s || !embeds_dev || n <= 0) return fail(VLO_E_INVALID, "bad llm_step arguments");
    vlo_engine *e = s->e;
    if (e->tp_size > 1) return fail(VLO_E_STATE, "tensor-parallel engine: step it through vlo_tp_llm_step");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    const int H = e->cfg.hidden_size, V = e->cfg.vocab_size;
    int rc;
    // inputs longer than one 16-row chunk go through the 64-token block path (one weight pass per 64 tokens); the live
    // frame / decode steps (n <= 16) keep the fused 16-row pipeline.  VLO_BLOCK_PATH=0 forces 16-row chunks everywhere.
    static const bool block_path = getenv("VLO_BLOCK_PATH") ? atoi(getenv("VLO_BLOCK_PATH")) != 0 : true;
    for (int c0 = 0; c0 < n;) {
        const int left = n - c0;
        const unsigned short *src = (const unsigned short *)embeds_dev + (size_t)c0 * H;
        unsigned short *all = all_logits_dev ? (unsigned short *)all_logits_dev + (size_t)c0 * V : nullptr;
        if (block_path && left >= VLO_PREFILL_MIN && prefill_ok(e)) {
            // long inputs: blocks of up to VLO_PREFILL_TOKENS tokens with the projections as GEMMs; what is left below VLO_PREFILL_MIN
            // goes through the 64-token block path (so a remainder never starts a new kind of weight pass for a handful of tokens)
            const int m = std::min(VLO_PREFILL_TOKENS, left);
            if ((rc = run_prefill(s, src, m, c0 + m == n, all, st))) return rc;
            c0 += m;
            continue;
        }
        if (block_path && left > 16) {
            const int m = std::min(VLO_BLOCK_TOKENS, left);
            if ((rc = run_block(s, src, m, c0 + m == n, all, st))) return rc;
            c0 += m;
            continue;
        }
        const int m = std::min(16, left);
        if ((rc = run_chunk(s, src, m, c0 + m == n, all != nullptr, st))) return rc;
        if (all) HIP_TRY(hipMemcpyAsync(all, s->logits, (size_t)m * V * 2, hipMemcpyDeviceToDevice, st));
        c0 += m;
    }
    if (last_logits_dev) HIP_TRY(hipMemcpyAsync(last_logits_dev, s->last_logits, (size_t)V * 2, hipMemcpyDeviceToDevice, st));
    return VLO_OK;
}

int vlo_stream_sample(vlo_session *s, float threshold, int interval_id, int64_t *tok_dev, float *p_interval_dev, void *stream) {
    if (!s || !tok_dev) return fail(VLO_E_INVALID, "bad stream_sample arguments");
    if (!s->has_logits) return fail(VLO_E_STATE, "no logits: call vlo_llm_step first");
    HIP_TRY(hipSetDevice(s->e->device));
    HIP_TRY(stream_sample_launch(s->last_logits, s->e->cfg.vocab_size, threshold, interval_id, tok_dev, p_interval_dev,
                                 s->sample_scratch, (hipStream_t)stream));
    return VLO_OK;
}

int vlo_embed(vlo_engine *e, const int64_t *ids_dev, int k, void *out_dev, void *stream) {
    if (!e || !ids_dev || !out_dev || k <= 0) return fail(VLO_E_INVALID, "bad embed arguments");
    if (!e->finalized) return fail(VLO_E_STATE, "engine not finalized");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(embed_gather_launch((const unsigned short *)e->embed, ids_dev, k, e->cfg.hidden_size, e->cfg.vocab_size,
                                (unsigned short *)out_dev, (hipStream_t)stream));
    return VLO_OK;
}

int vlo_step_input(vlo_engine *e, const int64_t *ids_host, int k, const void *frame_rows_dev, int rows, void *out_dev, void *stream) {
    if (!e || !out_dev || k < 0 || rows < 0 || k + rows <= 0 || (k > 0 && !ids_host) || (rows > 0 && !frame_rows_dev))
        return fail(VLO_E_INVALID, "bad step_input arguments");
    if (!e->finalized) return fail(VLO_E_STATE, "engine not finalized");
    HIP_TRY(hipSetDevice(e->device));
    const int H = e->cfg.hidden_size;
    unsigned short *out = (unsigned short *)out_dev;
    // ids travel as kernel arguments, VLO_STEP_IDS_MAX per launch; the frame rows ride on the last launch
    int done = 0;
    do {
        StepIds ids{};
        const int kk = std::min(VLO_STEP_IDS_MAX, k - done);
        for (int i = 0; i < kk; ++i) ids.v[i] = ids_host[done + i];
        const bool last = done + kk == k;
        HIP_TRY(step_input_launch((const unsigned short *)e->embed, ids, kk, (const unsigned short *)frame_rows_dev, last ? rows : 0, H,
                                  e->cfg.vocab_size, out + (size_t)done * H, (hipStream_t)stream));
        done += kk;
    } while (done < k);
    return VLO_OK;
}

int vlo_greedy_generate(vlo_session *s, const void *embeds_dev, int m, int eos_token_id, int64_t *out_ids_dev, int max_new,
                        int force_len, int *n_written, void *stream) {
    if (!s || !embeds_dev || !out_ids_dev || m <= 0 || max_new <= 0) return fail(VLO_E_INVALID, "bad greedy_generate arguments");
    vlo_engine *e = s->e;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    const int V = e->cfg.vocab_size;
    if (force_len > max_new) force_len = max_new;
    int rc = vlo_llm_step(s, embeds_dev, m, nullptr, nullptr, stream);
    if (rc) return rc;
    for (hipEvent_t &ev : s->tok_ev)
        if (!ev) HIP_TRY(hipEventCreate(&ev));
    const bool forced = force_len > 0;
    int i = 0;
    for (;; ++i) {
        int mode = 0;
        if (forced) mode = (i == force_len - 1) ? 2 : 1;
        HIP_TRY(greedy_sample_launch(s->last_logits, V, out_ids_dev + i, eos_token_id, mode, s->sample_scratch, st));
        const bool last = (i == max_new - 1) || (forced && i == force_len - 1);
        if (forced) {
            // scheduled speech (throughput runs): the token count is known, the host never has to look at a token
            if (last) break;
            HIP_TRY(embed_gather_launch((const unsigned short *)e->embed, out_ids_dev + i, 1, e->cfg.hidden_size, V, s->emb1, st));
            if ((rc = vlo_llm_step(s, s->emb1, 1, nullptr, nullptr, stream))) return rc;
            continue;
        }
        // the reference reads every token on the host (`if new_token_id == eos_token_id`, :179).  Here the read is asynchronous
        // and the NEXT step is enqueued before the host waits for it, so the GPU never idles across the host round trip; if the
        // token turns out to be EOS, the speculative step is undone by forgetting its one KV position (nothing reads it again).
        volatile int64_t *slot = s->host_tok + (i & 1);
        HIP_TRY(hipMemcpyAsync((void *)slot, out_ids_dev + i, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(s->tok_ev[i & 1], st));
        // speculate only when the step needs nothing it could fail to get: a position inside the pages the session already owns.
        // (An EOS on the last KV slot, or one page short of an exhausted pool, must end the response as the reference's loop
        // does, not fail on a step that would be thrown away.)
        const bool spec = !last && s->len + 1 <= (int64_t)s->pages.size() * VLO_PAGE_TOKENS && s->len + 1 <= e->max_positions;
        if (spec) {
            HIP_TRY(embed_gather_launch((const unsigned short *)e->embed, out_ids_dev + i, 1, e->cfg.hidden_size, V, s->emb1, st));
            if ((rc = vlo_llm_step(s, s->emb1, 1, nullptr, nullptr, stream))) return rc;
        }
        HIP_TRY(hipEventSynchronize(s->tok_ev[i & 1]));
        if (*slot == eos_token_id) {
            if (spec) s->len -= 1;           // the EOS token is never fed to the model (:179-181)
            // same session state after EOS on both paths: the logits that produced EOS are consumed (the speculative step has
            // overwritten them anyway), a sampler call needs a new step first
            s->has_logits = false;
            break;
        }
        if (last) break;
        if (!spec) {                         // the blocking order of the reference: token known, then the step (which may need a page)
            HIP_TRY(embed_gather_launch((const unsigned short *)e->embed, out_ids_dev + i, 1, e->cfg.hidden_size, V, s->emb1, st));
            if ((rc = vlo_llm_step(s, s->emb1, 1, nullptr, nullptr, stream))) { if (n_written) *n_written = i + 1; return rc; }
        }
    }
    HIP_TRY(hipStreamSynchronize(st));        // the ids are read by the caller right away
    if (n_written) *n_written = i + 1;
    return VLO_OK;
}

// connector scratch (allocated outside any stream capture)
int vlo_connector_reserve(vlo_engine *e) {
    if (e->conn_x) return VLO_OK;
    const int H = e->cfg.hidden_size, Hv = e->cfg.vision_hidden_size;
    int rc;
    if ((rc = dev_alloc(&e->conn_x, (size_t)2 * 32 * Hv * 2))) return rc;
    if ((rc = dev_alloc(&e->conn_mid, (size_t)2 * 32 * H * 2))) return rc;
    if ((rc = dev_alloc(&e->conn_out, (size_t)2 * 32 * H * 2))) return rc;
    e->owned.push_back(e->conn_x);
    e->owned.push_back(e->conn_mid);
    e->owned.push_back(e->conn_out);
    HIP_TRY(hipMemset(e->conn_x, 0, (size_t)2 * 32 * Hv * 2));
    HIP_TRY(hipMemset(e->conn_mid, 0, (size_t)2 * 32 * H * 2));
    return VLO_OK;
}

int connector_run(vlo_engine *e, int slot, const void *feats_dev, int rows, void *out_dev, hipStream_t st) {
    const int H = e->cfg.hidden_size, Hv = e->cfg.vision_hidden_size;
    int rc;
    if ((rc = vlo_connector_reserve(e))) return rc;
    unsigned short *cx = (unsigned short *)e->conn_x + (size_t)slot * 32 * Hv, *cm = (unsigned short *)e->conn_mid + (size_t)slot * 32 * H,
                   *co = (unsigned short *)e->conn_out + (size_t)slot * 32 * H;
    for (int r0 = 0; r0 < rows; r0 += 16) {
        const int m = std::min(16, rows - r0);
        HIP_TRY(copy_rows_launch((const unsigned short *)feats_dev + (size_t)r0 * Hv, cx, m, Hv, st));
        {
            GemvArgs a = gemv_args(e->conn0, cx, Hv, m);
            a.out_bf16 = cm; a.ldo = H; a.bias = (const unsigned short *)e->conn0_b;
            HIP_TRY(gemv_launch(a, e->conn0.plan, XSRC_PLAIN, EPI_BF16_GELU_ERF, st));
        }
        {
            GemvArgs a = gemv_args(e->conn2, cm, H, m);
            a.out_bf16 = co; a.ldo = H; a.bias = (const unsigned short *)e->conn2_b;
            HIP_TRY(gemv_launch(a, e->conn2.plan, XSRC_PLAIN, EPI_BF16, st));
        }
        HIP_TRY(copy_rows_launch(co, (unsigned short *)out_dev + (size_t)r0 * H, m, H, st));
    }
    return VLO_OK;
}

int vlo_connector(vlo_engine *e, const void *feats_dev, int rows, void *out_dev, void *stream) {
    if (!e || !feats_dev || !out_dev || rows <= 0) return fail(VLO_E_INVALID, "bad connector arguments");
    if (!e->finalized || !e->has_connector) return fail(VLO_E_STATE, "connector weights not loaded");
    HIP_TRY(hipSetDevice(e->device));
    return connector_run(e, 0, feats_dev, rows, out_dev, (hipStream_t)stream);
}

int vlo_visual_embed(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, void *stream) {
    if (!e || !frames_dev || !out_dev || B <= 0) return fail(VLO_E_INVALID, "bad visual_embed arguments");
    if (!e->finalized || !e->cfg.has_vit || !e->vit) return fail(VLO_E_STATE, "engine built without a vision tower");
    HIP_TRY(hipSetDevice(e->device));
    return vit_visual_embed(e, frames_dev, B, out_dev, (hipStream_t)stream);
}

int vlo_vision_tokens(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, void *stream) {
    if (!e || !frames_dev || !out_dev || B <= 0) return fail(VLO_E_INVALID, "bad vision_tokens arguments");
    if (!e->finalized || !e->cfg.has_vit || !e->vit) return fail(VLO_E_STATE, "engine built without a vision tower");
    HIP_TRY(hipSetDevice(e->device));
    return vit_vision_tokens(e, frames_dev, B, out_dev, (hipStream_t)stream);
}

int vlo_session_read_kv(vlo_session *s, int layer, int which, int kv_head, int64_t t0, int64_t t1, void *dst_dev, void *stream) {
    if (!s || !dst_dev || layer < 0 || layer >= s->e->cfg.num_layers || kv_head < 0 || kv_head >= s->e->nkv_l ||
        t0 < 0 || t1 > s->len || t1 < t0)
        return fail(VLO_E_INVALID, "bad read_kv arguments");
    HIP_TRY(hipSetDevice(s->e->device));
    HIP_TRY(read_kv_launch(kv_geom(s), layer, which, kv_head, t0, t1, (unsigned short *)dst_dev, (hipStream_t)stream));
    return VLO_OK;
}

// ---- teacher-forced evaluation surface (models/modeling_live.py:29-42, 44-168, 170-171) -------------------------------
int vlo_joint_embed(vlo_engine *e, const int64_t *ids_dev, int k, int64_t v_placeholder_id, const void *frame_rows_dev,
                    int n_frame_rows, void *out_dev, void *stream) {
    if (!e || !ids_dev || !out_dev || k <= 0 || n_frame_rows < 0 || (n_frame_rows > 0 && !frame_rows_dev))
        return fail(VLO_E_INVALID, "bad joint_embed arguments");
    if (!e->finalized) return fail(VLO_E_STATE, "engine not finalized");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    int *scratch = nullptr;
    HIP_TRY(hipMalloc((void **)&scratch, ((size_t)k + 1) * sizeof(int)));
    hipError_t he = joint_embed_launch((const unsigned short *)e->embed, ids_dev, k, v_placeholder_id, (const unsigned short *)frame_rows_dev,
                                       n_frame_rows, e->cfg.hidden_size, e->cfg.vocab_size, scratch, scratch + k,
                                       (unsigned short *)out_dev, st);
    int count = -1;
    if (he == hipSuccess) he = hipMemcpyAsync(&count, scratch + k, sizeof(int), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    hipFree(scratch);
    HIP_TRY(he);
    // `inputs_embeds[v_mask] = self.visual_embed(frames)` raises on a shape mismatch (:41)
    if (count != n_frame_rows)
        return fail(VLO_E_INVALID, std::to_string(count) + " placeholder positions but " + std::to_string(n_frame_rows) +
                                       " frame-token embeddings");
    return VLO_OK;
}

int vlo_logit_rows(vlo_engine *e, const void *logits_dev, int n, const int64_t *labels_dev, int interval_id, float *lse_dev,
                   int64_t *argmax_dev, float *label_logit_dev, float *p_interval_dev, int64_t *p_argmax_dev, void *stream) {
    if (!e || !logits_dev || n <= 0 || !lse_dev || !argmax_dev || !label_logit_dev || !p_interval_dev || !p_argmax_dev)
        return fail(VLO_E_INVALID, "bad logit_rows arguments");
    HIP_TRY(hipSetDevice(e->device));
    const int V = e->cfg.vocab_size;
    HIP_TRY(logit_rows_launch((const unsigned short *)logits_dev, n, V, V, labels_dev, interval_id, lse_dev, argmax_dev, label_logit_dev,
                              p_interval_dev, p_argmax_dev, (hipStream_t)stream));
    return VLO_OK;
}

// fork / crop of ONE KV shard (a whole TP = 1 session, or one rank's shard of a tensor-parallel session: tp.hip)
int session_fork_shard(vlo_session *src, int64_t n_tokens, vlo_session **out, void *stream) {
    if (!src || !out || n_tokens < 0 || n_tokens > src->len) return fail(VLO_E_INVALID, "bad session_fork arguments");
    vlo_engine *e = src->e;
    hipStream_t st = (hipStream_t)stream;
    vlo_session *d = nullptr;
    int rc = vlo_session_create(e, n_tokens, &d);
    if (rc) return rc;
    if ((rc = ensure_pages(d, n_tokens, st))) {
        vlo_session_destroy(d);
        return rc;
    }
    const int pages = (int)((n_tokens + VLO_PAGE_TOKENS - 1) / VLO_PAGE_TOKENS);
    hipError_t he = kv_copy_pages_launch(kv_geom(src), src->page_table, d->page_table, pages, e->cfg.num_layers, st);
    if (he != hipSuccess) {
        vlo_session_destroy(d);
        HIP_TRY(he);
    }
    d->len = n_tokens;
    *out = d;
    return VLO_OK;
}

int vlo_session_fork(vlo_session *src, int64_t n_tokens, vlo_session **out, void *stream) {
    if (src && src->e->tp_size > 1) return fail(VLO_E_STATE, "a tensor-parallel shard is forked through vlo_tp_session_fork");
    return session_fork_shard(src, n_tokens, out, stream);
}

int session_crop_shard(vlo_session *s, int64_t n_tokens) {
    if (!s || n_tokens < 0 || n_tokens > s->len) return fail(VLO_E_INVALID, "bad session_crop arguments");
    vlo_engine *e = s->e;
    const size_t keep = (size_t)((n_tokens + VLO_PAGE_TOKENS - 1) / VLO_PAGE_TOKENS);
    if (keep < s->pages.size()) {
        // pages go back to the pool (and page-table slots will be rewritten): let queued kernels that still read them drain
        HIP_TRY(hipSetDevice(e->device));
        HIP_TRY(hipDeviceSynchronize());
        std::lock_guard<std::mutex> g(e->pool_mu);
        while (s->pages.size() > keep) {
            e->free_pages.push_back(s->pages.back());
            s->pages.pop_back();
        }
    }
    s->len = n_tokens;
    s->has_logits = false;
    return VLO_OK;
}

int vlo_session_crop(vlo_session *s, int64_t n_tokens) {
    if (s && s->e->tp_size > 1) return fail(VLO_E_STATE, "a tensor-parallel shard is cropped through vlo_tp_session_crop");
    return session_crop_shard(s, n_tokens);
}

// device scratch of the test / micro-benchmark entry points: freed on every return path
struct ScratchBufs {
    std::vector<void *> ptrs;
    std::vector<hipEvent_t> events;
    ~ScratchBufs() {
        for (void *p : ptrs) hipFree(p);
        for (hipEvent_t e : events) hipEventDestroy(e);
    }
    hipError_t alloc(void **p, size_t bytes) {
        const hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    hipError_t event(hipEvent_t *ev) {
        const hipError_t e = hipEventCreate(ev);
        if (e == hipSuccess) events.push_back(*ev);
        return e;
    }
};

int vlo_test_gemv(const void *x_dev, const void *W_dev, float *y_dev, int n, int N, int K, void *stream) {
    if (!x_dev || !W_dev || !y_dev || n <= 0 || n > 16 || N <= 0 || (N & 3)) return fail(VLO_E_INVALID, "bad test_gemv arguments");
    hipStream_t st = (hipStream_t)stream;
    GemvPlan plan;
    // VLO_TEST_GEMV_WHOLE_K=1 (tests): one K slice per block, the waves walk KC chunks — the plan the step's whole-K launches use
    const bool whole_k = getenv("VLO_TEST_GEMV_WHOLE_K") && atoi(getenv("VLO_TEST_GEMV_WHOLE_K"));
    if (gemv_plan(K, !whole_k, &plan)) return fail(VLO_E_UNSUPPORTED, "no GEMV plan for K");
    const int NT = (N + 15) / 16;
    ScratchBufs sc;
    void *Wp = nullptr, *xp = nullptr;
    float *P = nullptr, *Y = nullptr;
    HIP_TRY(sc.alloc(&Wp, (size_t)NT * 16 * K * 2));
    HIP_TRY(sc.alloc(&xp, (size_t)32 * K * 2));
    HIP_TRY(sc.alloc((void **)&P, (size_t)plan.ksplit * 16 * NT * 16 * 4));
    HIP_TRY(sc.alloc((void **)&Y, (size_t)16 * NT * 16 * 4));
    HIP_TRY(hipMemsetAsync(xp, 0, (size_t)32 * K * 2, st));
    HIP_TRY(hipMemcpyAsync(xp, x_dev, (size_t)n * K * 2, hipMemcpyDeviceToDevice, st));
    HIP_TRY(pack_weight_launch(W_dev, Wp, N, K, K, NT, 1, 0, -1, st));
    GemvArgs a{};
    a.Wp = Wp; a.x = (const unsigned short *)xp; a.out_f32 = P;
    a.K = K; a.ldx = K; a.ldo = NT * 16; a.NT = NT; a.N_valid = N; a.n_rows = n;
    HIP_TRY(gemv_launch(a, plan, XSRC_PLAIN, EPI_PARTIAL_F32, st));
    // partial layout is [ksplit][16][ldo]; y is [n][N]: sum into a padded buffer, then copy the rows (ldo may exceed N)
    hipLaunchKernelGGL(sum_partials_kernel, dim3((n * NT * 16 + 255) / 256), dim3(256), 0, st, P, plan.ksplit, NT * 16, Y, n, NT * 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(y_dev, (size_t)N * 4, Y, (size_t)NT * 16 * 4, (size_t)N * 4, n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));      // the scratch is freed when `sc` goes out of scope
    return VLO_OK;
}

int vlo_test_gemv_fp8(const void *x_dev, const void *Wq_dev, const float *scale_dev, float *y_dev, int n, int N, int K, void *stream) {
    if (!x_dev || !Wq_dev || !scale_dev || !y_dev || n <= 0 || n > 16 || N <= 0 || (N & 3)) return fail(VLO_E_INVALID, "bad test_gemv_fp8 arguments");
    hipStream_t st = (hipStream_t)stream;
    GemvPlan plan;
    const bool whole_k = getenv("VLO_TEST_GEMV_WHOLE_K") && atoi(getenv("VLO_TEST_GEMV_WHOLE_K"));
    if (gemv_plan(K, !whole_k, &plan)) return fail(VLO_E_UNSUPPORTED, "no GEMV plan for K");
    if ((plan.KF & 1) || !(plan.NW == 8 || (plan.NW == 4 && plan.KF == 14)) || (K & 63)) return fail(VLO_E_UNSUPPORTED, "fp8 weight image needs an even fragment count per wave for this K");
    const int NT = (N + 15) / 16;
    ScratchBufs sc;
    void *Wp = nullptr, *xp = nullptr;
    float *P = nullptr, *Y = nullptr, *sp = nullptr;
    HIP_TRY(sc.alloc(&Wp, (size_t)NT * 16 * K));
    HIP_TRY(sc.alloc(&xp, (size_t)32 * K * 2));
    HIP_TRY(sc.alloc((void **)&sp, (size_t)NT * 16 * 4));
    HIP_TRY(sc.alloc((void **)&P, (size_t)plan.ksplit * 16 * NT * 16 * 4));
    HIP_TRY(sc.alloc((void **)&Y, (size_t)16 * NT * 16 * 4));
    HIP_TRY(hipMemsetAsync(xp, 0, (size_t)32 * K * 2, st));
    HIP_TRY(hipMemcpyAsync(xp, x_dev, (size_t)n * K * 2, hipMemcpyDeviceToDevice, st));
    HIP_TRY(pack_weight_fp8_launch(Wq_dev, scale_dev, Wp, sp, N, K, K, NT, 1, 0, -1, st));
    GemvArgs a{};
    a.Wp = Wp; a.wq = 1; a.wscale = sp; a.x = (const unsigned short *)xp; a.out_f32 = P;
    a.K = K; a.ldx = K; a.ldo = NT * 16; a.NT = NT; a.N_valid = N; a.n_rows = n;
    HIP_TRY(gemv_launch(a, plan, XSRC_PLAIN, EPI_PARTIAL_F32, st));
    hipLaunchKernelGGL(sum_partials_kernel, dim3((n * NT * 16 + 255) / 256), dim3(256), 0, st, P, plan.ksplit, NT * 16, Y, n, NT * 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(y_dev, (size_t)N * 4, Y, (size_t)NT * 16 * 4, (size_t)N * 4, n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return VLO_OK;
}

int vlo_test_gemm_fp8(const void *x_dev, const void *Wq_dev, const float *scale_dev, float *y_dev, void *xq_dev, float *xscale_dev, int M, int N, int K,
                      int iters, double *avg_us, void *stream) {
    if (!x_dev || !Wq_dev || !scale_dev || !y_dev || M <= 0 || !llm_gemm_fp8_ok(N, K)) return fail(VLO_E_INVALID, "bad test_gemm_fp8 arguments (N, K multiples of 256)");
    hipStream_t st = (hipStream_t)stream;
    const int NT = N / 16;
    const size_t RX = (size_t)((M + 255) / 256) * 256 + 256;
    ScratchBufs sc;
    void *Wp = nullptr, *xq = nullptr;
    float *sp = nullptr, *xs = nullptr;
    HIP_TRY(sc.alloc(&Wp, (size_t)N * K));
    HIP_TRY(sc.alloc(&xq, RX * K));
    HIP_TRY(sc.alloc((void **)&sp, (size_t)N * 4));
    HIP_TRY(sc.alloc((void **)&xs, RX * 4));
    HIP_TRY(hipMemsetAsync(xq, 0, RX * K, st));
    HIP_TRY(hipMemsetAsync(xs, 0, RX * 4, st));
    HIP_TRY(pack_weight_fp8_launch(Wq_dev, scale_dev, Wp, sp, N, K, K, NT, 1, 0, -1, st));
    HIP_TRY(quantize_rows_fp8_launch((const unsigned short *)x_dev, M, K, xq, xs, st));
    HIP_TRY(llm_gemm_fp8_launch(xq, xs, Wp, sp, M, N, K, y_dev, N, LLM_GEMM_F32, st));
    if (xq_dev) HIP_TRY(hipMemcpyAsync(xq_dev, xq, (size_t)M * K, hipMemcpyDeviceToDevice, st));
    if (xscale_dev) HIP_TRY(hipMemcpyAsync(xscale_dev, xs, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
    if (iters > 0 && avg_us) {                   // the GEMM alone, back to back (tools/probe_prefill.py --gemm)
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) HIP_TRY(llm_gemm_fp8_launch(xq, xs, Wp, sp, M, N, K, y_dev, N, LLM_GEMM_F32, st));
        HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *avg_us = (double)ms * 1e3 / iters;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    HIP_TRY(hipStreamSynchronize(st));
    return VLO_OK;
}

int vlo_bench_gemv(int N, int K, int n_rows, int epi, int iters, int nbuf, double *avg_us) {
    const int fp8 = (epi & 0x100) ? 1 : 0;          // bit 8 of `epi`: time the fp8 e4m3 weight image instead of the bf16 one
    epi &= 0xff;
    if (N <= 0 || K <= 0 || n_rows <= 0 || n_rows > 16 || iters <= 0 || nbuf <= 0 || !avg_us) return fail(VLO_E_INVALID, "bad bench_gemv arguments");
    GemvPlan plan;
    if (gemv_plan(K, epi == EPI_PARTIAL_F32, &plan)) return fail(VLO_E_UNSUPPORTED, "no GEMV plan for K");
    const int NT = (N + 15) / 16;
    const size_t wbytes = (size_t)NT * 16 * K * (fp8 ? 1 : 2);
    if (fp8 && ((plan.KF & 1) || plan.NW != 8)) return fail(VLO_E_UNSUPPORTED, "no fp8 GEMV for this K");
    ScratchBufs sc;
    std::vector<void *> Wp(nbuf, nullptr);
    void *x = nullptr, *o32 = nullptr, *o16 = nullptr, *hbuf = nullptr, *sq = nullptr, *nw = nullptr, *tab = nullptr, *kvp = nullptr;
    int *pt = nullptr;
    for (int i = 0; i < nbuf; ++i) {
        HIP_TRY(sc.alloc(&Wp[i], wbytes));
        HIP_TRY(hipMemset(Wp[i], 0x3c, wbytes));      // 0x3c3c = a small finite bf16
    }
    const size_t wide = (size_t)std::max(K, NT * 16);
    HIP_TRY(sc.alloc(&x, 32 * wide * 2));
    HIP_TRY(hipMemset(x, 0x3c, 32 * wide * 2));
    HIP_TRY(sc.alloc(&hbuf, 32 * wide * 2));
    HIP_TRY(hipMemset(hbuf, 0x3c, 32 * wide * 2));
    HIP_TRY(sc.alloc(&nw, wide * 2));
    HIP_TRY(hipMemset(nw, 0x3c, wide * 2));
    HIP_TRY(sc.alloc(&sq, 1024 * 16 * 4));
    HIP_TRY(hipMemset(sq, 0, 1024 * 16 * 4));
    HIP_TRY(sc.alloc(&o32, (size_t)plan.ksplit * 16 * NT * 16 * 4));
    HIP_TRY(sc.alloc(&o16, (size_t)16 * NT * 16 * 2));
    HIP_TRY(sc.alloc(&tab, (size_t)VLO_PAGE_TOKENS * 64 * 2));
    HIP_TRY(hipMemset(tab, 0x3c, (size_t)VLO_PAGE_TOKENS * 64 * 2));
    HIP_TRY(sc.alloc(&kvp, (size_t)NT * 16 * VLO_PAGE_TOKENS * 2));
    HIP_TRY(sc.alloc((void **)&pt, 64));
    HIP_TRY(hipMemset(pt, 0, 64));
    GemvArgs a{};
    void *wsc = nullptr;
    if (fp8) {
        HIP_TRY(sc.alloc(&wsc, (size_t)NT * 16 * 4));
        HIP_TRY(hipMemset(wsc, 0x3c, (size_t)NT * 16 * 4));
        a.wq = 1; a.wscale = (const float *)wsc;
    }
    a.x = (const unsigned short *)x; a.out_f32 = (float *)o32; a.out_bf16 = (unsigned short *)o16;
    a.K = K; a.ldx = K; a.ldo = (epi == EPI_SWIGLU) ? NT * 8 : NT * 16; a.NT = NT; a.N_valid = N; a.n_rows = n_rows;   // SwiGLU: 8 output columns per tile
    int xsrc = XSRC_PLAIN;
    if (epi == EPI_SWIGLU) {
        xsrc = XSRC_NORM;
        a.norm_w = (const unsigned short *)nw; a.sq_in = (const float *)sq; a.sq_in_parts = 128; a.eps = 1e-5f;
    }
    if (epi == EPI_RESID) { a.h = (unsigned short *)hbuf; a.sq_out = (float *)sq; }
    if (epi == EPI_ROPE) {
        // N = (nh + 2*nkv) * 128 with nkv = nh / 4 (Llama-3 GQA); one page, positions 0..n_rows-1
        const int heads = N / 128, nkv = heads / 6, nh = heads - 2 * nkv;
        a.cos_tab = a.sin_tab = (const unsigned short *)tab;
        a.kv.k_pool = a.kv.vt_pool = (unsigned short *)kvp; a.kv.page_table = pt; a.kv.layer_stride = 0;
        a.kv.page_elems = (int64_t)nkv * VLO_PAGE_TOKENS * 128; a.kv.num_kv_heads = nkv; a.kv.head_dim = 128;
        a.layer = 0; a.num_heads = nh; a.pos0 = 0;
    }
    hipEvent_t e0, e1;
    HIP_TRY(sc.event(&e0));
    HIP_TRY(sc.event(&e1));
    for (int i = 0; i < 3; ++i) { a.Wp = Wp[i % nbuf]; HIP_TRY(gemv_launch(a, plan, xsrc, epi, 0)); }
    HIP_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) { a.Wp = Wp[i % nbuf]; HIP_TRY(gemv_launch(a, plan, xsrc, epi, 0)); }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / iters;
    return VLO_OK;                         // `sc` frees the scratch and the events
}

int vlo_debug_read(vlo_session *s, int which, void *dst_dev, int64_t bytes, void *stream) {
    if (!s || !dst_dev || bytes <= 0) return fail(VLO_E_INVALID, "bad debug_read arguments");
    const void *src = which == 0 ? (void *)s->q : which == 1 ? (void *)s->attn : which == 2 ? (void *)s->h : which == 3 ? (void *)s->act : (void *)s->x;
    HIP_TRY(hipMemcpyAsync(dst_dev, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VLO_OK;
}

/* host-side geometry of the 64-token block path, for unit tests: plan (NW, KF, KC) for a K, packed-64 element offset */
int vlo_debug_gemm64_plan(int K, int *out3) {
    Gemm64Plan p;
    if (!out3 || gemm64_plan(K, &p)) return fail(VLO_E_UNSUPPORTED, "no block-GEMM plan for K=" + std::to_string(K));
    out3[0] = p.NW; out3[1] = p.KF; out3[2] = p.KC;
    return VLO_OK;
}
int64_t vlo_debug_pack64_elem(int row, int k) { return (row < 0 || row >= VLO_BLOCK_TOKENS || k < 0) ? -1 : (int64_t)vlo_pack64_elem(row, k); }

int vlo_debug_gemv_plan(int K, int allow_ksplit, int *out4) {
    GemvPlan p;
    if (!out4 || gemv_plan(K, allow_ksplit != 0, &p)) return fail(VLO_E_UNSUPPORTED, "no GEMV plan for K=" + std::to_string(K));
    out4[0] = p.NW; out4[1] = p.KF; out4[2] = p.KC; out4[3] = p.ksplit;
    return VLO_OK;
}
