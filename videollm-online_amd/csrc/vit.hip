// vit.hip — SigLIP vision tower + token selection on gfx950 (MFMA-bound part of the hot path).
//
// Restates, as hand-written HIP:
//   _siglip_vision_encode                 models/vision_live.py:10-30   (rescale, normalize, ViT, CLS + 3x3 pool)
//   SiglipVisionEmbeddings.forward        HF:models/siglip/modeling_siglip.py:175-186  (patch conv == GEMM, + pos)
//   SiglipEncoderLayer.forward  x L       HF:...:335-357  (LN, MHSA :273-307, MLP :318-322 with tanh-GELU)
//   post_layernorm + MAP head             HF:...:612-614, :633-644
//   LiveMixin.visual_embed                models/modeling_live.py:21-27 (-> bf16 -> connector, gemv.hip)
//
// Precision follows the reference's GPU path (torch.cuda.amp.autocast, models/vision_live.py:13):
// matmul operands and outputs fp16, accumulation fp32, LayerNorm / softmax / residual stream fp32.
//
// GEMM: out[m][n] = sum_k X[m][k] * W[n][k]; both operands K-contiguous.  v_mfma_f32_16x16x32_f16 with the
// WEIGHT tile as the A operand and the ACTIVATION tile as the B operand, so a lane ends up holding 4
// consecutive output columns of one token row (8-byte row-major stores).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>

#include "common.cuh"
#include "glds_asm.cuh"
#include "vit.h"

#include "vit_gemm.inc"

#include "vit_ln.inc"

#include "vit_attn.inc"
#include "vit_tall.inc"

// MAP head attention: one probe query per head over S keys.  grid = (heads, B), 256 threads.
// kv: fp16 [B*S][2D] (K | V row-major), qp: fp16 [D] (probe @ Wq + bq, precomputed at load)
// Scores: a thread per key (its K row: hd / 8 sixteen-byte loads, all in flight); P.V: threads = (hd / 4 dim groups) x (key groups), a thread walks
// every nkg-th key with one 8-byte load of V per key, the key groups' partial sums meet in LDS.  (Round 5's form gave a thread ONE dim and a
// quarter of the keys, two-byte loads a 4 KiB stride apart: 48.7 us for one frame.)
__global__ __launch_bounds__(256) void map_attn_kernel(const f16_t *__restrict__ kv, const f16_t *__restrict__ qp,
                                                       f16_t *__restrict__ out, int S, int D, int hd, float scale) {
    extern __shared__ float prob[];            // [max(S, 256 * 4)]
    __shared__ float sm[16];
    const int head = blockIdx.x, b = blockIdx.y;
    const f16_t *kb = kv + (size_t)b * S * 2 * D + (size_t)head * hd;
    const f16_t *vb = kb + D;
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        float s = 0.f;
        for (int d0 = 0; d0 < hd; d0 += 8) {       // 16-byte loads; hd % 8 == 0
            const uint4 qv = *reinterpret_cast<const uint4 *>(qp + head * hd + d0);
            const uint4 kv4 = *reinterpret_cast<const uint4 *>(kb + (size_t)t * 2 * D + d0);
            const f16_t *qe = reinterpret_cast<const f16_t *>(&qv), *ke = reinterpret_cast<const f16_t *>(&kv4);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += h2f(qe[j]) * h2f(ke[j]);
        }
        s *= scale;
        prob[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, sm);
    float sum = 0.f;
    for (int t = threadIdx.x; t < S; t += blockDim.x) {
        const float p = __expf(prob[t] - mx);
        prob[t] = p;
        sum += p;
    }
    sum = block_sum(sum, sm);
    __syncthreads();
    // threads = ndg dim groups (4 dims each) x nkg key groups
    const int ndg = hd >> 2, nkg = (int)blockDim.x / ndg;
    const int dg = threadIdx.x % ndg, kg = threadIdx.x / ndg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kg < nkg) {
#pragma unroll 4
        for (int t = kg; t < S; t += nkg) {
            const float p = rh(prob[t] / sum);
            const uint2 v4 = *reinterpret_cast<const uint2 *>(vb + (size_t)t * 2 * D + dg * 4);
            acc.x += p * h2f((f16_t)(v4.x & 0xffffu)); acc.y += p * h2f((f16_t)(v4.x >> 16));
            acc.z += p * h2f((f16_t)(v4.y & 0xffffu)); acc.w += p * h2f((f16_t)(v4.y >> 16));
        }
    }
    __syncthreads();
    float4 *red = reinterpret_cast<float4 *>(prob);          // reuse: [nkg][ndg]
    if (kg < nkg) red[kg * ndg + dg] = acc;
    __syncthreads();
    if (threadIdx.x < ndg) {
        float4 tot = red[threadIdx.x];
        for (int g = 1; g < nkg; ++g) { const float4 u = red[g * ndg + threadIdx.x]; tot.x += u.x; tot.y += u.y; tot.z += u.z; tot.w += u.w; }
        ushort4 o;
        o.x = f2h(tot.x); o.y = f2h(tot.y); o.z = f2h(tot.z); o.w = f2h(tot.w);
        *reinterpret_cast<ushort4 *>(out + (size_t)b * D + head * hd + threadIdx.x * 4) = o;
    }
}

// Linear on a handful of rows (the MAP head's out-proj / fc1 / fc2 run on ONE row per frame): a tile GEMM would walk K = 4096 in 64 steps on 16
// workgroups (21.8 us for one frame).  Here a wave owns one output column: lanes stride K in 16-byte chunks against up to 8 rows at a time (fp32
// sums, one wave reduction per row, epilogue on lane 0), rows beyond 8 in further passes over the column — every row's sum is the same sequence of
// operations whatever the batch size, so a frame encodes to the same bits alone or in a batch.  grid = N / 4, 256 threads.  Epilogues as
// gemm_store4 (same rounding points).
template <int EP>
__global__ __launch_bounds__(256) void vit_rowvec_kernel(GemmArgs a) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const f16_t *wr = a.W + (size_t)n * a.K;
    const float bias = a.bias[n];
    for (int m0 = 0; m0 < a.M; m0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 8; c < a.K; c += 512) {
            const uint4 wv = *reinterpret_cast<const uint4 *>(wr + c);
            const f16_t *we = reinterpret_cast<const f16_t *>(&wv);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m0 + m < a.M) {
                    const uint4 xv = *reinterpret_cast<const uint4 *>(a.X + (size_t)(m0 + m) * a.ldx + c);
                    const f16_t *xe = reinterpret_cast<const f16_t *>(&xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[m] += h2f(xe[j]) * h2f(we[j]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = wave_sum(acc[m]);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m0 + m >= a.M) break;
                const float v = acc[m] + bias;
                if (EP == EP_F16) a.out16[(size_t)(m0 + m) * a.ldo + n] = f2h(v);
                else if (EP == EP_F16_GELU) a.out16[(size_t)(m0 + m) * a.ldo + n] = f2h(gelu_tanh_f(rh(v)));
                else if (EP == EP_RESID) a.out32[(size_t)(m0 + m) * a.N + n] += rh(v);
            }
        }
    }
}
template <int EP>
static hipError_t rowvec_launch(const GemmArgs &a, hipStream_t st) {
    if (a.M < 1 || (a.K & 7) || (a.ldx & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((vit_rowvec_kernel<EP>), dim3((a.N + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

// frames (uint8 NCHW) -> the patch-embed GEMM's A operand, fp16 [M][Kp] (k = ch P P + py P + px, zero columns from Kreal up): the rescale + normalise of
// vision_live.py:12 exactly as the fused A-tile load of vit_gemm_kernel<EP_PATCH> computes it (same fp32 operations, same fp16 rounding).  One frame:
// this launch + the tall GEMM replace the register-ring kernel (30 us -> ~11).  8 elements per thread.
__global__ __launch_bounds__(256) void vit_im2col_kernel(const uint8_t *__restrict__ frames, f16_t *__restrict__ X, int M, int S, int G, int R, int P, int Kreal, int Kp) {
    const int cpr = Kp / 8;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * cpr) return;
    const int m = idx / cpr, k0 = (idx - m * cpr) * 8;
    const int b = m / S, t = m % S, gy = t / G, gx = t % G;
    f16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kj = k0 + j;
        const bool live = kj < Kreal;
        const int ch = kj / (P * P), rem = kj % (P * P), py = rem / P, px = rem % P;
        const uint8_t e = live ? frames[(((size_t)b * 3 + ch) * R + gy * P + py) * R + gx * P + px] : (uint8_t)0;
        const float x = (float)e * 0.00392156862745098f;
        o[j] = live ? f2h((x - 0.5f) / 0.5f) : (f16_t)0;
    }
    *reinterpret_cast<uint4 *>(X + (size_t)m * Kp + k0) = *reinterpret_cast<const uint4 *>(o);
}

// CLS + pooled tokens -> bf16 [B][1 + ph*pw][D]   (adaptive_avg_pool2d with exact G/ph blocks; vision_live.py:16-30)
__global__ __launch_bounds__(256) void pool_concat_kernel(const float *__restrict__ last, const float *__restrict__ cls,
                                                          bf16_t *__restrict__ out, int G, int D, int ph, int pw) {
    const int b = blockIdx.y, tok = blockIdx.x, T = 1 + ph * pw, S = G * G;
    const int d = blockIdx.z * 256 + threadIdx.x;
    if (d >= D) return;
    float v;
    if (tok == 0) {
        v = cls[(size_t)b * D + d];
    } else {
        const int py = (tok - 1) / pw, px = (tok - 1) % pw;
        // adaptive pooling windows: [floor(i*G/p), ceil((i+1)*G/p))
        const int y0 = (py * G) / ph, y1 = ((py + 1) * G + ph - 1) / ph;
        const int x0 = (px * G) / pw, x1 = ((px + 1) * G + pw - 1) / pw;
        float s = 0.f;
        for (int y = y0; y < y1; ++y) {
            const float *row = last + ((size_t)b * S + (size_t)y * G) * D + d;
#pragma unroll 8
            for (int x = x0; x < x1; ++x) s += row[(size_t)x * D];
        }
        v = s / (float)((y1 - y0) * (x1 - x0));
    }
    out[((size_t)b * T + tok) * D + d] = f2bf(v);       // frames.to(self.dtype)  (modeling_live.py:25)
}

// residual for the MAP head: out32[b][d] = a16[b][d] + (acc16 computed by EP_F32 gemm) — done inline via EP_F32 + add
__global__ void add_f16_f32_kernel(const f16_t *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = h2f(a[i]) + b[i];
}
__global__ void f16_to_f32_kernel(const f16_t *__restrict__ a, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = h2f(a[i]);
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static size_t attn_lds(int hdv, int qs = 4) { return (size_t)4 * qs * (hdv / 16) * 64 * 16 + 4 * qs * 16 * 2 * 4; }     // O partials + (m, l) of vit_attn_kernel
static const size_t kAttnLds = attn_lds(64);
static const size_t kAttn8Lds = (size_t)8 * 4 * 4 * 64 * 16 + 8 * 4 * 16 * 2 * 4;     // vit_attn_split8_kernel: 8 waves' O partials + (m, l)

struct VitLayer {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    f16_t *wqkv;
    float *bqkv;
    const f16_t *wo, *w1, *w2;
    const float *bo, *b1, *b2;
};

struct VitState {
    int D, I, L, nh, hd, R, P, G, S, Sp, ph, pw;             // Sp: S rounded up to 32 (row length of V^T); I: the MLP width the kernels see (padded)
    int hdk = 64, hdv = 64;                                  // columns per head in the q | k buffer (hd up to 32) / rows per head in V^T (hd up to 16)
    int Kpe = 0, Kpe_real = 0;                               // patch-embed K: 3 P P, and rounded up to 64 (zero columns in the padded weight)
    float eps;
    const f16_t *wpe;
    const float *bpe, *pos;
    std::vector<VitLayer> layers;
    const float *post_w, *post_b;
    const f16_t *in_proj_w;      // [3D][D]
    const float *in_proj_b;
    const f16_t *hout_w, *hfc1_w, *hfc2_w;
    const float *hout_b, *hln_w, *hln_b, *hfc1_b, *hfc2_b;
    f16_t *q_probe;              // [D]
    // workspace for up to Bcap frames
    int Bcap = 0;
    float *h = nullptr, *last = nullptr, *cls32 = nullptr, *tmp32 = nullptr;
    float *slab = nullptr;               // split-K partial sums of the residual GEMMs at few frames: [4][slab_rows][D] fp32
    int slab_rows = 0;
    f16_t *x16 = nullptr, *qk16 = nullptr, *vT = nullptr, *att16 = nullptr, *mid16 = nullptr, *kv16 = nullptr;
    f16_t *hx16 = nullptr, *hmid16 = nullptr, *hatt16 = nullptr, *ho16 = nullptr;
    bf16_t *tokens = nullptr;
    uint8_t *frames_in = nullptr;        // graph-stable staging of the input frames
    bf16_t *out_stage = nullptr;         // graph-stable staging of the output embeddings
    std::map<int, hipGraphExec_t> graphs;     // batch size -> captured encode
    int attn_vrs = 0;                         // vit_attn_head_kernel: bytes per V^T row in LDS ((vrs / 16) % 16 == 10: conflict-free fragment reads)
    size_t attn_head_lds = 0;                 // its LDS footprint, 0 = the head does not fit (Sp > 608)
    hipStream_t st2 = nullptr;                // second branch of the captured encode (two half-batches run concurrently)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<void *> ws;
};


#define VIT_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) return vlo_fail(VLO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static int vtake(vlo_engine *e, const std::string &name, std::vector<int64_t> shape, int dtype, const void **out) {
    auto it = e->raw.find(name);
    if (it == e->raw.end()) return vlo_fail(VLO_E_MISSING, "missing weight: " + name);
    if (it->second.shape != shape || it->second.dtype != dtype) return vlo_fail(VLO_E_INVALID, "bad shape/dtype for " + name);
    *out = it->second.ptr;
    e->owned.push_back(it->second.ptr);
    size_t n = 1;
    for (auto d : shape) n *= (size_t)d;
    e->weight_bytes += (int64_t)(n * (dtype == VLO_DT_F32 ? 4 : 2));
    e->raw.erase(it);
    return VLO_OK;
}

int vit_finalize(vlo_engine *e) {
    const vlo_config &c = e->cfg;
    VitState *v = new VitState();
    v->D = c.vit_hidden_size; v->I = c.vit_intermediate_size; v->L = c.vit_num_layers; v->nh = c.vit_num_heads;
    v->hd = v->D / v->nh; v->R = c.vit_image_size; v->P = c.vit_patch_size; v->G = v->R / v->P; v->S = v->G * v->G; v->Sp = (v->S + 31) & ~31;
    v->ph = c.pool_h; v->pw = c.pool_w; v->eps = c.vit_ln_eps;
    // Shapes the kernels take as they are: head dim 64, MLP width % 256 == 0, 3 P P % 64 == 0 (SigLIP-L/16: 64 / 4096 / 768).  Others are PADDED
    // with zeros where the padding is arithmetically inert (SigLIP-so400m/14: head dim 72, MLP 4304, 3 P P = 588):
    //   * MLP width I -> Ip = I up to 256: fc1 gets zero rows + zero bias (GELU(0) = 0), fc2 zero columns;
    //   * patch-embed K = 3 P P -> up to 64: zero weight columns, zero A-tile elements;
    //   * head dim hd -> hdk = hd up to 32 columns per head in the q | k buffer (zeros add 0 to q.k) and hdv = hd up to 16 rows per head in V^T
    //     (zero rows: outputs never stored).  The hidden size itself (LayerNorm width, residual stream) is never padded.
    const int D = v->D, I = c.vit_intermediate_size, Ip = (I + 255) / 256 * 256;
    v->I = Ip;
    v->Kpe_real = 3 * v->P * v->P; v->Kpe = (v->Kpe_real + 63) / 64 * 64;
    v->hdk = (v->hd + 31) / 32 * 32; v->hdv = (v->hd + 15) / 16 * 16;
    // head dim 64 exactly, or 68..80 (padded to 96 q|k columns / 80 V^T rows).  52..60 would also round up to 64 / 64, but the
    // unpadded kernel stores 64 columns per head at a stride of hd: rejected, not silently wrong
    const bool heads_ok = v->hd == 64 || (v->hd >= 68 && v->hd <= 80 && v->hdk == 96 && v->hdv == 80);
    if (v->nh <= 0 || v->hd * v->nh != D || (v->hd & 3) || !heads_ok || (D % 64) || D > 2048 || (I & 3) || c.vision_hidden_size != D ||
        c.frame_num_tokens != 1 + v->ph * v->pw || v->G <= 0)          // G = R / P rounded down: a strided 'valid' conv ignores the remainder (384 / 14)
        { delete v; return vlo_fail(VLO_E_UNSUPPORTED, "vision tower shape not covered by the kernels (head dim 64 or 68..80 in steps of 4, hidden % 64 == 0 and <= 2048)"); }
    int rc;
    // [rows][cols] fp16 (or a float vector) -> zero-padded [rows_p][cols_p] copy owned by the engine
    auto padded16 = [&](const f16_t *&w, int rows, int cols, int rows_p, int cols_p) -> int {
        if (rows == rows_p && cols == cols_p) return VLO_OK;
        f16_t *d;
        int r2 = dev_alloc((void **)&d, (size_t)rows_p * cols_p * 2);
        if (r2) return r2;
        e->owned.push_back(d);
        if (hipMemset(d, 0, (size_t)rows_p * cols_p * 2) != hipSuccess ||
            hipMemcpy2D(d, (size_t)cols_p * 2, w, (size_t)cols * 2, (size_t)cols * 2, rows, hipMemcpyDeviceToDevice) != hipSuccess)
            return vlo_fail(VLO_E_HIP, "padding a vision weight failed");
        w = d;
        return VLO_OK;
    };
    auto padded32 = [&](const float *&b, int n, int n_p) -> int {
        if (n == n_p) return VLO_OK;
        float *d;
        int r2 = dev_alloc((void **)&d, (size_t)n_p * 4);
        if (r2) return r2;
        e->owned.push_back(d);
        if (hipMemset(d, 0, (size_t)n_p * 4) != hipSuccess || hipMemcpy(d, b, (size_t)n * 4, hipMemcpyDeviceToDevice) != hipSuccess)
            return vlo_fail(VLO_E_HIP, "padding a vision bias failed");
        b = d;
        return VLO_OK;
    };
#define PAD(call) if ((rc = (call))) { delete v; return rc; }
#define TK(name, shape, dt, field) if ((rc = vtake(e, name, shape, dt, (const void **)&field))) { delete v; return rc; }
    TK("vision.embeddings.patch_embedding.weight", (std::vector<int64_t>{D, 3, v->P, v->P}), VLO_DT_F16, v->wpe);
    PAD(padded16(v->wpe, D, v->Kpe_real, D, v->Kpe));
    TK("vision.embeddings.patch_embedding.bias", (std::vector<int64_t>{D}), VLO_DT_F32, v->bpe);
    TK("vision.embeddings.position_embedding.weight", (std::vector<int64_t>{v->S, D}), VLO_DT_F32, v->pos);
    v->layers.resize(v->L);
    for (int l = 0; l < v->L; ++l) {
        VitLayer &Ly = v->layers[l];
        const std::string p = "vision.encoder.layers." + std::to_string(l) + ".";
        TK(p + "layer_norm1.weight", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.ln1_w);
        TK(p + "layer_norm1.bias", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.ln1_b);
        TK(p + "layer_norm2.weight", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.ln2_w);
        TK(p + "layer_norm2.bias", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.ln2_b);
        // fuse q,k,v into one [3D][D] weight
        if ((rc = dev_alloc((void **)&Ly.wqkv, (size_t)3 * D * D * 2))) { delete v; return rc; }
        if ((rc = dev_alloc((void **)&Ly.bqkv, (size_t)3 * D * 4))) { delete v; return rc; }
        e->owned.push_back(Ly.wqkv);
        e->owned.push_back(Ly.bqkv);
        const char *nm[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            const f16_t *wsrc;
            const float *bsrc;
            TK(p + "self_attn." + nm[j] + ".weight", (std::vector<int64_t>{D, D}), VLO_DT_F16, wsrc);
            TK(p + "self_attn." + nm[j] + ".bias", (std::vector<int64_t>{D}), VLO_DT_F32, bsrc);
            VIT_TRY(hipMemcpy(Ly.wqkv + (size_t)j * D * D, wsrc, (size_t)D * D * 2, hipMemcpyDeviceToDevice));
            VIT_TRY(hipMemcpy(Ly.bqkv + (size_t)j * D, bsrc, (size_t)D * 4, hipMemcpyDeviceToDevice));
        }
        TK(p + "self_attn.out_proj.weight", (std::vector<int64_t>{D, D}), VLO_DT_F16, Ly.wo);
        TK(p + "self_attn.out_proj.bias", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.bo);
        TK(p + "mlp.fc1.weight", (std::vector<int64_t>{I, D}), VLO_DT_F16, Ly.w1);
        TK(p + "mlp.fc1.bias", (std::vector<int64_t>{I}), VLO_DT_F32, Ly.b1);
        TK(p + "mlp.fc2.weight", (std::vector<int64_t>{D, I}), VLO_DT_F16, Ly.w2);
        TK(p + "mlp.fc2.bias", (std::vector<int64_t>{D}), VLO_DT_F32, Ly.b2);
        PAD(padded16(Ly.w1, I, D, Ip, D));
        PAD(padded32(Ly.b1, I, Ip));
        PAD(padded16(Ly.w2, D, I, D, Ip));
    }
    TK("vision.post_layernorm.weight", (std::vector<int64_t>{D}), VLO_DT_F32, v->post_w);
    TK("vision.post_layernorm.bias", (std::vector<int64_t>{D}), VLO_DT_F32, v->post_b);
    TK("vision.head.attention.in_proj_weight", (std::vector<int64_t>{3 * D, D}), VLO_DT_F16, v->in_proj_w);
    TK("vision.head.attention.in_proj_bias", (std::vector<int64_t>{3 * D}), VLO_DT_F32, v->in_proj_b);
    TK("vision.head.attention.out_proj.weight", (std::vector<int64_t>{D, D}), VLO_DT_F16, v->hout_w);
    TK("vision.head.attention.out_proj.bias", (std::vector<int64_t>{D}), VLO_DT_F32, v->hout_b);
    TK("vision.head.layernorm.weight", (std::vector<int64_t>{D}), VLO_DT_F32, v->hln_w);
    TK("vision.head.layernorm.bias", (std::vector<int64_t>{D}), VLO_DT_F32, v->hln_b);
    TK("vision.head.mlp.fc1.weight", (std::vector<int64_t>{I, D}), VLO_DT_F16, v->hfc1_w);
    TK("vision.head.mlp.fc1.bias", (std::vector<int64_t>{I}), VLO_DT_F32, v->hfc1_b);
    TK("vision.head.mlp.fc2.weight", (std::vector<int64_t>{D, I}), VLO_DT_F16, v->hfc2_w);
    TK("vision.head.mlp.fc2.bias", (std::vector<int64_t>{D}), VLO_DT_F32, v->hfc2_b);
    PAD(padded16(v->hfc1_w, I, D, Ip, D));
    PAD(padded32(v->hfc1_b, I, Ip));
    PAD(padded16(v->hfc2_w, D, I, D, Ip));
    // probe query is input-independent: q = probe @ Wq^T + bq, once (fp16 like the autocast Linear)
    {
        const float *probe32;
        TK("vision.head.probe", (std::vector<int64_t>{1, 1, D}), VLO_DT_F32, probe32);
        f16_t *probe16;
        if ((rc = dev_alloc((void **)&probe16, (size_t)D * 2)) || (rc = dev_alloc((void **)&v->q_probe, (size_t)D * 2))) { delete v; return rc; }
        e->owned.push_back(probe16);
        e->owned.push_back(v->q_probe);
        std::vector<float> hp(D);
        VIT_TRY(hipMemcpy(hp.data(), probe32, (size_t)D * 4, hipMemcpyDeviceToHost));
        std::vector<f16_t> hp16(D);
        for (int i = 0; i < D; ++i) { _Float16 t = (_Float16)hp[i]; memcpy(&hp16[i], &t, 2); }
        VIT_TRY(hipMemcpy(probe16, hp16.data(), (size_t)D * 2, hipMemcpyHostToDevice));
        GemmArgs a{};
        a.X = probe16; a.W = v->in_proj_w; a.bias = v->in_proj_b; a.out16 = v->q_probe;
        a.M = 1; a.N = D; a.K = D; a.ldx = D; a.ldo = D;
        VIT_TRY(gemm_launch<EP_F16>(a, 0));
        VIT_TRY(hipDeviceSynchronize());
    }
#undef TK
#undef PAD
    VIT_TRY(hipFuncSetAttribute((const void *)vit_attn_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnLds));
    VIT_TRY(hipFuncSetAttribute((const void *)vit_attn_split8_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttn8Lds));
    VIT_TRY(hipFuncSetAttribute((const void *)vit_attn_kernel<96, 80>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(80)));
    {
        const int cpr = v->Sp / 8;                                   // 16-byte chunks per V^T row
        v->attn_vrs = (cpr + ((10 - cpr % 16) + 16) % 16) * 16;
        const size_t lds = (size_t)v->Sp * 128 + (size_t)64 * v->attn_vrs;
        if (lds <= 160 * 1024 && v->Sp <= 608 && v->hd == 64) {
            v->attn_head_lds = lds;
            VIT_TRY(hipFuncSetAttribute((const void *)vit_attn_head_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
    }
    e->vit = v;
    return VLO_OK;
}

static int vit_reserve(vlo_engine *e, VitState *v, int B) {
    if (B <= v->Bcap) return VLO_OK;
    hipDeviceSynchronize();
    for (auto &g : v->graphs) hipGraphExecDestroy(g.second);     // captured pointers die with the workspace
    v->graphs.clear();
    for (void *p : v->ws) hipFree(p);
    v->ws.clear();
    v->Bcap = 0;                 // nothing usable until every buffer below exists (a failed allocation must not leave stale pointers live)
    const size_t M = (size_t)B * v->S, D = v->D, I = v->I;
    int rc = 0;
    auto A = [&](void **p, size_t bytes) {
        if (rc) return;
        rc = dev_alloc(p, bytes);
        if (!rc) v->ws.push_back(*p);
    };
    A((void **)&v->h, M * D * 4);
    A((void **)&v->last, M * D * 4);
    v->slab_rows = (int)std::min<size_t>(M, 2048);               // split-K only ever runs on fewer rows than that (vit_resid_ksplit)
    A((void **)&v->slab, (size_t)4 * v->slab_rows * D * 4);
    // GEMM operands carry 256 extra rows: the ping-pong GEMM (vit_gemm.inc) reads whole 256-row tiles, the rows past M are computed
    // and dropped
    const size_t Mp = M + 256;
    A((void **)&v->x16, Mp * D * 2);
    const size_t Dk = (size_t)v->nh * v->hdk, Dv = (size_t)v->nh * v->hdv;     // padded head layouts (== D for head dim 64)
    A((void **)&v->qk16, M * 2 * Dk * 2);                   // q | k, [M][2][heads][hdk]; pad columns never written: zeroed below
    A((void **)&v->vT, (size_t)B * Dv * v->Sp * 2);         // V^T [B][heads][hdv][Sp]; the pad columns [S, Sp) and pad rows are never written: zeroed below
    A((void **)&v->att16, Mp * D * 2);
    A((void **)&v->mid16, Mp * I * 2);
    A((void **)&v->kv16, M * 2 * D * 2);
    A((void **)&v->hatt16, (size_t)B * D * 2);
    A((void **)&v->ho16, (size_t)B * D * 2);
    A((void **)&v->hx16, (size_t)B * D * 2);
    A((void **)&v->hmid16, (size_t)B * I * 2);
    A((void **)&v->cls32, (size_t)B * D * 4);
    A((void **)&v->tmp32, (size_t)B * D * 4);
    A((void **)&v->tokens, (size_t)B * (1 + v->ph * v->pw) * D * 2);
    A((void **)&v->frames_in, (size_t)B * 3 * v->R * v->R);
    A((void **)&v->out_stage, (size_t)B * (1 + v->ph * v->pw) * e->cfg.hidden_size * 2);
    if (rc) return rc;
    if (hipMemset(v->vT, 0, (size_t)B * Dv * v->Sp * 2) != hipSuccess || (Dk != D && hipMemset(v->qk16, 0, M * 2 * Dk * 2) != hipSuccess))
        return vlo_fail(VLO_E_HIP, "vit workspace memset failed");
    v->Bcap = B;
    (void)e;
    return VLO_OK;
}

int vlo_connector_reserve(vlo_engine *e);      // engine.hip

// the ~180-launch encode of B frames: frames (uint8, device) -> out (bf16 [B*T][H], device)
static int vit_run(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st, bool with_connector = true, int b0 = 0,
                   int conn_slot = 0) {
    VitState *v = e->vit;
    const int D = v->D, I = v->I, S = v->S, M = B * S;
    // this call's slice of the workspace: frames [b0, b0 + B) (every buffer is frame-major), so that two calls on disjoint frame
    // ranges can run concurrently as parallel branches of one captured graph
    const size_t r0 = (size_t)b0 * S;
    float *const w_h = v->h + r0 * D, *const w_last = v->last + r0 * D, *const w_tmp32 = v->tmp32 + (size_t)b0 * D;
    f16_t *const w_x16 = v->x16 + r0 * D, *const w_qk16 = v->qk16 + r0 * 2 * v->nh * v->hdk, *const w_vT = v->vT + (size_t)b0 * v->nh * v->hdv * v->Sp, *const w_att16 = v->att16 + r0 * D,
          *const w_mid16 = v->mid16 + r0 * I, *const w_kv16 = v->kv16 + r0 * 2 * D, *const w_hatt16 = v->hatt16 + (size_t)b0 * D,
          *const w_ho16 = v->ho16 + (size_t)b0 * D, *const w_hx16 = v->hx16 + (size_t)b0 * D, *const w_hmid16 = v->hmid16 + (size_t)b0 * I;
    bf16_t *const w_tokens = v->tokens + (size_t)b0 * (1 + v->ph * v->pw) * D;
    const float scale = 1.0f / sqrtf((float)v->hd);
    // few frames: out-proj / fc2 as split-K slices into v->slab, reduced (+ bias + residual) by the LayerNorm that follows either of them
    float *const w_slab = v->slab + r0 * D;
    const size_t slab_stride = (size_t)v->slab_rows * D;
    const bool slab_ok = r0 + (size_t)M <= (size_t)v->slab_rows;
    // one frame (two): every GEMM on the tall tiles, one workgroup per CU (vit_tall.inc).
    // K slices of the two residual GEMMs on the tall tiles: the most (<= 4) that leave a slice >= 3 K tiles — out-proj 4 x 4 tiles, fc2 4 x 16 at SigLIP-L
    // (64 output tiles x 4 = one workgroup per CU; measured: out-proj 11.2 us straight into the residual stream on 64 CUs vs 5.9 us as 4 slices, and the
    // LayerNorm behind it takes the slabs for +1 us)
    constexpr int tall_max_rows = 1152;                          // one or two frames of SigLIP-L (three: the 64 x 64 tiles and the ping-pong kernel from four)
    const int tall_tiles = ((M + 143) / 144) * (D / 64);       // output tiles of a residual GEMM: slices fill the chip once (one frame: 64 x 4, two: 128 x 2)
    auto tall_ks = [&](int K) { for (int ks = 4; ks > 1; ks >>= 1) if (tall_tiles * ks <= vit_num_cus() && K % (GEMM_BK * ks) == 0 && K / (GEMM_BK * ks) >= 3) return ks; return 1; };
    const bool tall = M <= tall_max_rows && slab_ok && D % 64 == 0 && I % 64 == 0 && D >= 3 * GEMM_BK && I >= 3 * GEMM_BK;
    const int ks_out = tall ? tall_ks(D) : (slab_ok ? vit_resid_ksplit(M, D, D) : 1), ks_fc2 = tall ? tall_ks(I) : (slab_ok ? vit_resid_ksplit(M, D, I) : 1);
    int pend_ks = 0;                          // K slices waiting in the slab for the next LayerNorm (0 = none)
    const float *pend_bias = nullptr;
    auto layernorm = [&](const float *g, const float *b, f16_t *o16, float *o32) {
        if (pend_ks > 0)
            hipLaunchKernelGGL((vit_layernorm_kernel<true>), dim3((M + 3) / 4), dim3(256), 0, st, w_h, g, b, o16, o32, M, D, v->eps,
                               (const float *)w_slab, pend_ks, slab_stride, pend_bias);
        else
            hipLaunchKernelGGL((vit_layernorm_kernel<false>), dim3((M + 3) / 4), dim3(256), 0, st, w_h, g, b, o16, o32, M, D, v->eps,
                               (const float *)nullptr, 0, (size_t)0, (const float *)nullptr);
        pend_ks = 0;
    };
    auto resid_gemm = [&](const f16_t *X, const f16_t *W, const float *bias, int K, int ks) -> hipError_t {
        GemmArgs a{};
        a.xpad = 1; a.X = X; a.W = W; a.bias = bias; a.M = M; a.N = D; a.K = K; a.ldx = K;
        if (ks > 1) {
            a.out32 = w_slab; a.ldo = v->slab_rows;
            pend_ks = ks; pend_bias = bias;
            return tall ? gemm_launch_tall<EP_SLAB>(a, ks, st) : gemm_launch_slab(a, ks, st);
        }
        a.out32 = w_h;
        return tall ? gemm_launch_tall<EP_RESID>(a, 1, st) : gemm_launch<EP_RESID>(a, st);
    };
    {   // patch embed + pos  -> residual stream h (fp32)
        GemmArgs a{};
        a.frames = frames_dev; a.W = v->wpe; a.bias = v->bpe; a.out32 = w_h; a.pos = v->pos;
        a.M = M; a.N = D; a.K = v->Kpe; a.Kreal = v->Kpe_real; a.S = S; a.R = v->R; a.P = v->P; a.G = v->G;
        if (tall && v->Kpe >= 5 * GEMM_BK && (size_t)v->Kpe <= (size_t)I) {
            // one frame: the A operand as its own launch into the (still unused) MLP buffer, then the tall GEMM
            const int chunks = M * (v->Kpe / 8);
            hipLaunchKernelGGL(vit_im2col_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, frames_dev, w_mid16, M, S, v->G, v->R, v->P, v->Kpe_real, v->Kpe);
            a.X = w_mid16; a.ldx = v->Kpe;
            VIT_TRY(gemm_launch_tall<EP_PATCH>(a, 1, st));
        } else {
            VIT_TRY(gemm_launch<EP_PATCH>(a, st));
        }
    }
    for (int l = 0; l < v->L; ++l) {
        const VitLayer &Ly = v->layers[l];
        layernorm(Ly.ln1_w, Ly.ln1_b, w_x16, nullptr);
        {
            GemmArgs a{};
            a.xpad = 1; a.X = w_x16; a.W = Ly.wqkv; a.bias = Ly.bqkv; a.out16 = w_qk16; a.outVT = w_vT;
            a.M = M; a.N = 3 * D; a.K = D; a.ldx = D; a.ldo = 2 * v->nh * v->hdk; a.S = S; a.Sp = v->Sp; a.D = D; a.hd = v->hd; a.hdk = v->hdk; a.hdv = v->hdv;
            VIT_TRY(tall ? gemm_launch_tall<EP_QKV>(a, 1, st) : gemm_launch<EP_QKV>(a, st));
        }
        // batched frames: one workgroup per (frame, head) with K and V^T resident in LDS; few frames: 64-query tiles, keys split over 4 waves
        static const int head_min = getenv("VLO_VIT_ATTN_HEAD_MIN") ? atoi(getenv("VLO_VIT_ATTN_HEAD_MIN")) : 96;     // workgroups; 0 = never
        static const int tiles_min = getenv("VLO_VIT_ATTN_TILES_MIN") ? atoi(getenv("VLO_VIT_ATTN_TILES_MIN")) : 192;   // workgroups of the tile-streamed padded-head kernel; 0 = never
        if (head_min > 0 && B * v->nh >= head_min && v->attn_head_lds > 0)
            hipLaunchKernelGGL((vit_attn_head_kernel<0>), dim3((S + 575) / 576, v->nh, B), dim3(768), v->attn_head_lds, st, w_qk16, w_vT, w_att16, S, D, v->nh,
                               scale * 1.4426950408889634f, v->attn_vrs);
        else if (v->hdk == 64 && tall && S <= 768)       // one frame (two): 8 waves per 64 queries, every load up front (32-query blocks measured 12.8 vs 10.5 us)
            hipLaunchKernelGGL((vit_attn_split8_kernel<3, 4>), dim3((S + 63) / 64, v->nh, B), dim3(512), kAttn8Lds, st, w_qk16, w_vT, w_att16, S, D, v->nh, scale);
        else if (v->hdk == 64)
            hipLaunchKernelGGL((vit_attn_kernel<64>), dim3((S + 63) / 64, v->nh, B), dim3(256), kAttnLds, st, w_qk16, w_vT, w_att16, S, D, v->nh, scale);
        else if (B * v->nh * ((S + 255) / 256) >= tiles_min && tiles_min > 0) {
            // padded heads, batched frames: 256-query workgroups over LDS-staged key tiles (vit_attn.inc::vit_attn_tiles_kernel)
            hipLaunchKernelGGL((vit_attn_tiles_kernel<96, 80, 4>), dim3((S + 255) / 256, v->nh, B), dim3(512), 0, st, w_qk16, w_vT, w_att16, S, D, v->nh, scale);
        } else {
            hipLaunchKernelGGL((vit_attn_kernel<96, 80>), dim3((S + 63) / 64, v->nh, B), dim3(256), attn_lds(80), st, w_qk16, w_vT, w_att16, S, D, v->nh, scale);
        }
        VIT_TRY(resid_gemm(w_att16, Ly.wo, Ly.bo, D, ks_out));
        layernorm(Ly.ln2_w, Ly.ln2_b, w_x16, nullptr);
        {
            GemmArgs a{};
            a.xpad = 1; a.X = w_x16; a.W = Ly.w1; a.bias = Ly.b1; a.out16 = w_mid16;
            a.M = M; a.N = I; a.K = D; a.ldx = D; a.ldo = I;
            VIT_TRY(tall ? gemm_launch_tall<EP_F16_GELU>(a, 1, st) : gemm_launch<EP_F16_GELU>(a, st));
        }
        VIT_TRY(resid_gemm(w_mid16, Ly.w2, Ly.b2, I, ks_fc2));
    }
    // post layernorm: fp32 (pooling input) + fp16 (head K/V operand)
    layernorm(v->post_w, v->post_b, w_x16, w_last);
    {   // MAP head: K,V = last @ Wkv^T + bkv
        GemmArgs a{};
        a.xpad = 1; a.X = w_x16; a.W = v->in_proj_w + (size_t)D * D; a.bias = v->in_proj_b + D; a.out16 = w_kv16;
        a.M = M; a.N = 2 * D; a.K = D; a.ldx = D; a.ldo = 2 * D;
        VIT_TRY(gemm_launch<EP_F16>(a, st));
    }
    hipLaunchKernelGGL(map_attn_kernel, dim3(v->nh, B), dim3(256), (size_t)std::max(S, 1024) * 4, st, w_kv16, v->q_probe, w_hatt16, S, D, v->hd, scale);
    {   // out_proj -> attention output a (fp16), kept as the residual
        GemmArgs a{};
        a.X = w_hatt16; a.W = v->hout_w; a.bias = v->hout_b; a.out16 = w_ho16;
        a.M = B; a.N = D; a.K = D; a.ldx = D; a.ldo = D;
        VIT_TRY(rowvec_launch<EP_F16>(a, st));
    }
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((B * D + 255) / 256), dim3(256), 0, st, w_ho16, w_tmp32, B * D);
    hipLaunchKernelGGL((vit_layernorm_kernel<false>), dim3((B + 3) / 4), dim3(256), 0, st, w_tmp32, v->hln_w, v->hln_b, w_hx16, (float *)nullptr, B, D, v->eps,
                       (const float *)nullptr, 0, (size_t)0, (const float *)nullptr);
    {
        GemmArgs a{};
        a.X = w_hx16; a.W = v->hfc1_w; a.bias = v->hfc1_b; a.out16 = w_hmid16;
        a.M = B; a.N = I; a.K = D; a.ldx = D; a.ldo = I;
        VIT_TRY(rowvec_launch<EP_F16_GELU>(a, st));
    }
    {   // cls = residual + mlp(...)
        GemmArgs a{};
        a.X = w_hmid16; a.W = v->hfc2_w; a.bias = v->hfc2_b; a.out32 = w_tmp32;
        a.M = B; a.N = D; a.K = I; a.ldx = I;
        VIT_TRY(rowvec_launch<EP_RESID>(a, st));
    }
    hipLaunchKernelGGL(pool_concat_kernel, dim3(1 + v->ph * v->pw, B, (D + 255) / 256), dim3(256), 0, st, w_last, w_tmp32, w_tokens, v->G, D, v->ph, v->pw);
    VIT_TRY(hipGetLastError());
    if (!with_connector) {     // offline feature extraction: the CLS + pooled tokens themselves
        VIT_TRY(hipMemcpyAsync(out_dev, w_tokens, (size_t)B * (1 + v->ph * v->pw) * D * 2, hipMemcpyDeviceToDevice, st));
        return VLO_OK;
    }
    // connector (bf16 skinny GEMMs, gemv.hip)
    return connector_run(e, conn_slot, w_tokens, B * (1 + v->ph * v->pw), out_dev, st);
}

// B frames as ONE branch, or — from VLO_VIT_SPLIT_MIN frames up (default 4) — as TWO parallel half-batch branches: first half on the
// caller's stream, second half on an internal one, each on its own slice of the workspace and its own connector scratch, forked and
// joined with events (inside a stream capture they become parallel branches of the graph).  One half's tails, ramps and
// under-filled kernels overlap the other's.  Measured per call, one branch -> two: 4 frames 3.91 -> 3.57 ms, 6: 5.12 -> 4.45, 8: 5.90 ->
// 5.45, 12: 6.96 -> 6.56, 16: 9.41 -> 8.0, 28: 13.1 -> 12.4 (profiles/r3_vit_small_and_mid_batches.txt).  Every row is computed from its own frame only and all GEMM tiles
// accumulate K in the same order, so the split changes results only where a half-batch selects the other attention kernel (1 fp16 ulp).
static int vit_run_branches(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st, bool with_connector) {
    VitState *v = e->vit;
    static const int split_min = getenv("VLO_VIT_SPLIT_MIN") ? atoi(getenv("VLO_VIT_SPLIT_MIN")) : 4;
    if (split_min <= 0 || B < split_min || st == nullptr) return vit_run(e, frames_dev, B, out_dev, st, with_connector);
    if (!v->st2) {
        VIT_TRY(hipStreamCreateWithFlags(&v->st2, hipStreamNonBlocking));
        VIT_TRY(hipEventCreate(&v->ev_fork));
        VIT_TRY(hipEventCreate(&v->ev_join));
    }
    const int B0 = B / 2;
    const size_t frame_bytes = (size_t)3 * v->R * v->R;
    const size_t out_elems = (size_t)(1 + v->ph * v->pw) * (with_connector ? e->cfg.hidden_size : v->D);
    VIT_TRY(hipEventRecord(v->ev_fork, st));
    VIT_TRY(hipStreamWaitEvent(v->st2, v->ev_fork, 0));
    int rc = vit_run(e, frames_dev, B0, out_dev, st, with_connector, 0, 0);
    if (!rc) rc = vit_run(e, frames_dev + B0 * frame_bytes, B - B0, (bf16_t *)out_dev + B0 * out_elems, v->st2, with_connector, B0, 1);
    VIT_TRY(hipEventRecord(v->ev_join, v->st2));          // joined even after a failed launch: the second stream must leave the capture
    VIT_TRY(hipStreamWaitEvent(st, v->ev_join, 0));
    return rc;
}

// Entry point.  The launch sequence is static for a given B, so it is captured once into a hipGraph and
// replayed (one host call instead of ~180; the host thread also feeds the Llama stream).  Frames are
// staged into a fixed input buffer and the embeddings leave through a fixed output buffer so the
// captured kernel arguments stay valid.  (The null stream cannot be captured: eager launches.)
int vit_visual_embed(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st) {
    constexpr bool use_graph = true;
    VitState *v = e->vit;
    int rc;
    if ((rc = vit_reserve(e, v, B))) return rc;
    if ((rc = vlo_connector_reserve(e))) return rc;
    if (!use_graph || st == nullptr) return vit_run_branches(e, frames_dev, B, out_dev, st, true);    // the null stream cannot be captured (nor forked: one branch)
    const size_t in_bytes = (size_t)B * 3 * v->R * v->R;
    const size_t out_bytes = (size_t)B * (1 + v->ph * v->pw) * e->cfg.hidden_size * 2;
    VIT_TRY(hipMemcpyAsync(v->frames_in, frames_dev, in_bytes, hipMemcpyDeviceToDevice, st));
    auto it = v->graphs.find(B);
    if (it == v->graphs.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        VIT_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        rc = vit_run_branches(e, v->frames_in, B, v->out_stage, st, true);
        hipError_t ce = hipStreamEndCapture(st, &graph);
        if (rc) {
            if (graph) hipGraphDestroy(graph);
            return rc;
        }
        VIT_TRY(ce);
        VIT_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipGraphDestroy(graph);
        it = v->graphs.emplace(B, exec).first;
    }
    VIT_TRY(hipGraphLaunch(it->second, st));
    VIT_TRY(hipMemcpyAsync(out_dev, v->out_stage, out_bytes, hipMemcpyDeviceToDevice, st));
    return VLO_OK;
}

int vit_vision_tokens(vlo_engine *e, const uint8_t *frames_dev, int B, void *out_dev, hipStream_t st) {
    VitState *v = e->vit;
    int rc;
    if ((rc = vit_reserve(e, v, B))) return rc;
    return vit_run_branches(e, frames_dev, B, out_dev, st, false);
}

void vit_destroy(vlo_engine *e) {
    if (!e->vit) return;
    for (auto &g : e->vit->graphs) hipGraphExecDestroy(g.second);
    for (void *p : e->vit->ws) hipFree(p);
    if (e->vit->ev_fork) hipEventDestroy(e->vit->ev_fork);
    if (e->vit->ev_join) hipEventDestroy(e->vit->ev_join);
    if (e->vit->st2) hipStreamDestroy(e->vit->st2);
    delete e->vit;
    e->vit = nullptr;
}
