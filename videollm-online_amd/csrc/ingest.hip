// ingest.hip — frame preparation in front of the vision tower (the part that needs no video decoder).
//
// The reference prepares a video with an external ffmpeg binary before `LiveInfer.load_video` reads it back
// (data/utils.py:51-66 `ffmpeg_once`, called from demo/cli.py:13-22):
//     -sws_flags bicubic -vf "scale='if(gt(iw,ih),R,-2)':'if(gt(iw,ih),-2,R)',pad=R:R:(ow-iw)/2:(oh-ih)/2:color='#000000'"
// then `read_video(..., output_format='TCHW')` (demo/inference.py:112).  Here decoded RGB frames of any size already in
// HBM (decoder output [T,H,W,3] or planar [T,3,H,W]) become the uint8 [T,3,R,R] tensor the encode consumes:
//   geometry   longer side -> R, the other side av_rescale'd to a multiple of 2, centred on a black R x R canvas with the
//              offsets rounded down to the 2x2 chroma grid (libavfilter scale_eval.c / vf_pad.c on a yuv420p frame);
//   resampling separable antialiased Keys cubic (support stretched by the down-scale factor, pixel-centre aligned);
//              a = -0.6 is libswscale's SWS_BICUBIC default (B = 0, C = 0.6), a = -0.5 PIL / torch antialias.
// HBM-bound byte work (a 1080p frame is 6.2 MB in, 0.44 MB out): pass 1 reads every input row ONCE (coalesced 16-byte
// loads into LDS) and writes the horizontally resampled row as fp32 planes; pass 2 resamples vertically with lanes along x
// (coalesced), rounds, pads and writes NCHW.  Tap tables are built on the host in double precision once per geometry.
#include <math.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/vlo.h"
#include "common.cuh"
#include "engine.h"

struct PadRGB { uint8_t r, g, b; };
struct Taps {                     // device tables of one axis: first input index, tap count, normalised weights [out][kmax]
    int *first = nullptr, *count = nullptr;
    float *w = nullptr;
    int out = 0, kmax = 0;
};

static long long av_rescale_near(long long a, long long b, long long c) { return (a * b + c / 2) / c; }

int vlo_frame_ingest_geometry(int W, int H, int R, int *ow, int *oh, int *x0, int *y0) {
    if (W <= 0 || H <= 0 || R <= 1 || (R & 1)) return vlo_fail(VLO_E_INVALID, "bad ingest geometry");
    int w, h;
    if (W > H) { w = R; h = (int)av_rescale_near(w, H, (long long)W * 2) * 2; }     // scale=R:-2
    else       { h = R; w = (int)av_rescale_near(h, W, (long long)H * 2) * 2; }     // scale=-2:R
    if (w < 2) w = 2;
    if (h < 2) h = 2;
    if (ow) *ow = w;
    if (oh) *oh = h;
    if (x0) *x0 = ((R - w) / 2) & ~1;
    if (y0) *y0 = ((R - h) / 2) & ~1;
    return VLO_OK;
}

static double keys_cubic(double x, double a) {
    x = fabs(x);
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
    if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
    return 0.0;
}

// host tables of one axis (the tap selection / normalisation of PIL's and torch's antialiased resize)
static void build_taps(int in_size, int out_size, double a, std::vector<int> &first, std::vector<int> &count, std::vector<float> &w, int &kmax) {
    const double scale = (double)in_size / out_size;
    const double support = scale >= 1.0 ? 2.0 * scale : 2.0, inv = scale >= 1.0 ? 1.0 / scale : 1.0;
    kmax = (int)ceil(support) * 2 + 2;
    first.assign(out_size, 0);
    count.assign(out_size, 0);
    w.assign((size_t)out_size * kmax, 0.f);
    std::vector<double> t(kmax);
    for (int i = 0; i < out_size; ++i) {
        const double center = scale * (i + 0.5);
        int lo = (int)(center - support + 0.5), hi = (int)(center + support + 0.5);
        if (lo < 0) lo = 0;
        if (hi > in_size) hi = in_size;
        int n = hi - lo;
        if (n > kmax) n = kmax;
        double s = 0.0;
        for (int j = 0; j < n; ++j) { t[j] = keys_cubic((lo + j - center + 0.5) * inv, a); s += t[j]; }
        for (int j = 0; j < n; ++j) w[(size_t)i * kmax + j] = (float)(s != 0.0 ? t[j] / s : t[j]);
        first[i] = lo;
        count[i] = n;
    }
}

struct IngestPlan { Taps h, v; int ow, oh, x0, y0; };
struct IngestState {
    std::mutex mu;
    std::map<std::tuple<int, int, int, int>, IngestPlan> plans;      // (H, W, R, a * 1e4)
    float *tmp = nullptr;                 // ONE fp32 scratch per engine, shared by every caller: calls are ordered through `done`
    size_t tmp_bytes = 0;
    hipEvent_t done = nullptr;            // recorded after the last launch that uses tmp; the next call's stream waits for it
    std::vector<void *> owned;
};

void ingest_create(vlo_engine *e) { e->ingest = new IngestState(); }     // with the engine (vlo_engine_create): push() may come from a feeder thread

void ingest_destroy(vlo_engine *e) {
    IngestState *s = (IngestState *)e->ingest;
    if (!s) return;
    for (void *p : s->owned) hipFree(p);
    if (s->tmp) hipFree(s->tmp);
    if (s->done) hipEventDestroy(s->done);
    delete s;
    e->ingest = nullptr;
}

static int upload_taps(IngestState *s, int in_size, int out_size, double a, Taps *t) {
    std::vector<int> first, count;
    std::vector<float> w;
    int kmax = 0;
    build_taps(in_size, out_size, a, first, count, w, kmax);
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    if (hipMalloc(&p0, first.size() * 4) != hipSuccess || hipMalloc(&p1, count.size() * 4) != hipSuccess || hipMalloc(&p2, w.size() * 4) != hipSuccess)
        return vlo_fail(VLO_E_NOMEM, "ingest tap tables: hipMalloc failed");
    s->owned.push_back(p0); s->owned.push_back(p1); s->owned.push_back(p2);
    if (hipMemcpy(p0, first.data(), first.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p1, count.data(), count.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p2, w.data(), w.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return vlo_fail(VLO_E_HIP, "ingest tap tables: upload failed");
    t->first = (int *)p0; t->count = (int *)p1; t->w = (float *)p2; t->out = out_size; t->kmax = kmax;
    return VLO_OK;
}

// ---- pass 1: horizontal.  grid (H, frames); the input row (3 channels) is staged in LDS as bytes.
// layout 0: src [T][H][W][3] -> LDS holds the row as it lies (x*3 + c);  layout 1: src [T][3][H][W] -> LDS [c][W]
__global__ __launch_bounds__(256) void ingest_h_kernel(const uint8_t *__restrict__ src, int H, int W, int layout, Taps th,
                                                       float *__restrict__ tmp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t row[];
    const int y = blockIdx.x, t = blockIdx.y;
    const size_t frame = (size_t)t * 3 * H * W;
    // the row lies in LDS at byte offset `shift` (= its global address mod 16), so that every 16-byte global chunk that is
    // entirely inside the row is one aligned 16-byte LDS store; the ragged head and tail go byte by byte
    int shift = 0;
    if (layout == 0) {
        const uint8_t *g = src + frame + (size_t)y * W * 3;
        const int nb = W * 3;
        shift = (int)((size_t)g & 15);
        const uint8_t *gb = g - shift;
        const int k0 = shift ? 1 : 0, k1 = (shift + nb) / 16;              // whole chunks [k0, k1)
        for (int k = k0 + threadIdx.x; k < k1; k += blockDim.x)
            *reinterpret_cast<uint4 *>(row + k * 16) = *reinterpret_cast<const uint4 *>(gb + (size_t)k * 16);
        const int head_end = min(k0 * 16, shift + nb), tail_beg = max(k1 * 16, head_end);
        for (int i = shift + threadIdx.x; i < head_end; i += blockDim.x) row[i] = gb[i];
        for (int i = tail_beg + threadIdx.x; i < shift + nb; i += blockDim.x) row[i] = gb[i];
    } else {
        for (int c = 0; c < 3; ++c) {
            const uint8_t *g = src + frame + ((size_t)c * H + y) * W;
            for (int i = threadIdx.x; i < W; i += blockDim.x) row[c * W + i] = g[i];
        }
    }
    __syncthreads();
    const int ow = th.out;
    for (int ox = threadIdx.x; ox < ow; ox += blockDim.x) {
        const int lo = th.first[ox], n = th.count[ox];
        const float *w = th.w + (size_t)ox * th.kmax;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (layout == 0) {
            const uint8_t *p = row + shift + lo * 3;
            for (int j = 0; j < n; ++j) {
                const float wj = w[j];
                a0 += wj * (float)p[j * 3]; a1 += wj * (float)p[j * 3 + 1]; a2 += wj * (float)p[j * 3 + 2];
            }
        } else {
            const uint8_t *p = row + lo;
            for (int j = 0; j < n; ++j) {
                const float wj = w[j];
                a0 += wj * (float)p[j]; a1 += wj * (float)p[W + j]; a2 += wj * (float)p[2 * W + j];
            }
        }
        float *o = tmp + ((size_t)t * 3 * H + y) * ow + ox;              // planes [t][c][H][ow]
        o[0] = a0;
        o[(size_t)H * ow] = a1;
        o[(size_t)2 * H * ow] = a2;
    }
}

// ---- pass 2: vertical + round + pad + NCHW.  grid (R, 3, frames); lanes along x
__global__ __launch_bounds__(128) void ingest_v_kernel(const float *__restrict__ tmp, int H, int R, int ow, int oh, int x0, int y0,
                                                       Taps tv, PadRGB pad, uint8_t *__restrict__ out) {
    const int oy = blockIdx.x, c = blockIdx.y, t = blockIdx.z;
    uint8_t *o = out + (((size_t)t * 3 + c) * R + oy) * R;
    const uint8_t padv = c == 0 ? pad.r : (c == 1 ? pad.g : pad.b);
    const int yy = oy - y0;
    const bool in_y = yy >= 0 && yy < oh;
    const int lo = in_y ? tv.first[yy] : 0, n = in_y ? tv.count[yy] : 0;
    const float *w = tv.w + (size_t)(in_y ? yy : 0) * tv.kmax;
    const float *plane = tmp + ((size_t)t * 3 + c) * H * ow;
    for (int x = threadIdx.x; x < R; x += blockDim.x) {
        const int xx = x - x0;
        uint8_t v = padv;
        if (in_y && xx >= 0 && xx < ow) {
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += w[j] * plane[(size_t)(lo + j) * ow + xx];
            const float r = floorf(acc + 0.5f);
            v = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
        }
        o[x] = v;
    }
}

int vlo_frame_ingest(vlo_engine *e, const uint8_t *src_dev, int T, int H, int W, int layout, int resolution, float cubic_a,
                     uint8_t *out_dev, void *stream) {
    if (!e || !src_dev || !out_dev || T <= 0 || H <= 0 || W <= 0 || (layout != 0 && layout != 1))
        return vlo_fail(VLO_E_INVALID, "bad frame_ingest arguments");
    const int R = resolution > 0 ? resolution : (e->cfg.has_vit ? e->cfg.vit_image_size : 0);
    if (R <= 1 || (R & 1)) return vlo_fail(VLO_E_INVALID, "frame_ingest: resolution must be a positive even number");
    if ((size_t)W * 3 > 150 * 1024) return vlo_fail(VLO_E_UNSUPPORTED, "frame_ingest: rows wider than 51200 pixels");
    if (hipSetDevice(e->device) != hipSuccess) return vlo_fail(VLO_E_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    IngestState *s = (IngestState *)e->ingest;
    if (!s) return vlo_fail(VLO_E_STATE, "frame_ingest: engine has no ingest state");
    std::lock_guard<std::mutex> g(s->mu);
    // the scratch is shared: whatever stream the previous call ran on, its kernels finish before this call's kernels start (the host
    // mutex only serialises the enqueueing — two FrameRings, or a ring's feeder thread next to load_video, use different streams)
    if (!s->done) {
        if (hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess) return vlo_fail(VLO_E_HIP, "frame_ingest: hipEventCreate failed");
    } else if (hipStreamWaitEvent(st, s->done, 0) != hipSuccess) {
        return vlo_fail(VLO_E_HIP, "frame_ingest: hipStreamWaitEvent failed");
    }
    const auto key = std::make_tuple(H, W, R, (int)lrintf(cubic_a * 10000.f));
    auto it = s->plans.find(key);
    if (it == s->plans.end()) {
        IngestPlan p{};
        int rc = vlo_frame_ingest_geometry(W, H, R, &p.ow, &p.oh, &p.x0, &p.y0);
        if (rc) return rc;
        if ((rc = upload_taps(s, W, p.ow, cubic_a, &p.h))) return rc;
        if ((rc = upload_taps(s, H, p.oh, cubic_a, &p.v))) return rc;
        it = s->plans.emplace(key, p).first;
    }
    const IngestPlan &p = it->second;
    // fp32 planes of the horizontally resampled frames, a bounded number of frames at a time
    const size_t per_frame = (size_t)3 * H * p.ow * sizeof(float);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, ((size_t)256 << 20) / per_frame));
    if (s->tmp_bytes < per_frame * chunk) {
        if (s->tmp) {
            if (hipDeviceSynchronize() != hipSuccess) return vlo_fail(VLO_E_HIP, "frame_ingest: sync failed");   // queued launches still use it
            hipFree(s->tmp);
            s->tmp = nullptr;
            s->tmp_bytes = 0;
        }
        if (hipMalloc((void **)&s->tmp, per_frame * chunk) != hipSuccess) return vlo_fail(VLO_E_NOMEM, "frame_ingest: scratch hipMalloc failed");
        s->tmp_bytes = per_frame * chunk;
    }
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)ingest_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) (void)hipGetLastError();
        attr_done = true;
    }
    const PadRGB pad{0, 0, 0};                            // color='#000000' (data/utils.py:51)
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int n = std::min(chunk, T - t0);
        const uint8_t *src = src_dev + (size_t)t0 * 3 * H * W;
        hipLaunchKernelGGL(ingest_h_kernel, dim3(H, n), dim3(256), (size_t)W * 3 + 32, st, src, H, W, layout, p.h, s->tmp);
        hipLaunchKernelGGL(ingest_v_kernel, dim3(R, 3, n), dim3(128), 0, st, s->tmp, H, R, p.ow, p.oh, p.x0, p.y0, p.v, pad,
                           out_dev + (size_t)t0 * 3 * R * R);
        const hipError_t he = hipGetLastError();
        if (he != hipSuccess) return vlo_fail(VLO_E_HIP, std::string("frame_ingest launch: ") + hipGetErrorString(he));
    }
    if (hipEventRecord(s->done, st) != hipSuccess) return vlo_fail(VLO_E_HIP, "frame_ingest: hipEventRecord failed");
    return VLO_OK;
}
