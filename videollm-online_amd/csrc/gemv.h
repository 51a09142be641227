// gemv.h — host-visible interface of gemv.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "llm_ops.h"

// epilogues
enum {
    EPI_PARTIAL_F32 = 0,     // fp32 split-K partials [ksplit][16][ldo]            (unit tests)
    EPI_BF16 = 1,            // bf16(acc + bias)                                     (lm_head, connector.2)
    EPI_BF16_GELU_ERF = 2,   // HF python-GELU on bf16(acc + bias)                   (connector.0)
    EPI_SWIGLU = 3,          // tile = 8 gate rows + 8 up rows of the same columns -> bf16(silu(g) * u)   (gate_up)
    EPI_RESID = 4,           // h[m][col] = bf16(h + bf16(acc)); per-row sum of squares partials (o_proj, down_proj)
    EPI_ROPE = 5,            // q/k/v split, RoPE, q buffer + paged K / V^T append   (qkv)
    EPI_PARTIAL_MBOX = 6     // tensor parallel, ksplit == 1: a rank's o / down partial sums published as {epoch, fp32} granules straight into EVERY
                             // rank's p2p mailbox (tp.hip "p2p exchange"): no partial matrix in HBM, no publish pass in the exchange kernel
};
// activation operand source
enum {
    XSRC_PLAIN = 0,          // x is a bf16 [16][ldx] tile
    XSRC_NORM = 1            // x = LlamaRMSNorm(h) computed on the fly: bf16(w[k] * bf16(h[m][k] * rs[m]))
};

struct GemvArgs {
    const void *Wp;               // packed weights (see gemv.hip)
    int wq;                       // 0: bf16 image;  1: fp8 e4m3 image + per-output-channel scales
    const float *wscale;          // wq: fp32 [NT * 16] in packed row order (row r of tile t at t * 16 + r)
    const unsigned short *x;      // XSRC_PLAIN: bf16 [16][ldx];  XSRC_NORM: residual stream h, bf16 [16][ldx]
    float *out_f32;               // EPI_PARTIAL_F32
    unsigned short *out_bf16;     // EPI_BF16 / GELU / SWIGLU: [16][ldo];  EPI_ROPE: q buffer [16][nh*hd]
    const unsigned short *bias;   // bf16 [N] or null
    int K, ldx, ldo;
    int NT;                       // column tiles (N padded to 16)
    int N_valid;                  // real N (multiple of 4)
    int n_rows;                   // valid token rows (<= 16)
    int CT;                       // column tiles per group (0 = default)
    int KC;                       // K chunks walked sequentially by each wave (K = ksplit*NW*KC*KF*32)
    // XSRC_NORM
    const unsigned short *norm_w; // bf16 [K]
    const float *sq_in;           // [sq_in_parts][16] per-row sum-of-squares partials of h
    int sq_in_parts;
    float eps;
    // EPI_RESID
    unsigned short *h;            // residual stream, bf16 [16][ldo], updated in place
    float *sq_out;                // [gridDim.x][16] partials of the updated rows
    // The two epilogues below never meet in one launch and share their bytes: the kernel-argument block keeps the size it had before
    // EPI_PARTIAL_MBOX existed (88 more bytes of it cost the gate/up GEMV 0.5 us on the same box: 42.5 vs 42.0 us, tools/bench_gemv.py).
    union {
        struct {                      // EPI_ROPE
            const unsigned short *cos_tab, *sin_tab;   // bf16 [pos][hd/2]
            KvGeom kv;
            int layer, num_heads;
            long long pos0;
        };
        struct {                      // EPI_PARTIAL_MBOX
            unsigned long long *mbox[8];  // every rank's mailbox, by global rank (tp_p2p.cuh::P2PPeers)
            unsigned long long mbox_off;  // granule offset of [slot][this rank][row 0][column 0] in the reduce region; rows are ldo granules apart
            unsigned mbox_epoch;
            int mbox_T;
        };
    };
};
static_assert(sizeof(GemvArgs) <= 232, "GemvArgs grew: see the union above");

struct GemvPlan { int NW, KF, KC, ksplit; };

int gemv_plan(int K, bool allow_ksplit, GemvPlan *p);
// grid.x the launch will use (= number of sq_out partial rows an EPI_RESID launch writes)
int gemv_grid_x(const GemvArgs &a, const GemvPlan &p, int epi);
hipError_t gemv_launch(GemvArgs a, const GemvPlan &p, int xsrc, int epi, hipStream_t st);
// what gemv_launch does before it launches: validates, fills the launch-dependent fields of `a` (CT, KC) and returns the grid
// and the dynamic LDS size — for callers that run the kernel body themselves (layer.hip)
hipError_t gemv_prepare(GemvArgs *a, const GemvPlan &p, int epi, int *grid_x, int *grid_y, size_t *lds_bytes);
// source tiles [0,NT) of row-major W[N_valid][K] (row stride ldw elements) -> packed tiles t*tile_stride + tile_offset of Wp
// half = -1: 16-row tiles; half = 0/1: 8-row interleave of two matrices into one tile (gate / up)
// fp8 e4m3 source [N_valid][K] (row stride ldw BYTES) + per-row scales -> fp8 image and scales in packed row order
hipError_t pack_weight_fp8_launch(const void *W, const float *scale, void *Wp, float *scale_p, int N_valid, int K, int ldw, int NT,
                                  int tile_stride, int tile_offset, int half, hipStream_t st);
hipError_t pack_weight_launch(const void *W, void *Wp, int N_valid, int K, int ldw, int NT, int tile_stride, int tile_offset,
                              int half, hipStream_t st);
