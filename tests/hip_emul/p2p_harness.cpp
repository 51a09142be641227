// Executes the SOURCE of the peer-to-peer exchange kernels (videollm-online_amd/csrc/tp_p2p.cuh) on the CPU through the
// HIP-on-threads shim, in the configuration the single-process group uses (every rank publishes, then every rank
// collects).  Built and driven by tests/test_tp_p2p_kernels_emul_cpu.py.  Test infrastructure only.
#include "tp_p2p.cuh"

extern "C" {

// partials [T][ks][16][H] f32; h [T][16][H] bf16 in/out; w [H] bf16; x [T][16][H] bf16 out; mbox [T][granules] u64
// skip_rank >= 0: that rank never publishes (its peers must time out, not hang).  Returns the OR of the error words.
int emul_p2p_exchange(int T, int m, int H, int ks, const float *partials, unsigned short *h, const unsigned short *w,
                      unsigned short *x, unsigned long long *mbox, unsigned long long granules, unsigned long long slot_off,
                      unsigned epoch, float eps, long long timeout_ticks, int skip_rank, unsigned *err_words /*[T] in/out*/) {
    unsigned err_host = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int r = 0; r < T; ++r) {
            if (pass == 0 && r == skip_rank) continue;
            XchgArgs a{};
            a.partial = partials + (size_t)r * ks * 16 * H;
            a.ks = ks; a.ld = H; a.T = T; a.me = r;
            for (int p = 0; p < T; ++p) a.peers.mbox[p] = mbox + (size_t)p * granules;
            a.slot_off = slot_off; a.epoch = epoch; a.mode = pass == 0 ? 1 : 2;
            a.h = h + (size_t)r * 16 * H; a.w = w; a.x = x + (size_t)r * 16 * H; a.H = H; a.ldx = H; a.eps = eps;
            a.err_dev = err_words + r; a.err_host = &err_host; a.timeout_ticks = timeout_ticks;
            emul_launch(dim3(m), dim3(XCHG_THREADS), [a]() { tp_xchg_norm_kernel(a); });
        }
    return (int)err_host;
}

// local [T][nr][Vl] bf16 -> out [T][nr][V] bf16 (every rank's copy of the gathered rows)
int emul_p2p_gather(int T, int nr, int Vl, const unsigned short *local, unsigned short *out, unsigned long long *mbox,
                    unsigned long long granules, unsigned long long slot_off, unsigned epoch, long long timeout_ticks, int blocks,
                    unsigned *err_words) {
    unsigned err_host = 0;
    const int V = T * Vl;
    for (int pass = 0; pass < 2; ++pass)
        for (int r = 0; r < T; ++r) {
            GatherArgs a{};
            a.local = local + (size_t)r * nr * Vl; a.out = out + (size_t)r * nr * V;
            for (int p = 0; p < T; ++p) a.peers.mbox[p] = mbox + (size_t)p * granules;
            a.T = T; a.me = r; a.Vl = Vl; a.V = V; a.slot_off = slot_off; a.epoch = epoch; a.mode = pass == 0 ? 1 : 2;
            a.err_dev = err_words + r; a.err_host = &err_host; a.timeout_ticks = timeout_ticks;
            emul_launch(dim3(blocks, nr), dim3(64), [a]() { tp_gather_kernel(a); });
        }
    return (int)err_host;
}

int emul_xchg_threads(void) { return XCHG_THREADS; }
}
