// One flow of the small-M chain as one launch (round 5): the kernel in flow_persist.h, the host side here; a translation unit of its
// own (five instantiations of a large kernel: flow_kernels.hip already takes four minutes to compile).
#include "common.h"
#include "gemm_ring.h"
#include "fwn_internal.h"
#include "tail_zero_prob.h"
#include "flow_persist.h"

// ---- host side ----
int g_fwn_opt_persist_spin_us = 0;         // fwn_set_option("persist_spin_us", ..): 0 = the default bound (2 s)
int fwn_flow_persist_sync_words(int M, int L) {
    const int RT = (M + 63) / 64;
    int w = (FWN_PS_HDR + (2 * L + 3) * RT + 3) & ~3;     // multiple of 16 bytes
#ifdef FWN_PS_STAMP
    w += RT * (4 + L * 8 + (L - 1) * 4 + 8 + 4) * 16;     // [ticket][8] 64-bit stamps
#endif
    return w;
}
// whether this flow at this shape runs as one launch: hoisted conditioning, few rows, the N-split tail's row range (so that
// the log-det partial slots and the S / U buffers are the launch-per-stage path's), n_layer <= 2
int fwn_flow_persist_ok(int M, int Ch, int L, int npt, bool has_w2, bool xa_aligned) {
    if (M > FWN_TUNE(FWN_PERSIST_MAX_ROWS, FWN_PERSIST_MAX_ROWS) || L > FWN_PS_MAXL || L < 1) return 0;
    if (fwn_tail_chain_xb_out(M, npt, 0)) return 0;   // only where the tail (without a fragment stream) is the three N-split ring GEMMs
    if (Ch > 128 || npt < 1 || npt > 4) return 0;
    if (Ch >= 16 && !(has_w2 && xa_aligned)) return 0;
    return 1;
}
int fwn_flow_persist_front_inside(int Ch) { return Ch >= 16; }

static void fwn_launch_flow_persist(const PersistArgs& a, hipStream_t st);
void fwn_launch_flow_persist_desc(const fwn_flow_desc* d, float* xa, float* xb, void* hA, void* hB, void* o, const float* P,
                                  float* partial, unsigned* sync, int M, int Ti, int inverse, int has_front, hipStream_t st) {
    PersistArgs a;
    memset(&a, 0, sizeof(a));
    a.xa = xa; a.xb = xb; a.an = d->an;
    a.W2 = (const bf16*)d->Wfront2; a.bfront = d->bfront;
    for (int l = 0; l < d->L && l < FWN_PS_MAXL; ++l) {
        a.Wd[l] = (const bf16*)d->Wd[l]; a.bgate[l] = d->bgate[l];
        a.Wres[l] = (const bf16*)d->Wres[l]; a.bres[l] = d->bres[l];
    }
    a.Wskip = (const bf16*)d->Wskip; a.bskip = d->bskip;
    a.Wfinal = (const bf16*)d->Wfinal; a.bfinal = d->bfinal;
    a.Wzero = (const bf16*)d->Wzero; a.bzero = d->bzero; a.ezero = d->ezero;
    a.hA = (bf16*)hA; a.hB = (bf16*)hB; a.o = (bf16*)o; a.P = P; a.partial = partial; a.sync = sync;
    a.M = M; a.Ti = Ti; a.Ch = d->Ch; a.npt = d->npt; a.L = d->L; a.inverse = inverse; a.has_front = has_front;
    a.spin_ticks = g_fwn_opt_persist_spin_us > 0 ? (unsigned)(g_fwn_opt_persist_spin_us > 40000000 ? 4000000000u : (unsigned)g_fwn_opt_persist_spin_us * 100u) : 200000000u;
    fwn_launch_flow_persist(a, st);
}

static void fwn_launch_flow_persist(const PersistArgs& a, hipStream_t st) {
    const int ncu = fwn_device_cus();
    // one workgroup per ticket while the chip has room: a workgroup that holds a ticket of a LATER stage requests that
    // ticket's weights at once and waits for its producers with them in LDS - the run-ahead that hides the weight stream.
    // From 8 row tiles on (288+ tickets) only one level's worth of workgroups (8 per row tile): 256 workgroups that mostly wait
    // would hold every CU's LDS against the other lanes' kernels (8-clip overlapped step, block 7 = 504 rows: 44.6 M samples/s
    // with the full grid, 45.7 with 64 workgroups, 45.3 with a launch per stage; one stream: no difference)
    const int RT = (a.M + 63) / 64;
    const int total = RT * ((a.has_front ? 4 : 0) + a.L * 8 + (a.L - 1) * 4 + 8 + a.npt);
    const int cap = FWN_TUNE(FWN_PERSIST_GRID, 0) > 0 ? FWN_TUNE(FWN_PERSIST_GRID, 0) : RT >= 8 && 8 * RT < ncu ? 8 * RT : ncu;
    const int grid = total < cap ? total : cap;
    hipLaunchKernelGGL(flow_persist_kernel, dim3(grid), dim3(512), 0, st, a);
}
