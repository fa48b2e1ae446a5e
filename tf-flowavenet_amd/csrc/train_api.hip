// fwn_train_loss_and_grads: loss = -(log_p + logdet) (train.py:56-60) and its gradient with respect to every trainable
// tensor of the reference (train.py:63-66, one tf.gradients call there) for one batch, sequenced here - training forward
// with everything the backward needs kept per flow, then the flows in reverse - over the stage kernels of
// flow_kernels.hip / train_kernels.hip.  No allocation, no synchronisation: every buffer is carved from the caller's
// workspace (fwn_train_workspace_bytes), every launch goes to `stream`, so a caller may record the whole call into a
// hipGraph.  DESIGN.md section 8 describes the arithmetic; the host side (tf-flowavenet_amd/training.py) only builds the
// descriptors.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <vector>

#include "../../include/fwn.h"
#include "common.h"
#include "fwn_internal.h"

int fwn_set_error(int code, const char* fmt, ...);      // api.hip: thread-local message of fwn_last_error()
#define TREQUIRE(cond, ...) do { if (!(cond)) return fwn_set_error(FWN_ERR_ARG, __VA_ARGS__); } while (0)

namespace {

constexpr float SQH = 0.70710678118654752440f;          // sqrt(1/2), modules.py:128

// ---- small kernels that replace the host-framework tensor ops of the Python sequencing -----------------------------
__global__ __launch_bounds__(256) void scale_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, float s) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (src[i] + 0.0f) * s;
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(bf16* __restrict__ dst, const float* __restrict__ src, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = (bf16)src[i];
}
// mel planes [2][B][T][half] -> rows [B][T][2 half]: the gradient image (fp32) and the up-sampled conditioning (bf16 -> fp32)
__global__ __launch_bounds__(256) void planes_to_rows_kernel(const float* __restrict__ dplanes, const bf16* __restrict__ cplanes,
                                                             long BT, int half, float* __restrict__ dy, float* __restrict__ y) {
    const long total = BT * 2 * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long bt = i / (2 * half);
        const int w = (int)(i - bt * 2 * half), q = w / half, c = w - q * half;
        const long src = ((long)q * BT + bt) * half + c;
        dy[i] = dplanes[src];
        y[i] = (float)cplanes[src];
    }
}
__global__ void up_prepare_kernel(const float* __restrict__ g, float* __restrict__ g3) {
    if (threadIdx.x < 3) g3[threadIdx.x] = g[0];        // the three kw columns share one scalar g (convolutional.py:186)
}
__global__ void up_finish_kernel(const float* __restrict__ dg3, const float* __restrict__ dwb, int s, float* __restrict__ dg,
                                 float* __restrict__ dbias) {
    if (threadIdx.x == 0) {
        dg[0] = (dg3[0] + dg3[1]) + dg3[2];
        dbias[0] = dwb[6 * s];
    }
}
__global__ void finish_loss_kernel(const float* __restrict__ out2, const float* __restrict__ an_logdet, float* __restrict__ out3) {
    if (threadIdx.x == 0) {
        // (the tail's log-det partials carry the ActNorm terms: an_logdet, the parameter-only scalar of the un-fused tail of
        // round 2, is no longer added)
        const float log_p = out2[0], logdet = out2[1];
        (void)an_logdet;
        out3[0] = -(log_p + logdet);
        out3[1] = log_p;
        out3[2] = logdet;
    }
}
inline unsigned grid_of(long n) { long b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// ---- workspace carving (the same walk sizes the workspace and hands out the pointers) ------------------------------
struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off = (off + bytes + 255) & ~(size_t)255;
        return p;
    }
};
inline long roundup(long v, long m) { return (v + m - 1) / m * m; }
inline int dilation_of(int layer) { int d = 1; for (int i = 0; i < layer; ++i) d *= 3; return d; }
inline int hop_of(const fwn_model_desc* m) { int h = 1; for (int i = 0; i < m->n_up; ++i) h *= m->up_scale[i]; return h; }

// Fork / join markers of the side stream: created on first use (the eager first call), reused by every later call - a
// recorded call creates nothing.  One pool per host thread and device.
struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t next = 0;
    hipEvent_t get() {
        if (next == ev.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[next++];
    }
};
EventPool* event_pool() {
    thread_local std::map<int, EventPool> pools;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return &pools[dev];
}

struct TnSpec { int kx, n, ntap; };
inline int tn_group_splits(const TnSpec* sp, int n, int m) {          // training.tn_group_splits
    const int e = fwn_tn_tile(m);
    long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long)sp[i].ntap * ((sp[i].kx + e - 1) / e) * ((sp[i].n + e - 1) / e);
    const long a = (m + 63) / 64, b = 256 / (tiles > 1 ? tiles : 1);
    const long r = a < b ? a : b;
    return (int)(r < 1 ? 1 : r);
}

// Round 4: the split count of a flow's weight-gradient GEMMs is planned for its BLOCK - with a side stream the groups of all
// n_flow flows of a block run as one launch (fwn_tn_multi_launch), and the one-stream path uses the same count so that both
// sum the same partials in the same order (bit-identical results).  `sp`: the jobs of ONE flow.
inline int tn_plan_flows(int n_flow) { return FWN_TUNE(FWN_TN_BLOCK, 1) ? n_flow : 1; }     // developer switch (tunable build): 0 = every flow on its own, as before
inline int tn_block_splits(const TnSpec* sp, int n, int n_flow, int m) {
    const int e = fwn_tn_tile(m);
    long tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (long)sp[i].ntap * ((sp[i].kx + e - 1) / e) * ((sp[i].n + e - 1) / e);
    tiles *= n_flow;
    // the fewest splits that fill the rounds of workgroups the launch needs to 85 % (256 CUs, one 256 x 256 tile each): the
    // six flows of block 0 are 144 tiles - one split would leave 112 CUs idle, two would run a second round for 32 of them,
    // five make 720 workgroups = three rounds at 94 %
    // (few rows: the launch is bound by writing its output tiles, and every extra split is another copy of them for the
    // weight-norm backward to read - one split per tile there.  Measured per block at 8 x 6400, TN + weight-norm kernel time:
    // blocks 0-2 fill-aware 728 / 555 / 339 us against 1030 / 653 / 371 with one split; blocks 3-7 one split 261 / 239 / 283 /
    // 376 / 562 us against 287 / 236 / 302 / 422 / 672 fill-aware)
    if (m < 4096) return 1;
    const long a = (m + 63) / 64 < 32 ? (m + 63) / 64 : 32;
    for (long n = 1; n <= a; ++n) {
        const long w = tiles * n, rounds = (w + 255) / 256;
        if (100 * w >= 85 * 256 * rounds) return (int)n;
    }
    return (int)(a < 1 ? 1 : a);
}

struct FlowSaved {          // what the training forward keeps of one flow
    void* h[FWN_MAX_LAYERS]; void* o[FWN_MAX_LAYERS]; void* aux[FWN_MAX_LAYERS];
    void* s_act; void* u_act; float* z; float* part; int nb, p;
};

struct BwdSet { void* dz; void* du; void* ds; void* dpre[FWN_MAX_LAYERS]; void* dh[FWN_MAX_LAYERS]; void* ya_bf; double* sg; };
struct Plan {               // every buffer of one call
    void* cplanes; float* ups[FWN_MAX_UPSAMPLE]; float* planes; float* gplanes; float* dcplanes; float* P; float* Ppart;
    float* partial_all; float* out2; float* an_dummy;
    FlowSaved* saved;       // host array, owned by the caller of plan()
    // backward temporaries, sized for the largest block and reused flow after flow
    void* xhl; float* dzz; void* d_all; void* d_o[FWN_MAX_LAYERS]; float* tn_part; void* tn_table; double* wn_scratch; double* up_wn;
    // the ones the weight-gradient GEMMs read.  One set without a side stream; with one, a set per flow: its weight
    // gradients run on the side stream while the main stream goes on differentiating
    std::vector<BwdSet> sets;
    float* up_dy; float* up_y; float* up_dx[2]; float* up_dwb; float* up_scr; float* up_g3; float* up_dv; float* up_dg3;
    size_t total;
    int npart;
};

// Number of fp32 elements of the grouped weight-gradient partials of ALL flows of block i (they coexist: one launch per block).
long tn_partial_floats(const fwn_model_desc* md, int i, long m) {
    const int ch = 1 << i, L = md->n_layer, half = md->num_mels / 2, cin = half * (2 << i);
    const int ldz = 2 * ch > 8 ? 2 * ch : 8;
    TnSpec sp[4 + 5 * FWN_MAX_LAYERS];
    int n = 0;
    sp[n++] = {256, ldz, 1};
    sp[n++] = {256, 256, 1};
    for (int l = 0; l < L; ++l) sp[n++] = {256, 256, 1};
    for (int l = L - 1; l >= 0; --l) {
        if (l < L - 1) sp[n++] = {256, 256, 1};
        sp[n++] = {256, 512, 3};
        sp[n++] = {cin, 512, 1};
    }
    sp[n++] = {ch < 8 ? 8 : ch, 256, 3};
    long tot = 0;
    const int ns = tn_block_splits(sp, n, tn_plan_flows(md->n_flow), (int)m);
    for (int j = 0; j < n; ++j) tot += (long)ns * ((long)sp[j].ntap * sp[j].kx + 1) * sp[j].n;
    return tot * md->n_flow;
}

void plan(const fwn_train_desc* t, long B, long T, void* ws, Plan& pl) {
    const fwn_model_desc* md = t->model;
    const int L = md->n_layer, half = md->num_mels / 2, nmel = 2 * half;
    Bump b(ws);
    pl.cplanes = b.take((size_t)2 * B * T * half * 2);
    long H = T / hop_of(md);
    for (int n = 0; n < md->n_up; ++n) {
        H *= md->up_scale[n];
        pl.ups[n] = n + 1 < md->n_up ? (float*)b.take((size_t)B * H * nmel * 4) : nullptr;      // output of stage n (fp32), except the last
    }
    pl.planes = (float*)b.take((size_t)B * T * 4);
    pl.gplanes = (float*)b.take((size_t)B * T * 4);
    pl.dcplanes = (float*)b.take((size_t)2 * B * T * half * 4);
    size_t pbytes = 0, ppart = 0;
    int npart = 0;
    long mmax = B * T / 2;
    for (int i = 0; i < md->n_block; ++i) {
        const long ch = 1L << i, m = B * T / (2 * ch);
        if (m < 4096) {
            const size_t need = (size_t)md->n_flow * L * m * 512 * 4;
            if (need > pbytes) pbytes = need;
            const size_t sp = (size_t)(fwn_cond_nsplit((int)m, ((md->n_flow + 1) / 2) * L, md->flows[i * md->n_flow].kcpad) - 1) * need;
            if (sp > ppart) ppart = sp;
        }
        for (int j = 0; j < md->n_flow; ++j) {
            FlowSaved& s = pl.saved[i * md->n_flow + j];
            for (int l = 0; l < L; ++l) {
                s.h[l] = b.take((size_t)m * 256 * 2);
                s.o[l] = b.take((size_t)m * 256 * 2);
                s.aux[l] = b.take((size_t)m * 512 * 2);
            }
            s.s_act = b.take((size_t)m * 256 * 2);
            s.u_act = b.take((size_t)m * 256 * 2);
            s.z = (float*)b.take((size_t)m * 2 * ch * 4);
            {       // log-det partial slots of the flow's tail launch (the register-streamed tail tiles differently: tail_rs.hip)
                const fwn_flow_desc* fd = &md->flows[i * md->n_flow + j];
                s.nb = fwn_tail_npartials((int)m, fwn_tail_rs_mt((int)m, fd->L, fd->Ch, fd->npt, fd->Wts != nullptr));
            }
            npart += s.nb;
        }
    }
    pl.P = (float*)b.take(pbytes);
    pl.Ppart = (float*)b.take(ppart);          // split-K partials of the hoisted conditioning (few rows)
    pl.partial_all = (float*)b.take((size_t)npart * 4);
    pl.npart = npart;
    {   // partial slices in flow order
        int off = 0;
        for (int f = 0; f < md->n_block * md->n_flow; ++f) { pl.saved[f].part = pl.partial_all ? pl.partial_all + off : nullptr; off += pl.saved[f].nb; }
    }
    pl.out2 = (float*)b.take(16);
    // backward temporaries: maxima over the blocks
    size_t dz_b = 0, dzz_b = 0, ya_b = 0, tn_b = 0, wn_b = 0, sg_b = 0;
    for (int i = 0; i < md->n_block; ++i) {
        const long ch = 1L << i, m = B * T / (2 * ch), cin = (long)half * (2 << i);
        const long ldz = 2 * ch > 8 ? 2 * ch : 8;
        dz_b = dz_b > (size_t)m * ldz * 2 ? dz_b : (size_t)m * ldz * 2;
        dzz_b = dzz_b > (size_t)m * 2 * ch * 4 ? dzz_b : (size_t)m * 2 * ch * 4;
        ya_b = ya_b > (size_t)m * (ch < 8 ? 8 : ch) * 2 ? ya_b : (size_t)m * (ch < 8 ? 8 : ch) * 2;      // rows padded to 8 channels (16 bytes)
        const size_t tnb = (size_t)tn_partial_floats(md, i, m) * 4;
        tn_b = tn_b > tnb ? tn_b : tnb;
        // weight-norm scratch of a group of <= 16 jobs: bounded by the 16 largest jobs (K/32 + 1) N 2 doubles; the
        // conditioning convs dominate: (cin/32 + 1) 256 2 each; take all jobs of the flow as the bound
        size_t wn = 0;
        wn += (size_t)(256 / 32 + 1) * 256 * 2 * (1 + 2 * L);                         // final, skip, res
        wn += (size_t)(768 / 32 + 1) * 256 * 2 * 2 * L;                               // filter, gate
        wn += (size_t)(cin / 32 + 2) * 256 * 2 * 2 * L;                               // filter_c, gate_c
        wn += (size_t)(3 * ch / 32 + 2) * 256 * 2 + 64;                               // front
        wn_b = wn_b > wn * 8 ? wn_b : wn * 8;
        const size_t sg = (size_t)fwn_small_grads_blocks(m, (int)ch) * 6 * ch * 8;
        sg_b = sg_b > sg ? sg_b : sg;
    }
    pl.xhl = b.take((size_t)B * T * 2);        // front conv of the blocks with Ch >= 32: bf16 (hi | lo) image of the plane
    pl.dzz = (float*)b.take(dzz_b);
    pl.d_all = b.take((size_t)mmax * L * 256 * 2);
    for (int l = 0; l < L; ++l) pl.d_o[l] = b.take((size_t)mmax * 256 * 2);
    // one set per flow with a side stream (a set stays live until its block is joined), sized for the flow's block
    const int nsets = t->side_stream ? md->n_block * md->n_flow : 1;
    pl.sets.resize(nsets);
    for (int k = 0; k < nsets; ++k) {
        const long mset = t->side_stream ? B * T / (2L << (k / md->n_flow)) : mmax;
        BwdSet& w = pl.sets[k];
        w.dz = b.take(dz_b);
        w.du = b.take((size_t)mset * 256 * 2);
        w.ds = b.take((size_t)mset * 256 * 2);
        for (int l = 0; l < L; ++l) {
            w.dpre[l] = b.take((size_t)mset * 512 * 2);
            w.dh[l] = b.take((size_t)mset * 256 * 2);
        }
        w.ya_bf = b.take(ya_b);
        w.sg = (double*)b.take(sg_b);          // row-range partials of the flow's small gradients (totalled on the side stream)
    }
    pl.tn_part = (float*)b.take(tn_b);
    pl.tn_table = b.take(fwn_tn_table_bytes());       // group tables of the block-wide weight-gradient launch
    pl.wn_scratch = (double*)b.take(wn_b);
    pl.up_wn = (double*)b.take(1024 * 8);
    // up-sampling backward
    pl.up_dy = (float*)b.take((size_t)B * T * nmel * 4);
    pl.up_y = (float*)b.take((size_t)B * T * nmel * 4);
    long Hs = T;
    int smax = 2;
    for (int n = 0; n < md->n_up; ++n) smax = md->up_scale[n] > smax ? md->up_scale[n] : smax;
    for (int k = 0; k < 2; ++k) { pl.up_dx[k] = (float*)b.take((size_t)B * (T / md->up_scale[md->n_up - 1]) * nmel * 4); (void)Hs; }
    pl.up_dwb = (float*)b.take((size_t)(6 * smax + 1) * 4);
    pl.up_scr = (float*)b.take((size_t)64 * (6 * smax + 1) * 4);
    pl.up_g3 = (float*)b.take(16);
    pl.up_dv = (float*)b.take((size_t)2 * smax * 3 * 4);
    pl.up_dg3 = (float*)b.take(16);
    pl.total = b.off;
}

// ---- fwn_gemm descriptors (the keyword arguments of training.gemm) ---------------------------------------------------
struct Seg { const void* x; long rows; int ld, k, shift, koff; };
fwn_gemm_desc gemm_desc(const Seg* segs, int nseg, const void* W, int ldw, int N, long M, int Ti, void* Y, int ldy, bool out_f32) {
    fwn_gemm_desc d;
    memset(&d, 0, sizeof(d));
    for (int s = 0; s < nseg; ++s) {
        d.seg[s].x = segs[s].x; d.seg[s].rows = (int)segs[s].rows; d.seg[s].ld = segs[s].ld; d.seg[s].k = segs[s].k;
        d.seg[s].shift = segs[s].shift; d.seg[s].koff = segs[s].koff;
    }
    d.nseg = nseg; d.M = (int)M; d.N = N; d.Ti = Ti;
    d.W = W; d.ldw = ldw;
    d.Y = Y; d.ldy = ldy; d.out_f32 = out_f32 ? 1 : 0;
    d.nsplit = 1; d.oscale = 1.0f;
    return d;
}

struct TnList {
    fwn_tn_job job[4 + 5 * FWN_MAX_LAYERS];
    int n = 0;
    int add(const void* x, int ldx, const void* dy, int ldy, int kx, int N, int ntap, int shift0, int dshift) {
        fwn_tn_job& q = job[n];
        memset(&q, 0, sizeof(q));
        q.x = x; q.dy = dy; q.ldx = ldx; q.Kx = kx; q.ntap = ntap; q.shift0 = shift0; q.dshift = dshift; q.ldy = ldy; q.N = N; q.bias_row = 1;
        return n++;
    }
};
struct WnItem { int tn; const float* part_direct; int part_nsplit; long part_stride; int part_rows, part_ld;   // tn < 0: part_direct
                 const fwn_conv_grad* c; int k, n, col0; float scale; const int32_t* row_src; const int32_t* col_src; };

}  // namespace

extern "C" {

size_t fwn_train_workspace_bytes(const fwn_train_desc* t, int64_t B, int64_t T) {
    if (!t || !t->model || !t->flows || B <= 0 || T <= 0) return 0;
    const fwn_model_desc* md = t->model;
    if (md->n_block < 1 || md->n_block > 16 || md->n_flow < 1 || md->n_layer < 1 || md->n_layer > FWN_MAX_LAYERS) return 0;
    if (T % hop_of(md) || T % (1L << md->n_block)) return 0;
    if (md->n_block * md->n_flow > 256) return 0;
    FlowSaved saved[256];
    Plan pl{};
    pl.saved = saved;
    plan(t, (long)B, (long)T, nullptr, pl);
    return pl.total;
}

int fwn_train_loss_and_grads(const fwn_train_desc* t, int64_t B_, int64_t T_, const float* x, const float* mel, void* workspace,
                             size_t workspace_bytes, float* out3, fwn_block_done_fn on_block_done, void* user, void* stream) {
    TREQUIRE(t && t->model && t->flows && t->model->flows, "fwn_train_loss_and_grads: null descriptor");
    TREQUIRE(x && mel && workspace && out3, "fwn_train_loss_and_grads: null pointer");
    TREQUIRE((((uintptr_t)workspace) & 255) == 0, "fwn_train_loss_and_grads: workspace must be 256-byte aligned");
    const fwn_model_desc* md = t->model;
    const long B = (long)B_, T = (long)T_;
    const int L = md->n_layer, half = md->num_mels / 2, nmel = 2 * half, NF = md->n_flow;
    TREQUIRE(md->n_block >= 1 && md->n_block <= 16 && NF >= 1 && md->n_block * NF <= 256 && L >= 1 && L <= FWN_MAX_LAYERS,
             "fwn_train_loss_and_grads: bad n_block / n_flow / n_layer");
    TREQUIRE(B > 0 && T > 0 && T % hop_of(md) == 0 && T % (1L << md->n_block) == 0, "fwn_train_loss_and_grads: T must be a multiple of hop_size and 2^n_block");
    TREQUIRE((int64_t)L * (B * T / 2) * 512 < ((int64_t)1 << 31), "fwn_train_loss_and_grads: B*T too large (2 GiB per activation buffer)");
    TREQUIRE(t->an_logdet, "fwn_train_loss_and_grads: an_logdet (device scalar) required");
    for (int i = 0; i < md->n_block; ++i)
        TREQUIRE(t->cond_rows[i] && t->front_rows[i] && t->zinv32[i] && t->br[i] && t->zcol[i], "fwn_train_loss_and_grads: missing index table (block %d)", i);
    FlowSaved saved[256];
    Plan pl{};
    pl.saved = saved;
    plan(t, B, T, workspace, pl);
    if (workspace_bytes < pl.total) return fwn_set_error(FWN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, pl.total);
    hipStream_t st = (hipStream_t)stream;

    // ---------------- forward, keeping what the backward needs ----------------
    {
        long H = T / hop_of(md);
        const float* in = mel;
        for (int n = 0; n < md->n_up; ++n) {
            const bool last = n == md->n_up - 1;
            TREQUIRE(t->up_bias_dev[n], "fwn_train_loss_and_grads: up_bias_dev[%d] is null", n);
            fwn_launch_upsample(in, (int)B, (int)H, nmel, md->up_w[n], 0.0f, t->up_bias_dev[n], md->up_scale[n], last ? nullptr : pl.ups[n],
                                last ? pl.cplanes : nullptr, st);
            H *= md->up_scale[n];
            in = pl.ups[n];
        }
    }
    fwn_launch_split(x, B, T, pl.planes, st);
    const size_t plane_elems = (size_t)B * T / 2, cplane_elems = (size_t)B * T * half;
    int p = 0;
    for (int i = 0; i < md->n_block; ++i) {
        const int ch = 1 << i;
        const long ti = T / (2 * ch), m = B * ti;
        const int cin = half * (2 << i);
        const fwn_flow_desc* d0 = &md->flows[i * NF];
        const bool hoist = m < 4096;
        if (hoist) {
            const long pn = (long)NF * L * m * 512;
            const int ns = fwn_cond_nsplit((int)m, ((NF + 1) / 2) * L, d0->kcpad);
            const bf16* ca0 = (const bf16*)pl.cplanes + (size_t)p * cplane_elems;
            const bf16* ca1 = (const bf16*)pl.cplanes + (size_t)(p ^ 1) * cplane_elems;
            if (NF > 1 && fwn_cond_merge((int)m, ((NF + 1) / 2) * L, ns)) {
                fwn_launch_cond2(ca0, ca1, d0->Wc[0], pl.P, (long)512 * d0->kcpad, m * 512, 0, 1, NF, L, (int)m, cin, d0->kcpad, pl.Ppart, pn, ns, st);
            } else {
                for (int g_ = 0; g_ < 2 && g_ < NF; ++g_)
                    fwn_launch_cond(g_ ? ca1 : ca0, d0->Wc[0], pl.P, (long)512 * d0->kcpad, m * 512, g_, 2, (NF - g_ + 1) / 2, L, (int)m, cin, d0->kcpad,
                                    pl.Ppart, pn, ns, st);
            }
            fwn_launch_cond_reduce(pl.P, pl.Ppart, pn, ns, pn, st);
        }
        for (int j = 0; j < NF; ++j) {
            const fwn_flow_desc* d = &md->flows[i * NF + j];
            FlowSaved& s = saved[i * NF + j];
            s.p = p;
            float* xa = pl.planes + (size_t)p * plane_elems;
            float* xb = pl.planes + (size_t)(p ^ 1) * plane_elems;
            const void* ca = (const bf16*)pl.cplanes + (size_t)p * cplane_elems;
            // the inference kernels, keeping what the backward needs: the front conv applies the flow's ActNorm on the fly, the
            // tail (skip sum -> final conv -> ZeroConv -> coupling + ActNorm of both planes, csrc/tail_chain.h) leaves
            // y_a / out_b in the planes - the state the backward starts from - and writes S, U and Z on its way
            fwn_launch_front(xa, d->an, d->Wfront, d->Wfront2, d->bfront, s.h[0], ch >= 32 ? pl.xhl : nullptr, (int)m, (int)ti, ch, d->kfpad, 1, nullptr, st);
            for (int l = 0; l < L; ++l) {
                const float* Pl = hoist ? pl.P + ((size_t)j * L + l) * m * 512 : nullptr;
                fwn_launch_gate(s.h[l], hoist ? nullptr : ca, Pl, d->Wd[l], d->Wc[l], nullptr, d->bgate[l], s.o[l], (int)m, (int)ti, dilation_of(l), d->cin,
                                d->kcpad, s.aux[l], st);
                if (l + 1 < L) fwn_launch_res(s.o[l], s.h[l], d->Wres[l], d->bres[l], s.h[l + 1], (int)m, nullptr, st);
            }
            {
                const long o_stride = L > 1 ? (long)(((const char*)s.o[1] - (const char*)s.o[0]) / 2) : 0;
                for (int l = 1; l < L; ++l)
                    TREQUIRE((const char*)s.o[l] == (const char*)s.o[0] + (size_t)l * o_stride * 2, "fwn_train_loss_and_grads: o buffers not evenly spaced");
                // the tail addresses o through ONE 32-bit buffer descriptor (byte span of all L layers, as fwn_tail_train checks)
                TREQUIRE(((long)(L - 1) * o_stride + m * 256) * 2 < (1L << 31), "fwn_train_loss_and_grads: o rows of a flow span %ld bytes: beyond the tail's 32-bit buffer offsets", ((long)(L - 1) * o_stride + m * 256) * 2);
                fwn_tail_chain tc;
                memset(&tc, 0, sizeof(tc));
                tc.save_s = s.s_act; tc.save_u = s.u_act; tc.save_z = s.z;
                fwn_launch_tail(s.o[0], o_stride, L, d->Wskip, d->bskip, d->Wfinal, d->bfinal, d->Wzero, d->bzero, d->ezero, d->an, xa, xb, s.part,
                                (int)m, ch, d->npt, 0, nullptr, nullptr, &tc, d->Wts, st);
            }
            p ^= 1;
        }
    }
    fwn_launch_prior(pl.planes, B * T, pl.partial_all, pl.npart, 1.0 / (double)(B * T), pl.out2, st);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, st, pl.out2, t->an_logdet, out3);

    // ---------------- backward ----------------
    // The data gradients of a flow are one dependent chain of small launches; its weight gradients (grouped TN GEMM,
    // weight-norm backward) and the conditioning gradient hang off that chain and nothing on it waits for them.  With
    // t->side_stream they run there, block by block: block i's under the main stream's chain through block i - 1, joined
    // before on_block_done(i).  Measured alternatives, all slower: handing over flow by flow (every fork / join is a
    // cross-queue edge), issuing a block's side launches one flow at a time between the next chain's flows, never
    // joining before the end.  The two queues overlap only partly (tools/diag/train_timeline.py: a chain launch of a few
    // workgroups waits behind the dispatch of a side kernel that fills the chip - tools/probe/queue_overlap.hip).
    // Without a side stream they follow their flow on the one stream.
    hipStream_t side = (hipStream_t)t->side_stream;
    EventPool* evp = nullptr;
    if (side) {
        evp = event_pool();
        TREQUIRE(evp, "fwn_train_loss_and_grads: hipEventCreate failed");
        evp->next = 0;
    }
    struct Deferred {
        TnList tn; WnItem wn[4 + 7 * FWN_MAX_LAYERS]; int nwn; long m, ti; int ch, i; const BwdSet* w;
        const fwn_flow_train_desc* td;
        const float* an;                                   // the flow's ActNorm table (totals of the small gradients)
        fwn_gemm_desc dca[FWN_MAX_LAYERS]; int ndca;      // conditioning-gradient GEMMs (accumulate into the mel image: order kept)
    };
    std::vector<Deferred> pending;        // the flows of the block whose weight gradients are still to be enqueued
    // Error exits: once anything has been handed to the side stream, EVERY return path joins it into `st` first -
    // otherwise weight-gradient kernels of up to n_block blocks (defer_block_done: all of them) would keep reading and
    // writing the caller's workspace and gradient buffers after the call has returned an error.  Event-based, so it is also
    // legal while the stream is being captured.
    struct SideJoinGuard {
        hipStream_t st, side; EventPool* evp; bool armed;
        ~SideJoinGuard() {
            if (!armed || !side || !evp) return;
            hipEvent_t e = evp->get();
            if (!e || hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) (void)hipStreamSynchronize(side);
        }
    } side_guard{st, side, evp, false};
    int pending_block = -1;               // block whose weight gradients run on the side stream, not yet joined
    // a hook that returns non-zero stops the sequencing where it stands (the side stream is joined at that point)
    bool hook_failed = false;
    auto hook_stop = [&]() -> int { return fwn_set_error(FWN_ERR_CALLBACK, "fwn_train_loss_and_grads: the on_block_done callback asked to stop"); };
    // The weight gradients of a flow in three stages: (1) totals of the small gradients + the conditioning-gradient GEMMs,
    // (2) the grouped TN GEMM(s) into fp32 partials, (3) the grouped weight-norm backward(s) that total them.  One stream:
    // (1)-(3) behind the flow's chain.  Side stream: (1) of every flow of the block, then (2) of ALL its flows as ONE launch
    // (fwn_tn_multi_launch: the block fills the chip with a sixth of the splits a flow on its own would need - a sixth of the
    // partials to write and to read back), then (3) flow by flow.  `part`: where the flow's partials go (advanced).
    auto wg_small = [&](Deferred& D, hipStream_t s_) {
        if (side)       // (one stream: totalled right behind their first pass, on the chain)
            fwn_small_grads_final(D.an, D.m, D.ch, (const long long*)t->br[D.i], (const long long*)t->zcol[D.i], D.w->sg, D.td->d_an_b, D.td->d_an_logs,
                                  D.td->d_zscale, s_);
        for (int k = 0; k < D.ndca; ++k) fwn_gemm_launch(&D.dca[k], s_);
    };
    auto wg_place = [&](Deferred& D, float*& part) {          // split count of the block, partial regions of this flow's jobs
        TnList& tn = D.tn;
        TnSpec sp[4 + 5 * FWN_MAX_LAYERS];
        for (int k = 0; k < tn.n; ++k) sp[k] = {tn.job[k].Kx, tn.job[k].N, tn.job[k].ntap};
        const int ns = tn_block_splits(sp, tn.n, tn_plan_flows(NF), (int)D.m);
        for (int k = 0; k < tn.n; ++k) {
            fwn_tn_job& q = tn.job[k];
            const long size = ((long)q.ntap * q.Kx + 1) * q.N;
            q.part = part; q.split_stride = size; q.nsplit = ns;
            part += (size_t)ns * size;
        }
    };
    auto wg_wn = [&](Deferred& D, hipStream_t s_) -> int {
        TnList& tn = D.tn;
        for (int w0 = 0; w0 < D.nwn; w0 += FWN_MAX_GROUP) {
            const int cnt = D.nwn - w0 < FWN_MAX_GROUP ? D.nwn - w0 : FWN_MAX_GROUP;
            fwn_wn_job jobs[FWN_MAX_GROUP];
            for (int k = 0; k < cnt; ++k) {
                const WnItem& it = D.wn[w0 + k];
                fwn_wn_job& q = jobs[k];
                memset(&q, 0, sizeof(q));
                if (it.tn >= 0) {
                    const fwn_tn_job& tj = tn.job[it.tn];
                    q.part = tj.part; q.split_stride = tj.split_stride; q.nsplit = tj.nsplit; q.ldp = tj.N;
                    q.bias_row = tj.ntap * tj.Kx;
                } else {
                    q.part = it.part_direct; q.split_stride = it.part_stride; q.nsplit = it.part_nsplit; q.ldp = it.part_ld;
                    q.bias_row = it.part_rows - 1;
                }
                q.row_src = it.row_src; q.col_src = it.col_src; q.col0 = it.col0; q.K = it.k; q.N = it.n; q.scale = it.scale;
                q.V = it.c->g ? it.c->V : nullptr; q.g = it.c->g; q.dV = it.c->dV; q.dg = it.c->g ? it.c->dg : nullptr; q.db = it.c->db;
                TREQUIRE(q.dV && (!q.g || (q.V && q.dg)), "fwn_train_loss_and_grads: block %d: missing gradient / master pointer", D.i);
            }
            fwn_wn_group_launch(jobs, cnt, pl.wn_scratch, s_);
        }
        return FWN_OK;
    };
    auto weight_grads = [&](Deferred& D, hipStream_t s_) -> int {        // one flow, start to end (the one-stream path)
        wg_small(D, s_);
        float* part = pl.tn_part;
        wg_place(D, part);
        for (int g0 = 0; g0 < D.tn.n; g0 += FWN_MAX_GROUP)
            fwn_tn_group_launch(D.tn.job + g0, D.tn.n - g0 < FWN_MAX_GROUP ? D.tn.n - g0 : FWN_MAX_GROUP, (int)D.m, (int)D.ti, s_);
        return wg_wn(D, s_);
    };
    // after_small: recorded on s_ behind stage (1) of the block's last flow (block 0: the conditioning gradient is complete there)
    auto weight_grads_block = [&](std::vector<Deferred>& flows, hipStream_t s_, hipEvent_t after_small) -> int {      // all flows of a block (side stream)
        if (flows.empty()) return FWN_OK;
        const fwn_tn_job* gj[FWN_MAX_GROUP * 2];
        int gn[FWN_MAX_GROUP * 2], ng = 0;
        bool fits = true;
        for (size_t k = 0; k < flows.size(); ++k) ng += (flows[k].tn.n + FWN_MAX_GROUP - 1) / FWN_MAX_GROUP;
        if (ng > fwn_tn_multi_max() || ng > (int)(sizeof(gn) / sizeof(gn[0])) || !FWN_TUNE(FWN_TN_BLOCK, 1)) fits = false;
        float* part = pl.tn_part;
        for (size_t k = 0; k < flows.size(); ++k) {
            wg_small(flows[k], s_);
            wg_place(flows[k], part);
        }
        if (after_small && hipEventRecord(after_small, s_) != hipSuccess)
            return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: hipEventRecord failed");
        TREQUIRE(part - pl.tn_part <= tn_partial_floats(md, flows[0].i, flows[0].m),
                 "fwn_train_loss_and_grads: block %d: weight-gradient partials need %ld floats, planned %ld", flows[0].i,
                 (long)(part - pl.tn_part), tn_partial_floats(md, flows[0].i, flows[0].m));
        if (fits) {
            ng = 0;
            for (size_t k = 0; k < flows.size(); ++k)
                for (int g0 = 0; g0 < flows[k].tn.n; g0 += FWN_MAX_GROUP) {
                    gj[ng] = flows[k].tn.job + g0;
                    gn[ng++] = flows[k].tn.n - g0 < FWN_MAX_GROUP ? flows[k].tn.n - g0 : FWN_MAX_GROUP;
                }
            fwn_tn_multi_launch(gj, gn, ng, (int)flows[0].m, (int)flows[0].ti, pl.tn_table, s_);
        } else {
            for (size_t k = 0; k < flows.size(); ++k)
                for (int g0 = 0; g0 < flows[k].tn.n; g0 += FWN_MAX_GROUP)
                    fwn_tn_group_launch(flows[k].tn.job + g0, flows[k].tn.n - g0 < FWN_MAX_GROUP ? flows[k].tn.n - g0 : FWN_MAX_GROUP,
                                        (int)flows[k].m, (int)flows[k].ti, s_);
        }
        for (size_t k = 0; k < flows.size(); ++k) {
            const int rc = wg_wn(flows[k], s_);
            if (rc != FWN_OK) return rc;
        }
        return FWN_OK;
    };
    // side stream: everything enqueued on `st` so far happens before what `side` gets next
    auto fork_side = [&]() -> bool {
        hipEvent_t e = evp->get();
        side_guard.armed = true;
        return e && hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(side, e, 0) == hipSuccess;
    };
    auto join_side = [&]() -> bool {
        hipEvent_t e = evp->get();
        return e && hipEventRecord(e, side) == hipSuccess && hipStreamWaitEvent(st, e, 0) == hipSuccess;
    };
    // fwn_train_desc.defer_block_done: no consumer of a block's gradients before the end of the call - the chain never
    // waits for the side stream; ONE join behind block 0, then every block is reported (same order)
    const bool defer = side && t->defer_block_done != 0;
    auto join_pending = [&]() -> bool {      // the block handed over last: join it into `st`, then report it
        if (pending_block < 0) return true;
        if (defer && pending_block > 0) { pending_block = -1; return true; }
        if (!join_side()) return false;
        if (defer) {
            for (int b_ = md->n_block - 1; b_ >= 0 && !hook_failed; --b_)
                if (on_block_done && on_block_done(user, b_) != 0) hook_failed = true;
        } else if (on_block_done && on_block_done(user, pending_block) != 0) hook_failed = true;
        pending_block = -1;
        return true;
    };
    hipEvent_t dca_done = nullptr;          // side stream, block 0: behind the last conditioning-gradient GEMM of the call
    auto hand_over = [&](bool last_block) -> int {          // fork, then everything in `pending` to the side stream
        if (!fork_side()) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: forking the side stream failed");
        // (FWN_SKIP_WG: developer builds only - times the data-gradient chain alone; the gradients are then wrong)
        if (!FWN_TUNE(FWN_SKIP_WG, 0)) {
            if (last_block) {
                dca_done = evp->get();
                if (!dca_done) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: hipEventCreate failed");
            }
            const int rc = weight_grads_block(pending, side, dca_done);
            if (rc != FWN_OK) return rc;
        }
        pending.clear();
        return FWN_OK;
    };
    // d loss / d z = z / (B T)   (log_p = mean 0.5 (-log 2 pi - z^2))
    hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_of(B * T)), dim3(256), 0, st, pl.gplanes, pl.planes, B * T, (float)(1.0 / (double)(B * T)));
    if (hipMemsetAsync(pl.dcplanes, 0, (size_t)2 * B * T * half * 4, st) != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "hipMemsetAsync failed");
    for (int i = md->n_block - 1; i >= 0; --i) {
        const int ch = 1 << i;
        const long ti = T / (2 * ch), m = B * ti;
        const int cin = half * (2 << i);
        const int ldz = 2 * ch > 8 ? 2 * ch : 8;
        for (int j = NF - 1; j >= 0; --j) {
            const fwn_flow_desc* d = &md->flows[i * NF + j];
            const fwn_flow_train_desc* td = &t->flows[i * NF + j];
            const FlowSaved& s = saved[i * NF + j];
            const BwdSet& w = side ? pl.sets[(size_t)i * NF + j] : pl.sets[0];
            const int pp = s.p;
            float* xa = pl.planes + (size_t)pp * plane_elems;          // y_a
            float* xb = pl.planes + (size_t)(pp ^ 1) * plane_elems;    // out_b
            float* ga = pl.gplanes + (size_t)pp * plane_elems;
            float* gb = pl.gplanes + (size_t)(pp ^ 1) * plane_elems;
            const void* ca = (const bf16*)pl.cplanes + (size_t)pp * cplane_elems;
            float* dca = pl.dcplanes + (size_t)pp * cplane_elems;
            // coupling
            // (also: the bf16 copy of y_a for the front conv's weight gradient, and the zero padding of dZ's rows)
            fwn_ew_coupling_bwd_ex(gb, xb, s.z, td->ez, m * ch, ch, (float)(1.0 / (2.0 * (double)m * ch)), w.dz, ldz, pl.dzz, xa, w.ya_bf, ch < 8 ? 8 : ch, st);
            {
                Seg a{w.dz, m, ldz, ldz, 0, 0};
                fwn_gemm_desc g = gemm_desc(&a, 1, td->WzT, ldz, 256, m, 0, w.du, 256, false);
                g.mask = s.u_act; g.ldmask = 256;
                fwn_gemm_launch(&g, st);
            }
            pending.emplace_back();
            Deferred& D = pending.back();
            D.nwn = 0; D.m = m; D.ti = ti; D.ch = ch; D.i = i; D.w = &w; D.td = td; D.ndca = 0; D.an = d->an;
            TnList& tn = D.tn;
            auto add_wn = [&](int tnj, const fwn_conv_grad* c, int k, int n, int col0, float scale, const int32_t* row_src, const int32_t* col_src) {
                D.wn[D.nwn++] = WnItem{tnj, nullptr, 0, 0, 0, 0, c, k, n, col0, scale, row_src, col_src};
            };
            add_wn(tn.add(s.u_act, 256, w.dz, ldz, 256, ldz, 1, 0, 0), &td->zero, 256, 2 * ch, 0, 1.0f, nullptr, t->zinv32[i]);
            add_wn(tn.add(s.s_act, 256, w.du, 256, 256, 256, 1, 0, 0), &td->final_, 256, 256, 0, 1.0f, nullptr, nullptr);
            {
                Seg a{w.du, m, 256, 256, 0, 0};
                fwn_gemm_desc g = gemm_desc(&a, 1, td->WfinT, 256, 256, m, 0, w.ds, 256, false);
                g.mask = s.s_act; g.ldmask = 256;
                fwn_gemm_launch(&g, st);
                Seg a2{w.ds, m, 256, 256, 0, 0};
                g = gemm_desc(&a2, 1, td->WskipT_all, 256, L * 256, m, 0, pl.d_all, L * 256, false);      // do_l = dS Wskip_l for every layer at once
                // the last layer's do is complete here: its gate derivative rides this epilogue (no store + fwn_gate_bwd)
                g.gate_aux = s.aux[L - 1]; g.gate_out = w.dpre[L - 1]; g.gate_col0 = (L - 1) * 256;
                fwn_gemm_launch(&g, st);
            }
            for (int l = 0; l < L; ++l) add_wn(tn.add(s.o[l], 256, w.ds, 256, 256, 256, 1, 0, 0), &td->skip[l], 256, 256, 0, 1.0f, nullptr, nullptr);
            const void* dh_next = nullptr;
            for (int l = L - 1; l >= 0; --l) {
                const int dil = dilation_of(l);
                const void* d_ol = (const bf16*)pl.d_all + (size_t)l * 256;
                int ld_do = L * 256;
                if (dh_next) {      // h_{l+1} = (h_l + res(o_l)) sqrt(1/2)
                    add_wn(tn.add(s.o[l], 256, dh_next, 256, 256, 256, 1, 0, 0), &td->res[l], 256, 256, 0, SQH, nullptr, nullptr);
                    Seg a{dh_next, m, 256, 256, 0, 0};
                    fwn_gemm_desc g = gemm_desc(&a, 1, td->WresT[l], 256, 256, m, 0, pl.d_o[l], 256, false);
                    g.R = d_ol; g.ldr = ld_do; g.rscale = (float)(1.0 / 0.7071067811865476); g.oscale = SQH;
                    g.gate_aux = s.aux[l]; g.gate_out = w.dpre[l]; g.gate_col0 = 0;      // do_l goes straight through the gate derivative
                    fwn_gemm_launch(&g, st);
                } else if (t->zero_dead_res) {      // dead res_conv of the last layer (modules.py:126-128): zero gradients
                    const fwn_conv_grad& c = td->res[l];
                    if (c.dV && hipMemsetAsync(c.dV, 0, (size_t)256 * 256 * 4, st) != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "hipMemsetAsync failed");
                    if (c.dg && hipMemsetAsync(c.dg, 0, 256 * 4, st) != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "hipMemsetAsync failed");
                    if (c.db && hipMemsetAsync(c.db, 0, 256 * 4, st) != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "hipMemsetAsync failed");
                }
                const int jd = tn.add(s.h[l], 256, w.dpre[l], 512, 256, 512, 3, -dil, dil);
                add_wn(jd, &td->filt[l], 768, 256, 0, 1.0f, nullptr, nullptr);
                add_wn(jd, &td->gate[l], 768, 256, 256, 1.0f, nullptr, nullptr);
                const int jc = tn.add(ca, cin, w.dpre[l], 512, cin, 512, 1, 0, 0);
                add_wn(jc, &td->filt_c[l], cin, 256, 0, 1.0f, t->cond_rows[i], nullptr);
                add_wn(jc, &td->gate_c[l], cin, 256, 256, 1.0f, t->cond_rows[i], nullptr);
                {
                    const bool merged = td->wct_ld == L * 512;          // the layers' WcT side by side: one GEMM below, after the loop
                    if (!merged) {
                        Seg a{w.dpre[l], m, 512, 512, 0, 0};
                        fwn_gemm_desc g = gemm_desc(&a, 1, td->WcT[l], td->wct_ld ? td->wct_ld : 512, cin, m, 0, dca, cin, true);
                        g.accumulate = 1;
                        // nothing in the flow chain reads the conditioning gradient: with a side stream it leaves the chain
                        if (side) D.dca[D.ndca++] = g;
                        else fwn_gemm_launch(&g, st);
                    }
                    Seg sg[3];
                    for (int tap = 0; tap < 3; ++tap) sg[tap] = {w.dpre[l], m, 512, 512, -(tap - 1) * dil, tap * 512};
                    fwn_gemm_desc g = gemm_desc(sg, 3, td->WdT[l], 1536, 256, m, (int)ti, w.dh[l], 256, false);
                    if (dh_next) { g.R = dh_next; g.ldr = 256; g.rscale = SQH; }
                    if (l == 0) { g.mask = s.h[0]; g.ldmask = 256; }
                    fwn_gemm_launch(&g, st);
                }
                dh_next = w.dh[l];
            }
            // front conv
            if (td->wct_ld == L * 512) {      // conditioning gradient of the flow: dca += [dpre_0 | dpre_1 | ..] [Wc_0 ; Wc_1 ; ..]
                Seg sg[FWN_MAX_LAYERS];
                for (int l = 0; l < L; ++l) {
                    TREQUIRE((const char*)td->WcT[l] == (const char*)td->WcT[0] + (size_t)l * 512 * 2, "fwn_train_loss_and_grads: wct_ld = L*512 needs WcT[l] = WcT[0] + l*512");
                    sg[l] = {w.dpre[l], m, 512, 512, 0, l * 512};
                }
                fwn_gemm_desc g = gemm_desc(sg, L, td->WcT[0], L * 512, cin, m, 0, dca, cin, true);
                g.accumulate = 1;
                if (side) D.dca[D.ndca++] = g;
                else fwn_gemm_launch(&g, st);
            }
            // front conv: y_a rows are padded to 8 channels (16 bytes: the unit the TN GEMM moves), so every block's front weight
            // gradient is one more job of the group; front_rows maps logical row tap Ch + c to GEMM row tap max(Ch, 8) + c'
            const int kxp = ch < 8 ? 8 : ch;
            add_wn(tn.add(w.ya_bf, kxp, dh_next, 256, kxp, 256, 3, -1, 1), &td->front, 3 * ch, 256, 0, 1.0f, t->front_rows[i], nullptr);
            if (!side) {
                const int rc = weight_grads(D, st);
                if (rc != FWN_OK) return rc;
                pending.clear();
            }
            {
                Seg sg[3];
                for (int tap = 0; tap < 3; ++tap) sg[tap] = {dh_next, m, 256, 256, -(tap - 1), tap * 256};
                fwn_gemm_desc g = gemm_desc(sg, 3, td->WfT, 768, ch, m, (int)ti, ga, ch, true);
                g.accumulate = 1;
                fwn_gemm_launch(&g, st);
            }
            // ActNorm of both planes back to the flow's inputs, with its b / logs gradients and the ZeroConv scale gradient
            TREQUIRE(td->d_an_b && td->d_an_logs && td->d_zscale, "fwn_train_loss_and_grads: flow (%d,%d): missing small gradient pointers", i, j);
            fwn_small_grads_main(ga, xa, gb, xb, pl.dzz, d->an, m, ch, w.sg, st);
            if (!side) fwn_small_grads_final(d->an, m, ch, (const long long*)t->br[i], (const long long*)t->zcol[i], w.sg, td->d_an_b, td->d_an_logs,
                                             td->d_zscale, st);
            if (!side && j == 0 && on_block_done && on_block_done(user, i) != 0) hook_failed = true;
            if (hook_failed) return hook_stop();
            if (side && j == 0) {      // this block's weight gradients: under the next block's chain
                if (!join_pending()) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: joining the side stream failed");
                if (hook_failed) return hook_stop();
                const int rc = hand_over(i == 0);
                if (rc != FWN_OK) return rc;
                pending_block = i;
            }
        }
    }
    // The conditioning gradient is complete behind stage (1) of block 0's weight gradients (side stream): the up-sampling
    // backward waits for THAT event and runs under the rest of block 0's side work (its grouped TN GEMM and weight-norm
    // backward: ~1 ms in which the chain's queue had nothing to do); block 0 is joined and reported behind it.
    if (side && dca_done) {
        if (hipStreamWaitEvent(st, dca_done, 0) != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: hipStreamWaitEvent failed");
    } else if (side) {
        if (!join_pending()) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: joining the side stream failed");
        if (hook_failed) return hook_stop();
    }
    // up-sampling transposed convolutions (model.py:301-311), last stage first
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3(grid_of(B * T * nmel)), dim3(256), 0, st, pl.dcplanes, (const bf16*)pl.cplanes, B * T, half, pl.up_dy,
                       pl.up_y);
    {
        float* dy = pl.up_dy;
        const float* y = pl.up_y;
        for (int n = md->n_up - 1; n >= 0; --n) {
            const int s_ = md->up_scale[n];
            long hh = T / hop_of(md);
            for (int k = 0; k < n; ++k) hh *= md->up_scale[k];                 // rows of the stage's input
            const float* xin = n == 0 ? mel : pl.ups[n - 1];
            float* dx = n > 0 ? pl.up_dx[n & 1] : nullptr;
            const fwn_conv_grad& c = t->up[n];
            TREQUIRE(c.V && c.g && c.dV && c.dg && c.db, "fwn_train_loss_and_grads: up-sampling stage %d: missing pointer", n);
            fwn_up_bwd_launch(dy, y, xin, (int)B, (int)hh, nmel, s_, md->up_w[n], dx, pl.up_dwb, pl.up_scr, st);
            hipLaunchKernelGGL(up_prepare_kernel, dim3(1), dim3(64), 0, st, c.g, pl.up_g3);
            fwn_wn_job q;
            memset(&q, 0, sizeof(q));
            q.part = pl.up_dwb; q.split_stride = 0; q.nsplit = 1; q.ldp = 3; q.col0 = 0; q.bias_row = -1; q.K = 2 * s_; q.N = 3; q.scale = 1.0f;
            q.V = c.V; q.g = pl.up_g3; q.dV = c.dV; q.dg = pl.up_dg3; q.db = nullptr;
            fwn_wn_group_launch(&q, 1, pl.up_wn, st);
            hipLaunchKernelGGL(up_finish_kernel, dim3(1), dim3(64), 0, st, pl.up_dg3, pl.up_dwb, s_, c.dg, c.db);
            dy = dx;
            y = xin;
        }
    }
    if (side && !join_pending()) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: joining the side stream failed");
    if (hook_failed) return hook_stop();
    if (on_block_done && on_block_done(user, -1) != 0) return hook_stop();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fwn_set_error(FWN_ERR_HIP, "fwn_train_loss_and_grads: %s", hipGetErrorString(e));
    side_guard.armed = false;             // every block was joined on the way (join_pending)
    return FWN_OK;
}

}  // extern "C"
