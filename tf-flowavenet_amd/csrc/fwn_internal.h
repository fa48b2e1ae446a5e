// Internal launcher prototypes shared by the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
// Developer builds (-DFWN_TUNABLE, tools/tune.py) read dispatch thresholds from the environment once; the product
// build folds them to their defaults.
#ifdef FWN_TUNABLE
#include <stdlib.h>
static inline int fwn_tune_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#define FWN_TUNE(name, dflt) ([]() -> int { static const int v = fwn_tune_env(#name, dflt); return v; }())
#else
#define FWN_TUNE(name, dflt) (dflt)
#endif


// h8out (may be NULL): also write the e4m3 copy of the output the fp8 gate reads (front: Ch <= 16 only)
void fwn_launch_front(const float* xa, const float* an_a, const void* W, const void* W2, const float* bias,
                      void* hout, void* scratch, int M, int Ti, int Ch, int kpad, int apply_an, void* h8out, hipStream_t st);
// Wgs: the same weights in fragment order (fwn_launch_gate_stream_pack) or nullptr; used instead of Wd / Wc where
// fwn_gate_stream_ok says so
int fwn_launch_gate_clock(const void* h, const void* ca, const void* Wgs, const float* bias, void* o, int M, int Ti, int dil, int cin,
                          unsigned long long* clk, hipStream_t st);
void fwn_launch_gate(const void* h, const void* ca, const float* P, const void* Wd, const void* Wc, const void* Wgs,
                     const float* bias, void* o, int M, int Ti, int dil, int cin, int kcpad, void* aux,
                     hipStream_t st);
long fwn_gate_stream_size(int cin);        // bytes, 0: no kernel for this cin
int fwn_gate_stream_min_rows();
int fwn_gate_stream_ok(int M, int Ti, int dil, int cin, bool fused_cond, bool aux);
void fwn_launch_gate_stream_pack(const void* Wd, const void* Wc, int cin, int kcpad, void* out, hipStream_t st);
// gate_rs.hip: the register-streamed launch itself (the caller has checked fwn_gate_stream_ok)
void fwn_launch_gate_rs(const void* h, const void* ca, const void* Wgs, const float* bias, void* o, int M, int Ti, int dil, int cin,
                        hipStream_t st);
// the gate with its dilated taps in fp8 (h8 e4m3 [M][256], Wd8 e4m3 [512][768] stored as W 2^wexp); fwn_gate_fp8_ok says
// whether this shape has such a kernel (the tap-sharing tiles: M >= 12288 rows, dilation <= 3, conditioning fused)
int fwn_gate_fp8_ok(int M, int dil);
void fwn_launch_gate_fp8(const void* h8, const void* ca, const void* Wd8, int wexp, const void* Wc, const float* bias, void* o,
                         int M, int Ti, int dil, int cin, int kcpad, hipStream_t st);
void fwn_launch_res(const void* o, const void* hin, const void* W, const float* bias, void* hout, int M, void* h8out,
                    hipStream_t st);
void fwn_launch_cond(const void* ca, const void* Wc_base, float* P_base, long w_stride, long p_stride,
                     int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part_base, long part_stride,
                     int nsplit, hipStream_t st);
// whether both parity groups of a block go into ONE launch (cheaper by the stream model of flow_kernels.hip, or split K)
bool fwn_cond_merge(int M, int nz_group, int nsplit);
void fwn_launch_cond2(const void* ca, const void* ca_odd, const void* Wc_base, float* P_base, long w_stride, long p_stride,
                      int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part_base, long part_stride,
                      int nsplit, hipStream_t st);
// split-K of the hoisted conditioning for few rows: the split count for nz matrices per launch, and the in-order sum of
// the partial outputs (part: [nsplit - 1][..] laid out like P) into P[0..n)
int fwn_cond_nsplit(int M, int nz, int kcpad);
void fwn_launch_cond_reduce(float* P, const float* part, long part_stride, int nsplit, long n, hipStream_t st);
// register-streamed form (cond_rs.h / cond_rs.hip): whole blocks of nz = nflow * L matrices from their fragment streams
long fwn_cond_stream_size(int kcpad);
int fwn_cond_stream_min_rows();
bool fwn_cond_rs_ok(int M, int cin, int kcpad, bool have_stream);               // the kernel serves the shape
bool fwn_cond_rs_wanted(int M, int cin, int kcpad, int nz, bool have_stream);   // ... and the model-level calls use it there
int fwn_cond_rs_nsplit(int M, int nz, int kcpad);
void fwn_launch_cond_stream_pack(const void* Wc_base, long w_stride, int kcpad, int nz, void* out, hipStream_t st);
void fwn_launch_cond_rs(const void* ca, const void* ca_odd, const void* Ws, float* P, int nz, int L, int M, int cin, int kcpad,
                        float* part, long part_stride, int nsplit, hipStream_t st);
// Chaining the flows of a block (whole-model calls): out_b to a third plane buffer, and the NEXT flow's front conv computed
// by this tail (csrc/tail_chain.h).  NULL / all-zero = the plain in-place tail.
struct fwn_tail_chain {
    float* xb_out;          // out_b destination; NULL: in place (xb)
    void* h0_next;          // != NULL: also the next flow's h0 [M][256] bf16 (needs xb_out: the tiles then overlap by one row)
    const void* Wfn;        // next flow's chained front weights [256][kfn] (fwn_flow_desc.Wfront3)
    const float* bfn;       // its bias [256]
    const float* an_next;   // forward: the next flow's ActNorm table; inverse: NULL
    int kfn, Ti;
    // what the training backward keeps of the tail (each optional): S = ReLU(skip sum), U = ReLU(final conv) as bf16
    // [M][256] in natural channel order, Z = U Wz + bz as fp32 [M][2 Ch] (log_s channels, then t channels, plane order)
    void* save_s;
    void* save_u;
    float* save_z;
};
void fwn_launch_tail(const void* o, long o_stride, int L, const void* Ws, const float* bs, const void* Wf,
                     const float* bfin, const void* Wz, const float* bz, const float* ez, const float* an,
                     float* xa, float* xb, float* partial, int M, int Ch, int npt, int inverse, void* scratch_s,
                     void* scratch_u, const fwn_tail_chain* chain, const void* Wts, hipStream_t st);
// register-streamed tail (tail_rs.h / tail_rs.hip).  Wts: Wskip | Wfinal in fragment order (fwn_launch_tail_stream_pack) or
// nullptr; fwn_tail_rs_mt: 32-row tiles per workgroup of that kernel at this shape, 0 = another tail serves it
struct TailArgs;
long fwn_tail_stream_size(int L);          // bytes, 0: no kernel for this layer count
int fwn_tail_stream_min_rows();
int fwn_tail_rs_mt(int M, int L, int Ch, int npt, bool have_stream);
void fwn_launch_tail_stream_pack(const void* Ws, const void* Wf, void* out, hipStream_t st);
void fwn_launch_tail_stream_pack_jobs(const void* jobs, int njobs, hipStream_t st);     // jobs: device array of {Wskip, Wfinal, out}
void fwn_launch_tail_rs(const TailArgs& a, const void* Wts, int mt, hipStream_t st);

// one flow of the small-M chain as one launch (flow_persist.h)
struct fwn_flow_desc;
int fwn_flow_persist_sync_words(int M, int L);
int fwn_flow_persist_ok(int M, int Ch, int L, int npt, bool has_w2, bool xa_aligned);
int fwn_flow_persist_front_inside(int Ch);
void fwn_launch_flow_persist_desc(const fwn_flow_desc* d, float* xa, float* xb, void* hA, void* hB, void* o, const float* P,
                                  float* partial, unsigned* sync, int M, int Ti, int inverse, int has_front, hipStream_t st);
// process-wide developer options (fwn_set_option): -1 = auto
extern int g_fwn_opt_rs_persist;
extern int g_fwn_opt_persist_spin_us;     // flow_persist.h: bound of the one-launch flow's spins in microseconds (0 = 2 s)
int fwn_device_cus();            // compute units of the current device (cached per device)

int fwn_tail_rows(int M);        // rows per fused-tail workgroup
int fwn_tail_is_split(int M);    // the N-split tail (ring GEMMs; needs [2][M][256] bf16 scratch) serves this M
// rs_mt: fwn_tail_rs_mt of the launch (0: the register-streamed tail does not run it)
int fwn_tail_npartials(int M, int rs_mt);   // log-det partial slots a plain (un-chained) tail launch writes
int fwn_tail_npartials_chain(int M, int Ch, int front, int rs_mt);   // ... a chained launch (fwn_tail_chain given; front: h0_next set)
int fwn_tail_chain_xb_out(int M, int npt, int rs_mt);         // whether the tail at this shape can write out_b elsewhere (xb_out)
int fwn_tail_chain_front(int M, int Ch, int npt, int rs_mt);  // ... and can compute the next flow's front conv

void fwn_launch_wn_scale(const float* v, const float* g, int k_src, int n_src, float* scale, hipStream_t st);
void fwn_launch_pack(const float* v, const float* scale, const int* src_k, const int* src_n, int n_src,
                     int k_dst, int n_dst, long ld_dst, void* out, hipStream_t st);
void fwn_launch_wn_absmax(const float* v, const float* scale, int k_src, int n_src, float mul, float* amax, hipStream_t st);
void fwn_launch_pack_e4m3(const float* v, const float* scale, const int* src_k, const int* src_n, int n_src, int k_dst,
                          int n_dst, long ld_dst, float mul, const float* amax, void* out, int* exp_out, hipStream_t st);
void fwn_launch_cast_e4m3(const void* src, void* dst, long n, hipStream_t st);
void fwn_launch_gather_tables(const float* flat, const long long* idx, int nterm, long total, const double* post,
                              const unsigned char* expflag, float* out, hipStream_t st);
void fwn_launch_sum_f32(const float* in, long n, float* out, hipStream_t st);
void fwn_launch_upsample_wn(const float* v, const float* g, int s, float* out, hipStream_t st);
void fwn_launch_upsample(const float* in, int B, int H, int W, const float* wk, float bias, const float* bias_dev, int s,
                         float* out_f32, void* out_planes, hipStream_t st);
void fwn_launch_split(const float* x, long B, long T, float* planes, hipStream_t st);
void fwn_launch_merge(const float* planes, long B, long T, float* x, hipStream_t st);
void fwn_launch_ddi(const float* xa, const float* xb, int M, int Ch, float* an, hipStream_t st);
void fwn_launch_ddi_moments(const float* xa, const float* xb, int M, int Ch, double* mom, hipStream_t st);
void fwn_launch_ddi_from_moments(const double* mom, int Ch, float* an, hipStream_t st);
void fwn_launch_prior(const float* planes, long n, const float* partial, int n_partial, double inv_bt,
                      float* out2, hipStream_t st);

int fwn_sqnorm_blocks(long n);
void fwn_launch_mel(const float* wav, long B, long T, const float* window, const float* fb, int n_fft, int hop,
                    int n_mels, float ref_db, float min_db, float* mel, hipStream_t st);
void fwn_launch_grad_norm(const float* g, long n, float gscale, double* partial, float* out, hipStream_t st);
void fwn_launch_adam(float* w, const float* g, float* m, float* v, long n, const float* gnorm, float gscale,
                     float clip, float lr_t, const float* lr_dev, float b1, float b2, float eps, hipStream_t st);

// train_kernels.hip
struct fwn_gemm_desc;
int fwn_gemm_launch(const fwn_gemm_desc* g, hipStream_t st);
void fwn_transpose_launch(const void* src, int M, int C, int ld_src, int shift0, int dshift, int ntap, int Ti, void* dst,
                          int ld_dst, int ones_row, hipStream_t st);
void fwn_reduce_splits_launch(const float* partial, int nsplit, long stride, long n, float scale, float* out,
                              hipStream_t st);
void fwn_ew_actnorm_fwd(float* x, const float* an, long n, int Ch, hipStream_t st);
void fwn_ew_actnorm_fwd2(float* xa, float* xb, const float* an2, long n, int Ch, hipStream_t st);
void fwn_ew_coupling_fwd(float* yb, const float* Z, const float* ez, long n, int Ch, float* partial, int nblocks,
                         hipStream_t st);
void fwn_ew_coupling_bwd(float* g, float* ob, const float* Z, const float* ez, long n, int Ch, float cls, void* dZ,
                         int ldz, float* dzz, hipStream_t st);
void fwn_ew_coupling_bwd_ex(float* g, float* ob, const float* Z, const float* ez, long n, int Ch, float cls, void* dZ,
                            int ldz, float* dzz, const float* ya, void* ya_bf, int ldya, hipStream_t st);
void fwn_ew_gate_bwd(const void* do_, int ld_do, const void* aux, long n, void* dpre, hipStream_t st);
int fwn_colsum_blocks(long M, int C);
void fwn_ew_colsum_prod(const float* A, const float* B, long M, int C, float scale, float* partial, float* out,
                        hipStream_t st);
void fwn_ew_actnorm_bwd(float* dy, float* y, const float* an, long n, int Ch, hipStream_t st);
int fwn_small_grads_blocks(long M, int Ch);
void fwn_small_grads_main(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, long M, int Ch,
                          double* partial, hipStream_t st);
void fwn_small_grads_final(const float* an, long M, int Ch, const long long* br, const long long* zc, const double* partial, float* db,
                           float* dlogs, float* dzscale, hipStream_t st);
void fwn_small_grads_launch(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, long M, int Ch,
                            const long long* br, const long long* zc, double* partial, float* db, float* dlogs,
                            float* dzscale, hipStream_t st);
struct fwn_wn_job;
struct fwn_tn_job;
long fwn_wn_group_scratch_doubles(const fwn_wn_job* jobs, int njobs);
void fwn_wn_group_launch(const fwn_wn_job* jobs, int njobs, double* scratch, hipStream_t st);
int fwn_up_bwd_chunks(int B, int H);
void fwn_up_bwd_launch(float* dy, const float* y, const float* x, int B, int H, int W, int s, const float* wk,
                       float* dx, float* dwk_bias, float* partial, hipStream_t st);
struct fwn_scale_job;
struct fwn_pack_job;
void fwn_launch_pack_jobs(const fwn_scale_job* sjobs, int nsjobs, const fwn_pack_job* jobs, int njobs, float* scales,
                          int scale_ld, hipStream_t st);
int fwn_tn_tile(int M);
void fwn_tn_group_launch(const fwn_tn_job* jobs, int njobs, int M, int Ti, hipStream_t st);
size_t fwn_tn_table_bytes(void);
int fwn_tn_multi_max(void);
void fwn_tn_multi_launch(const fwn_tn_job* const* jobs, const int* njobs, int ngroups, int M, int Ti, void* table, hipStream_t st);
void fwn_colsum_bf16_launch(const void* dy, long M, int C, int ld, float scale, float* partial, float* out, hipStream_t st);
