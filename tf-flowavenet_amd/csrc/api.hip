// C-ABI of libfwn.so (declared in include/fwn.h): argument validation, workspace carving and
// launch sequencing for the flow forward / inverse path.  No allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#include "../../include/fwn.h"
#include "fwn_internal.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// the same for the other translation units of the library (train_api.hip)
int fwn_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FWN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return FWN_OK;
}
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(FWN_ERR_ARG, __VA_ARGS__); } while (0)
#define ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" {

int fwn_version(void) { return FWN_VERSION; }
const char* fwn_last_error(void) { return g_err; }

int fwn_wn_scale(const float* v, const float* g, int k_src, int n_src, float* scale, void* stream) {
    REQUIRE(v && g && scale, "fwn_wn_scale: null pointer");
    REQUIRE(k_src > 0 && n_src > 0, "fwn_wn_scale: bad shape %d x %d", k_src, n_src);
    fwn_launch_wn_scale(v, g, k_src, n_src, scale, (hipStream_t)stream);
    return check_launch("fwn_wn_scale");
}

int fwn_pack_bf16(const float* v, const float* scale, const int32_t* src_k, const int32_t* src_n,
                  int n_src, int k_dst, int n_dst, int64_t ld_dst, void* out_bf16, void* stream) {
    REQUIRE(v && src_k && src_n && out_bf16, "fwn_pack_bf16: null pointer");
    REQUIRE(n_src > 0 && k_dst > 0 && n_dst > 0 && ld_dst >= k_dst, "fwn_pack_bf16: bad shape");
    fwn_launch_pack(v, scale, src_k, src_n, n_src, k_dst, n_dst, (long)ld_dst, out_bf16, (hipStream_t)stream);
    return check_launch("fwn_pack_bf16");
}

int fwn_pack_jobs(const fwn_scale_job* scale_jobs, int n_scale_jobs, const fwn_pack_job* pack_jobs, int n_pack_jobs,
                  float* scales, int scale_ld, void* stream) {
    REQUIRE(n_scale_jobs >= 0 && n_pack_jobs >= 0 && (n_scale_jobs == 0 || (scale_jobs && scales)) &&
                (n_pack_jobs == 0 || pack_jobs) && scale_ld > 0 && scale_ld % 32 == 0, "fwn_pack_jobs: bad argument");
    REQUIRE(n_scale_jobs < 65536 * 32 && n_pack_jobs < 65536, "fwn_pack_jobs: too many jobs (the pack job index is gridDim.y)");
    fwn_launch_pack_jobs(scale_jobs, n_scale_jobs, pack_jobs, n_pack_jobs, scales, scale_ld, (hipStream_t)stream);
    return check_launch("fwn_pack_jobs");
}

int fwn_gather_tables(const float* flat, const int64_t* idx, int nterm, int64_t total, const double* post,
                      const unsigned char* mode, float* out, void* stream) {
    REQUIRE(flat && idx && post && mode && out && nterm >= 1 && total > 0, "fwn_gather_tables: bad argument");
    fwn_launch_gather_tables(flat, (const long long*)idx, nterm, (long)total, post, mode, out, (hipStream_t)stream);
    return check_launch("fwn_gather_tables");
}
int fwn_sum_f32(const float* in, int64_t n, float* out, void* stream) {
    REQUIRE(in && out && n > 0, "fwn_sum_f32: bad argument");
    fwn_launch_sum_f32(in, (long)n, out, (hipStream_t)stream);
    return check_launch("fwn_sum_f32");
}
int fwn_upsample_wn(const float* v, const float* g, int s, float* out, void* stream) {
    REQUIRE(v && g && out && s > 0, "fwn_upsample_wn: bad argument");
    fwn_launch_upsample_wn(v, g, s, out, (hipStream_t)stream);
    return check_launch("fwn_upsample_wn");
}

int fwn_upsample_stage(const float* in, int B, int H, int W, const float* wk, float bias, int s,
                       float* out_f32, void* out_cplanes, void* stream) {
    REQUIRE(in && wk, "fwn_upsample_stage: null pointer");
    REQUIRE((out_f32 != nullptr) != (out_cplanes != nullptr), "fwn_upsample_stage: exactly one output");
    REQUIRE(B > 0 && H > 0 && W > 0 && s > 0 && (s % 2) == 0, "fwn_upsample_stage: bad shape (s must be even)");
    REQUIRE(!out_cplanes || (W % 2) == 0, "fwn_upsample_stage: W must be even for planes");
    fwn_launch_upsample(in, B, H, W, wk, bias, nullptr, s, out_f32, out_cplanes, (hipStream_t)stream);
    return check_launch("fwn_upsample_stage");
}
int fwn_upsample_stage_dev(const float* in, int B, int H, int W, const float* wk, const float* bias, int s,
                           float* out_f32, void* out_cplanes, void* stream) {
    REQUIRE(in && wk && bias, "fwn_upsample_stage_dev: null pointer");
    REQUIRE((out_f32 != nullptr) != (out_cplanes != nullptr), "fwn_upsample_stage_dev: exactly one output");
    REQUIRE(B > 0 && H > 0 && W > 0 && s > 0 && (s % 2) == 0, "fwn_upsample_stage_dev: bad shape (s must be even)");
    REQUIRE(!out_cplanes || (W % 2) == 0, "fwn_upsample_stage_dev: W must be even for planes");
    fwn_launch_upsample(in, B, H, W, wk, 0.0f, bias, s, out_f32, out_cplanes, (hipStream_t)stream);
    return check_launch("fwn_upsample_stage_dev");
}

int fwn_split_planes(const float* x, int64_t B, int64_t T, float* planes, void* stream) {
    REQUIRE(x && planes && B > 0 && T > 0 && (T % 2) == 0, "fwn_split_planes: bad argument");
    fwn_launch_split(x, B, T, planes, (hipStream_t)stream);
    return check_launch("fwn_split_planes");
}
int fwn_merge_planes(const float* planes, int64_t B, int64_t T, float* x, void* stream) {
    REQUIRE(x && planes && B > 0 && T > 0 && (T % 2) == 0, "fwn_merge_planes: bad argument");
    fwn_launch_merge(planes, B, T, x, (hipStream_t)stream);
    return check_launch("fwn_merge_planes");
}

int fwn_actnorm_ddi(const float* xa, const float* xb, int M, int Ch, float* an, void* stream) {
    REQUIRE(xa && xb && an && M > 0 && Ch > 0, "fwn_actnorm_ddi: bad argument");
    fwn_launch_ddi(xa, xb, M, Ch, an, (hipStream_t)stream);
    return check_launch("fwn_actnorm_ddi");
}

int fwn_actnorm_moments(const float* xa, const float* xb, int M, int Ch, double* mom, void* stream) {
    REQUIRE(xa && xb && mom && M > 0 && Ch > 0, "fwn_actnorm_moments: bad argument");
    fwn_launch_ddi_moments(xa, xb, M, Ch, mom, (hipStream_t)stream);
    return check_launch("fwn_actnorm_moments");
}
int fwn_actnorm_from_moments(const double* mom, int Ch, float* an, void* stream) {
    REQUIRE(mom && an && Ch > 0, "fwn_actnorm_from_moments: bad argument");
    fwn_launch_ddi_from_moments(mom, Ch, an, (hipStream_t)stream);
    return check_launch("fwn_actnorm_from_moments");
}

static int check_desc(const fwn_flow_desc* d) {
    REQUIRE(d, "flow desc is null");
    REQUIRE(d->Ch >= 1 && (d->Ch & (d->Ch - 1)) == 0, "flow desc: Ch=%d must be a power of two", d->Ch);
    REQUIRE(d->L >= 1 && d->L <= FWN_MAX_LAYERS, "flow desc: L=%d out of range", d->L);
    REQUIRE(d->cin > 0 && d->cin % 8 == 0, "flow desc: cin=%d must be a multiple of 8", d->cin);
    REQUIRE(d->kcpad % 64 == 0 && d->kcpad >= d->cin, "flow desc: kcpad=%d", d->kcpad);
    REQUIRE(d->kfpad % 64 == 0 && d->kfpad >= 3 * d->Ch, "flow desc: kfpad=%d", d->kfpad);
    REQUIRE(d->npt >= 1 && d->npt * 32 >= d->Ch, "flow desc: npt=%d", d->npt);
    REQUIRE(d->Wfront && d->bfront && d->Wskip && d->bskip && d->Wfinal && d->bfinal && d->Wzero &&
            d->bzero && d->ezero && d->an, "flow desc: null weight pointer");
    for (int l = 0; l < d->L; ++l) {
        REQUIRE(d->Wd[l] && d->Wc[l] && d->bgate[l], "flow desc: null gate weights (layer %d)", l);
        REQUIRE(l == d->L - 1 || (d->Wres[l] && d->bres[l]), "flow desc: null res weights (layer %d)", l);
        REQUIRE(ALIGNED16(d->Wd[l]) && ALIGNED16(d->Wc[l]), "flow desc: weights must be 16-byte aligned");
    }
    REQUIRE(ALIGNED16(d->Wfront) && ALIGNED16(d->Wskip) && ALIGNED16(d->Wfinal) && ALIGNED16(d->Wzero),
            "flow desc: weights must be 16-byte aligned");
    for (int l = 0; l < d->L; ++l)
        REQUIRE(!d->Wgs[l] || (ALIGNED16(d->Wgs[l]) && fwn_gate_stream_size(d->cin) != 0),
                "flow desc: Wgs[%d] given but no register-streamed gate kernel exists for cin = %d (or misaligned)", l, d->cin);
    REQUIRE(!d->Wts || (ALIGNED16(d->Wts) && fwn_tail_stream_size(d->L) != 0),
            "flow desc: Wts given but no register-streamed tail kernel exists for L = %d (or misaligned)", d->L);
    if (d->Wfront3)
        REQUIRE(d->Ch <= 8 && d->kf3 == (6 * d->Ch + 15) / 16 * 16 && ALIGNED16(d->Wfront3),
                "flow desc: Wfront3 needs Ch <= 8 and kf3 = 6 Ch rounded up to 16 (Ch %d, kf3 %d)", d->Ch, d->kf3);
    return FWN_OK;
}

int fwn_front(const fwn_flow_desc* d, const float* xa, void* h_out, void* scratch, int M, int Ti, int apply_an,
              void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(xa && h_out && M > 0 && Ti > 0 && M % Ti == 0, "fwn_front: bad argument");
    REQUIRE(ALIGNED16(xa) && ALIGNED16(h_out), "fwn_front: buffers must be 16-byte aligned");
    REQUIRE(!scratch || ALIGNED16(scratch), "fwn_front: scratch must be 16-byte aligned");
    fwn_launch_front(xa, d->an, d->Wfront, d->Wfront2, d->bfront, h_out, scratch, M, Ti, d->Ch, d->kfpad, apply_an,
                     nullptr, (hipStream_t)stream);
    return check_launch("fwn_front");
}

static int dilation_of(int layer) {  // kernel_size ** n, modules.py:152
    int dil = 1;
    for (int i = 0; i < layer; ++i) dil *= 3;
    return dil;
}

int fwn_gate(const fwn_flow_desc* d, int layer, const void* h, const void* ca, const float* P, void* o,
             int M, int Ti, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(layer >= 0 && layer < d->L, "fwn_gate: layer %d out of range", layer);
    REQUIRE(h && o && M > 0 && Ti > 0 && M % Ti == 0, "fwn_gate: bad argument");
    REQUIRE((ca != nullptr) != (P != nullptr), "fwn_gate: exactly one of ca / P");
    REQUIRE(ALIGNED16(h) && ALIGNED16(o) && ALIGNED16(ca), "fwn_gate: buffers must be 16-byte aligned");
    fwn_launch_gate(h, ca, P, d->Wd[layer], d->Wc[layer], d->Wgs[layer], d->bgate[layer], o, M, Ti, dilation_of(layer),
                    d->cin, d->kcpad, nullptr, (hipStream_t)stream);
    return check_launch("fwn_gate");
}

int fwn_gate_train(const fwn_flow_desc* d, int layer, const void* h, const void* ca, const float* P, void* o, void* aux,
                   int M, int Ti, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(layer >= 0 && layer < d->L, "fwn_gate_train: layer %d out of range", layer);
    REQUIRE(h && o && aux && M > 0 && Ti > 0 && M % Ti == 0, "fwn_gate_train: bad argument");
    REQUIRE((ca != nullptr) != (P != nullptr), "fwn_gate_train: exactly one of ca / P");
    REQUIRE(ALIGNED16(h) && (!ca || ALIGNED16(ca)), "fwn_gate_train: buffers must be 16-byte aligned");
    fwn_launch_gate(h, ca, P, d->Wd[layer], d->Wc[layer], nullptr, d->bgate[layer], o, M, Ti, dilation_of(layer),
                    d->cin, d->kcpad, aux, (hipStream_t)stream);
    return check_launch("fwn_gate_train");
}

int fwn_gate_clock(const fwn_flow_desc* d, int layer, const void* h, const void* ca, void* o, int M, int Ti, uint64_t* stamps,
                   void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(layer >= 0 && layer < d->L && h && ca && o && stamps && M > 0 && Ti > 0 && M % Ti == 0, "fwn_gate_clock: bad argument");
    REQUIRE(ALIGNED16(h) && ALIGNED16(ca) && ALIGNED16(o) && (((uintptr_t)stamps) & 7) == 0, "fwn_gate_clock: misaligned buffer");
    const int n = fwn_launch_gate_clock(h, ca, d->Wgs[layer], d->bgate[layer], o, M, Ti, dilation_of(layer), d->cin,
                                        (unsigned long long*)stamps, (hipStream_t)stream);
    REQUIRE(n > 0, "fwn_gate_clock: no 256-row register-streamed gate for this shape (M = %d, cin = %d, Wgs %s)", M, d->cin,
            d->Wgs[layer] ? "set" : "NULL");
    rc = check_launch("fwn_gate_clock");
    return rc ? rc : n;
}

int64_t fwn_gate_stream_bytes(int cin) { return cin > 0 ? (int64_t)fwn_gate_stream_size(cin) : 0; }
int fwn_gate_stream_rows(void) { return fwn_gate_stream_min_rows(); }
int fwn_pack_gate_stream(const void* Wd, const void* Wc, int cin, int kcpad, void* out, void* stream) {
    REQUIRE(Wd && Wc && out && cin > 0 && kcpad >= cin && kcpad % 64 == 0, "fwn_pack_gate_stream: bad argument");
    REQUIRE(fwn_gate_stream_size(cin) != 0, "fwn_pack_gate_stream: no register-streamed gate kernel for cin = %d", cin);
    REQUIRE(ALIGNED16(Wd) && ALIGNED16(Wc) && ALIGNED16(out), "fwn_pack_gate_stream: buffers must be 16-byte aligned");
    fwn_launch_gate_stream_pack(Wd, Wc, cin, kcpad, out, (hipStream_t)stream);
    return check_launch("fwn_pack_gate_stream");
}

int fwn_gate_fp8_supported(int M, int layer) {
    return layer >= 0 && layer < FWN_MAX_LAYERS && M > 0 && fwn_gate_fp8_ok(M, dilation_of(layer)) ? 1 : 0;
}
int fwn_gate_fp8(const fwn_flow_desc* d, int layer, const void* h8, const void* ca, void* o, int M, int Ti, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(layer >= 0 && layer < d->L, "fwn_gate_fp8: layer %d out of range", layer);
    REQUIRE(h8 && ca && o && M > 0 && Ti > 0 && M % Ti == 0, "fwn_gate_fp8: bad argument");
    REQUIRE(d->Wd8[layer], "fwn_gate_fp8: the flow has no e4m3 weights (fwn_pack_e4m3)");
    REQUIRE(fwn_gate_fp8_supported(M, layer), "fwn_gate_fp8: no fp8 kernel for M=%d, layer %d (needs M >= 12288, dilation <= 3)", M, layer);
    REQUIRE(ALIGNED16(h8) && ALIGNED16(o) && ALIGNED16(ca) && ALIGNED16(d->Wd8[layer]), "fwn_gate_fp8: buffers must be 16-byte aligned");
    fwn_launch_gate_fp8(h8, ca, d->Wd8[layer], d->wd8_exp[layer], d->Wc[layer], d->bgate[layer], o, M, Ti, dilation_of(layer),
                        d->cin, d->kcpad, (hipStream_t)stream);
    return check_launch("fwn_gate_fp8");
}
int fwn_cast_e4m3(const void* src_bf16, void* dst_u8, int64_t n, void* stream) {
    REQUIRE(src_bf16 && dst_u8 && n > 0, "fwn_cast_e4m3: bad argument");
    fwn_launch_cast_e4m3(src_bf16, dst_u8, (long)n, (hipStream_t)stream);
    return check_launch("fwn_cast_e4m3");
}
int fwn_wn_absmax(const float* v, const float* scale, int k_src, int n_src, float mul, float* amax, void* stream) {
    REQUIRE(v && amax && k_src > 0 && n_src > 0, "fwn_wn_absmax: bad argument");
    fwn_launch_wn_absmax(v, scale, k_src, n_src, mul, amax, (hipStream_t)stream);
    return check_launch("fwn_wn_absmax");
}
int fwn_pack_e4m3(const float* v, const float* scale, const int32_t* src_k, const int32_t* src_n, int n_src, int k_dst,
                  int n_dst, int64_t ld_dst, float mul, const float* amax, void* out_u8, int32_t* exp_out, void* stream) {
    REQUIRE(v && src_k && src_n && amax && out_u8 && exp_out, "fwn_pack_e4m3: null pointer");
    REQUIRE(n_src > 0 && k_dst > 0 && n_dst > 0 && ld_dst >= k_dst, "fwn_pack_e4m3: bad shape");
    fwn_launch_pack_e4m3(v, scale, src_k, src_n, n_src, k_dst, n_dst, (long)ld_dst, mul, amax, out_u8, exp_out, (hipStream_t)stream);
    return check_launch("fwn_pack_e4m3");
}

int fwn_res(const fwn_flow_desc* d, int layer, const void* o, const void* h_in, void* h_out, int M,
            void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(layer >= 0 && layer < d->L - 1, "fwn_res: layer %d has no live res_conv", layer);
    REQUIRE(o && h_in && h_out && M > 0, "fwn_res: bad argument");
    fwn_launch_res(o, h_in, d->Wres[layer], d->bres[layer], h_out, M, nullptr, (hipStream_t)stream);
    return check_launch("fwn_res");
}

int fwn_cond(const void* ca, const void* Wc_base, float* P_base, int64_t w_stride, int64_t p_stride,
             int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, void* stream) {
    REQUIRE(ca && Wc_base && P_base, "fwn_cond: null pointer");
    REQUIRE(nflow > 0 && L > 0 && M > 0 && cin > 0 && cin % 8 == 0 && kcpad % 64 == 0 && kcpad >= cin,
            "fwn_cond: bad shape");
    REQUIRE(flow0 >= 0 && flow_step > 0, "fwn_cond: bad flow group");
    fwn_launch_cond(ca, Wc_base, P_base, (long)w_stride, (long)p_stride, flow0, flow_step, nflow, L, M, cin,
                    kcpad, nullptr, 0, 1, (hipStream_t)stream);
    return check_launch("fwn_cond");
}

int fwn_cond_splits(int M, int nz, int kcpad) { return (M > 0 && nz > 0 && kcpad > 0) ? fwn_cond_nsplit(M, nz, kcpad) : 1; }
int fwn_cond_split(const void* ca, const void* Wc_base, float* P_base, int64_t w_stride, int64_t p_stride,
                   int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part, int64_t part_stride,
                   int nsplit, void* stream) {
    REQUIRE(ca && Wc_base && P_base, "fwn_cond_split: null pointer");
    REQUIRE(nflow > 0 && L > 0 && M > 0 && cin > 0 && cin % 8 == 0 && kcpad % 64 == 0 && kcpad >= cin,
            "fwn_cond_split: bad shape");
    REQUIRE(flow0 >= 0 && flow_step > 0, "fwn_cond_split: bad flow group");
    REQUIRE(nsplit >= 1 && nsplit <= kcpad / 64 && (nsplit == 1 || (part && part_stride > 0 && part_stride % 4 == 0)),
            "fwn_cond_split: nsplit=%d needs a partial buffer and at most one split per 64-wide chunk", nsplit);
    fwn_launch_cond(ca, Wc_base, P_base, (long)w_stride, (long)p_stride, flow0, flow_step, nflow, L, M, cin,
                    kcpad, part, (long)part_stride, nsplit, (hipStream_t)stream);
    return check_launch("fwn_cond_split");
}
int fwn_cond_reduce(float* P, const float* part, int64_t part_stride, int nsplit, int64_t n, void* stream) {
    REQUIRE(P && n > 0 && n % 4 == 0 && nsplit >= 1 && (nsplit == 1 || (part && part_stride >= n)), "fwn_cond_reduce: bad argument");
    fwn_launch_cond_reduce(P, part, (long)part_stride, nsplit, (long)n, (hipStream_t)stream);
    return check_launch("fwn_cond_reduce");
}
int64_t fwn_cond_stream_bytes(int kcpad) { return (int64_t)fwn_cond_stream_size(kcpad); }
int fwn_cond_stream_rows(void) { return fwn_cond_stream_min_rows(); }
int fwn_cond_stream_splits(int M, int nz, int kcpad) { return (M > 0 && nz > 0 && kcpad >= 64) ? fwn_cond_rs_nsplit(M, nz, kcpad) : 1; }
int fwn_pack_cond_stream(const void* Wc_base, int64_t w_stride, int kcpad, int nz, void* out, void* stream) {
    REQUIRE(Wc_base && out && nz > 0 && fwn_cond_stream_size(kcpad) != 0 && w_stride >= (int64_t)512 * kcpad && w_stride % 8 == 0,
            "fwn_pack_cond_stream: bad argument");
    REQUIRE(((uintptr_t)Wc_base & 15) == 0 && ((uintptr_t)out & 15) == 0, "fwn_pack_cond_stream: 16-byte aligned operands");
    fwn_launch_cond_stream_pack(Wc_base, (long)w_stride, kcpad, nz, out, (hipStream_t)stream);
    return check_launch("fwn_pack_cond_stream");
}
int fwn_cond_stream(const void* ca, const void* ca_odd, const void* Ws, float* P, int nflow, int L, int M, int cin, int kcpad,
                    float* part, int64_t part_stride, int nsplit, void* stream) {
    REQUIRE(ca && Ws && P && nflow > 0 && L > 0, "fwn_cond_stream: null pointer / bad counts");
    REQUIRE(fwn_cond_rs_ok(M, cin, kcpad, true) && kcpad >= cin, "fwn_cond_stream: no register-streamed kernel for M=%d cin=%d kcpad=%d (fwn_cond_stream_rows)", M, cin, kcpad);
    REQUIRE(nsplit >= 1 && nsplit * 8 <= kcpad / 64 * 8 && nsplit <= kcpad / 64 && (nsplit == 1 || (part && part_stride >= (int64_t)nflow * L * M * 512)),
            "fwn_cond_stream: nsplit=%d needs a partial buffer and at most one split per 64-wide chunk", nsplit);
    fwn_launch_cond_rs(ca, ca_odd, Ws, P, nflow * L, L, M, cin, kcpad, part, (long)part_stride, nsplit, (hipStream_t)stream);
    return check_launch("fwn_cond_stream");
}

// 32-row tiles per workgroup of the register-streamed tail when that kernel serves flow d at M rows, else 0
static int desc_rs_mt(const fwn_flow_desc* d, int M) { return fwn_tail_rs_mt(M, d->L, d->Ch, d->npt, d->Wts != nullptr); }

int64_t fwn_tail_stream_bytes(int L) { return L > 0 ? (int64_t)fwn_tail_stream_size(L) : 0; }
int fwn_tail_stream_rows(void) { return fwn_tail_stream_min_rows(); }
int fwn_pack_tail_stream(const void* Wskip, const void* Wfinal, int L, void* out, void* stream) {
    REQUIRE(Wskip && Wfinal && out, "fwn_pack_tail_stream: bad argument");
    REQUIRE(fwn_tail_stream_size(L) != 0, "fwn_pack_tail_stream: no register-streamed tail kernel for L = %d", L);
    REQUIRE(ALIGNED16(Wskip) && ALIGNED16(Wfinal) && ALIGNED16(out), "fwn_pack_tail_stream: buffers must be 16-byte aligned");
    fwn_launch_tail_stream_pack(Wskip, Wfinal, out, (hipStream_t)stream);
    return check_launch("fwn_pack_tail_stream");
}

int fwn_pack_tail_stream_jobs(const fwn_tail_stream_job* jobs, int njobs, int L, void* stream) {
    REQUIRE(jobs && njobs > 0 && njobs < 65536, "fwn_pack_tail_stream_jobs: bad argument");
    REQUIRE(fwn_tail_stream_size(L) != 0, "fwn_pack_tail_stream_jobs: no register-streamed tail kernel for L = %d", L);
    fwn_launch_tail_stream_pack_jobs(jobs, njobs, (hipStream_t)stream);
    return check_launch("fwn_pack_tail_stream_jobs");
}

int fwn_tail(const fwn_flow_desc* d, const void* o, float* xa, float* xb, float* partial, int M,
             int inverse, void* scratch, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(o && xa && xb && M > 0, "fwn_tail: bad argument");
    REQUIRE(!fwn_tail_is_split(M) || scratch || desc_rs_mt(d, M), "fwn_tail: M=%d runs the N-split tail: pass scratch [2][M][256] bf16", M);
    REQUIRE(!scratch || ALIGNED16(scratch), "fwn_tail: scratch must be 16-byte aligned");
    fwn_launch_tail(o, (long)M * 256, d->L, d->Wskip, d->bskip, d->Wfinal, d->bfinal, d->Wzero, d->bzero,
                    d->ezero, d->an, xa, xb, partial, M, d->Ch, d->npt, inverse, scratch,
                    scratch ? (char*)scratch + (size_t)M * 512 : nullptr, nullptr, d->Wts, (hipStream_t)stream);
    return check_launch("fwn_tail");
}

int fwn_tail_can_chain(const fwn_flow_desc* d, int M, int with_front) {
    if (!d || M <= 0 || !fwn_tail_chain_xb_out(M, d->npt, desc_rs_mt(d, M))) return 0;
    return with_front ? (fwn_tail_chain_front(M, d->Ch, d->npt, desc_rs_mt(d, M)) ? 1 : 0) : 1;
}
// the public counts are upper bounds over the kernels that may serve the shape (which one runs depends on the flow's packed
// operands): callers zero the buffer, a launch writes its first n slots
static int tail_partials_bound(int M, int Ch, int front) {
    int n = fwn_tail_npartials_chain(M, Ch, front, 0);
    const int mt = fwn_tail_rs_mt(M, 2, 1, 1, true);
    if (mt) { const int b = fwn_tail_npartials_chain(M, Ch, front, mt); n = b > n ? b : n; }
    return n;
}
int fwn_tail_partials_chained(int M, int Ch, int with_front) { return M > 0 ? tail_partials_bound(M, Ch, with_front != 0) : 0; }
// exact: the slots the tail of flow d writes at M rows - mode -1: fwn_tail / fwn_tail_train (in place), 0: fwn_tail_chained
// without a next flow, 1: with one
int fwn_tail_partials_desc(const fwn_flow_desc* d, int M, int mode) {
    if (!d || M <= 0) return 0;
    const int mt = desc_rs_mt(d, M);
    return mode < 0 ? fwn_tail_npartials(M, mt) : fwn_tail_npartials_chain(M, d->Ch, mode != 0, mt);
}
int fwn_tail_chained(const fwn_flow_desc* d, const fwn_flow_desc* next, const void* o, float* xa, const float* xb, float* xb_out,
                     void* h0_next, float* partial, int M, int Ti, int inverse, void* scratch, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (next && (rc = check_desc(next))) return rc;
    REQUIRE(o && xa && xb && xb_out && xb_out != xb && M > 0 && Ti > 0 && M % Ti == 0, "fwn_tail_chained: bad argument");
    REQUIRE(fwn_tail_can_chain(d, M, next != nullptr), "fwn_tail_chained: the tail at M=%d, Ch=%d cannot chain%s", M, d->Ch, next ? " a front conv" : "");
    REQUIRE(!next || (h0_next && ALIGNED16(h0_next) && next->Wfront3 && next->kf3 > 0 && next->Ch == d->Ch),
            "fwn_tail_chained: next needs Wfront3 / kf3, the same Ch, and h0_next");
    REQUIRE(!fwn_tail_is_split(M) || scratch || desc_rs_mt(d, M), "fwn_tail_chained: M=%d runs the N-split tail: pass scratch [2][M][256] bf16", M);
    fwn_tail_chain tc;
    memset(&tc, 0, sizeof(tc));
    tc.xb_out = xb_out;
    if (next) {
        tc.h0_next = h0_next;
        tc.Wfn = next->Wfront3; tc.bfn = next->bfront; tc.an_next = inverse ? nullptr : next->an; tc.kfn = next->kf3;
    }
    tc.Ti = Ti;
    // (xb is read only; the launcher's xb parameter is not written when xb_out is given)
    fwn_launch_tail(o, (long)M * 256, d->L, d->Wskip, d->bskip, d->Wfinal, d->bfinal, d->Wzero, d->bzero, d->ezero, d->an, xa,
                    const_cast<float*>(xb), inverse ? nullptr : partial, M, d->Ch, d->npt, inverse, scratch,
                    scratch ? (char*)scratch + (size_t)M * 512 : nullptr, &tc, d->Wts, (hipStream_t)stream);
    return check_launch("fwn_tail_chained");
}

int fwn_tail_train(const fwn_flow_desc* d, const void* o, int64_t o_stride, float* xa, float* xb, float* partial, int M,
                   void* save_s, void* save_u, float* save_z, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(o && xa && xb && M > 0 && o_stride >= 0, "fwn_tail_train: bad argument");
    REQUIRE(save_s && save_u && save_z && ALIGNED16(save_s) && ALIGNED16(save_u) && ALIGNED16(save_z),
            "fwn_tail_train: save_s / save_u / save_z (16-byte aligned) required");
    REQUIRE(((int64_t)(d->L - 1) * o_stride + (int64_t)M * 256) * 2 < ((int64_t)1 << 31), "fwn_tail_train: o spans more than 2 GiB");
    fwn_tail_chain tc;
    memset(&tc, 0, sizeof(tc));
    tc.save_s = save_s; tc.save_u = save_u; tc.save_z = save_z;
    fwn_launch_tail(o, (long)o_stride, d->L, d->Wskip, d->bskip, d->Wfinal, d->bfinal, d->Wzero, d->bzero, d->ezero, d->an, xa, xb,
                    partial, M, d->Ch, d->npt, 0, nullptr, nullptr, &tc, d->Wts, (hipStream_t)stream);
    return check_launch("fwn_tail_train");
}

int fwn_tail_partials(int M) { return M > 0 ? tail_partials_bound(M, 0, 0) : 0; }

// ddi: 0 none, 1 local two-pass init, 2 moments -> reduce callback (may be NULL) -> tables
// Chain context of one flow inside a whole-model call (NULL: a flow on its own, everything in place).
struct FlowChain {
    float* xb_out;               // third plane buffer that receives out_b, or NULL (in place)
    const fwn_flow_desc* next;   // the flow whose front conv this flow's tail also computes (same block), or NULL
    int have_h0;                 // the previous flow's tail has already written this flow's h0 into `h0`
    void* h0_next;               // out: the buffer the next flow's h0 was written to (one of h0 / h1), NULL if none
    int n_partial;               // out: log-det partial slots this flow's tail wrote
};

static int flow_run_impl(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, const void* ca,
                         void* h0, void* h1, void* o, const float* P, float* partial, int inverse, int ddi,
                         double* mom, fwn_reduce_fn reduce, void* user, void* h8a, void* h8b, FlowChain* chain, void* stream,
                         unsigned* sync = nullptr) {
    int rc = check_desc(d);
    if (rc) return rc;
    REQUIRE(B > 0 && T > 0 && T % (2 * (int64_t)d->Ch) == 0, "fwn_flow_run: T=%lld not divisible by 2*Ch=%d",
            (long long)T, 2 * d->Ch);
    REQUIRE(B * T < ((int64_t)1 << 31) && (int64_t)d->L * (B * T / 2) * 512 < ((int64_t)1 << 31),
            "fwn_flow_run: B*T too large (2 GiB per activation buffer)");
    REQUIRE(xa && xb && h0 && h1 && o, "fwn_flow_run: null buffer");
    REQUIRE((ca != nullptr) != (P != nullptr), "fwn_flow_run: exactly one of ca / P");
    REQUIRE(!(ddi && inverse), "fwn_flow_run: data-dependent init runs in the forward direction only");
    hipStream_t st = (hipStream_t)stream;
    const int Ti = (int)(T / (2 * d->Ch));
    const int M = (int)(B * Ti);
    if (ddi == 1) fwn_launch_ddi(xa, xb, M, d->Ch, d->an, st);
    if (ddi == 2) {
        REQUIRE(mom, "fwn_flow_run: no moment buffer");
        fwn_launch_ddi_moments(xa, xb, M, d->Ch, mom, st);
        rc = check_launch("fwn_actnorm_moments");
        if (rc) return rc;
        if (reduce && reduce(user, mom, 4 * d->Ch + 1, stream) != 0)
            return fail(FWN_ERR_ARG, "fwn_model_forward_init: the reduce callback failed");
        fwn_launch_ddi_from_moments(mom, d->Ch, d->an, st);
    }
    // fp8 dilated taps for layer l: e4m3 weights packed, the shape has an fp8 kernel, conditioning fused, and the
    // producer of h (VALU front conv: Ch <= 16; res kernel) can write the e4m3 copy
    auto fp8_layer = [&](int l) {
        return h8a && h8b && ca && l < d->L && d->Wd8[l] && fwn_gate_fp8_ok(M, dilation_of(l)) && (l > 0 || d->Ch <= 16);
    };
    // ---- the whole flow as ONE launch (flow_persist.h): small M, hoisted conditioning; `sync` zeroed by the caller ----
    if (sync && P && !ddi && !(h8a && ca) && !(chain && (chain->have_h0 || chain->xb_out)) &&
        fwn_flow_persist_ok(M, d->Ch, d->L, d->npt, d->Wfront2 != nullptr, ((uintptr_t)xa & 15) == 0)) {
        const int inside = fwn_flow_persist_front_inside(d->Ch);
        if (!inside)
            fwn_launch_front(xa, d->an, d->Wfront, d->Wfront2, d->bfront, h0, h1, M, Ti, d->Ch, d->kfpad, inverse ? 0 : 1, nullptr, st);
        fwn_launch_flow_persist_desc(d, xa, xb, h0, h1, o, P, inverse ? nullptr : partial, sync, M, Ti, inverse, inside, st);
        if (chain) { chain->h0_next = nullptr; chain->n_partial = fwn_tail_npartials_chain(M, d->Ch, false, 0); }
        return check_launch("fwn_flow_run_persist");
    }
    void* h8c = h8a;
    void* h8n = h8b;
    if (!(chain && chain->have_h0))
        fwn_launch_front(xa, d->an, d->Wfront, d->Wfront2, d->bfront, h0, h1, M, Ti, d->Ch, d->kfpad, inverse ? 0 : 1,
                         fp8_layer(0) ? h8c : nullptr, st);
    void* hc = h0;
    void* hn = h1;
    for (int l = 0; l < d->L; ++l) {
        void* ol = (char*)o + (size_t)l * M * 256 * 2;
        if (fp8_layer(l))
            fwn_launch_gate_fp8(h8c, ca, d->Wd8[l], d->wd8_exp[l], d->Wc[l], d->bgate[l], ol, M, Ti, dilation_of(l), d->cin,
                                d->kcpad, st);
        else
            fwn_launch_gate(hc, ca, P ? P + (size_t)l * M * 512 : nullptr, d->Wd[l], d->Wc[l], d->Wgs[l], d->bgate[l], ol, M,
                            Ti, dilation_of(l), d->cin, d->kcpad, nullptr, st);
        if (l + 1 < d->L) {
            fwn_launch_res(ol, hc, d->Wres[l], d->bres[l], hn, M, fp8_layer(l + 1) ? h8n : nullptr, st);
            void* t = hc; hc = hn; hn = t;
            t = h8c; h8c = h8n; h8n = t;
        }
    }
    // both h buffers are free once the last gate has run: the N-split tail (small M) keeps S and U there
    fwn_tail_chain tc;
    memset(&tc, 0, sizeof(tc));
    if (chain) {
        tc.xb_out = chain->xb_out;
        chain->h0_next = nullptr;
        if (chain->next && chain->xb_out) {      // the next flow's front conv rides this tail: its h0 goes to the buffer S does not use
            const fwn_flow_desc* nx = chain->next;
            tc.h0_next = hc;
            tc.Wfn = nx->Wfront3; tc.bfn = nx->bfront; tc.an_next = inverse ? nullptr : nx->an; tc.kfn = nx->kf3; tc.Ti = Ti;
            chain->h0_next = hc;
        }
        chain->n_partial = fwn_tail_npartials_chain(M, d->Ch, tc.h0_next != nullptr, desc_rs_mt(d, M));
    }
    fwn_launch_tail(o, (long)M * 256, d->L, d->Wskip, d->bskip, d->Wfinal, d->bfinal, d->Wzero, d->bzero,
                    d->ezero, d->an, xa, xb, inverse ? nullptr : partial, M, d->Ch, d->npt, inverse, hn, hc, chain ? &tc : nullptr, d->Wts, st);
    return check_launch("fwn_flow_run");
}

int fwn_flow_run(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, const void* ca,
                 void* h0, void* h1, void* o, const float* P, float* partial, int inverse, int ddi,
                 void* stream) {
    return flow_run_impl(d, B, T, xa, xb, ca, h0, h1, o, P, partial, inverse, ddi ? 1 : 0, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, stream);
}
int fwn_flow_run_fp8(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, const void* ca,
                     void* h0, void* h1, void* o, const float* P, float* partial, int inverse, int ddi,
                     void* h8a, void* h8b, void* stream) {
    REQUIRE(h8a && h8b && ALIGNED16(h8a) && ALIGNED16(h8b), "fwn_flow_run_fp8: h8 scratch buffers (16-byte aligned) required");
    return flow_run_impl(d, B, T, xa, xb, ca, h0, h1, o, P, partial, inverse, ddi ? 1 : 0, nullptr, nullptr, nullptr, h8a, h8b,
                         nullptr, stream);
}

int fwn_flow_persist_supported(const fwn_flow_desc* d, int64_t B, int64_t T) {
    if (!d || check_desc(d) != FWN_OK || B <= 0 || T <= 0 || T % (2 * (int64_t)d->Ch) != 0) return 0;
    const int64_t M = B * (T / (2 * d->Ch));
    if (M > ((int64_t)1 << 30)) return 0;
    return fwn_flow_persist_ok((int)M, d->Ch, d->L, d->npt, d->Wfront2 != nullptr, true);
}
int64_t fwn_flow_persist_sync_bytes(int M, int L) { return (M > 0 && L > 0) ? (int64_t)fwn_flow_persist_sync_words(M, L) * 4 : 0; }
int fwn_flow_run_persist(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, void* h0, void* h1, void* o,
                         const float* P, float* partial, int inverse, void* sync, void* stream) {
    REQUIRE(d && sync && P, "fwn_flow_run_persist: null pointer");
    REQUIRE(fwn_flow_persist_supported(d, B, T) && (((uintptr_t)xa) & 15) == 0,
            "fwn_flow_run_persist: this flow / shape has no one-launch form (fwn_flow_persist_supported)");
    return flow_run_impl(d, B, T, xa, xb, nullptr, h0, h1, o, P, partial, inverse, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, stream, (unsigned*)sync);
}
int fwn_flow_persist_status(const void* sync, void* stream) {
    REQUIRE(sync, "fwn_flow_persist_status: null pointer");
    unsigned w[2] = {0, 0};
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(w, sync, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(FWN_ERR_HIP, "fwn_flow_persist_status: copy failed");
    return (int)w[1];
}
int fwn_set_option(const char* name, int value) {
    REQUIRE(name, "fwn_set_option: null name");
    int* slot = !strcmp(name, "rs_persist") ? &g_fwn_opt_rs_persist : !strcmp(name, "persist_spin_us") ? &g_fwn_opt_persist_spin_us : nullptr;
    REQUIRE(slot, "fwn_set_option: unknown option '%s'", name);
    const int old = *slot;
    *slot = value;
    return old;
}

int fwn_prior_logp(const float* planes, int64_t n, const float* partial, int n_partial, float* out2,
                   void* stream) {
    REQUIRE(planes && out2 && n > 0 && n_partial >= 0 && (partial || n_partial == 0), "fwn_prior_logp: bad argument");
    fwn_launch_prior(planes, (long)n, partial, n_partial, 1.0 / (double)n, out2, (hipStream_t)stream);
    return check_launch("fwn_prior_logp");
}

// ---- mel front-end (the data format on the input side of the path) ------------------------------
int fwn_mel_spectrogram(const float* wav, int64_t B, int64_t T, const float* window, const float* fb, int n_fft,
                        int hop, int n_mels, float ref_level_db, float min_level_db, float* mel, void* stream) {
    REQUIRE(wav && window && fb && mel, "fwn_mel_spectrogram: null pointer");
    REQUIRE(n_fft >= 32 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "fwn_mel_spectrogram: n_fft=%d must be a power of two in [32, 4096]", n_fft);
    REQUIRE(B > 0 && B < 65536 && hop > 0 && n_mels > 0, "fwn_mel_spectrogram: bad shape");
    REQUIRE(T > n_fft / 2, "fwn_mel_spectrogram: T=%lld must exceed n_fft/2 (reflect padding)", (long long)T);
    REQUIRE(min_level_db < 0.0f, "fwn_mel_spectrogram: min_level_db must be negative");
    fwn_launch_mel(wav, (long)B, (long)T, window, fb, n_fft, hop, n_mels, ref_level_db, min_level_db, mel,
                   (hipStream_t)stream);
    return check_launch("fwn_mel_spectrogram");
}

// ---- training-side primitives --------------------------------------------------------------------
int fwn_gemm(const fwn_gemm_desc* g, void* stream) {
    REQUIRE(g && g->W && g->Y, "fwn_gemm: null pointer");
    REQUIRE(g->nseg >= 1 && g->nseg <= FWN_GEMM_MAXSEG, "fwn_gemm: nseg=%d", g->nseg);
    REQUIRE(g->M > 0 && g->N > 0 && g->ldw > 0 && g->ldy >= g->N && g->Ti >= 0, "fwn_gemm: bad shape");
    REQUIRE(ALIGNED16(g->W) && g->ldw % 8 == 0, "fwn_gemm: W rows must be 16-byte aligned");
    for (int s = 0; s < g->nseg; ++s) {
        const fwn_gemm_seg& sg = g->seg[s];
        REQUIRE(sg.x && sg.k > 0 && sg.k % 8 == 0 && sg.ld >= sg.k && sg.ld % 8 == 0 && ALIGNED16(sg.x) && sg.koff % 8 == 0 &&
                    sg.koff >= 0 && sg.koff + sg.k <= g->ldw && sg.rows > 0,
                "fwn_gemm: segment %d: k, ld, koff must be multiples of 8 within the operands", s);
        REQUIRE((int64_t)sg.rows * sg.ld * 2 < ((int64_t)1 << 31), "fwn_gemm: segment %d exceeds 2 GiB", s);
    }
    REQUIRE((int64_t)g->N * g->ldw * 2 < ((int64_t)1 << 31), "fwn_gemm: W exceeds 2 GiB");
    REQUIRE((int64_t)g->M * g->ldy * (g->out_f32 ? 4 : 2) < ((int64_t)1 << 31) &&
                (!g->R || (int64_t)g->M * g->ldr * 2 < ((int64_t)1 << 31)) &&
                (!g->mask || (int64_t)g->M * g->ldmask * 2 < ((int64_t)1 << 31)),
            "fwn_gemm: output / residual / mask exceed 2 GiB (32-bit buffer offsets)");
    REQUIRE(!g->R || g->ldr >= g->N, "fwn_gemm: ldr < N");
    REQUIRE(!g->mask || g->ldmask >= g->N, "fwn_gemm: ldmask < N");
    REQUIRE(g->nsplit >= 1 && g->nsplit <= 1024, "fwn_gemm: nsplit=%d", g->nsplit);
    REQUIRE(g->nsplit == 1 || (g->out_f32 && !g->bias && !g->R && !g->mask && !g->relu && !g->accumulate),
            "fwn_gemm: split-K writes plain fp32 partials");
    REQUIRE(!g->accumulate || g->out_f32, "fwn_gemm: accumulate needs an fp32 output");
    REQUIRE(!g->gate_aux || (g->gate_out && !g->out_f32 && g->gate_col0 >= 0 && g->gate_col0 % 256 == 0 && g->gate_col0 + 256 <= g->N &&
                             (int64_t)g->M * 512 * 2 < ((int64_t)1 << 31)),
            "fwn_gemm: gate_aux needs gate_out, a bf16 output and 256 columns from gate_col0 (a multiple of 256) inside N");
    fwn_gemm_launch(g, (hipStream_t)stream);
    return check_launch("fwn_gemm");
}
int fwn_transpose_shift(const void* src, int M, int C, int ld_src, int shift0, int dshift, int ntap, int Ti, void* dst,
                        int ld_dst, int ones_row, void* stream) {
    REQUIRE(src && dst && M > 0 && C > 0 && ld_src >= C && ld_dst >= M && Ti >= 0 && ntap >= 1 && ntap <= 64,
            "fwn_transpose_shift: bad argument");
    fwn_transpose_launch(src, M, C, ld_src, shift0, dshift, ntap, Ti, dst, ld_dst, ones_row, (hipStream_t)stream);
    return check_launch("fwn_transpose_shift");
}
int fwn_reduce_splits(const float* partial, int nsplit, int64_t stride, int64_t n, float scale, float* out,
                      void* stream) {
    REQUIRE(partial && out && nsplit >= 1 && n > 0 && stride >= n, "fwn_reduce_splits: bad argument");
    fwn_reduce_splits_launch(partial, nsplit, (long)stride, (long)n, scale, out, (hipStream_t)stream);
    return check_launch("fwn_reduce_splits");
}

int fwn_actnorm_apply(float* x, const float* an, int64_t n, int Ch, void* stream) {
    REQUIRE(x && an && n > 0 && Ch >= 1 && (Ch & (Ch - 1)) == 0, "fwn_actnorm_apply: bad argument");
    fwn_ew_actnorm_fwd(x, an, (long)n, Ch, (hipStream_t)stream);
    return check_launch("fwn_actnorm_apply");
}
int fwn_actnorm_apply2(float* xa, float* xb, const float* an2, int64_t n, int Ch, void* stream) {
    REQUIRE(xa && xb && an2 && n > 0 && Ch >= 1 && (Ch & (Ch - 1)) == 0, "fwn_actnorm_apply2: bad argument");
    fwn_ew_actnorm_fwd2(xa, xb, an2, (long)n, Ch, (hipStream_t)stream);
    return check_launch("fwn_actnorm_apply2");
}
int fwn_coupling_fwd(float* yb, const float* Z, const float* ez, int64_t M, int Ch, float* partial, int nblocks,
                     void* stream) {
    REQUIRE(yb && Z && ez && partial && M > 0 && Ch >= 1 && nblocks >= 1, "fwn_coupling_fwd: bad argument");
    fwn_ew_coupling_fwd(yb, Z, ez, (long)M * Ch, Ch, partial, nblocks, (hipStream_t)stream);
    return check_launch("fwn_coupling_fwd");
}
int fwn_coupling_bwd(float* g, float* out_b, const float* Z, const float* ez, int64_t M, int Ch, float cls, void* dZ,
                     int ldz, float* dzz, void* stream) {
    REQUIRE(g && out_b && Z && ez && dZ && dzz && M > 0 && Ch >= 1 && ldz >= 2 * Ch, "fwn_coupling_bwd: bad argument");
    fwn_ew_coupling_bwd(g, out_b, Z, ez, (long)M * Ch, Ch, cls, dZ, ldz, dzz, (hipStream_t)stream);
    return check_launch("fwn_coupling_bwd");
}
int fwn_gate_bwd(const void* d_o, int ld_do, const void* aux, int64_t M, void* dpre, void* stream) {
    REQUIRE(d_o && aux && dpre && M > 0 && ld_do >= 256, "fwn_gate_bwd: bad argument");
    fwn_ew_gate_bwd(d_o, ld_do, aux, (long)M * 256, dpre, (hipStream_t)stream);
    return check_launch("fwn_gate_bwd");
}
int fwn_colsum_partials(int64_t M, int C_) { return fwn_colsum_blocks((long)M, C_) * C_; }
int fwn_colsum_prod(const float* A, const float* B, int64_t M, int C_, float scale, float* partial, float* out,
                    void* stream) {
    REQUIRE(A && out && partial && M > 0 && C_ > 0, "fwn_colsum_prod: bad argument");
    fwn_ew_colsum_prod(A, B, (long)M, C_, scale, partial, out, (hipStream_t)stream);
    return check_launch("fwn_colsum_prod");
}
int fwn_actnorm_bwd(float* dy, float* y, const float* an, int64_t n, int Ch, void* stream) {
    REQUIRE(dy && y && an && n > 0 && Ch >= 1 && (Ch & (Ch - 1)) == 0, "fwn_actnorm_bwd: bad argument");
    fwn_ew_actnorm_bwd(dy, y, an, (long)n, Ch, (hipStream_t)stream);
    return check_launch("fwn_actnorm_bwd");
}
int64_t fwn_flow_small_grads_partials(int64_t M, int Ch) {
    if (M <= 0 || Ch <= 0) return 0;
    return (int64_t)fwn_small_grads_blocks((long)M, Ch) * 6 * Ch;
}
int fwn_flow_small_grads(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, int64_t M, int Ch,
                         const int64_t* br, const int64_t* zc, double* partial, float* db, float* dlogs, float* dzscale,
                         void* stream) {
    REQUIRE(ga && ya && gb && yb && dzz && an && br && zc && partial && db && dlogs && dzscale, "fwn_flow_small_grads: null pointer");
    REQUIRE(M > 0 && Ch >= 1 && Ch <= 128 && (Ch & (Ch - 1)) == 0, "fwn_flow_small_grads: Ch must be a power of two <= 128");
    fwn_small_grads_launch(ga, ya, gb, yb, dzz, an, (long)M, Ch, (const long long*)br, (const long long*)zc, partial, db, dlogs,
                           dzscale, (hipStream_t)stream);
    return check_launch("fwn_flow_small_grads");
}
static bool wn_job_ok(const fwn_wn_job& q) {
    return q.part && q.dV && q.K > 0 && q.N > 0 && q.nsplit >= 1 && q.col0 >= 0 && q.ldp >= q.col0 + q.N && (!q.g || (q.V && q.dg));
}
int64_t fwn_wn_group_scratch(const fwn_wn_job* jobs, int njobs) {
    if (!jobs || njobs < 1 || njobs > FWN_MAX_GROUP) return 0;
    return (int64_t)fwn_wn_group_scratch_doubles(jobs, njobs);
}
int fwn_wn_backward_group(const fwn_wn_job* jobs, int njobs, double* scratch, void* stream) {
    REQUIRE(jobs && njobs >= 1 && njobs <= FWN_MAX_GROUP, "fwn_wn_backward_group: 1..FWN_MAX_GROUP jobs");
    bool any_g = false;
    for (int j = 0; j < njobs; ++j) {
        REQUIRE(wn_job_ok(jobs[j]), "fwn_wn_backward_group: bad job (null pointer, shape, or weight norm without V / dg)");
        any_g = any_g || jobs[j].g;
    }
    REQUIRE(!any_g || scratch, "fwn_wn_backward_group: weight norm needs the scratch buffer");
    fwn_wn_group_launch(jobs, njobs, scratch, (hipStream_t)stream);
    return check_launch("fwn_wn_backward_group");
}

int fwn_upsample_bwd_partials(int B, int H, int s) { return fwn_up_bwd_chunks(B, H) * (6 * s + 1); }
int fwn_upsample_bwd(float* dy, const float* y, const float* x, int B, int H, int W, int s, const float* wk,
                     float* dx, float* dwk_bias, float* partial, void* stream) {
    REQUIRE(dy && y && x && wk && dwk_bias && partial, "fwn_upsample_bwd: null pointer");
    REQUIRE(B > 0 && H > 0 && W > 0 && s > 0 && (s % 2) == 0, "fwn_upsample_bwd: bad shape (s must be even)");
    fwn_up_bwd_launch(dy, y, x, B, H, W, s, wk, dx, dwk_bias, partial, (hipStream_t)stream);
    return check_launch("fwn_upsample_bwd");
}

static const char* tn_job_error(const fwn_tn_job& q, int M) {
    if (!(q.x && q.dy && q.part && q.Kx > 0 && q.N > 0 && q.ntap >= 1 && q.nsplit >= 1)) return "bad argument";
    if (!(q.ldx >= q.Kx && q.ldy >= q.N && q.ldx % 8 == 0 && q.ldy % 8 == 0 && q.Kx % 8 == 0 && q.N % 8 == 0 && ALIGNED16(q.x) &&
          ALIGNED16(q.dy)))
        return "rows must be 16-byte aligned and Kx, N multiples of 8";
    if (!((int64_t)M * q.ldx * 2 < ((int64_t)1 << 31) && (int64_t)M * q.ldy * 2 < ((int64_t)1 << 31) &&
          (int64_t)q.ntap * q.Kx * q.N * 4 < ((int64_t)1 << 31)))
        return "operand exceeds 2 GiB";
    if (!(q.nsplit == 1 || q.split_stride >= ((int64_t)q.ntap * q.Kx + (q.bias_row ? 1 : 0)) * q.N)) return "split_stride too small";
    return nullptr;
}
int fwn_tn_gemm_tile(int M) { return fwn_tn_tile(M); }
int fwn_tn_gemm_group(const fwn_tn_job* jobs, int njobs, int M, int Ti, void* stream) {
    REQUIRE(jobs && njobs >= 1 && njobs <= FWN_MAX_GROUP && M > 0 && Ti >= 0, "fwn_tn_gemm_group: 1..FWN_MAX_GROUP jobs, M > 0");
    for (int j = 0; j < njobs; ++j) {
        const char* e = tn_job_error(jobs[j], M);
        if (e) return fail(FWN_ERR_ARG, "fwn_tn_gemm_group: job %d: %s", j, e);
    }
    fwn_tn_group_launch(jobs, njobs, M, Ti, (hipStream_t)stream);
    return check_launch("fwn_tn_gemm_group");
}
int fwn_tn_gemm(const void* x, int ldx, int Kx, int ntap, int shift0, int dshift, const void* dy, int ldy, int N, int M,
                int Ti, int nsplit, float* part, int64_t split_stride, int bias_row, void* stream) {
    const fwn_tn_job q{x, dy, part, split_stride, ldx, Kx, ntap, shift0, dshift, ldy, N, nsplit, bias_row, 0};
    REQUIRE(M > 0 && Ti >= 0, "fwn_tn_gemm: bad argument");
    const char* e = tn_job_error(q, M);
    if (e) return fail(FWN_ERR_ARG, "fwn_tn_gemm: %s", e);
    fwn_tn_group_launch(&q, 1, M, Ti, (hipStream_t)stream);
    return check_launch("fwn_tn_gemm");
}
int fwn_colsum_bf16(const void* dy, int64_t M, int C_, int ld, float scale, float* partial, float* out, void* stream) {
    REQUIRE(dy && partial && out && M > 0 && C_ > 0 && ld >= C_, "fwn_colsum_bf16: bad argument");
    fwn_colsum_bf16_launch(dy, (long)M, C_, ld, scale, partial, out, (hipStream_t)stream);
    return check_launch("fwn_colsum_bf16");
}

// ---- data-parallel optimiser step -------------------------------------------------------------
int fwn_grad_norm_partials(int64_t n) { return fwn_sqnorm_blocks((long)n); }

int fwn_grad_norm(const float* g, int64_t n, float gscale, double* partial, float* gnorm_out, void* stream) {
    REQUIRE(g && partial && gnorm_out && n > 0, "fwn_grad_norm: bad argument");
    REQUIRE(ALIGNED16(g), "fwn_grad_norm: gradient buffer must be 16-byte aligned");
    fwn_launch_grad_norm(g, (long)n, gscale, partial, gnorm_out, (hipStream_t)stream);
    return check_launch("fwn_grad_norm");
}

int fwn_clip_adam(float* w, const float* g, float* m, float* v, int64_t n, const float* gnorm, float gscale,
                  float clip, float lr, int64_t step, float beta1, float beta2, float eps, void* stream) {
    REQUIRE(w && g && m && v && gnorm && n > 0, "fwn_clip_adam: bad argument");
    REQUIRE(ALIGNED16(w) && ALIGNED16(g) && ALIGNED16(m) && ALIGNED16(v), "fwn_clip_adam: buffers must be 16-byte aligned");
    REQUIRE(step >= 1 && clip > 0.0f && lr > 0.0f, "fwn_clip_adam: step must be >= 1, clip and lr positive");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    fwn_launch_adam(w, g, m, v, (long)n, gnorm, gscale, clip, (float)lr_t, nullptr, beta1, beta2, eps, (hipStream_t)stream);
    return check_launch("fwn_clip_adam");
}
double fwn_adam_rate(float lr, int64_t step, float beta1, float beta2) {
    return (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
}
int fwn_clip_adam_dev(float* w, const float* g, float* m, float* v, int64_t n, const float* gnorm, float gscale,
                      float clip, const float* lr_t, float beta1, float beta2, float eps, void* stream) {
    REQUIRE(w && g && m && v && gnorm && lr_t && n > 0, "fwn_clip_adam_dev: bad argument");
    REQUIRE(ALIGNED16(w) && ALIGNED16(g) && ALIGNED16(m) && ALIGNED16(v), "fwn_clip_adam_dev: buffers must be 16-byte aligned");
    REQUIRE(clip > 0.0f, "fwn_clip_adam_dev: clip must be positive");
    fwn_launch_adam(w, g, m, v, (long)n, gnorm, gscale, clip, 0.0f, lr_t, beta1, beta2, eps, (hipStream_t)stream);
    return check_launch("fwn_clip_adam_dev");
}

// ---------------------------------------------------------------------------------------------
// Whole-model sequencing
// ---------------------------------------------------------------------------------------------
struct Carve {
    size_t cplanes, up0, up1, planes, plane3, h0, h1, o, P, Ppart, partial, mom, h8a, h8b, sync, total;
    size_t sync_stride, sync_bytes;       // one block of counters per flow (flow_persist.h), zeroed once per pass
    int n_partial;
};
static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

static bool hoist_cond(const fwn_model_desc* m, int64_t M, int cin) {
    if (m->cond_mode == 1) return false;
    if (m->cond_mode == 2) return true;
    // small-M blocks: batch the weight-streaming cond GEMMs of all flows - where the conditioning K is long enough to pay for
    // the extra launch and the P round trip
    return M < FWN_TUNE(FWN_HOIST_M, 4096) && cin >= FWN_TUNE(FWN_HOIST_CIN, 256);
}
// the register-streamed conditioning projection (csrc/cond_rs.h) serves this block at M rows: its stream is given
static bool block_cond_rs(const fwn_model_desc* m, int blk, int64_t M) {
    const fwn_flow_desc* f0 = &m->flows[blk * m->n_flow];
    return blk < 16 && fwn_cond_rs_wanted((int)M, f0->cin, f0->kcpad, m->n_flow * m->n_layer, m->cond_stream[blk] != nullptr);
}
// K splits of the block's hoisted conditioning launch (sizes the partial buffer of the workspace, too)
static int block_cond_nsplit(const fwn_model_desc* m, int blk, int64_t M) {
    const fwn_flow_desc* f0 = &m->flows[blk * m->n_flow];
    if (block_cond_rs(m, blk, M)) return fwn_cond_rs_nsplit((int)M, m->n_flow * m->n_layer, f0->kcpad);
    return fwn_cond_nsplit((int)M, ((m->n_flow + 1) / 2) * m->n_layer, f0->kcpad);
}

// whether block blk's flows run as one launch each (flow_persist.h)
static bool persist_block(const fwn_model_desc* m, int64_t M, int blk) {
    // 0: where it measured ahead of the launch-per-stage path (<= 512 rows: blocks 4 - 7 of one clip, block 7 of the 8-clip
    // pass - there with one level's worth of workgroups, flow_persist.h's launcher), 1: nowhere,
    // 2: wherever the form exists.  Same results bit for bit either way.
    if (m->persist_mode == 1) return false;       // (gate_fp8 models too: fp8 taps need fused conditioning and >= 12288 rows - never these blocks)
    if (m->persist_mode != 2 && M > FWN_TUNE(FWN_PERSIST_AUTO_ROWS, 512)) return false;
    const fwn_flow_desc* d = &m->flows[blk * m->n_flow];
    return hoist_cond(m, M, d->cin) && fwn_flow_persist_ok((int)M, d->Ch, d->L, d->npt, d->Wfront2 != nullptr, true);
}

static int hop_of(const fwn_model_desc* m) {
    int hop = 1;
    for (int i = 0; i < m->n_up; ++i) hop *= m->up_scale[i];
    return hop;
}

static int check_model(const fwn_model_desc* m, int64_t B, int64_t T) {
    REQUIRE(m && m->flows, "model desc is null");
    REQUIRE(m->n_block >= 1 && m->n_block <= 16 && m->n_flow >= 1 && m->n_layer >= 1 &&
            m->n_layer <= FWN_MAX_LAYERS, "model desc: bad n_block/n_flow/n_layer");
    REQUIRE(m->n_up >= 1 && m->n_up <= FWN_MAX_UPSAMPLE, "model desc: n_up=%d", m->n_up);
    REQUIRE(m->num_mels > 0 && m->num_mels % 2 == 0, "model desc: num_mels must be even");
    REQUIRE(B > 0 && T > 0, "B and T must be positive");
    REQUIRE(T % hop_of(m) == 0, "T=%lld must be a multiple of hop_size=%d (model.py:231)", (long long)T, hop_of(m));
    REQUIRE(T % ((int64_t)1 << m->n_block) == 0, "T=%lld must be a multiple of 2^n_block (model.py:226)",
            (long long)T);
    REQUIRE(B * T < ((int64_t)1 << 31), "B*T too large");
    // every activation matrix is addressed through a raw buffer descriptor with 32-bit offsets and
    // an out-of-range sentinel at 2 GiB: the largest one is o = [n_layer][B*T/2][256] bf16.
    REQUIRE((int64_t)m->n_layer * (B * T / 2) * 512 < ((int64_t)1 << 31) &&
                B * T * (int64_t)(m->num_mels / 2) * 2 < ((int64_t)1 << 31),
            "B*T=%lld samples per call exceeds the 2 GiB-per-buffer limit (split the batch)", (long long)(B * T));
    for (int i = 0; i < m->n_block; ++i)
        for (int j = 0; j < m->n_flow; ++j) {
            const fwn_flow_desc* d = &m->flows[i * m->n_flow + j];
            int rc = check_desc(d);
            if (rc) return rc;
            REQUIRE(d->Ch == (1 << i) && d->L == m->n_layer && d->cin == (m->num_mels / 2) * (2 << i),
                    "model desc: flow (%d,%d) geometry mismatch", i, j);
        }
    for (int i = 0; i < m->n_up; ++i) REQUIRE(m->up_w[i], "model desc: null upsample kernel");
    return FWN_OK;
}

static int tail_partials_max(int M, int Ch) {       // a flow's tail runs plain or chained (overlapping tiles): room for either
    const int a = tail_partials_bound(M, Ch, 0), b = tail_partials_bound(M, Ch, 1);
    return a > b ? a : b;
}

// ---- plane bookkeeping of the chained flows ----
// The flow state lives in two fp32 planes (even / odd samples).  A chained tail writes out_b to a third buffer (its
// neighbours still read the old rows), so the three buffers rotate: `at[k]` = where logical plane k lives now.
struct Planes {
    float* at[2];
    float* spare;
    float* home[2];
};
static void planes_after_flow(Planes& pl, int p, bool rotated) {     // the flow read b = at[p ^ 1] and wrote out_b to spare
    if (!rotated) return;
    float* nb = pl.spare;
    pl.spare = pl.at[p ^ 1];
    pl.at[p ^ 1] = nb;
}
static int planes_go_home(Planes& pl, size_t plane_bytes, hipStream_t st) {   // back to the canonical buffers (no-op when the rotations cancel)
    auto cp = [&](float* dst, float* src) { return hipMemcpyAsync(dst, src, plane_bytes, hipMemcpyDeviceToDevice, st) == hipSuccess; };
    for (int k = 0; k < 2; ++k) {
        if (pl.at[k] == pl.home[k]) continue;
        if (pl.at[k ^ 1] == pl.home[k]) {            // the other plane sits in this one's home: move it to the spare buffer first
            if (!cp(pl.spare, pl.at[k ^ 1])) return -1;
            float* t = pl.spare; pl.spare = pl.at[k ^ 1]; pl.at[k ^ 1] = t;
        }
        if (!cp(pl.home[k], pl.at[k])) return -1;
        if (pl.spare == pl.home[k]) pl.spare = pl.at[k];
        pl.at[k] = pl.home[k];
    }
    return 0;
}

static Carve carve(const fwn_model_desc* m, int64_t B, int64_t T) {
    Carve c;
    size_t off = 0;
    const size_t half = m->num_mels / 2;
    c.cplanes = off; off = align_up(off + 2 * (size_t)B * T * half * 2);
    const size_t up_elems = (m->n_up > 1) ? (size_t)B * (T / m->up_scale[m->n_up - 1]) * m->num_mels : 0;
    c.up0 = off; off = align_up(off + up_elems * 4);
    c.up1 = off; off = align_up(off + (m->n_up > 2 ? up_elems * 4 : 0));
    c.planes = off; off = align_up(off + (size_t)B * T * 4);
    c.plane3 = off; off = align_up(off + (size_t)B * T * 2);      // third plane buffer: chained flows write out_b beside their inputs
    const size_t Mmax = (size_t)B * T / 2;
    c.h0 = off; off = align_up(off + Mmax * 256 * 2);
    c.h1 = off; off = align_up(off + Mmax * 256 * 2);
    c.o = off; off = align_up(off + (size_t)m->n_layer * Mmax * 256 * 2);
    size_t pbytes = 0, ppart = 0;
    int npart = 0;
    for (int i = 0; i < m->n_block; ++i) {
        const int64_t M = B * T / ((int64_t)2 << i);
        if (hoist_cond(m, M, m->flows[i * m->n_flow].cin)) {
            const size_t need = (size_t)m->n_flow * m->n_layer * M * 512 * 4;
            if (need > pbytes) pbytes = need;
            const size_t sp = (size_t)(block_cond_nsplit(m, i, M) - 1) * need;
            if (sp > ppart) ppart = sp;
        }
        npart += m->n_flow * tail_partials_max((int)M, m->flows[i * m->n_flow].Ch);
    }
    c.P = off; off = align_up(off + pbytes);
    c.Ppart = off; off = align_up(off + ppart);        // split-K partials of the hoisted conditioning (few rows)
    c.partial = off; off = align_up(off + (size_t)npart * 4);
    c.n_partial = npart;
    // per-flow moment buffers of the data-parallel ActNorm init: 4 Ch + 1 doubles each, Ch <= 2^(n_block-1)
    c.mom = off; off = align_up(off + (size_t)m->n_block * m->n_flow * (4 * ((size_t)1 << (m->n_block - 1)) + 1) * 8);
    c.h8a = off; off = align_up(off + (m->gate_fp8 ? Mmax * 256 : 0));      // e4m3 copies of h (fp8 gate path)
    c.h8b = off; off = align_up(off + (m->gate_fp8 ? Mmax * 256 : 0));
    c.sync_stride = 0;
    for (int i = 0; i < m->n_block; ++i) {
        const int64_t M = B * T / ((int64_t)2 << i);
        if (persist_block(m, M, i)) {
            const size_t w = (size_t)fwn_flow_persist_sync_words((int)M, m->n_layer) * 4;
            if (w > c.sync_stride) c.sync_stride = w;
        }
    }
    c.sync_bytes = c.sync_stride * (size_t)m->n_block * m->n_flow;
    c.sync = off; off = align_up(off + c.sync_bytes);
    c.total = off;
    return c;
}

size_t fwn_workspace_bytes(const fwn_model_desc* m, int64_t B, int64_t T) {
    if (check_model(m, B, T) != FWN_OK) return 0;
    return carve(m, B, T).total;
}

static void run_upsample(const fwn_model_desc* m, int64_t B, int64_t T, const float* mel, char* ws,
                         const Carve& c, hipStream_t st) {
    int H = (int)(T / hop_of(m));
    const float* in = mel;
    for (int i = 0; i < m->n_up; ++i) {
        const bool last = (i == m->n_up - 1);
        float* outf = last ? nullptr : (float*)(ws + ((i & 1) ? c.up1 : c.up0));
        fwn_launch_upsample(in, (int)B, H, m->num_mels, m->up_w[i], m->up_bias[i], nullptr, m->up_scale[i], outf,
                            last ? (void*)(ws + c.cplanes) : nullptr, st);
        H *= m->up_scale[i];
        in = outf;
    }
}

static void run_cond_groups(const fwn_model_desc* m, int blk, int64_t M, const int* parity_of_flow,
                            char* ws, const Carve& c, int64_t B, int64_t T, hipStream_t st) {
    const fwn_flow_desc* f0 = &m->flows[blk * m->n_flow];
    const size_t half = m->num_mels / 2;
    const size_t plane_elems = (size_t)B * T * half;
    const long pn = (long)m->n_flow * m->n_layer * M * 512;          // floats of the block's P matrices
    const int ns = block_cond_nsplit(m, blk, M);
    // even flows of the block read plane parity_of_flow[0], odd flows the other: one launch or one per parity group
    const char* ca0 = ws + c.cplanes + (size_t)parity_of_flow[0] * plane_elems * 2;
    const char* ca1 = ws + c.cplanes + (size_t)parity_of_flow[1] * plane_elems * 2;
    const int nzg = ((m->n_flow + 1) / 2) * m->n_layer;
    if (block_cond_rs(m, blk, M)) {        // the block's matrices from their fragment streams (csrc/cond_rs.h)
        fwn_launch_cond_rs(ca0, m->n_flow > 1 ? ca1 : nullptr, m->cond_stream[blk], (float*)(ws + c.P), m->n_flow * m->n_layer, m->n_layer,
                           (int)M, f0->cin, f0->kcpad, (float*)(ws + c.Ppart), pn, ns, st);
    } else if (m->n_flow > 1 && fwn_cond_merge((int)M, nzg, ns)) {
        fwn_launch_cond2(ca0, ca1, f0->Wc[0], (float*)(ws + c.P), (long)512 * f0->kcpad, (long)M * 512, 0, 1, m->n_flow,
                         m->n_layer, (int)M, f0->cin, f0->kcpad, (float*)(ws + c.Ppart), pn, ns, st);
    } else {
        for (int g = 0; g < 2 && g < m->n_flow; ++g)
            fwn_launch_cond(g ? ca1 : ca0, f0->Wc[0], (float*)(ws + c.P), (long)512 * f0->kcpad, (long)M * 512, g, 2, (m->n_flow - g + 1) / 2,
                            m->n_layer, (int)M, f0->cin, f0->kcpad, (float*)(ws + c.Ppart), pn, ns, st);
    }
    fwn_launch_cond_reduce((float*)(ws + c.P), (const float*)(ws + c.Ppart), pn, ns, pn, st);
}

static int check_block_contiguity(const fwn_model_desc* m, int blk) {
    const fwn_flow_desc* f0 = &m->flows[blk * m->n_flow];
    const size_t stride = (size_t)512 * f0->kcpad * 2;
    for (int j = 0; j < m->n_flow; ++j)
        for (int l = 0; l < m->n_layer; ++l)
            REQUIRE((const char*)m->flows[blk * m->n_flow + j].Wc[l] ==
                        (const char*)f0->Wc[0] + (size_t)(j * m->n_layer + l) * stride,
                    "block %d: Wc must be contiguous in (flow, layer) order for hoisted conditioning", blk);
    return FWN_OK;
}

// What flow `d` chains with inside a whole-model call: out_b to the spare plane wherever the tail at this shape can write
// it elsewhere, and the front conv of `next` (the flow that runs after d in this block and direction) in d's tail where
// that kernel exists (Ch <= 8) - not during the data-dependent init (next's ActNorm table does not exist yet) and not on
// the fp8 path (its first gate reads an e4m3 copy of h0 that only the stand-alone front conv writes).
static FlowChain flow_chain(const fwn_model_desc* m, const fwn_flow_desc* d, const fwn_flow_desc* next, int M, bool init, float* spare,
                            int have_h0, bool one_launch) {
    FlowChain ch;
    memset(&ch, 0, sizeof(ch));
    // (a block whose flows run as ONE launch each - flow_persist.h - does not chain: that form holds the whole flow already)
    const bool on = m->chain_mode != 1 && !one_launch;
    const int rs_mt = desc_rs_mt(d, M);
    ch.xb_out = (on && fwn_tail_chain_xb_out(M, d->npt, rs_mt)) ? spare : nullptr;
    ch.next = (ch.xb_out && next && !init && !m->gate_fp8 && next->Wfront3 && next->kf3 > 0 && fwn_tail_chain_front(M, d->Ch, d->npt, rs_mt)) ? next : nullptr;
    ch.have_h0 = have_h0;
    return ch;
}

static int record_block_event(const fwn_model_desc* m, int k, hipStream_t st) {
    if (!m->block_events || !m->block_events[k]) return 0;
    return hipEventRecord((hipEvent_t)m->block_events[k], st) != hipSuccess;
}

// Whether any one-launch flow of the LAST pass run in `workspace` (fwn_model_forward / fwn_model_reverse with the same m, B, T)
// gave up a bounded dependency wait (its outputs are NaN then: never silently wrong, but the pass itself returned FWN_OK):
// 0 = none, > 0 = the give-up code of the first such flow (1 + the stage that waited).  Synchronises the stream.  ADVICE r5: the
// per-flow sync blocks live inside the workspace, out of reach of fwn_flow_persist_status.
int fwn_model_persist_status(const fwn_model_desc* m, int64_t B, int64_t T, const void* workspace, void* stream) {
    int rc = check_model(m, B, T);
    if (rc) return rc;
    REQUIRE(workspace, "fwn_model_persist_status: null workspace");
    const Carve c = carve(m, B, T);
    if (!c.sync_bytes) return 0;
    const int nf = m->n_block * m->n_flow;
    unsigned w[1024];
    REQUIRE(nf <= 1024, "fwn_model_persist_status: more than 1024 flows");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
        hipMemcpy2D(w, sizeof(unsigned), (const char*)workspace + c.sync + sizeof(unsigned), c.sync_stride, sizeof(unsigned), (size_t)nf,
                    hipMemcpyDeviceToHost) != hipSuccess)
        return fail(FWN_ERR_HIP, "fwn_model_persist_status: copy failed");
    for (int k = 0; k < nf; ++k)
        if (w[k]) return (int)w[k];
    return 0;
}

static int model_forward_impl(const fwn_model_desc* m, int64_t B, int64_t T, const float* x, const float* mel,
                              void* workspace, size_t workspace_bytes, float* out2, float* z_planes, int init,
                              fwn_reduce_fn reduce, void* user, void* stream) {
    int rc = check_model(m, B, T);
    if (rc) return rc;
    REQUIRE(x && mel && workspace && out2, "fwn_model_forward: null pointer");
    REQUIRE((((uintptr_t)workspace) & 255) == 0, "workspace must be 256-byte aligned");
    const Carve c = carve(m, B, T);
    if (workspace_bytes < c.total)
        return fail(FWN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, c.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t half = m->num_mels / 2;
    const size_t cplane_bytes = (size_t)B * T * half * 2;
    const size_t plane_elems = (size_t)B * T / 2;
    float* planes = (float*)(ws + c.planes);

    run_upsample(m, B, T, mel, ws, c, st);
    fwn_launch_split(x, B, T, planes, st);
    if (c.sync_bytes && !init && hipMemsetAsync(ws + c.sync, 0, c.sync_bytes, st) != hipSuccess)
        return fail(FWN_ERR_HIP, "fwn_model_forward: hipMemsetAsync failed");
    int p = 0;
    float* partial = (float*)(ws + c.partial);
    int poff = 0;
    Planes pl{{planes, planes + plane_elems}, (float*)(ws + c.plane3), {planes, planes + plane_elems}};
    for (int i = 0; i < m->n_block; ++i) {
        if (record_block_event(m, i, st)) return fail(FWN_ERR_HIP, "fwn_model_forward: hipEventRecord failed");
        const int64_t M = B * T / ((int64_t)2 << i);
        const bool hoist = hoist_cond(m, M, m->flows[i * m->n_flow].cin);
        if (hoist) {
            rc = check_block_contiguity(m, i);
            if (rc) return rc;
            const int par[2] = {p, p ^ 1};
            run_cond_groups(m, i, M, par, ws, c, B, T, st);
        }
        void* hA = ws + c.h0;
        void* hB = ws + c.h1;
        int have_h0 = 0;
        for (int j = 0; j < m->n_flow; ++j) {
            const fwn_flow_desc* d = &m->flows[i * m->n_flow + j];
            const void* ca = hoist ? nullptr : (const void*)(ws + c.cplanes + (size_t)p * cplane_bytes);
            const float* P = hoist ? (const float*)(ws + c.P) + (size_t)j * m->n_layer * M * 512 : nullptr;
            double* mom = (double*)(ws + c.mom) + (size_t)(i * m->n_flow + j) * (4 * ((size_t)1 << (m->n_block - 1)) + 1);
            unsigned* sync = (!init && persist_block(m, M, i)) ? (unsigned*)(ws + c.sync + (size_t)(i * m->n_flow + j) * c.sync_stride) : nullptr;
            FlowChain ch = flow_chain(m, d, j + 1 < m->n_flow ? d + 1 : nullptr, (int)M, init != 0, pl.spare, have_h0, sync != nullptr);
            rc = flow_run_impl(d, B, T, pl.at[p], pl.at[p ^ 1], ca, hA, hB, ws + c.o, P, partial + poff, 0, init, mom, reduce, user,
                               m->gate_fp8 ? ws + c.h8a : nullptr, m->gate_fp8 ? ws + c.h8b : nullptr, &ch, stream, sync);
            if (rc) return rc;
            poff += ch.n_partial;
            planes_after_flow(pl, p, ch.xb_out != nullptr);
            have_h0 = ch.h0_next != nullptr;
            if (have_h0 && ch.h0_next != hA) { void* t = hA; hA = hB; hB = t; }
            p ^= 1;   // change_order (model.py:190)
        }
    }
    if (record_block_event(m, m->n_block, st)) return fail(FWN_ERR_HIP, "fwn_model_forward: hipEventRecord failed");
    if (planes_go_home(pl, plane_elems * 4, st)) return fail(FWN_ERR_HIP, "fwn_model_forward: plane copy failed");
    fwn_launch_prior(planes, (long)(B * T), partial, poff, 1.0 / (double)(B * T), out2, st);
    if (z_planes) {
        hipError_t e = hipMemcpyAsync(z_planes, planes, (size_t)B * T * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail(FWN_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    }
    return check_launch("fwn_model_forward");
}

int fwn_model_forward(const fwn_model_desc* m, int64_t B, int64_t T, const float* x, const float* mel,
                      void* workspace, size_t workspace_bytes, float* out2, float* z_planes, int init,
                      void* stream) {
    return model_forward_impl(m, B, T, x, mel, workspace, workspace_bytes, out2, z_planes, init ? 1 : 0, nullptr, nullptr,
                              stream);
}
int fwn_model_forward_init(const fwn_model_desc* m, int64_t B, int64_t T, const float* x, const float* mel,
                           void* workspace, size_t workspace_bytes, float* out2, float* z_planes,
                           fwn_reduce_fn reduce, void* user, void* stream) {
    return model_forward_impl(m, B, T, x, mel, workspace, workspace_bytes, out2, z_planes, 2, reduce, user, stream);
}

int fwn_model_reverse(const fwn_model_desc* m, int64_t B, int64_t T, const float* z, const float* mel,
                      void* workspace, size_t workspace_bytes, float* x_out, void* stream) {
    int rc = check_model(m, B, T);
    if (rc) return rc;
    REQUIRE(z && mel && workspace && x_out, "fwn_model_reverse: null pointer");
    REQUIRE((((uintptr_t)workspace) & 255) == 0, "workspace must be 256-byte aligned");
    REQUIRE(((m->n_block * m->n_flow) & 1) == 0,
            "reverse with odd n_block*n_flow ends in swapped channel order (model.py:199,254); unsupported");
    const Carve c = carve(m, B, T);
    if (workspace_bytes < c.total)
        return fail(FWN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, c.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t half = m->num_mels / 2;
    const size_t cplane_bytes = (size_t)B * T * half * 2;
    const size_t plane_elems = (size_t)B * T / 2;
    float* planes = (float*)(ws + c.planes);

    run_upsample(m, B, T, mel, ws, c, st);
    fwn_launch_split(z, B, T, planes, st);   // the n_block pre-squeezes of model.py:374-392 are index math
    if (c.sync_bytes && hipMemsetAsync(ws + c.sync, 0, c.sync_bytes, st) != hipSuccess)
        return fail(FWN_ERR_HIP, "fwn_model_reverse: hipMemsetAsync failed");
    int p = 0;
    Planes pl{{planes, planes + plane_elems}, (float*)(ws + c.plane3), {planes, planes + plane_elems}};
    for (int i = m->n_block - 1; i >= 0; --i) {
        if (record_block_event(m, m->n_block - 1 - i, st)) return fail(FWN_ERR_HIP, "fwn_model_reverse: hipEventRecord failed");
        const int64_t M = B * T / ((int64_t)2 << i);
        const bool hoist = hoist_cond(m, M, m->flows[i * m->n_flow].cin);
        if (hoist) {
            rc = check_block_contiguity(m, i);
            if (rc) return rc;
            // parity seen by flow j: p is flipped before each flow, visiting j = n_flow-1 .. 0
            int par[2] = {0, 0};
            int pc = p;
            for (int j = m->n_flow - 1; j >= 0; --j) {
                pc ^= 1;
                if (j < 2) par[j] = pc;
            }
            run_cond_groups(m, i, M, par, ws, c, B, T, st);
        }
        void* hA = ws + c.h0;
        void* hB = ws + c.h1;
        int have_h0 = 0;
        for (int j = m->n_flow - 1; j >= 0; --j) {
            p ^= 1;   // change_order first (model.py:199)
            const fwn_flow_desc* d = &m->flows[i * m->n_flow + j];
            const void* ca = hoist ? nullptr : (const void*)(ws + c.cplanes + (size_t)p * cplane_bytes);
            const float* P = hoist ? (const float*)(ws + c.P) + (size_t)j * m->n_layer * M * 512 : nullptr;
            unsigned* sync = persist_block(m, M, i) ? (unsigned*)(ws + c.sync + (size_t)(i * m->n_flow + j) * c.sync_stride) : nullptr;
            FlowChain ch = flow_chain(m, d, j > 0 ? d - 1 : nullptr, (int)M, false, pl.spare, have_h0, sync != nullptr);
            rc = flow_run_impl(d, B, T, pl.at[p], pl.at[p ^ 1], ca, hA, hB, ws + c.o, P, nullptr, 1, 0, nullptr, nullptr, nullptr,
                               m->gate_fp8 ? ws + c.h8a : nullptr, m->gate_fp8 ? ws + c.h8b : nullptr, &ch, stream, sync);
            if (rc) return rc;
            planes_after_flow(pl, p, ch.xb_out != nullptr);
            have_h0 = ch.h0_next != nullptr;
            if (have_h0 && ch.h0_next != hA) { void* t = hA; hA = hB; hB = t; }
        }
    }
    if (record_block_event(m, m->n_block, st)) return fail(FWN_ERR_HIP, "fwn_model_reverse: hipEventRecord failed");
    if (planes_go_home(pl, plane_elems * 4, st)) return fail(FWN_ERR_HIP, "fwn_model_reverse: plane copy failed");
    fwn_launch_merge(planes, B, T, x_out, st);
    return check_launch("fwn_model_reverse");
}

}  // extern "C"
