// Translation unit of the register-streamed conditioning projection (cond_rs.h): the kernel, its fragment-stream packing and
// the dispatch rule (rows, split count).
#include "cond_rs.h"
#include "fwn_internal.h"

// bytes of ONE matrix's stream ([8 waves][kcpad / 16][2][64][8] bf16 = 512 kcpad 2: the size of the matrix)
long fwn_cond_stream_size(int kcpad) { return kcpad > 0 && kcpad % 64 == 0 ? 512L * kcpad * 2 : 0; }
// the kernel serves whole 128-row tiles: from 384 rows on (blocks 4 - 7 of the 8-clip pass, 4 - 6 of a 4-clip one); below, the
// ring tiles with their 64-row form and deeper K splits stay
int fwn_cond_stream_min_rows() { return FWN_TUNE(FWN_CRS_MIN_ROWS, 384); }
bool fwn_cond_rs_ok(int M, int cin, int kcpad, bool have_stream) {
    return have_stream && FWN_TUNE(FWN_CRS, 1) && M >= fwn_cond_stream_min_rows() && kcpad % 64 == 0 && kcpad >= 256 && cin % 8 == 0 &&
           (long)M * cin < (1L << 30);
}
// Tile height (32 MT rows) and K splits for nz matrices per launch: the pair with the least estimated time.  Measured
// (tools/probe/cond_bench.py): a workgroup takes 1.375 us per 64-wide chunk at 128 rows (0.85 of a CU's matrix pipe) - in
// proportion for 96 rows (64-row tiles are built but not planned: a weight fragment then feeds four MFMAs only, the stream runs
// at the CU's load rate and, with the weights cold, block 6 of the 8-clip pass took 101 us in the pass against 74 alone) - plus
// ~11 us per launch that do not depend on K (first operands, and all workgroups of a round writing their
// P tiles at the same time); a round is as long as its workgroups whether or not it fills the chip; a split beyond the first
// costs a partial write and the in-order sum (fwn_launch_cond_reduce) reads it back: (ns + 1) P bytes at ~4 TB/s.
static double crs_cost(int M, int nz, int kcpad, int mt, int ns, int ncu) {
    const int nch = kcpad / 64, base = ((M + 32 * mt - 1) / (32 * mt)) * nz;
    const int rounds = (base * ns + ncu - 1) / ncu, per = (nch + ns - 1) / ns;
    const double tch = 1.375 * mt / 4.0 * (mt == 2 ? 1.3 : 1.0);
    return rounds * (per * tch + 4.0) + 7.0 + (ns > 1 ? (ns + 1) * ((double)nz * M * 512 * 4) / 4.0e6 : 0.0);      // us
}
void fwn_cond_rs_plan(int M, int nz, int kcpad, int* mt_out, int* ns_out) {
    const int fmt = FWN_TUNE(FWN_CRS_MT, 0), fns = FWN_TUNE(FWN_CRS_NSPLIT, 0);
    const int nch = kcpad / 64, ncu = fwn_device_cus() > 0 ? fwn_device_cus() : 256;
    int bm = 4, bn = 1;
    double bc = 1e30;
    for (int mt = 4; mt >= (fmt == 2 ? 2 : 3); --mt) {
        if (fmt && mt != fmt) continue;
        for (int ns = 1; ns <= 8 && (ns == 1 || ns * 8 <= nch); ++ns) {
            if (fns && ns != (fns < nch ? fns : nch)) continue;
            const double c = crs_cost(M, nz, kcpad, mt, ns, ncu);
            if (c < bc - 1e-9) { bc = c; bm = mt; bn = ns; }
        }
    }
    *mt_out = bm;
    *ns_out = bn;
}
// ... and the model-level calls use it where that estimate is under the ring tiles' (12 us + 1.11 us per GFLOP fits their
// launches at 1 / 2 / 4 / 8 clips: 21 / 30 / 47 / 82 us for 7.9 / 15.9 / 31.7 / 63.4 GFLOP) and K >= 640 (five chunks of K = 320
// are all prologue): profiles/r06_notes.md section 8
bool fwn_cond_rs_wanted(int M, int cin, int kcpad, int nz, bool have_stream) {
    if (!fwn_cond_rs_ok(M, cin, kcpad, have_stream) || kcpad < FWN_TUNE(FWN_CRS_MIN_K, 640)) return false;
    int mt, ns;
    fwn_cond_rs_plan(M, nz, kcpad, &mt, &ns);
    const int ncu = fwn_device_cus() > 0 ? fwn_device_cus() : 256;
    return crs_cost(M, nz, kcpad, mt, ns, ncu) < 12.0 + 1.11e-9 * (2.0 * M * cin * 512.0 * nz) || FWN_TUNE(FWN_CRS, 1) == 2;
}
int fwn_cond_rs_nsplit(int M, int nz, int kcpad) {
    int mt, ns;
    fwn_cond_rs_plan(M, nz, kcpad, &mt, &ns);
    return ns;
}
void fwn_launch_cond_stream_pack(const void* Wc_base, long w_stride, int kcpad, int nz, void* out, hipStream_t st) {
    hipLaunchKernelGGL(cond_stream_pack_kernel, dim3(1024), dim3(256), 0, st, (const bf16*)Wc_base, w_stride, kcpad, nz, (bf16*)out);
}
// all nz = nflow * L matrices of a block in one launch (ca_odd: the plane of the flows with an odd index, or NULL: all read ca)
void fwn_launch_cond_rs(const void* ca, const void* ca_odd, const void* Ws, float* P, int nz, int L, int M, int cin, int kcpad,
                        float* part, long part_stride, int nsplit, hipStream_t st) {
    int mt, ns_plan;
    fwn_cond_rs_plan(M, nz, kcpad, &mt, &ns_plan);           // (the tile height of the plan; the caller's split count)
    const int nrt = (M + 32 * mt - 1) / (32 * mt);
    CondRsArgs a{(const bf16*)ca, (const bf16*)ca_odd, (const bf16*)Ws, P, part, part_stride, M, cin, kcpad, L, nsplit > 1 ? nsplit : 1, nrt};
    const dim3 grid(nrt * a.nsplit * nz);
    if (mt == 4) hipLaunchKernelGGL((cond_rs_kernel<4>), grid, dim3(512), 0, st, a);
    else if (mt == 3) hipLaunchKernelGGL((cond_rs_kernel<3>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((cond_rs_kernel<2>), grid, dim3(512), 0, st, a);
}
