// Translation unit of the register-streamed tail (tail_rs.h): the kernel's instantiations, the fragment-stream packing and the
// dispatch rule.  Its own unit so that it builds beside flow_kernels.hip (four minutes of hipcc).
#include "tail_rs.h"
#include "fwn_internal.h"

// bytes of the fragment stream of one flow (Wskip | Wfinal), 0: no kernel for this layer count
long fwn_tail_stream_size(int L) { return L == 2 ? 8L * 48 * 1024 : 0; }

#ifndef FWN_TRS_MIN_ROWS
#define FWN_TRS_MIN_ROWS 1008        // fewer rows: the N-split ring GEMMs / the one-launch flow (flow_kernels.hip, flow_persist.h)
#endif
// Tile height: the larger the better down to a few dozen workgroups - a workgroup streams all 384 KB of Wskip | Wfinal from L2
// whatever its rows (at ~64 B / clock / CU that alone is 6 k cycles), so smaller tiles buy occupancy with L2 traffic and lose
// (tools/bench_tail_rs.hip, us per launch: 16 128 rows 17.1 as 126 workgroups of 128 rows / 19.0 as 252 of 64; 8 064 rows
// 11.8 as 126 of 64 / 14.6 as 252 of 32).
#ifndef FWN_TRS_ROWS128
#define FWN_TRS_ROWS128 12288        // from here on 128-row workgroups (one per CU: 140 KB of LDS)
#endif
#ifndef FWN_TRS_ROWS64
#define FWN_TRS_ROWS64 6144          // 64-row workgroups down to here; 32-row ones below (one clip's blocks 1 - 3, blocks 4 / 5 of the
                                     // 8-clip pass: a few dozen workgroups - 9.0 - 9.5 us per launch against 10.5 - 11 with 64 rows,
                                     // where the N-split tail took three launches)
#endif
// From 49 152 rows on (block 0 of the 8-clip pass: 504 workgroups of 128 rows = two rounds on 256 CUs) the 256-row register-chained
// tail_kernel (252 workgroups, one round, weights read once per 256 rows) is still ahead in situ: 38 against 41 us per launch
// (rocprofv3 per-block tables of the same box), although the stand-alone harness has this kernel ahead (42.8 against 46).
#ifndef FWN_TRS_MAX_ROWS
#define FWN_TRS_MAX_ROWS 49152
#endif
int fwn_tail_stream_min_rows() { return FWN_TUNE(FWN_TRS_MIN_ROWS, FWN_TRS_MIN_ROWS); }

// 32-row time tiles per workgroup of the register-streamed tail at this shape, 0: it does not serve the shape
int fwn_tail_rs_mt(int M, int L, int Ch, int npt, bool have_stream) {
    if (!have_stream || L != 2 || npt != 1 || Ch > 32 || M < fwn_tail_stream_min_rows() || M >= FWN_TUNE(FWN_TRS_MAX_ROWS, FWN_TRS_MAX_ROWS) ||
        !FWN_TUNE(FWN_TRS, 1)) return 0;
    if (M >= FWN_TUNE(FWN_TRS_ROWS128, FWN_TRS_ROWS128)) return 4;
    if (M >= FWN_TUNE(FWN_TRS_ROWS64, FWN_TRS_ROWS64)) return 2;
    return 1;
}

void fwn_launch_tail_stream_pack(const void* Ws, const void* Wf, void* out, hipStream_t st) {
    hipLaunchKernelGGL(tail_stream_pack_kernel, dim3(96), dim3(256), 0, st, (const bf16*)Ws, (const bf16*)Wf, (bf16*)out, (const TailStreamJob*)nullptr);
}
void fwn_launch_tail_stream_pack_jobs(const void* jobs, int njobs, hipStream_t st) {
    hipLaunchKernelGGL(tail_stream_pack_kernel, dim3(24, njobs), dim3(256), 0, st, (const bf16*)nullptr, (const bf16*)nullptr, (bf16*)nullptr,
                       (const TailStreamJob*)jobs);
}

void fwn_launch_tail_rs(const TailArgs& a, const void* Wts, int mt, hipStream_t st) {
    const bool front = a.h0_next != nullptr, save = a.save_s && a.save_u && a.save_z;
    const int rows = 32 * mt - (a.overlap ? 2 : 0);
    const dim3 grid((a.M + rows - 1) / rows), block(512);
#define TRS_LAUNCH(MT)                                                                                                       \
    do {                                                                                                                     \
        if (front) hipLaunchKernelGGL((tail_rs_kernel<MT, true, false>), grid, block, 0, st, a, (const bf16*)Wts);           \
        else if (save) hipLaunchKernelGGL((tail_rs_kernel<MT, false, true>), grid, block, 0, st, a, (const bf16*)Wts);       \
        else hipLaunchKernelGGL((tail_rs_kernel<MT, false, false>), grid, block, 0, st, a, (const bf16*)Wts);                \
    } while (0)
    if (mt == 4) TRS_LAUNCH(4);
    else if (mt == 2) TRS_LAUNCH(2);
    else TRS_LAUNCH(1);
#undef TRS_LAUNCH
}
