// Translation unit of the register-streamed gate (gate_rs.h): the kernel's instantiations, the fragment-stream packing, the
// dispatch rule and the clock-stamped diagnostic launch.  Its own unit since round 6 (it was half of flow_kernels.hip's four
// minutes of hipcc; the two now build side by side).
#include "gate_rs.h"
#include "fwn_internal.h"

#define FWN_RS_MAXDIL 3          // = FWN_HALO_MAXDIL (gate_halo.h): the dilations whose halo fits the activation slots

// ---- one instantiation per number of conditioning k-steps: cin = 80 (block 0), 160 (block 1), 320 (block 2), 640 (block 3) at
// num_mels = 80; the last one in the 64- and 128-row forms only (round 6)
#define FWN_RS_CASES(X) X(5) X(10) X(20) X(40)
static bool rs_has_mt8(int nkc) { return nkc != 40; }
long fwn_gate_stream_size(int cin) {
    const int nkc = (cin + 15) / 16;
#define X(n) if (nkc == n) return 16L * RsPlan<n>::NK * 1024;
    FWN_RS_CASES(X)
#undef X
    return 0;
}
// 256-row tiles from 24 576 rows on (x 2 channel halves: the chip is full, as for the 256 x 256 tap-sharing tile); 128-row
// tiles from 12 288 rows on (block 2 of the 8-clip pass, block 1 of a 4-clip pass: 256-row tiles would leave half the CUs empty);
// 64-row tiles from 6 144 rows on (round 6: block 3 of the 8-clip pass, block 0 of one clip - 252 workgroups of 64 rows where
// the 128 x 128 tap-sharing tile ran two 67 KB workgroups per CU at 0.22 of the MFMA peak)
int fwn_gate_stream_min_rows() { return FWN_TUNE(FWN_RS_MIN_ROWS, 6144); }
static int gate_stream_mt(int M, int nkc) {
    if (M >= FWN_TUNE(FWN_RS_MIN_ROWS256, 24576) && rs_has_mt8(nkc)) return 8;
    return M >= FWN_TUNE(FWN_RS_MIN_ROWS128, 12288) ? 4 : 2;
}
int fwn_gate_stream_ok(int M, int Ti, int dil, int cin, bool fused_cond, bool aux) {
    // a tile may cross one clip edge only; dilations whose halo fits the slot
    return fused_cond && !aux && dil <= FWN_RS_MAXDIL && Ti >= 256 && M >= fwn_gate_stream_min_rows() &&
           fwn_gate_stream_size(cin) != 0 && cin % 8 == 0;
}
void fwn_launch_gate_stream_pack(const void* Wd, const void* Wc, int cin, int kcpad, void* out, hipStream_t st) {
    const int nkc = (cin + 15) / 16;
#define X(n) if (nkc == n) hipLaunchKernelGGL(gate_stream_pack_kernel<n>, dim3(128), dim3(256), 0, st, (const bf16*)Wd, (const bf16*)Wc, kcpad, (bf16*)out);
    FWN_RS_CASES(X)
#undef X
}

// the caller has checked fwn_gate_stream_ok
void fwn_launch_gate_rs(const void* h, const void* ca, const void* Wgs, const float* bias, void* o, int M, int Ti, int dil, int cin,
                        hipStream_t st) {
    GateRsArgs a{(const bf16*)h, (const bf16*)ca, (const bf16*)Wgs, bias, (bf16*)o, M, Ti, dil, cin};
    const int nkc = (cin + 15) / 16;
    // gate_rs_kernel<.., PERSIST = true>: one workgroup per CU loops over its tiles, the next tile's first items and
    // weights issued under the tail of the current one (-10 % cycles per two-tile workgroup at block 0,
    // tools/bench_gate_rs.hip).  Its race of the first half of round 4 is root-caused and fixed (a register copy hipcc
    // placed in front of a branch-dependent asm wait: gate_rs.h; tools/check_async_loads.py); it
    // soaks clean inside overlapped passes at 8 / 16 / 32 clips.  Used where a workgroup gets three tiles or more (from
    // 13 clips of 16128 samples on at block 0: 32 clips -1.9 % on the one-stream pair, -0.8 % on the overlapped step);
    // at the bench's 8 clips (two tiles per workgroup) the overlapped step is 1.5 % slower with it: one tile per
    // workgroup there.  fwn_set_option("rs_persist", 0 / 1) forces either form (round 4 read an environment variable here, per launch).
    const int ncu = fwn_device_cus() & ~1;
    const int mt = gate_stream_mt(M, nkc);
    const int pe = g_fwn_opt_rs_persist;            // fwn_set_option("rs_persist", ..): -1 auto
    const bool persist = mt == 8 && (pe >= 0 ? pe == 1 : ((M + 255) / 256) * 2 >= 3 * ncu);
    const int ntiles = ((M + 32 * mt - 1) / (32 * mt)) * 2, grid = persist && ntiles > ncu ? ncu : ntiles;
    // 64-row tiles: a k-step is two MFMAs per wave - twelve ring stages keep the weight stream ahead of it (six at 128 / 256 rows)
#define X(n)                                                                                                                   \
    if (nkc == n) {                                                                                                            \
        if constexpr (n != 40) {                                                                                               \
            if (mt == 8 && !persist) hipLaunchKernelGGL((gate_rs_kernel<n, 8, false>), dim3(grid), dim3(512), 0, st, a, ntiles);  \
            else if (mt == 8) hipLaunchKernelGGL((gate_rs_kernel<n, 8, true>), dim3(grid), dim3(512), 0, st, a, ntiles);          \
        }                                                                                                                      \
        if (mt == 4) hipLaunchKernelGGL((gate_rs_kernel<n, 4, false>), dim3(grid), dim3(512), 0, st, a, ntiles);                  \
        else if (mt == 2) hipLaunchKernelGGL((gate_rs_kernel<n, 2, false, 12>), dim3(grid), dim3(512), 0, st, a, ntiles);         \
    }
    FWN_RS_CASES(X)
#undef X
}

// Diagnostic launch of the dominant kernel with two clock stamps per wave (fwn_gate_clock): the 256-row register-streamed
// gate, same code otherwise.  Returns the number of workgroups (8 stamp records each) or 0 if the shape has no such kernel.
int fwn_launch_gate_clock(const void* h, const void* ca, const void* Wgs, const float* bias, void* o, int M, int Ti, int dil, int cin,
                          unsigned long long* clk, hipStream_t st) {
    const int nkc = (cin + 15) / 16;
    if (!Wgs || !fwn_gate_stream_ok(M, Ti, dil, cin, ca != nullptr, false) || gate_stream_mt(M, nkc) != 8) return 0;
    GateRsArgs a{(const bf16*)h, (const bf16*)ca, (const bf16*)Wgs, bias, (bf16*)o, M, Ti, dil, cin};
    a.clk = clk;
    const int ntiles = ((M + 255) / 256) * 2;
#define X(n) if constexpr (n != 40) { if (nkc == n) hipLaunchKernelGGL((gate_rs_kernel<n, 8, false, FWN_RS_R, true>), dim3(ntiles), dim3(512), 0, st, a, ntiles); }
    FWN_RS_CASES(X)
#undef X
    return ntiles;
}
