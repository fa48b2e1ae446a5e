/* Plain-C consumer of the C ABI (include/fwn.h): no Python, no torch.
 *   gcc -std=c99 -I include tests/c/cabi_smoke.c -o cabi_smoke -ldl
 *   ./cabi_smoke path/to/libfwn.so            -> argument validation only (no GPU needed)
 *   ./cabi_smoke path/to/libfwn.so gpu        -> also x -> planes -> x round trip and a mel frame on the GPU
 * The HIP runtime is loaded with dlopen so the program also builds on a machine without ROCm headers. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fwn.h"

#define CHECK(c, msg) do { if (!(c)) { fprintf(stderr, "FAIL: %s (line %d)\n", msg, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libfwn.so [gpu]\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    int (*version)(void) = (int (*)(void))dlsym(lib, "fwn_version");
    const char* (*last_error)(void) = (const char* (*)(void))dlsym(lib, "fwn_last_error");
    int (*split)(const float*, int64_t, int64_t, float*, void*) = (int (*)(const float*, int64_t, int64_t, float*, void*))dlsym(lib, "fwn_split_planes");
    int (*merge)(const float*, int64_t, int64_t, float*, void*) = (int (*)(const float*, int64_t, int64_t, float*, void*))dlsym(lib, "fwn_merge_planes");
    size_t (*ws_bytes)(const fwn_model_desc*, int64_t, int64_t) = (size_t (*)(const fwn_model_desc*, int64_t, int64_t))dlsym(lib, "fwn_workspace_bytes");
    CHECK(version && last_error && split && merge && ws_bytes, "symbols missing");
    CHECK(version() == FWN_VERSION, "version mismatch between header and library");
    /* errors come back as codes + a thread-local message, never as a crash */
    CHECK(split(NULL, 1, 8, NULL, NULL) < 0 && strstr(last_error(), "fwn_split_planes"), "null pointers must be rejected");
    CHECK(split((const float*)16, 1, 7, (float*)16, NULL) < 0, "odd T must be rejected");
    fwn_model_desc md;
    memset(&md, 0, sizeof md);
    CHECK(ws_bytes(&md, 1, 256) == 0 && last_error()[0], "an empty model descriptor must be rejected");
    printf("C ABI ok: version %d, validation messages e.g. \"%s\"\n", version(), last_error());
    if (argc < 3) return 0;

    void* hip = dlopen("libamdhip64.so", RTLD_NOW);
    CHECK(hip, "libamdhip64.so not found");
    int (*hipMalloc)(void**, size_t) = (int (*)(void**, size_t))dlsym(hip, "hipMalloc");
    int (*hipMemcpy)(void*, const void*, size_t, int) = (int (*)(void*, const void*, size_t, int))dlsym(hip, "hipMemcpy");
    int (*hipDeviceSynchronize)(void) = (int (*)(void))dlsym(hip, "hipDeviceSynchronize");
    CHECK(hipMalloc && hipMemcpy && hipDeviceSynchronize, "hip symbols missing");
    const int64_t B = 3, T = 4096;
    const size_t n = (size_t)(B * T);
    float* h = (float*)malloc(n * sizeof(float));
    float* back = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) h[i] = sinf(0.001f * (float)i);
    void *dx, *dp, *dy;
    CHECK(!hipMalloc(&dx, n * 4) && !hipMalloc(&dp, n * 4) && !hipMalloc(&dy, n * 4), "hipMalloc");
    CHECK(!hipMemcpy(dx, h, n * 4, 1 /* HostToDevice */), "H2D");
    CHECK(split((const float*)dx, B, T, (float*)dp, NULL) == 0, last_error());
    CHECK(merge((const float*)dp, B, T, (float*)dy, NULL) == 0, last_error());
    CHECK(!hipDeviceSynchronize() && !hipMemcpy(back, dy, n * 4, 2 /* DeviceToHost */), "D2H");
    CHECK(memcmp(h, back, n * 4) == 0, "split -> merge must be the identity, bit for bit");
    printf("GPU round trip ok (%lld samples)\n", (long long)n);
    return 0;
}
