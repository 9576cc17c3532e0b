// Measurement tool (not part of the product): what does a CU-masked stream (hipExtStreamCreateWithCUMask) do on an 8-XCD part?
//  1. which XCDs / CUs the workgroups of a launch land on, for several masks;
// (the question behind it: could the write's side chain run on some XCDs while the latency-bound end of read_prepare_write keeps
// the others?  Answer on this part: no - a launch's workgroups go round-robin to ALL 8 XCDs whatever the mask says; a mask only
// thins the CUs inside every XCD, and the single-launch trace chain needs 24 CUs of ONE XCD per ciphertext.)
//   hipcc -O3 --offload-arch=gfx950 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(unsigned* out, int spin_us) {
    extern __shared__ double lds[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; lds[0] = 1.0; }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin_us * 100) { __builtin_amdgcn_s_sleep(4); }
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("CUs %d\n", cus);
    unsigned* d; CK(hipMalloc(&d, 4096 * 8));
    std::vector<unsigned> h(4096 * 2);
    struct M { const char* name; uint32_t w[8]; };
    M masks[] = {
        {"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
        {"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"bits = 0..3 mod 8", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}},
        {"bits = 4..7 mod 8", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}},
        {"words 0,2,4,6", {~0u, 0, ~0u, 0, ~0u, 0, ~0u, 0}},
    };
    CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    for (auto& m : masks) {
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, m.w));
        const int G = 512;
        CK(hipMemsetAsync(d, 0xff, G * 8, st));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, st);
        hipLaunchKernelGGL(k_where, dim3(G), dim3(512), 140 * 1024, st, d, 50);   // one workgroup per CU (LDS), 50 us each
        hipEventRecord(b, st);
        CK(hipStreamSynchronize(st));
        float ms; hipEventElapsedTime(&ms, a, b);
        CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0};
        std::set<unsigned> places;
        for (int i = 0; i < G; i++) { per_xcc[h[2 * i] & 15]++; places.insert((h[2 * i] << 16) | ((h[2 * i + 1] >> 8) & 0xff)); }
        printf("mask %-18s: %d workgroups of 50 us, one per CU: %.0f us (= %.1f rounds); per XCC:", m.name, G, ms * 1e3, ms * 1e3 / 50);
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("; distinct (xcc, se/sh/cu) places %zu; first blocks' XCC:", places.size());
        for (int i = 0; i < 16; i++) printf(" %u", h[2 * i]);
        printf("\n");
        hipStreamDestroy(st);
    }
    return 0;
}
