// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the FHE-RAM kernels use.
// MI355X_MICROARCH.md calibrates the x2 correction of FETCH_SIZE only for 16-B/lane streaming reads; the RAM
// rows are int32 and are read 4 B per lane (512 threads x 8 loads of one 16 KiB limb polynomial per
// workgroup), so this tool streams a buffer of KNOWN size with exactly that pattern, and with 16 B/lane as
// the control, under  rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE.
//   hipcc -O3 --offload-arch=gfx950 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int T = 512, E = 8, POLY = T * E;   // one workgroup reads / writes one 4096-element polynomial

__global__ __launch_bounds__(T) void k_read4(const int32_t* __restrict__ in, int32_t* __restrict__ sink) {
    const int32_t* p = in + (size_t)blockIdx.x * POLY;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < E; k++) acc += p[threadIdx.x + T * k];
    if (acc == 0x7fffffff) sink[blockIdx.x] = acc;   // never true for the test pattern: no write traffic
}
__global__ __launch_bounds__(T) void k_read16(const int4* __restrict__ in, int32_t* __restrict__ sink) {
    const int4* p = in + (size_t)blockIdx.x * (POLY / 4) * 4;   // 4 polynomials' worth per workgroup
    int acc = 0;
#pragma unroll
    for (int k = 0; k < E; k++) { const int4 v = p[threadIdx.x + T * k]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x7fffffff) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(T) void k_write4(int32_t* __restrict__ out) {
    int32_t* p = out + (size_t)blockIdx.x * POLY;
#pragma unroll
    for (int k = 0; k < E; k++) p[threadIdx.x + T * k] = (int)threadIdx.x + k;
}

int main() {
    const size_t bytes = (size_t)96 << 20;   // 96 MiB, three distinct buffers so that no launch re-reads a cached one
    int32_t *a[3], *sink;
    for (auto& p : a) { if (hipMalloc(&p, bytes) != hipSuccess) return 1; hipMemset(p, 1, bytes); }
    hipMalloc(&sink, 1 << 20);
    hipDeviceSynchronize();
    const int wg4 = (int)(bytes / (POLY * 4)), wg16 = (int)(bytes / (POLY * 16));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_read4, dim3(wg4), dim3(T), 0, 0, a[i], sink);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_read16, dim3(wg16), dim3(T), 0, 0, (const int4*)a[i], sink);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_write4, dim3(wg4), dim3(T), 0, 0, a[i]);
    hipDeviceSynchronize();
    printf("fetch_calib: %zu bytes per launch; k_read4 x3, k_read16 x3, k_write4 x3\n", bytes);
    return 0;
}
