// Measurement tool (not part of the product): inverse transforms per CU when TWO independent workgroups share a CU
// (one exchange buffer each: twiddles + 1 buffer = 67-68 KB of LDS) against one workgroup per CU, for 512-thread
// workgroups with 8 coefficients per thread (LOGE=3) and 256-thread workgroups with 16 (LOGE=4, build with -DFK_LOGE=4).
// Independent workgroups do not meet at barriers, so the LDS exchanges of one can overlap the butterflies of the other.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Ifhe-ram_amd/csrc [-DFK_LOGE=4] -o tools/ntt_occupancy tools/ntt_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ntt_dev.hpp"
using namespace fk;

template <int B>
__global__ __launch_bounds__(T) void k_inv(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    double x[B][E];
    for (int b = 0; b < B; b++)
        for (int k = 0; k < E; k++) x[b][k] = (double)(tid * 8 + k + b);
    for (int r = 0; r < reps; r++) ntt_inv<B>(x, tw, data, tid);
    double s = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + tid] = s;
}
template <int B> void run(const double* tw, double* sink, int blocks, size_t lds, const char* what) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv<B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_inv<B>, T, lds);
    const int reps = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) k_inv<B><<<blocks, T, lds>>>(tw, sink, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_inv<B><<<blocks, T, lds>>>(tw, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("E=%2d T=%3d B=%d %-34s blocks=%4d (resident per CU: %d): %7.3f us per transform per CU, %7.3f us per transform of one workgroup\n",
           E, T, B, what, blocks, per_cu, ms * 1e3 / (10.0 * reps * B) / (blocks / 256.0), ms * 1e3 / (10.0 * reps * B) / ((blocks / 256 + per_cu - 1) / per_cu));
}
int main() {
    std::vector<double> h(2 * N);
    for (int i = 0; i < 2 * N; i++) h[i] = (double)((i * 2654435761u) % 1000003);
    double *tw, *sink;
    hipMalloc(&tw, 2 * N * sizeof(double)); hipMalloc(&sink, 1024 * T * sizeof(double));
    hipMemcpy(tw, h.data(), 2 * N * sizeof(double), hipMemcpyHostToDevice);
    const size_t one = (size_t)(LDS_TW + LDS_DATA) * sizeof(double), big = 140 * 1024;
    run<1>(tw, sink, 256, big, "1 workgroup per CU (140 KB)");
    run<1>(tw, sink, 512, big, "1 workgroup per CU (140 KB)");
    run<1>(tw, sink, 256, one, "up to 2 per CU (68 KB), 1 each");
    run<1>(tw, sink, 512, one, "up to 2 per CU (68 KB), 2 each");
    run<1>(tw, sink, 1024, one, "up to 2 per CU (68 KB), 4 each");
    return 0;
}
