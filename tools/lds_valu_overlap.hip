// Measurement tool (not part of the product): can the LDS exchanges of the transforms overlap the FP64 butterflies of the
// SAME CU at all?  tools/ntt_bench.hip found "butterflies only" + "exchanges only" = "full transform" to the nanosecond
// (1.59 + 0.84 = 2.42 us), in every arrangement tried — but always with all eight waves of the workgroup in the same
// phase.  Here the two kinds of work are given to DIFFERENT waves of a SIMD: waves 0-3 (one per SIMD) run only the FP64
// instruction stream of two transforms per iteration, waves 4-7 only the LDS traffic of two transforms per iteration —
// the same totals per CU as eight waves doing one transform each.  If the hardware can overlap them the iteration costs
// max(butterflies, exchanges); if they exclude each other it costs the sum.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../fhe-ram_amd/csrc -o lds_valu_overlap lds_valu_overlap.hip
#include "ntt_dev.hpp"
#include <cstdio>
#include <vector>
using namespace fk;

// the FP64 stream of one inverse transform on this thread's 8 values (no LDS): 4 passes + the reductions
__device__ __forceinline__ void valu_transform(double (&x)[1][E], const TwPass& t) {
#pragma unroll
    for (int k = 0; k < E; k++) x[0][k] = reduce(x[0][k]);
    inv_pass<3>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
    inv_pass<2>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
    inv_pass<1>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
    inv_pass<0>(x[0], t);
#pragma unroll
    for (int k = 0; k < E; k++) x[0][k] = reduce(x[0][k]);
}
// the LDS traffic of one inverse transform: three exchanges (8 writes + 8 reads of 8 bytes each), wave local: every wave
// stays inside its own region, no workgroup barrier (the question is bandwidth and overlap, not synchronisation)
__device__ __forceinline__ void lds_transform(double (&x)[1][E], double* data, int tid) {
    exchange_inv<2, 1>(x, data, tid);
    exchange_inv<1, 1>(x, data, tid);
    // exchange 0 crosses waves in the real transform; here its traffic pattern inside the wave's own region (same bytes)
#pragma unroll
    for (int k = 0; k < E; k++) data[lay<1>(pat<2>(tid, k))] = x[0][k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < E; k++) x[0][k] = data[lay<1>(pat<1>(tid, k))];
}
// MODE 0: every wave: butterflies then exchanges (one transform per iteration)     -> the lock-step baseline
// MODE 1: every wave: butterflies only        MODE 2: every wave: exchanges only
// MODE 3: waves 0-3: two transforms' butterflies per iteration; waves 4-7: two transforms' exchanges per iteration
// MODE 4: as 3 with the roles by wave parity (waves of one SIMD: w and w + 4 -> parity puts both kinds on every SIMD too)
template <int MODE>
__global__ __launch_bounds__(T, T / 256) void k_overlap(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    TwPass t;
    inv_twiddles<3>(t, tw, tid);
    double x[1][E];
    for (int k = 0; k < E; k++) x[0][k] = (double)(tid * 8 + k);
    const int wave = tid >> 6;
    const bool valu_role = (MODE == 3) ? (wave < 4) : ((wave & 1) == 0);
    for (int r = 0; r < reps; r++) {
        if constexpr (MODE == 0) { valu_transform(x, t); lds_transform(x, data, tid); }
        if constexpr (MODE == 1) valu_transform(x, t);
        if constexpr (MODE == 2) lds_transform(x, data, tid);
        if constexpr (MODE == 3 || MODE == 4) {
            if (valu_role) { valu_transform(x, t); valu_transform(x, t); }
            else { lds_transform(x, data, tid); lds_transform(x, data, tid); }
        }
    }
    double s = 0;
    for (int k = 0; k < E; k++) s += x[0][k];
    sink[blockIdx.x * T + tid] = s;
}

template <int MODE> void run(const char* name, const double* tw, double* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_overlap<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    const int reps = 400, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) k_overlap<MODE><<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_overlap<MODE><<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-78s %7.3f us per transform-equivalent per CU (%s)\n", name, ms * 1e3 / (10.0 * reps), hipGetErrorString(hipGetLastError()));
}

int main() {
    std::vector<double> h(2 * N);
    for (int i = 0; i < 2 * N; i++) h[i] = (double)((i * 2654435761u) % 1000003);
    double *tw, *sink;
    hipMalloc(&tw, 2 * N * sizeof(double)); hipMalloc(&sink, 512 * T * sizeof(double));
    hipMemcpy(tw, h.data(), 2 * N * sizeof(double), hipMemcpyHostToDevice);
    run<0>("every wave: butterflies, then exchanges (one transform per iteration)", tw, sink);
    run<1>("every wave: butterflies only", tw, sink);
    run<2>("every wave: exchanges only", tw, sink);
    run<3>("waves 0-3 butterflies of two transforms | waves 4-7 exchanges of two transforms", tw, sink);
    run<4>("even waves butterflies of two transforms | odd waves exchanges of two transforms", tw, sink);
    hipFree(tw); hipFree(sink);
    return 0;
}
