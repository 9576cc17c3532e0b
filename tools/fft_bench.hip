// Measurement / self-check tool (not part of the product): the transforms of csrc/fft_dev.hpp.
//   1. correctness: forward transforms, pointwise products against prepared operands, inverse transforms == exact negacyclic
//      convolution (host, 128-bit integers), with the FP64 round-off before rounding reported;
//   2. steady-state cost of forward / inverse transforms on one CU (one 512-thread workgroup per CU, as in the evaluator).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../fhe-ram_amd/csrc fft_bench.hip -o fft_bench
#include "fft_dev.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace fk;

__device__ __forceinline__ void cmac(double (&acc)[E], const double (&x)[E], const double (&g)[E]) {
#pragma unroll
    for (int m = 0; m < 4; m++) {
        acc[2 * m] = __builtin_fma(x[2 * m], g[2 * m], acc[2 * m]);
        acc[2 * m] = __builtin_fma(-x[2 * m + 1], g[2 * m + 1], acc[2 * m]);
        acc[2 * m + 1] = __builtin_fma(x[2 * m], g[2 * m + 1], acc[2 * m + 1]);
        acc[2 * m + 1] = __builtin_fma(x[2 * m + 1], g[2 * m], acc[2 * m + 1]);
    }
}
// out0 = sum_r a_r * g_r,  out1 = sum_r a_r * g_{r^1}   (R terms each), raw doubles (not rounded)
template <int MODE>   // 0: two at a time, 1: one at a time
__global__ __launch_bounds__(T) void k_conv(const int* a, const int* g, double* out, const double* tw_g, int R) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double acc[2][E];
    for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) acc[b][k] = 0.0;
    for (int r = 0; r < R; r += 2) {
        double x[2][E], gg[2][E];
        for (int k = 0; k < E; k++) {
            x[0][k] = (double)a[(long)r * N + tid + T * k]; x[1][k] = (double)a[(long)(r + 1) * N + tid + T * k];
            gg[0][k] = (double)g[(long)r * N + tid + T * k]; gg[1][k] = (double)g[(long)(r + 1) * N + tid + T * k];
        }
        if (MODE == 1) {
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&x[0]), tw, data, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&x[1]), tw, data + LDS_DATA, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&gg[0]), tw, data, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&gg[1]), tw, data + LDS_DATA, tid);
        } else {
            ntt_fwd<2>(x, tw, data, tid);
            ntt_fwd<2>(gg, tw, data, tid);
        }
        for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) gg[b][k] *= (1.0 / NC);
        cmac(acc[0], x[0], gg[0]); cmac(acc[0], x[1], gg[1]);
        cmac(acc[1], x[0], gg[1]); cmac(acc[1], x[1], gg[0]);
    }
    if (MODE == 1) {
        double* const d0[1] = {data};
        double* const d1[1] = {data + LDS_DATA};
        fft_inv_skew<1, 1, false>(*reinterpret_cast<double(*)[1][E]>(&acc[0]), tw, d0, tid);
        fft_inv_skew<1, 1, false>(*reinterpret_cast<double(*)[1][E]>(&acc[1]), tw, d1, tid);
    } else { double* const d[2] = {data, data + LDS_DATA}; fft_inv_skew<2, 1, false>(acc, tw, d, tid); }
    for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) out[(long)b * N + tid + T * k] = acc[b][k];
}

// MODE: F1/F2/F3 forward 1..3, I1/I2/I3 inverse 1..3
template <int MODE>
__global__ __launch_bounds__(T, T / 256) void k_time(const double* tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double x[3][E];
    for (int b = 0; b < 3; b++) for (int k = 0; k < E; k++) x[b][k] = (double)((tid * 8 + k + b) & 1023);
    double* const d3[3] = {data, data + LDS_DATA, data + 2 * LDS_DATA};
    for (int r = 0; r < reps; r++) {
        if (MODE == 1) ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&x[0]), tw, data, tid);
        if (MODE == 2) ntt_fwd<2>(*reinterpret_cast<double(*)[2][E]>(&x[0]), tw, data, tid);
        if (MODE == 3) ntt_fwd<3>(x, tw, data, tid);
        if (MODE == 11) ntt_inv<1, true>(*reinterpret_cast<double(*)[1][E]>(&x[0]), tw, data, tid);
        if (MODE == 12) ntt_inv<2, true>(*reinterpret_cast<double(*)[2][E]>(&x[0]), tw, data, tid);
        if (MODE == 13) fft_inv_skew<3, 1>(x, tw, d3, tid);
        if (MODE == 22) ntt_inv2_loop(*reinterpret_cast<double(*)[2][E]>(&x[0]), tw, data, data + LDS_DATA, tid);
        for (int b = 0; b < 3; b++) for (int k = 0; k < E; k++) x[b][k] *= 0.001;
    }
    double s = 0;
    for (int b = 0; b < 3; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + threadIdx.x] = s;
}

// The same single transforms from TWO independent workgroups per CU (<= 128 registers, one exchange buffer each): what phase
// diversity between the waves of a SIMD would be worth (the evaluator's kernels cannot run this way: they need > 128 registers).
template <int MODE>
__global__ __launch_bounds__(T, 2) void k_time_occ2(const double* tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double x[1][E];
    for (int k = 0; k < E; k++) x[0][k] = (double)((tid * 8 + k) & 1023);
    for (int r = 0; r < reps; r++) {
        if (MODE == 1) ntt_fwd<1>(x, tw, data, tid);
        if (MODE == 11) ntt_inv<1, true>(x, tw, data, tid);
        for (int k = 0; k < E; k++) x[0][k] *= 0.001;
    }
    double s = 0;
    for (int k = 0; k < E; k++) s += x[0][k];
    sink[blockIdx.x * T + threadIdx.x] = s;
}

// Does the operand stream overlap a transform?  Per repetition: P operand polynomials (32 KB each, from a 768 KB key every workgroup
// shares: the XCD's L2) are REQUESTED, an inverse pair runs, the operands are consumed by multiply-accumulates.
//   WHAT 0: loads + transforms + MACs   1: loads + MACs (no transform)   2: transforms + MACs on stale registers (no loads)
template <int P, int WHAT>
__global__ __launch_bounds__(T, T / 256) void k_stream(const double* tw_g, double* sink, int reps, const double* key) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double x[2][E], acc[2][E];
    for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) { x[b][k] = (double)((tid * 8 + k + b) & 1023); acc[b][k] = 0.0; }
    double2 g[P][E / 2];
    for (int q = 0; q < P; q++) for (int kk = 0; kk < E / 2; kk++) { g[q][kk].x = 1e-3; g[q][kk].y = 2e-3; }
    int poly = 0;
    for (int r = 0; r < reps; r++) {
        if (WHAT != 2) {
#pragma unroll
            for (int q = 0; q < P; q++) {
                const double2* gp = reinterpret_cast<const double2*>(key + (long)poly * N);
#pragma unroll
                for (int kk = 0; kk < E / 2; kk++) g[q][kk] = gp[kk * T + tid];
                poly = (poly + 1) % 24;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (WHAT != 1) ntt_inv2_loop(x, tw, data, data + LDS_DATA, tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < P; q++)
#pragma unroll
            for (int kk = 0; kk < E / 2; kk++) {
                acc[q & 1][2 * kk] = __builtin_fma(x[q & 1][2 * kk], g[q][kk].x, acc[q & 1][2 * kk]);
                acc[q & 1][2 * kk] = __builtin_fma(-x[q & 1][2 * kk + 1], g[q][kk].y, acc[q & 1][2 * kk]);
                acc[q & 1][2 * kk + 1] = __builtin_fma(x[q & 1][2 * kk], g[q][kk].y, acc[q & 1][2 * kk + 1]);
                acc[q & 1][2 * kk + 1] = __builtin_fma(x[q & 1][2 * kk + 1], g[q][kk].x, acc[q & 1][2 * kk + 1]);
            }
        for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) x[b][k] = acc[b][k] * 1e-6;
    }
    double s_ = 0;
    for (int b = 0; b < 2; b++) for (int k = 0; k < E; k++) s_ += x[b][k];
    sink[blockIdx.x * T + threadIdx.x] = s_;
}

// The chain kernels' streamed form: ONE inverse transform per repetition with the multiply-accumulates of six operand polynomials between
// its phases (fft_inv1_hooked), each from a register set refilled at once with the polynomial W places further on.  W = in-flight window.
template <int W, bool LOADS>
__global__ __launch_bounds__(T, T / 256) void k_rolling(const double* tw_g, double* sink, int reps, const double* key) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double x[1][E], xs[E], accn[E];
    for (int k = 0; k < E; k++) { x[0][k] = (double)((tid * 8 + k) & 1023); xs[k] = x[0][k] * 1e-3; accn[k] = 0.0; }
    double2 w[W][E / 2];
    int poly = 0;
    auto request = [&](int slot) {
        if (LOADS) {
            const double2* gp = reinterpret_cast<const double2*>(key + (long)poly * N);
#pragma unroll
            for (int kk = 0; kk < E / 2; kk++) w[slot][kk] = gp[kk * T + tid];
        }
        poly = (poly + 1) % 24;
    };
    for (int i = 0; i < W; i++) { for (int kk = 0; kk < E / 2; kk++) { w[i][kk].x = 1e-3; w[i][kk].y = 2e-3; } request(i); }
    for (int r = 0; r < reps; r++) {
        auto step = [&](auto qtag) {
            constexpr int q = decltype(qtag)::value;
            constexpr int sl = q % W;
#pragma unroll
            for (int kk = 0; kk < E / 2; kk++) {
                accn[2 * kk] = __builtin_fma(xs[2 * kk], w[sl][kk].x, accn[2 * kk]);
                accn[2 * kk] = __builtin_fma(-xs[2 * kk + 1], w[sl][kk].y, accn[2 * kk]);
                accn[2 * kk + 1] = __builtin_fma(xs[2 * kk], w[sl][kk].y, accn[2 * kk + 1]);
                accn[2 * kk + 1] = __builtin_fma(xs[2 * kk + 1], w[sl][kk].x, accn[2 * kk + 1]);
            }
#pragma unroll
            for (int k = 0; k < E; k++) asm volatile("" : "+v"(accn[k]));
            __builtin_amdgcn_sched_barrier(0);
            request(sl);
        };
        auto hook = [&](auto stag) {
            step(std::integral_constant<int, decltype(stag)::value>{});
        };
        fft_inv1_hooked<2>(x, tw, data, tid, hook);
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        for (int k = 0; k < E; k++) { x[0][k] = accn[k] * 1e-6 + x[0][k] * 1e-3; accn[k] = 0.0; }
    }
    double s_ = 0;
    for (int k = 0; k < E; k++) s_ += x[0][k];
    for (int i = 0; i < W; i++) for (int kk = 0; kk < E / 2; kk++) s_ += w[i][kk].x;
    sink[blockIdx.x * T + threadIdx.x] = s_;
}

static void exact_negacyclic(const int* a, const int* g, long long* c) {
    std::vector<__int128> t(2 * N, 0);
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) t[i + j] += (__int128)a[i] * g[j];
    for (int i = 0; i < N; i++) c[i] = (long long)(t[i] - t[i + N]);
}
template <typename K>
static void timeit(K kern, const char* name, const double* tw, double* sink, int blocks, double per, size_t lds_bytes = LDS_BYTES) {
    const int reps = 2000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds_bytes, 0, tw, sink, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds_bytes, 0, tw, sink, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s blocks=%4d: %7.3f us per call, %7.3f us per polynomial transform per CU\n", name, blocks, ms * 1e3 / reps, ms * 1e3 / reps / per);
}
int main() {
    std::vector<double> twh = make_fft_twiddles();
    double *tw, *sink, *dout;
    int *da, *dg;
    const int R = 6;
    hipMalloc(&tw, N * sizeof(double)); hipMalloc(&sink, 512 * T * sizeof(double)); hipMalloc(&dout, 2 * N * sizeof(double));
    hipMalloc(&da, R * N * sizeof(int)); hipMalloc(&dg, R * N * sizeof(int));
    hipMemcpy(tw, twh.data(), N * sizeof(double), hipMemcpyHostToDevice);
    std::vector<int> a(R * N), g(R * N);
    int bad = 0;
    for (int pat = 0; pat < 4; pat++) {
        srand(12345 + pat);
        for (int i = 0; i < R * N; i++) {
            if (pat == 0) { a[i] = (rand() % 131072) - 65536; g[i] = (rand() % 131072) - 65536; }
            else if (pat == 1) { a[i] = (rand() & 1) ? -65536 : 65535; g[i] = (rand() & 1) ? -65536 : 65535; }
            else if (pat == 2) { a[i] = -65536; g[i] = -65536; }
            else { a[i] = (i % N == 0) ? 65535 : -65536; g[i] = -65536; }
        }
        hipMemcpy(da, a.data(), R * N * sizeof(int), hipMemcpyHostToDevice);
        hipMemcpy(dg, g.data(), R * N * sizeof(int), hipMemcpyHostToDevice);
        std::vector<long long> c0(N, 0), c1(N, 0), tmp(N);
        for (int r = 0; r < R; r++) {
            exact_negacyclic(&a[r * N], &g[r * N], tmp.data()); for (int i = 0; i < N; i++) c0[i] += tmp[i];
            exact_negacyclic(&a[r * N], &g[(r ^ 1) * N], tmp.data()); for (int i = 0; i < N; i++) c1[i] += tmp[i];
        }
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 1) hipLaunchKernelGGL(k_conv<1>, dim3(1), dim3(T), LDS_BYTES, 0, da, dg, dout, tw, R);
            else hipLaunchKernelGGL(k_conv<0>, dim3(1), dim3(T), LDS_BYTES, 0, da, dg, dout, tw, R);
            std::vector<double> o(2 * N);
            hipMemcpy(o.data(), dout, 2 * N * sizeof(double), hipMemcpyDeviceToHost);
            double e = 0, mx = 0;
            for (int i = 0; i < N; i++) {
                e = std::max(e, std::abs(o[i] - (double)c0[i])); e = std::max(e, std::abs(o[N + i] - (double)c1[i]));
                mx = std::max(mx, std::abs((double)c0[i]));
            }
            printf("pattern %d %s: max |exact| = 2^%.2f, max round-off = %.3g (2^%.2f) %s\n", pat, mode == 0 ? "two at a time" : "one at a time", log2(mx), e, log2(e + 1e-300), e < 0.25 ? "ok" : "BAD");
            if (!(e < 0.25)) bad++;
        }
    }
    timeit(k_time<1>, "forward, 1", tw, sink, 256, 1);
    timeit(k_time<2>, "forward, 2 half a phase apart", tw, sink, 256, 2);
    timeit(k_time<3>, "forward, 3 half a phase apart", tw, sink, 256, 3);
    timeit(k_time<11>, "inverse, 1", tw, sink, 256, 1);
    timeit(k_time<12>, "inverse, 2 half a phase apart", tw, sink, 256, 2);
    timeit(k_time<13>, "inverse, 3 half a phase apart", tw, sink, 256, 3);
    timeit(k_time<22>, "inverse, 2, fenced by the free counter", tw, sink, 256, 2);
    {
        double* key;
        hipMalloc(&key, 24 * N * sizeof(double));
        hipMemset(key, 0, 24 * N * sizeof(double));
        auto run = [&](auto kern, const char* name) {
            const int reps = 2000;
            hipLaunchKernelGGL(kern, dim3(256), dim3(T), LDS_BYTES, 0, tw, sink, 10, key);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(T), LDS_BYTES, 0, tw, sink, reps, key);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-78s %7.3f us per repetition\n", name, ms * 1e3 / reps);
        };
        run(k_stream<3, 2>, "inverse pair + 3 MACs, no loads");
        run(k_stream<3, 1>, "3 operand polynomials (96 KB per CU) + 3 MACs, no transform");
        run(k_stream<3, 0>, "3 operand polynomials requested in front of the inverse pair, + 3 MACs");
        run(k_stream<6, 2>, "inverse pair + 6 MACs, no loads");
        run(k_stream<6, 1>, "6 operand polynomials (192 KB per CU) + 6 MACs, no transform");
        run(k_stream<6, 0>, "6 operand polynomials requested in front of the inverse pair, + 6 MACs");
        run(k_rolling<2, false>, "one inverse transform + 6 MACs between its phases, no loads");
        run(k_rolling<1, true>, "  ... 6 operand polynomials (192 KB) through a window of 1");
        run(k_rolling<2, true>, "  ... through a window of 2");
        run(k_rolling<3, true>, "  ... through a window of 3");
        run(k_rolling<4, true>, "  ... through a window of 4");
        run(k_rolling<6, true>, "  ... through a window of 6");
    }
    const size_t lds1 = (size_t)(LDS_TW + LDS_DATA) * sizeof(double);   // 69 632 B: two workgroups fit a CU's 160 KB
    timeit(k_time_occ2<1>, "forward, 1, ONE workgroup per CU (same kernel)", tw, sink, 256, 1, lds1);
    timeit(k_time_occ2<1>, "forward, 1, TWO workgroups per CU", tw, sink, 512, 2, lds1);
    timeit(k_time_occ2<11>, "inverse, 1, ONE workgroup per CU (same kernel)", tw, sink, 256, 1, lds1);
    timeit(k_time_occ2<11>, "inverse, 1, TWO workgroups per CU", tw, sink, 512, 2, lds1);
    return bad ? 1 : 0;
}
