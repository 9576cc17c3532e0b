// Measurement tool (not part of the product): steady-state cost of one inverse transform of
// csrc/ntt_dev.hpp on one CU (one 512-thread workgroup per CU, as in the evaluator), with parts of it
// switched off to see what bounds it.   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../fhe-ram_amd/csrc
#include "ntt_dev.hpp"
#include <cstdio>
#include <vector>
using namespace fk;

// exchange with un-merged ds_read_b64 (2 LDS cycles each; the compiler's ds_read2_b64 takes 8 for two)
__device__ __forceinline__ double lds_read64(const double* p) {
    double v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return v;
}
template <int X, int B>
__device__ __forceinline__ void exchange_inv_asm(double (&x)[B][E], double* data, int tid) {
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) data[b * LDS_DATA + lay<X>(pat<X + 1>(tid, k))] = x[b][k];
    if constexpr (!wave_local<X>()) lds_barrier();
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = lds_read64(&data[b * LDS_DATA + lay<X>(pat<X>(tid, k))]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int B, int FULL>
__global__ __launch_bounds__(T, T / 256) void k_bench_asm(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    double x[B][E];
    for (int b = 0; b < B; b++)
        for (int k = 0; k < E; k++) x[b][k] = (double)(tid * 8 + k + b);
    for (int r = 0; r < reps; r++) {
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
        TwPass t;
        if (FULL) { inv_twiddles<3>(t, tw, tid); lds_barrier(); for (int b = 0; b < B; b++) inv_pass<3>(x[b], t); inv_twiddles<2>(t, tw, tid); }
        exchange_inv_asm<2, B>(x, data, tid);
        if (FULL) { for (int b = 0; b < B; b++) inv_pass<2>(x[b], t); inv_twiddles<1>(t, tw, tid); }
        exchange_inv_asm<1, B>(x, data, tid);
        if (FULL) { for (int b = 0; b < B; b++) inv_pass<1>(x[b], t); inv_twiddles<0>(t, tw, tid); }
        exchange_inv_asm<0, B>(x, data, tid);
        if (FULL) { for (int b = 0; b < B; b++) inv_pass<0>(x[b], t); }
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
    }
    double s = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + tid] = s;
}

// passes 0 and 1 take their (workgroup- / wave-uniform) twiddles from scalar registers, loaded once
template <int Q>
__device__ __forceinline__ void inv_twiddles_uniform(TwPass& t, const double* __restrict__ tw_g, int wave) {
    constexpr int HQ = 1 << (LOGE * Q);
    const int hm = HQ - 1 - (Q == 0 ? 0 : wave);
#pragma unroll
    for (int u = LOGE - 1; u >= 0; u--)
#pragma unroll
        for (int j = 0; j < (1 << u); j++) t.w[(1 << u) - 1 + j] = tw_g[(HQ << u) + ((1 << u) - 1 - j) * HQ + hm];
}
template <int B>
__global__ __launch_bounds__(T, T / 256) void k_bench_sgpr(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TwPass u0, u1;
    inv_twiddles_uniform<0>(u0, tw_g, wave);
    inv_twiddles_uniform<1>(u1, tw_g, wave);
    double x[B][E];
    for (int b = 0; b < B; b++)
        for (int k = 0; k < E; k++) x[b][k] = (double)(tid * 8 + k + b);
    for (int r = 0; r < reps; r++) {
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
        TwPass t;
        inv_twiddles<3>(t, tw, tid);
        double* buf = data + (r & 1) * LDS_DATA;
        for (int b = 0; b < B; b++) inv_pass<3>(x[b], t);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        inv_twiddles<2>(t, tw, tid);
        exchange_inv<2, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<2>(x[b], t);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        exchange_inv<1, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<1>(x[b], u1);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        exchange_inv<0, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<0>(x[b], u0);
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
    }
    double s = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + tid] = s;
}

// all twiddles from global memory (L1/L2) or scalar registers: none in LDS
template <int B>
__global__ __launch_bounds__(T, T / 256) void k_bench_gtw(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    TwPass u0, u1;
    inv_twiddles_uniform<0>(u0, tw_g, wave);
    inv_twiddles_uniform<1>(u1, tw_g, wave);
    double x[B][E];
    for (int b = 0; b < B; b++)
        for (int k = 0; k < E; k++) x[b][k] = (double)(tid * 8 + k + b);
    for (int r = 0; r < reps; r++) {
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
        TwPass t3, t2;
        inv_twiddles<3>(t3, tw_g, tid);      // same index arithmetic, global table
        inv_twiddles<2>(t2, tw_g, tid);
        double* buf = data + (r & 1) * LDS_DATA;
        for (int b = 0; b < B; b++) inv_pass<3>(x[b], t3);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        exchange_inv<2, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<2>(x[b], t2);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        exchange_inv<1, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<1>(x[b], u1);
        for (int b = 0; b < B; b++) { x[b][0] = reduce(x[b][0]); x[b][1] = reduce(x[b][1]); }
        exchange_inv<0, B>(x, buf, tid);
        for (int b = 0; b < B; b++) inv_pass<0>(x[b], u0);
        for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
    }
    double s = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + tid] = s;
}

// Split-phase workgroup barrier in LDS (arrive = one ds_add per wave, wait = bounded spin) with independent
// VALU work (a MAC-sized block: 24 mulmod-accumulates) between the two, against the same work after s_barrier.
__device__ __forceinline__ void sw_arrive(unsigned* cnt) {
    if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned)(size_t)cnt), "v"(1u) : "memory");
}
__device__ __forceinline__ void sw_wait(unsigned* cnt, unsigned target) {
    for (int spin = 0; spin < (1 << 20); spin++) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)cnt) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
template <int SPLIT>
__global__ __launch_bounds__(T, T / 256) void k_bench_split(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    unsigned* cnt = reinterpret_cast<unsigned*>(data + 2 * LDS_DATA);   // two counters in the third buffer
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    double x[1][E], op[3][E], xh[3][E], acc[E];
    for (int k = 0; k < E; k++) { x[0][k] = (double)(tid * 8 + k); acc[k] = 0; for (int r = 0; r < 3; r++) { op[r][k] = 1000.0 + r + k + tid; xh[r][k] = 77.0 * r + k + tid; } }
    for (int r = 0; r < reps; r++) {
        for (int k = 0; k < E; k++) x[0][k] = reduce(x[0][k]);
        TwPass t;
        inv_twiddles<3>(t, tw, tid);
        double* buf = data + (r & 1) * LDS_DATA;
        inv_pass<3>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
        inv_twiddles<2>(t, tw, tid);
        exchange_inv<2, 1>(x, buf, tid);
        inv_pass<2>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
        inv_twiddles<1>(t, tw, tid);
        exchange_inv<1, 1>(x, buf, tid);
        inv_pass<1>(x[0], t); x[0][0] = reduce(x[0][0]); x[0][1] = reduce(x[0][1]);
        inv_twiddles<0>(t, tw, tid);
        for (int k = 0; k < E; k++) buf[lay<0>(pat<1>(tid, k))] = x[0][k];
        if (SPLIT) {
            sw_arrive(&cnt[r & 1]);
            __builtin_amdgcn_sched_barrier(0);
            for (int k = 0; k < E; k++) acc[k] = 0.0;
            for (int q = 0; q < 3; q++) for (int k = 0; k < E; k++) acc[k] = macmod(acc[k], xh[q][k], op[q][k]);   // the next limb's MAC
            __builtin_amdgcn_sched_barrier(0);
            sw_wait(&cnt[r & 1], 8u * (unsigned)(r / 2 + 1));
        } else {
            lds_barrier();
        }
        for (int k = 0; k < E; k++) x[0][k] = buf[lay<0>(pat<0>(tid, k))];
        inv_pass<0>(x[0], t);
        for (int k = 0; k < E; k++) x[0][k] = reduce(x[0][k]);
        if (!SPLIT) {
            for (int k = 0; k < E; k++) acc[k] = 0.0;
            for (int q = 0; q < 3; q++) for (int k = 0; k < E; k++) acc[k] = macmod(acc[k], xh[q][k], op[q][k]);
        }
        for (int k = 0; k < E; k++) xh[0][k] += acc[k] * 1e-30;   // keep the MAC alive
    }
    double s = 0;
    for (int k = 0; k < E; k++) s += x[0][k] + xh[0][k];
    sink[blockIdx.x * T + tid] = s;
}

// The NEXT limb's pointwise MAC (3 operand polynomials, 168 FP64 instructions per thread) as filler of the LDS
// round trips of the current inverse transform: one operand polynomial's worth after the DS operations of each of
// the three exchanges have been issued, before the pass that consumes the exchanged data.
//   FILL 0: transform, then the MAC (what the fused kernels do);  FILL 1: interleaved.
template <int X>
__device__ __forceinline__ void xchg_issue(double (&x)[E], double (&y)[E], double* buf, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) buf[lay<X>(pat<X + 1>(tid, k))] = x[k];
    if constexpr (!wave_local<X>()) lds_barrier(); else wave_lds_fence();
#pragma unroll
    for (int k = 0; k < E; k++) y[k] = buf[lay<X>(pat<X>(tid, k))];
    __builtin_amdgcn_sched_barrier(0);   // the DS operations are issued before whatever follows
}
// FILL 2: the same filler, but spread between the individual DS operations (one DS write / read, then 3-4 FP64
// instructions): a wave issues in order, so a DS operation that finds the LDS queue full holds up everything behind it
template <int X, typename F>
__device__ __forceinline__ void xchg_spread(double (&x)[E], double (&y)[E], double* buf, int tid, F&& half) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < E; k++) buf[lay<X>(pat<X + 1>(tid, k))] = x[k];
    half(0);
#pragma unroll
    for (int i = 0; i < E; i++) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 4, 0); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!wave_local<X>()) lds_barrier(); else wave_lds_fence();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < E; k++) y[k] = buf[lay<X>(pat<X>(tid, k))];
    half(1);
#pragma unroll
    for (int i = 0; i < E; i++) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 4, 0); }
    __builtin_amdgcn_sched_barrier(0);
}
template <int FILL>
__global__ __launch_bounds__(T, T / 256) void k_bench_fill(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    double x[E], y[E], op[3][E], xh[3][E], acc[E];
    for (int k = 0; k < E; k++) { x[k] = (double)(tid * 8 + k); acc[k] = 0; for (int r = 0; r < 3; r++) { op[r][k] = 1000.0 + r + k + tid; xh[r][k] = 77.0 * r + k + tid; } }
    auto chunk = [&](int q) {
#pragma unroll
        for (int k = 0; k < E; k++) acc[k] = macmod(acc[k], xh[q][k], op[q][k]);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int r = 0; r < reps; r++) {
        for (int k = 0; k < E; k++) { x[k] = reduce(x[k]); acc[k] = 0.0; }
        TwPass t;
        double* buf = data + (r & 1) * LDS_DATA;
        inv_twiddles<3>(t, tw, tid);
        inv_pass<3>(x, t); x[0] = reduce(x[0]); x[1] = reduce(x[1]);
        inv_twiddles<2>(t, tw, tid);
        auto halfq = [&](int q, int h) {
#pragma unroll
            for (int k = 4 * h; k < 4 * h + 4; k++) acc[k] = macmod(acc[k], xh[q][k], op[q][k]);
        };
        if (FILL == 2) xchg_spread<2>(x, y, buf, tid, [&](int h) { halfq(0, h); });
        else { xchg_issue<2>(x, y, buf, tid); if (FILL) chunk(0); }
        inv_pass<2>(y, t); y[0] = reduce(y[0]); y[1] = reduce(y[1]);
        inv_twiddles<1>(t, tw, tid);
        if (FILL == 2) xchg_spread<1>(y, x, buf, tid, [&](int h) { halfq(1, h); });
        else { xchg_issue<1>(y, x, buf, tid); if (FILL) chunk(1); }
        inv_pass<1>(x, t); x[0] = reduce(x[0]); x[1] = reduce(x[1]);
        inv_twiddles<0>(t, tw, tid);
        if (FILL == 2) xchg_spread<0>(x, y, buf, tid, [&](int h) { halfq(2, h); });
        else { xchg_issue<0>(x, y, buf, tid); if (FILL) chunk(2); }
        inv_pass<0>(y, t);
        for (int k = 0; k < E; k++) x[k] = reduce(y[k]);
        if (!FILL) { chunk(0); chunk(1); chunk(2); }
        for (int k = 0; k < E; k++) xh[0][k] += acc[k] * 1e-30;   // keep the MAC alive
    }
    double s_ = 0;
    for (int k = 0; k < E; k++) s_ += x[k] + xh[0][k];
    sink[blockIdx.x * T + tid] = s_;
}

// Prototype of the other decomposition: one WAVE per polynomial as a 64 x 64 four-step transform.  Lane b holds
// the 64 coefficients {64 a + b}: a 64-point transform over a in registers (twiddles wave-uniform: scalar
// loads), a twist by a per-element factor, one transpose through LDS, a second 64-point transform.  No
// workgroup barrier.  Arbitrary twiddle values: this measures time, not a usable transform.
__device__ __forceinline__ void ntt64_regs(double (&x)[64], const double* __restrict__ w) {
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int half = 32 >> s;
#pragma unroll
        for (int j = 0; j < (1 << s); j++) {
            const double tw = w[(1 << s) - 1 + j];      // uniform address: scalar load
#pragma unroll
            for (int i = 0; i < half; i++) gs(x[2 * j * half + i], x[2 * j * half + i + half], tw);
        }
#pragma unroll
        for (int k = 0; k < 64; k += 8) x[k] = reduce(x[k]);   // keep magnitudes bounded (cost model only)
    }
}
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_wave_ntt(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* buf = lds + wave * (64 * 65);
    double x[64];
    for (int a = 0; a < 64; a++) x[a] = (double)(lane + 64 * a);
    for (int r = 0; r < reps; r++) {
        ntt64_regs(x, tw_g);
#pragma unroll
        for (int a = 0; a < 64; a++) x[a] = mulmod(reduce(x[a]), tw_g[64 + 64 * a + lane]);   // twist, coalesced table read
#pragma unroll
        for (int a = 0; a < 64; a++) buf[a * 65 + lane] = x[a];
#pragma unroll
        for (int b = 0; b < 64; b++) x[b] = buf[lane * 65 + b];
        ntt64_regs(x, tw_g);
    }
    double s = 0;
    for (int a = 0; a < 64; a++) s += x[a];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES> void run_wave(const double* tw, double* sink, int blocks) {
    const size_t ldsb = (size_t)WAVES * 64 * 65 * sizeof(double);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_ntt<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int reps = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; i++) k_wave_ntt<WAVES><<<blocks, 64 * WAVES, ldsb>>>(tw, sink, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_wave_ntt<WAVES><<<blocks, 64 * WAVES, ldsb>>>(tw, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("wave-per-polynomial 64x64 prototype, %d waves per CU: %7.3f us per transform per CU (%s)\n", WAVES,
           ms * 1e3 / (10.0 * reps * WAVES * ((blocks + 255) / 256)), hipGetErrorString(hipGetLastError()));
}

// VARIANT 0: full ntt_inv<B>;  1: no workgroup barriers;  2: no LDS exchanges (butterflies + twiddle reads only);
//         3: exchanges only (no butterflies)
template <int B, int VARIANT>
__global__ __launch_bounds__(T, T / 256) void k_bench(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    if constexpr (VARIANT == 5) { if (tid >= T / 2) __builtin_amdgcn_s_setprio(3); }
    if constexpr (VARIANT == 6) { if ((tid >> 6) & 1) __builtin_amdgcn_s_setprio(3); }
    double x[B][E];
    for (int b = 0; b < B; b++)
        for (int k = 0; k < E; k++) x[b][k] = (double)(tid * 8 + k + b);
    for (int r = 0; r < reps; r++) {
        if constexpr (VARIANT == 0 || VARIANT == 5 || VARIANT == 6) {
            ntt_inv<B>(x, tw, data, tid);
        } else {
            for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
            TwPass t;
            if constexpr (VARIANT != 3) { inv_twiddles<3>(t, tw, tid); for (int b = 0; b < B; b++) inv_pass<3>(x[b], t); }
            if constexpr (VARIANT != 2) exchange_inv<2, B>(x, data, tid);
            if constexpr (VARIANT != 3) { inv_twiddles<2>(t, tw, tid); for (int b = 0; b < B; b++) inv_pass<2>(x[b], t); }
            if constexpr (VARIANT != 2) exchange_inv<1, B>(x, data, tid);
            if constexpr (VARIANT != 3) { inv_twiddles<1>(t, tw, tid); for (int b = 0; b < B; b++) inv_pass<1>(x[b], t); }
            if constexpr (VARIANT != 2) {
                // exchange 0 with or without its barrier
                for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) data[b * LDS_DATA + lay<0>(pat<1>(tid, k))] = x[b][k];
                if constexpr (VARIANT != 1) lds_barrier();   // VARIANT 7 = this barrier only (no barrier at the start of the transform)
                for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = data[b * LDS_DATA + lay<0>(pat<0>(tid, k))];
            }
            if constexpr (VARIANT != 3) { inv_twiddles<0>(t, tw, tid); for (int b = 0; b < B; b++) inv_pass<0>(x[b], t); }
            for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
        }
    }
    double s = 0;
    for (int b = 0; b < B; b++) for (int k = 0; k < E; k++) s += x[b][k];
    sink[blockIdx.x * T + tid] = s;
}

// Two transforms software-pipelined inside each wave: while the exchange of one is in flight in the LDS
// pipe, the butterflies of the other occupy the VALU.
__device__ __forceinline__ void slots01(double (&x)[E]) { x[0] = reduce(x[0]); x[1] = reduce(x[1]); }
template <int X> __device__ __forceinline__ void xchg_issue_local(double (&x)[E], double* buf, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) buf[lay<X>(pat<X + 1>(tid, k))] = x[k];
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = buf[lay<X>(pat<X>(tid, k))];
}
__global__ __launch_bounds__(T, T / 256) void k_pipe2(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    double* bufA = data;
    double* bufB = data + LDS_DATA;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    double a[E], b[E];
    for (int k = 0; k < E; k++) { a[k] = (double)(tid * 8 + k); b[k] = (double)(tid * 8 + k + 1); }
    for (int r = 0; r < reps; r++) {
        for (int k = 0; k < E; k++) { a[k] = reduce(a[k]); b[k] = reduce(b[k]); }
        TwPass t3, t2, t1, t0;
        inv_twiddles<3>(t3, tw, tid);
        lds_barrier();
        inv_pass<3>(a, t3); slots01(a);
        inv_twiddles<2>(t2, tw, tid);
        xchg_issue_local<2>(a, bufA, tid);
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<3>(b, t3); slots01(b);
        xchg_issue_local<2>(b, bufB, tid);
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<2>(a, t2); slots01(a);
        inv_twiddles<1>(t1, tw, tid);
        xchg_issue_local<1>(a, bufA, tid);
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<2>(b, t2); slots01(b);
        xchg_issue_local<1>(b, bufB, tid);
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<1>(a, t1); slots01(a);
        inv_twiddles<0>(t0, tw, tid);
#pragma unroll
        for (int k = 0; k < E; k++) bufA[lay<0>(pat<1>(tid, k))] = a[k];
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<1>(b, t1); slots01(b);
#pragma unroll
        for (int k = 0; k < E; k++) bufB[lay<0>(pat<1>(tid, k))] = b[k];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < E; k++) a[k] = bufA[lay<0>(pat<0>(tid, k))];
#pragma unroll
        for (int k = 0; k < E; k++) b[k] = bufB[lay<0>(pat<0>(tid, k))];
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<0>(a, t0);
        inv_pass<0>(b, t0);
        for (int k = 0; k < E; k++) { a[k] = reduce(a[k]); b[k] = reduce(b[k]); }
    }
    double s = 0;
    for (int k = 0; k < E; k++) s += a[k] + b[k];
    sink[blockIdx.x * T + tid] = s;
}
#define MIX()                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++) {               \
        __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);            \
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);            \
    }
__global__ __launch_bounds__(T, T / 256) void k_pipe2g(const double* __restrict__ tw_g, double* sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    double* bufA = data;
    double* bufB = data + LDS_DATA;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    double a[E], b[E];
    for (int k = 0; k < E; k++) { a[k] = (double)(tid * 8 + k); b[k] = (double)(tid * 8 + k + 1); }
    for (int r = 0; r < reps; r++) {
        for (int k = 0; k < E; k++) { a[k] = reduce(a[k]); b[k] = reduce(b[k]); }
        TwPass t3, t2, t1, t0;
        inv_twiddles<3>(t3, tw, tid);
        lds_barrier();
        inv_pass<3>(a, t3); slots01(a);
        inv_twiddles<2>(t2, tw, tid);
        __builtin_amdgcn_sched_barrier(0);
        xchg_issue_local<2>(a, bufA, tid);
        inv_pass<3>(b, t3); slots01(b);
        MIX();
        __builtin_amdgcn_sched_barrier(0);
        xchg_issue_local<2>(b, bufB, tid);
        inv_pass<2>(a, t2); slots01(a);
        MIX();
        __builtin_amdgcn_sched_barrier(0);
        inv_twiddles<1>(t1, tw, tid);
        xchg_issue_local<1>(a, bufA, tid);
        inv_pass<2>(b, t2); slots01(b);
        MIX();
        __builtin_amdgcn_sched_barrier(0);
        xchg_issue_local<1>(b, bufB, tid);
        inv_pass<1>(a, t1); slots01(a);
        MIX();
        __builtin_amdgcn_sched_barrier(0);
        inv_twiddles<0>(t0, tw, tid);
#pragma unroll
        for (int k = 0; k < E; k++) bufA[lay<0>(pat<1>(tid, k))] = a[k];
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<1>(b, t1); slots01(b);
#pragma unroll
        for (int k = 0; k < E; k++) bufB[lay<0>(pat<1>(tid, k))] = b[k];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < E; k++) a[k] = bufA[lay<0>(pat<0>(tid, k))];
#pragma unroll
        for (int k = 0; k < E; k++) b[k] = bufB[lay<0>(pat<0>(tid, k))];
        __builtin_amdgcn_sched_barrier(0);
        inv_pass<0>(a, t0);
        inv_pass<0>(b, t0);
        for (int k = 0; k < E; k++) { a[k] = reduce(a[k]); b[k] = reduce(b[k]); }
    }
    double s = 0;
    for (int k = 0; k < E; k++) s += a[k] + b[k];
    sink[blockIdx.x * T + tid] = s;
}
template <typename K> void run_pipe2(K kern, const char* name, const double* tw, double* sink, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    const int reps = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) kern<<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) kern<<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s B=2 blocks=%4d: %7.3f us per transform per CU\n", name, blocks, ms * 1e3 / (10.0 * reps * 2));
}

template <int B, int VARIANT> void run(const char* name, const double* tw, double* sink, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bench<B, VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    const int reps = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) k_bench<B, VARIANT><<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_bench<B, VARIANT><<<blocks, T, LDS_BYTES>>>(tw, sink, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves = (blocks + 255) / 256;   // workgroups per CU, sequential (LDS allows one at a time)
    printf("%-44s B=%d blocks=%4d: %7.3f us per transform per CU\n", name, B, blocks, ms * 1e3 / (10.0 * reps * B * waves));
}

int main() {
    std::vector<double> h(2 * N);
    for (int i = 0; i < 2 * N; i++) h[i] = (double)((i * 2654435761u) % 1000003);   // any values: timing only
    double *tw, *sink;
    hipMalloc(&tw, 2 * N * sizeof(double)); hipMalloc(&sink, 512 * T * sizeof(double));
    hipMemcpy(tw, h.data(), 2 * N * sizeof(double), hipMemcpyHostToDevice);
    run<1, 0>("inverse transform, full", tw, sink, 256);
    run<1, 1>("  without the workgroup barrier", tw, sink, 256);
    run<1, 2>("  butterflies + twiddle reads only", tw, sink, 256);
    run<1, 3>("  exchanges only", tw, sink, 256);
    run<1, 7>("  exchange-0 barrier only (double buffered)", tw, sink, 256);
    run<2, 7>("  exchange-0 barrier only (double buffered)", tw, sink, 256);
    run<1, 5>("  full, waves 4-7 at high priority", tw, sink, 256);
    run<1, 6>("  full, odd waves at high priority", tw, sink, 256);
    run<2, 0>("inverse transform, full", tw, sink, 256);
    run<2, 2>("  butterflies + twiddle reads only", tw, sink, 256);
    run<2, 3>("  exchanges only", tw, sink, 256);
    run<3, 0>("inverse transform, full", tw, sink, 256);
    run<3, 2>("  butterflies + twiddle reads only", tw, sink, 256);
    run<3, 3>("  exchanges only", tw, sink, 256);
    run_wave<4>(tw, sink, 256);
    run_wave<2>(tw, sink, 256);
    run_pipe2(k_bench_split<0>, "transform + MAC, s_barrier then MAC (x2)", tw, sink, 256);
    run_pipe2(k_bench_split<1>, "transform + MAC in a split software barrier (x2)", tw, sink, 256);
    run_pipe2(k_bench_fill<0>, "transform, then the next limb's MAC (x2)", tw, sink, 256);
    run_pipe2(k_bench_fill<1>, "next limb's MAC inside the exchanges' round trips (x2)", tw, sink, 256);
    run_pipe2(k_bench_fill<2>, "next limb's MAC spread between the DS operations (x2)", tw, sink, 256);
    run_pipe2(k_bench_gtw<1>, "double buffered, no twiddles in LDS (x2)", tw, sink, 256);
    run_pipe2(k_bench_sgpr<1>, "double buffered + scalar twiddles in passes 0,1 (x2)", tw, sink, 256);
    run_pipe2(k_bench_asm<1, 1>, "full, un-merged ds_read_b64 (x2 = per transform)", tw, sink, 256);
    run_pipe2(k_bench_asm<1, 0>, "exchanges only, un-merged ds_read_b64 (x2)", tw, sink, 256);
    run_pipe2(k_pipe2, "two transforms, coarse pipeline", tw, sink, 256);
    run_pipe2(k_pipe2g, "two transforms, DS ops spread over VALU", tw, sink, 256);
    hipFree(tw); hipFree(sink);
    return 0;
}
