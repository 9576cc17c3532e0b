// Negacyclic NTT over Z_p at N = 4096 for gfx950, p = 2^48 + 57345 (p = 1 mod 8192), carried in
// FP64: CDNA4 has no 64-bit integer multiplier and v_mul_{lo,hi}_u32 costs 1.75x an FP64 FMA
// (tools/valu_rate.hip), so residues live in doubles and the modular product is
// the error-free FMA form  a*b - rint(a*b/p)*p  (5 FP64 ops).  All values are exact integers
// below 2^53 in magnitude; arithmetic is "lazy" (values drift up to a few p and are pulled back
// with reduce()).
//
// Why this is the same arithmetic as the reference: the reference multiplies limb polynomials
// with Poulpy's FFT64 backend and rounds to i64 (/root/reference/examples/fhe-ram.rs:3-7,
// SURVEY.md §0.4); every product on the RAM path is an exact integer negacyclic convolution
// bounded by 6*2^44 < p/2, so the centred lift of the NTT result is that integer.
//
// Decomposition: one workgroup of 512 threads per polynomial, 8 coefficients per thread, four
// radix-8 register passes (3 butterfly stages each) separated by three LDS exchanges.  Twiddles
// (4096 doubles, psi^bitrev order, re-laid per stage so that lanes read consecutive or identical
// addresses) sit in LDS next to the 4096(+pad)-element exchange buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fk {

constexpr int LOGN = 12;
constexpr int N = 1 << LOGN;
#ifndef FK_LOGE
#define FK_LOGE 3
#endif
#ifndef FK_XCHG_REG
#define FK_XCHG_REG 0   // 1: both wave-local exchanges as register-lane transposes (DPP / v_permlane*_swap) instead of LDS; 2: only exchange 1 (lane bits 3..5: the permlane swaps)
#endif
constexpr int LOGE = FK_LOGE;    // log2(coefficients per thread): 3 (512 threads) or 4 (256 threads)
static_assert(LOGN % LOGE == 0, "radix must divide log N");
constexpr int E = 1 << LOGE;     // coefficients per thread
constexpr int T = N / E;         // threads per workgroup
constexpr int NPASS = LOGN / LOGE;
constexpr int LDS_TW = N;        // doubles
constexpr int LDS_DATA = N + N / E;  // doubles (exchange buffer incl. padding)
constexpr int BMAX = 3;          // polynomials transformed together (3 exchange buffers fit 160 KB of LDS)
constexpr size_t LDS_BYTES = (size_t)(LDS_TW + BMAX * LDS_DATA) * sizeof(double);

constexpr double P = 281474976768001.0;   // 2^48 + 57345, prime
constexpr double PINV = 1.0 / 281474976768001.0;
constexpr uint64_t P_U64 = 281474976768001ULL;
constexpr uint64_t PSI_8192 = 40653067404937ULL;   // primitive 8192-th root of unity mod p

// a*b mod p, |result| < ~(0.5 + |a*b|/p * 2^-51) p.  Exact for |a*b| < 2^101, see DESIGN.md.
__device__ __forceinline__ double mulmod(double a, double b) {
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double q = __builtin_rint(h * PINV);
    const double r = __builtin_fma(-q, P, h);
    return r + l;
}
// acc + a*b mod p (lazy)
__device__ __forceinline__ double macmod(double acc, double a, double b) {
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double q = __builtin_rint(h * PINV);
    const double r = __builtin_fma(-q, P, h);
    return (acc + r) + l;
}
// centred representative in [-p/2, p/2]
__device__ __forceinline__ double reduce(double x) {
    return __builtin_fma(-__builtin_rint(x * PINV), P, x);
}

// ---- index patterns --------------------------------------------------------------------
// Pass Q holds, per thread, the E elements  hi*E*S + k*S + lo  with S = N >> LOGE*(Q+1),
// tid = hi*S + lo.  Pass 0 = natural coefficients tid + T*k; the last pass = E*tid + k.
template <int Q>
__device__ __forceinline__ int pat(int tid, int k) {
    constexpr int LS = LOGN - LOGE * (Q + 1);  // log2(S)
    const int hi = tid >> LS;
    const int lo = tid & ((1 << LS) - 1);
    return (hi << (LS + LOGE)) + (k << LS) + lo;
}
// LDS layout of exchange X (between pass X and X+1).  Every layout keeps the 64*E indices owned by
// wave w inside the padded region [64*(E+1)*w, 64*(E+1)*(w+1)):
//  * the read side of exchange X walks runs of R = N >> LOGE*(X+2) consecutive elements; for
//    R < 64 a pad of R per E*R block keeps both sides bank-conflict free;
//  * otherwise (runs of whole waves) a pad of 64 per 64*E block only aligns the regions.
template <int X>
__device__ __forceinline__ int lay(int idx) {
    constexpr int LR = LOGN - LOGE * (X + 2);  // log2(R)
    if constexpr (LR >= 6) return idx + ((idx >> (6 + LOGE)) << 6);
    else return idx + ((idx >> (LR + LOGE)) << LR);
}
// Exchange X stays inside one wave when the pass-X pattern already has hi >= wave granularity
// (S_X = N >> LOGE*(X+1) <= 64): thread t = 64w + l then touches only indices of region w before
// and after, so no workgroup barrier is needed — LDS operations of one wave execute in order.
template <int X>
constexpr bool wave_local() { return (LOGN - LOGE * (X + 1)) <= 6; }

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) and then
// synchronises, but leaves global loads/stores in flight.  __syncthreads() also drains vmcnt,
// which would serialise the operand prefetches and the output stores behind every exchange.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// Hand-off inside ONE wave (wave-local exchanges): the lanes' LDS writes must be ordered before the other lanes'
// reads.  The hardware executes a wave's DS operations in order; this makes the ordering a guarantee of the
// memory model too (no instruction is emitted for wavefront scope: the compiler merely may not move or merge
// LDS accesses across it).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// B polynomials are transformed together: one LDS exchange moves all of them, and the B
// independent butterfly streams give the FP64 pipe more ILP.
// Hazard bookkeeping: positions of pattern 0 (lay<0>(tid + T*k)) are private to a thread.  The only
// accesses that read what another WAVE wrote are (a) the far side of exchange 0 and (b) the
// staged-limb gathers in the key-switch kernel; a barrier separates each of them from the writes
// before it, and a barrier separates them from the writes after it (forward: right after the reads
// of exchange 0; inverse: at the first LDS write of the next transform).
template <int X, int B> __device__ __forceinline__ void exchange_reg(double (&x)[B][E], int tid);
template <int X, int B>
__device__ __forceinline__ void exchange_fwd(double (&x)[B][E], double* data, int tid) {
    if constexpr ((FK_XCHG_REG == 1 && wave_local<X>()) || (FK_XCHG_REG == 2 && X == 1)) { exchange_reg<X, B>(x, tid); return; }
    if constexpr (!wave_local<X>()) lds_barrier();   // cross-wave readers of the previous transform are done
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) data[b * LDS_DATA + lay<X>(pat<X>(tid, k))] = x[b][k];
    if constexpr (!wave_local<X>()) lds_barrier(); else wave_lds_fence();
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = data[b * LDS_DATA + lay<X>(pat<X + 1>(tid, k))];
    // these reads cross waves, and the next exchange of the forward transform is wave local: it writes
    // this wave's region without a barrier of its own, so every wave's reads must have landed first
    if constexpr (!wave_local<X>() && wave_local<X + 1>()) lds_barrier();
}
template <int X, int B>
__device__ __forceinline__ void exchange_inv(double (&x)[B][E], double* data, int tid) {
    if constexpr ((FK_XCHG_REG == 1 && wave_local<X>()) || (FK_XCHG_REG == 2 && X == 1)) { exchange_reg<X, B>(x, tid); return; }
    // no barrier before the write when everything since the fence at the start of ntt_inv was wave
    // local; a second cross-wave exchange (radix 4 only) must fence the readers of the one before it
    if constexpr (!wave_local<X>() && !wave_local<X + 1>()) lds_barrier();
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) data[b * LDS_DATA + lay<X>(pat<X + 1>(tid, k))] = x[b][k];
    if constexpr (!wave_local<X>()) lds_barrier(); else wave_lds_fence();
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = data[b * LDS_DATA + lay<X>(pat<X>(tid, k))];
}

// ---- wave-local exchanges in registers (no LDS) -----------------------------------------------
// Inside wave w the 512 elements i = 64a + 8b + c (a, b, c in 0..7) are held as
//   pass 1: lane 8b + c, register a      pass 2: lane 8a + c, register b      pass 3: lane 8a + b, register c
// so exchange 1 (pass 1 <-> 2) is an 8x8 transpose between the register index and lane bits 3..5, and
// exchange 2 (pass 2 <-> 3) one between the register index and lane bits 0..2.  Each is three rounds of
// "registers k and k + 2^t swap halves between lanes l and l ^ 2^(s+t)":
//   lane bit 5 / 4 : v_permlane32_swap / v_permlane16_swap (gfx950), one instruction per dword pair;
//   lane bit 3 / 2 : two DPP moves per dword pair (row_shr / row_shl by 8 / 4, the bank mask selecting
//                    the half of each row that receives);
//   lane bit 1 / 0 : DPP quad permutes + selects.
// Selected with -DFK_XCHG_REG=1 (measured against the LDS form: make XREG=1, tools/ntt_bench.hip).
template <int LANEBIT>
__device__ __forceinline__ void lane_swap_u32(unsigned& a, unsigned& b, bool hi) {
    // after: lanes with bit LANEBIT clear: b = a of lane ^ 2^LANEBIT;  lanes with it set: a = b of lane ^ 2^LANEBIT
    if constexpr (LANEBIT == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (LANEBIT == 4) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else if constexpr (LANEBIT == 3) {
        const unsigned na = __builtin_amdgcn_update_dpp(a, b, 0x118 /*row_shr:8*/, 0xF, 0xC, false);
        const unsigned nb = __builtin_amdgcn_update_dpp(b, a, 0x108 /*row_shl:8*/, 0xF, 0x3, false);
        a = na; b = nb;
    } else if constexpr (LANEBIT == 2) {
        const unsigned na = __builtin_amdgcn_update_dpp(a, b, 0x114 /*row_shr:4*/, 0xF, 0xA, false);
        const unsigned nb = __builtin_amdgcn_update_dpp(b, a, 0x104 /*row_shl:4*/, 0xF, 0x5, false);
        a = na; b = nb;
    } else {
        constexpr int ctrl = (LANEBIT == 1) ? 0x4E /*quad_perm:[2,3,0,1]*/ : 0xB1 /*quad_perm:[1,0,3,2]*/;
        const unsigned pb = __builtin_amdgcn_update_dpp(0u, b, ctrl, 0xF, 0xF, false);
        const unsigned pa = __builtin_amdgcn_update_dpp(0u, a, ctrl, 0xF, 0xF, false);
        a = hi ? pb : a;
        b = hi ? b : pa;
    }
}
template <int LANEBIT>
__device__ __forceinline__ void lane_swap(double& a, double& b, bool hi) {
    unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
    unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
    lane_swap_u32<LANEBIT>(al, bl, hi);
    lane_swap_u32<LANEBIT>(ah, bh, hi);
    a = __hiloint2double((int)ah, (int)al);
    b = __hiloint2double((int)bh, (int)bl);
}
// 8x8 transpose between the register index and lane bits S0 .. S0+2
template <int S0>
__device__ __forceinline__ void transpose_reg_lane(double (&x)[E], int lane) {
    static_assert(E == 8, "register transposes are written for radix 8");
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const bool hi = (lane >> (S0 + t)) & 1;
#pragma unroll
        for (int k = 0; k < E; k++) {
            if (k & (1 << t)) continue;
            if (t == 0) lane_swap<S0 + 0>(x[k], x[k | 1], hi);
            if (t == 1) lane_swap<S0 + 1>(x[k], x[k | 2], hi);
            if (t == 2) lane_swap<S0 + 2>(x[k], x[k | 4], hi);
        }
    }
}
template <int X, int B>
__device__ __forceinline__ void exchange_reg(double (&x)[B][E], int tid) {
    static_assert(LOGE == 3 && (X == 1 || X == 2), "wave-local exchanges of the radix-8 transform");
#pragma unroll
    for (int b = 0; b < B; b++) transpose_reg_lane<(X == 1) ? 3 : 0>(x[b], tid & 63);
}

// Cooley-Tukey butterfly: (x, y) <- (x + w*y, x - w*y)
__device__ __forceinline__ void bf(double& x, double& y, double w) {
    const double t = mulmod(y, w);
    y = x - t;
    x = x + t;
}
// Gentleman-Sande butterfly with the mirrored forward twiddle: (a, b) <- (a + b, (b - a)*w)
__device__ __forceinline__ void gs(double& a, double& b, double w) {
    const double s = a + b;
    const double d = b - a;
    a = s;
    b = mulmod(d, w);
}

// Twiddle table position of W[2^s + J] (s = LOGE*Q + u, J = hi*2^u + j): 2^s + j*E^Q + hi, so
// the lanes of a wave read consecutive (last pass) or identical (first passes) addresses.
// The E-1 twiddles of a pass are read into registers as one group, issued BEFORE the exchange that
// precedes the pass: LDS operations complete in order, so by the time the exchanged data has
// arrived the twiddles are there too.  (Read one by one next to their butterflies, as the compiler
// schedules them when left alone, every read exposes a full LDS round trip: measured at 40 % of
// an inverse transform.)
struct TwPass { double w[E - 1]; };
template <int Q>
__device__ __forceinline__ void fwd_twiddles(TwPass& t, const double* tw, int tid) {
    constexpr int LS = LOGN - LOGE * (Q + 1);
    constexpr int HQ = 1 << (LOGE * Q);
    const int hi = tid >> LS;
#pragma unroll
    for (int u = 0; u < LOGE; u++)
#pragma unroll
        for (int j = 0; j < (1 << u); j++) t.w[(1 << u) - 1 + j] = tw[(HQ << u) + j * HQ + hi];
    __builtin_amdgcn_sched_barrier(0);
}
template <int Q>
__device__ __forceinline__ void fwd_pass(double (&x)[E], const TwPass& t) {
#pragma unroll
    for (int u = 0; u < LOGE; u++) {
        const int half = E >> (u + 1);
#pragma unroll
        for (int j = 0; j < (1 << u); j++) {
            const double w = t.w[(1 << u) - 1 + j];
#pragma unroll
            for (int i = 0; i < half; i++) bf(x[2 * j * half + i], x[2 * j * half + i + half], w);
        }
    }
}
// inverse of fwd_pass<Q> up to the factor E (the total 1/N is folded into prepared operands):
// w^-1 of forward twiddle W[m + J] is -W[2m - 1 - J]; the sign is absorbed by using (b - a).
template <int Q>
__device__ __forceinline__ void inv_twiddles(TwPass& t, const double* tw, int tid) {
    constexpr int LS = LOGN - LOGE * (Q + 1);
    constexpr int HQ = 1 << (LOGE * Q);
    const int hm = HQ - 1 - (tid >> LS);
#pragma unroll
    for (int u = LOGE - 1; u >= 0; u--)
#pragma unroll
        for (int j = 0; j < (1 << u); j++) t.w[(1 << u) - 1 + j] = tw[(HQ << u) + ((1 << u) - 1 - j) * HQ + hm];
    __builtin_amdgcn_sched_barrier(0);
}
template <int Q>
__device__ __forceinline__ void inv_pass(double (&x)[E], const TwPass& t) {
#pragma unroll
    for (int u = LOGE - 1; u >= 0; u--) {
        const int half = E >> (u + 1);
#pragma unroll
        for (int j = 0; j < (1 << u); j++) {
            const double w = t.w[(1 << u) - 1 + j];
#pragma unroll
            for (int i = 0; i < half; i++) gs(x[2 * j * half + i], x[2 * j * half + i + half], w);
        }
    }
}

// Magnitude bookkeeping (all values are exact integers, so the only requirements are |v| < 2^53 =
// 32p for every intermediate and |d*w| < 2^101 for every product):
//  * forward (Cooley-Tukey): a stage maps (x, y) to x +- w*y with |w*y mod p| <= 0.5p + |y|*2^-3.4,
//    so magnitudes grow by at most 0.9p per stage and stay below 11p after the 12 stages
//    starting from limb inputs (< 2^18): no reduce() is needed for LOGE <= 3.
//  * inverse (Gentleman-Sande): a pass of LOGE stages turns inputs bounded by I into
//    [E*I, 4p, 2p, 2p, p, ...] (slot 0 is the sum of all E inputs, slot 1 a sum of products);
//    after the LDS exchange all E values of a thread come from the same slot, so I is uniform per
//    thread.  A product d*w with |d| <= 8I, |w| <= p/2 comes back within (0.5 + |d| * 1.5 * 2^-53) p of zero (q is
//    off by at most 0.5 + 3 * 2^-53 * |d*w|/p), i.e. <= 1.8p for |d| <= 27p; so a pass leaves at most
//    [8I, 7.2p, 3.6p, 3.6p, 1.8p, ...]; reducing slots 0 and 1 after each pass keeps I <= 3.6p, the next slot 0 below
//    29p < 32p = 2^53 / p and every product below 8 * 3.6p * p/2 < 2^101.  (Typical values are far smaller: the
//    bound is the worst case the self-test drives.)
template <int Q, int B>
__device__ __forceinline__ void fwd_rec(double (&x)[B][E], const TwPass& t, const double* tw, double* data, int tid) {
#pragma unroll
    for (int b = 0; b < B; b++) fwd_pass<Q>(x[b], t);
    if constexpr (Q + 1 < NPASS) {
        if constexpr (LOGE >= 4) {   // radix 16: a pass adds up to 3.6p; keep the classic per-pass reduce
#pragma unroll
            for (int b = 0; b < B; b++)
#pragma unroll
                for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
        }
        TwPass tn;
        fwd_twiddles<Q + 1>(tn, tw, tid);
        exchange_fwd<Q, B>(x, data, tid);
        fwd_rec<Q + 1, B>(x, tn, tw, data, tid);
    }
}
#ifndef FK_INV_SERIAL
#define FK_INV_SERIAL 1   // batched inverse transforms: the polynomials' passes one after the other (no interleaving by the scheduler: the butterflies' temporaries are shared instead of doubled); the batch still shares exchanges and barriers
#endif
template <int Q, int B>
__device__ __forceinline__ void inv_rec(double (&x)[B][E], const TwPass& t, const double* tw, double* data, int tid) {
#pragma unroll
    for (int b = 0; b < B; b++) {
        inv_pass<Q>(x[b], t);
        if constexpr (B > 1 && FK_INV_SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (Q > 0) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            if constexpr (LOGE <= 3) {
                x[b][0] = reduce(x[b][0]);
                x[b][1] = reduce(x[b][1]);
            } else {
#pragma unroll
                for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
            }
        }
        TwPass tn;
        inv_twiddles<Q - 1>(tn, tw, tid);
        exchange_inv<Q - 1, B>(x, data, tid);
        inv_rec<Q - 1, B>(x, tn, tw, data, tid);
    }
}

// Forward negacyclic NTT of B polynomials.  in: x[b][k] = coefficient tid + T*k (|x| < 2^20).
// out: x[b][k] = transform value at position E*tid + k (bit-reversed order), |x| < 11p.
template <int B>
__device__ __forceinline__ void ntt_fwd(double (&x)[B][E], const double* tw, double* data, int tid) {
    TwPass t;
    fwd_twiddles<0>(t, tw, tid);
    fwd_rec<0, B>(x, t, tw, data, tid);
}
// Inverse negacyclic NTT without the 1/N factor.  in: x[b][k] at position E*tid + k, |x| < 16p.
// out: x[b][k] = N * coefficient(tid + T*k) mod p, centred in [-p/2, p/2].
// FENCE: barrier before the first LDS write.  It is needed when other waves may still be reading this
// buffer across waves (the far side of exchange 0 of the previous inverse transform in the SAME
// buffer); callers that alternate between two buffers, or whose previous transform was a forward one
// (its cross-wave reads are fenced inside exchange_fwd), pass false.
// PRE: reduce the inputs first.  Needed when |x| may exceed ~3.4p (the external product accumulates 6 MAC terms of
// up to ~1.02p each); the key-switch family accumulates at most 4 terms (|x| <= 4.1p is NOT enough for E = 8: 8 * 4.1p
// > 32p) — so PRE = false is used for at most 3 terms (|x| <= 3.06p: slot 0 of the first pass <= 24.5p < 32p = 2^53,
// every product |d*w| <= 8 * 3.06p * p/2 < 2^100), checked on the device with worst-case operands by
// tests/test_gpu_modarith.py.
template <int B, bool FENCE = true, bool PRE = true>
__device__ __forceinline__ void ntt_inv(double (&x)[B][E], const double* tw, double* data, int tid) {
    if constexpr (PRE) {
#pragma unroll
        for (int b = 0; b < B; b++)
#pragma unroll
            for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
    }
    TwPass t;
    inv_twiddles<NPASS - 1>(t, tw, tid);
    if constexpr (FENCE) lds_barrier();   // first LDS write of this transform: earlier cross-wave readers are done
    inv_rec<NPASS - 1, B>(x, t, tw, data, tid);
#pragma unroll
    for (int b = 0; b < B; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
}

// ---- two inverse transforms, software pipelined against each other (round 4) ------------------------------------
// ntt_inv<2> runs the two polynomials in step: both passes, then both exchanges (all writes, then all reads), so a wave's
// LDS traffic and its butterflies alternate and the LDS round trips are exposed.  Here polynomial B runs HALF A PHASE
// behind A: the exchange of one (its writes and, right behind them, its reads: a wave's DS operations execute in order,
// and the wave-local exchanges 2 and 1 touch only the wave's own region) is in flight while the wave computes the other's
// pass.  Same operations per polynomial, same values; exchange 0 crosses waves and keeps its barrier.
template <int X>
__device__ __forceinline__ void xw_inv(const double (&x)[E], double* data, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) data[lay<X>(pat<X + 1>(tid, k))] = x[k];
}
template <int X>
__device__ __forceinline__ void xr_inv(double (&x)[E], const double* data, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = data[lay<X>(pat<X>(tid, k))];
}
__device__ __forceinline__ void reduce01(double (&x)[E]) { x[0] = reduce(x[0]); x[1] = reduce(x[1]); }
// FENCE / PRE as ntt_inv.  d0 / d1: the exchange buffers of the two polynomials.
template <bool FENCE = true, bool PRE = true>
__device__ __forceinline__ void ntt_inv2_skew(double (&x)[2][E], const double* tw, double* d0, double* d1, int tid) {
    static_assert(LOGE == 3 && NPASS == 4, "written for the radix-8 transform");
    if constexpr (PRE) {
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
    }
    TwPass t3, t2, t1, t0;
    inv_twiddles<3>(t3, tw, tid);
    if constexpr (FENCE) lds_barrier();
    inv_pass<3>(x[0], t3); reduce01(x[0]);
    xw_inv<2>(x[0], d0, tid); wave_lds_fence(); xr_inv<2>(x[0], d0, tid);
    inv_twiddles<2>(t2, tw, tid);
    inv_pass<3>(x[1], t3); reduce01(x[1]);
    xw_inv<2>(x[1], d1, tid); wave_lds_fence(); xr_inv<2>(x[1], d1, tid);
    __builtin_amdgcn_sched_barrier(0);
    inv_pass<2>(x[0], t2); reduce01(x[0]);
    xw_inv<1>(x[0], d0, tid); wave_lds_fence(); xr_inv<1>(x[0], d0, tid);
    inv_twiddles<1>(t1, tw, tid);
    inv_pass<2>(x[1], t2); reduce01(x[1]);
    xw_inv<1>(x[1], d1, tid); wave_lds_fence(); xr_inv<1>(x[1], d1, tid);
    __builtin_amdgcn_sched_barrier(0);
    inv_pass<1>(x[0], t1); reduce01(x[0]);
    xw_inv<0>(x[0], d0, tid);
    inv_twiddles<0>(t0, tw, tid);
    inv_pass<1>(x[1], t1); reduce01(x[1]);
    xw_inv<0>(x[1], d1, tid);
    lds_barrier();
    xr_inv<0>(x[0], d0, tid);
    xr_inv<0>(x[1], d1, tid);
    inv_pass<0>(x[0], t0);
    inv_pass<0>(x[1], t0);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = reduce(x[b][k]);
}

// ---- three forward transforms, software pipelined against each other (round 4) --------------------------------------
// As ntt_inv2_skew for the forward direction: the cross-wave exchange 0 stays common to the three polynomials (its barriers
// are shared), the wave-local exchanges 1 and 2 of one polynomial are in flight while the wave computes the next one's pass.
template <int X>
__device__ __forceinline__ void xw_fwd(const double (&x)[E], double* data, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) data[lay<X>(pat<X>(tid, k))] = x[k];
}
template <int X>
__device__ __forceinline__ void xr_fwd(double (&x)[E], const double* data, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = data[lay<X>(pat<X + 1>(tid, k))];
}
__device__ __forceinline__ void ntt_fwd3_skew(double (&x)[3][E], const double* tw, double* data, int tid) {
    static_assert(LOGE == 3 && NPASS == 4, "written for the radix-8 transform");
    TwPass t0, t1, t2, t3;
    fwd_twiddles<0>(t0, tw, tid);
#pragma unroll
    for (int b = 0; b < 3; b++) fwd_pass<0>(x[b], t0);
    fwd_twiddles<1>(t1, tw, tid);
    exchange_fwd<0, 3>(x, data, tid);            // cross-wave: common barriers
#pragma unroll
    for (int b = 0; b < 3; b++) {
        fwd_pass<1>(x[b], t1);
        xw_fwd<1>(x[b], data + b * LDS_DATA, tid); wave_lds_fence(); xr_fwd<1>(x[b], data + b * LDS_DATA, tid);
        if (b == 0) fwd_twiddles<2>(t2, tw, tid);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < 3; b++) {
        fwd_pass<2>(x[b], t2);
        xw_fwd<2>(x[b], data + b * LDS_DATA, tid); wave_lds_fence(); xr_fwd<2>(x[b], data + b * LDS_DATA, tid);
        if (b == 0) fwd_twiddles<3>(t3, tw, tid);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < 3; b++) fwd_pass<3>(x[b], t3);
}

// copy the 4096-entry twiddle table into LDS
__device__ __forceinline__ void load_twiddles(double* tw_lds, const double* __restrict__ tw_g, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) tw_lds[tid + T * k] = tw_g[tid + T * k];
    __syncthreads();
}
// the same in two halves, so that a kernel can issue its coefficient loads between them and pay
// one global round trip instead of two at start-up
struct TwRegs { double v[E]; };
__device__ __forceinline__ void twiddles_issue(TwRegs& r, const double* __restrict__ tw_g, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) r.v[k] = tw_g[tid + T * k];
}
__device__ __forceinline__ void twiddles_commit(const TwRegs& r, double* tw_lds, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) tw_lds[tid + T * k] = r.v[k];
    __syncthreads();
}

// ---- base-2^17 limb arithmetic on exact-integer doubles ------------------
constexpr int BASE2K = 17;
constexpr double TWO_B = 131072.0;         // 2^17
constexpr double INV_TWO_B = 1.0 / 131072.0;

// carry(x) = floor((x + 2^16) / 2^17);  digit(x) = x - carry*2^17 in [-2^16, 2^16)
__device__ __forceinline__ double carry_of(double x) { return __builtin_floor(__builtin_fma(x, INV_TWO_B, 0.5)); }
__device__ __forceinline__ double digit_of(double x, double c) { return __builtin_fma(-c, TWO_B, x); }

// vec_znx_big_normalize for one coefficient: in[0..SI) -> out[0..SO), SI >= SO.
template <int SI, int SO>
__device__ __forceinline__ void normalize_coeff(const double (&in)[SI], double (&out)[SO]) {
    double c = 0.0;
#pragma unroll
    for (int j = SI - 1; j >= 0; j--) {
        const double v = in[j] + c;
        c = carry_of(v);
        if (j < SO) out[j] = digit_of(v, c);
    }
}

// integer digit helpers (int32 is enough: all operands are sums of a few 17-bit digits)
__device__ __forceinline__ int digit17(int v) { return (int)((unsigned)v << 15) >> 15; }

// vec_znx_rsh_inplace(k = 1) on one coefficient of S limbs (see oracle/znx.hpp rsh_inplace):
// shift by one limb, then normalise with lsh = 16.
template <int S>
__device__ __forceinline__ void rsh1_coeff(const int (&x)[S], int (&y)[S]) {
    int d = -(x[S - 1] & 1);
    int c = (x[S - 1] - d) >> 1;
#pragma unroll
    for (int j = S - 1; j >= 1; j--) {
        const int src = x[j - 1];
        d = -(src & 1);
        const int cr = (src - d) >> 1;
        const int dpc = d * 65536 + c;
        y[j] = digit17(dpc);
        c = cr + ((dpc - y[j]) >> 17);
    }
    y[0] = digit17(c);
}

}  // namespace fk
