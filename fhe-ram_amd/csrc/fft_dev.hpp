// Negacyclic FFT64 at N = 4096 for gfx950: the arithmetic of the reference's own backend (Poulpy FFT64 / spqlios "reim"
// FFT, /root/reference/examples/fhe-ram.rs:3-7) restated for CDNA4.
//
// A real polynomial a of degree < N is folded into n = N/2 complex points z_j = a_j + i a_{j+n}; multiplying modulo
// X^N + 1 is multiplying z modulo X^n - i, i.e. pointwise in the n roots of X^n - i.  The forward transform is the
// "merged" Cooley-Tukey recursion on that modulus (X^m - r splits into X^(m/2) -+ sqrt(r): no separate twist pass),
// natural order in, bit-reversed order out; the inverse is its Gentleman-Sande mirror with conjugate twiddles, the 1/n
// folded into the prepared operands.  Every product on the RAM path is an exact integer negacyclic convolution of
// normalised 17-bit limbs bounded by 6 * 4096 * 2^32 < 2^47; the inverse transform's output is rounded to
// the nearest integer, which is that integer as long as the accumulated FP64 round-off stays below 1/2 — the reference
// backend's own contract.  Measured round-off (tests/test_gpu_fft.py, tools/fft_bench.hip): <= 2^-9 on uniformly random
// limbs at any magnitude, 0.11 on the worst coherent pattern (every coefficient -2^16, six terms).
//
// Decomposition (one 512-thread workgroup, 8 waves):
//   * the unit of work is a PAIR of polynomials.  In natural order thread t holds coefficients t + 512 k (k < 8) of both,
//     i.e. 4 complex points of each; one round of v_permlane32_swap turns that into 8 complex points of ONE polynomial
//     (lanes 0-31: polynomial A, lanes 32-63: polynomial B), so the 11 butterfly stages run as radix-8 register passes
//     3 + 3 + 3 + 2 with THREE LDS exchanges (one across waves, two inside a wave) — the exchange count of a radix-8
//     transform on 4096 reals, at 6 (forward) / 8 (inverse) FP64 instructions per complex butterfly instead of 8 per
//     modular one on twice as many points, and 4 FMAs per complex multiply-accumulate instead of 7 per modular one.
//   * for that the "thread id" of every kernel that calls a transform is the VIRTUAL id vt() below: index bit 8 sits on
//     lane bit 5 and bits 7..5 on the wave id, so that the swap partner (lane ^ 32) holds the points 256 further on.
//     Consecutive lanes 0..31 still hold consecutive coefficients (coalesced 128-byte rows, conflict-free odd-stride LDS
//     gathers).
//   * a single polynomial runs on all 64 lanes too (fft_fwd1 / fft_inv1): four points per lane, each three-stage pass split
//     around a swap round; same exchanges, same order of the results, half a pair's work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fk {

constexpr int LOGN = 12;
constexpr int N = 1 << LOGN;
constexpr int LOGE = 3;
constexpr int E = 1 << LOGE;     // coefficients per thread and polynomial (= 4 complex points)
constexpr int T = N / E;         // threads per workgroup
constexpr int NC = N / 2;        // complex points per polynomial
constexpr int LDS_TW = N;        // doubles: NC complex twiddles W[h] = (re, im), heap order (W[0] unused)
constexpr int LDS_DATA = N + N / E;  // doubles per polynomial exchange buffer: 2304 complex incl. padding
constexpr int BMAX = 3;          // exchange buffers (a pair transform uses two, a single one)
constexpr size_t LDS_BYTES = (size_t)(LDS_TW + BMAX * LDS_DATA) * sizeof(double);

typedef double d2 __attribute__((ext_vector_type(2)));   // (re, im)

// Virtual thread id: hardware thread h = 64 w + 32 l5 + ll  ->  256 l5 + 32 w + ll.  Every kernel that calls a transform
// owns coefficients vt + 512 k and MUST pass vt(threadIdx.x) as `tid`.
__device__ __forceinline__ int vt(int h) { return ((h & 32) << 3) | ((h >> 6) << 5) | (h & 31); }

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

// lanes l and l ^ 32 exchange: after the call lanes 0-31 hold (a, a of lane + 32), lanes 32-63 (b of lane - 32, b)
__device__ __forceinline__ void swap32(double& a, double& b) {
    unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
    unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]);
    b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap32(d2& a, d2& b) {
    double ar = a.x, ai = a.y, br = b.x, bi = b.y;
    swap32(ar, br);
    swap32(ai, bi);
    a.x = ar; a.y = ai; b.x = br; b.y = bi;
}

// Cooley-Tukey butterfly: (u, v) <- (u + w v, u - w v), 6 FP64 instructions (the second output as 2u - first)
__device__ __forceinline__ void bf(d2& u, d2& v, const d2 w) {
    double lr = __builtin_fma(w.x, v.x, u.x);
    lr = __builtin_fma(-w.y, v.y, lr);
    double li = __builtin_fma(w.x, v.y, u.y);
    li = __builtin_fma(w.y, v.x, li);
    v.x = __builtin_fma(2.0, u.x, -lr);
    v.y = __builtin_fma(2.0, u.y, -li);
    u.x = lr;
    u.y = li;
}
// Gentleman-Sande butterfly with the conjugate of the forward twiddle: (a, b) <- (a + b, (a - b) conj(w)), 8 FP64 instructions
__device__ __forceinline__ void gs(d2& a, d2& b, const d2 w) {
    const double dr = a.x - b.x, di = a.y - b.y;
    a.x = a.x + b.x;
    a.y = a.y + b.y;
    b.x = __builtin_fma(di, w.y, dr * w.x);
    b.y = __builtin_fma(-dr, w.y, di * w.x);
}

// Twiddles of one register pass over the subtree rooted at heap node H: stage u (u = 0 .. ST-1) uses W[(H << u) + jb],
// jb < 2^u.  Read as one group, ahead of the exchange that precedes the pass (LDS operations complete in order).
template <int ST> struct TwPass { d2 w[(1 << ST) - 1]; };
template <int ST>
__device__ __forceinline__ void load_tw(TwPass<ST>& t, const d2* tw, int H) {
#pragma unroll
    for (int u = 0; u < ST; u++)
#pragma unroll
        for (int jb = 0; jb < (1 << u); jb++) t.w[(1 << u) - 1 + jb] = tw[(H << u) + jb];
}
// ST butterfly stages on the 2^ST values y[0 .. 2^ST)
template <int ST>
__device__ __forceinline__ void fwd_pass(d2* y, const TwPass<ST>& t) {
    constexpr int R = 1 << ST;
#pragma unroll
    for (int u = 0; u < ST; u++) {
        const int half = R >> (u + 1);
#pragma unroll
        for (int jb = 0; jb < (1 << u); jb++)
#pragma unroll
            for (int i = 0; i < half; i++) bf(y[2 * jb * half + i], y[2 * jb * half + i + half], t.w[(1 << u) - 1 + jb]);
    }
}
template <int ST>
__device__ __forceinline__ void inv_pass(d2* y, const TwPass<ST>& t) {
    constexpr int R = 1 << ST;
#pragma unroll
    for (int u = ST - 1; u >= 0; u--) {
        const int half = R >> (u + 1);
#pragma unroll
        for (int jb = 0; jb < (1 << u); jb++)
#pragma unroll
            for (int i = 0; i < half; i++) gs(y[2 * jb * half + i], y[2 * jb * half + i + half], t.w[(1 << u) - 1 + jb]);
    }
}

// ---- LDS layouts of the three exchanges (complex units inside one polynomial's buffer of 2304) -------------------------------
// Index bits of a complex point j (11 bits), c = the 3 bits a thread holds in registers, w = wave, ll = lane & 31:
//   after pass 0 (bits 10..8 done):  j = c 256 + w 32 + ll          exchange 0: written there, read at  w 256 + c 32 + ll
//   after pass 1 (bits 7..5 done) :  j = w 256 + c 32 + ll          exchange 1: written there, read at  w 256 + hi3 32 + c 4 + lo2   (ll = hi3 4 + lo2)
//   after pass 2 (bits 4..2 done) :  j = w 256 + hi3 32 + c 4 + lo2 exchange 2: written there, read at  w 256 + ll 8 + c
// Exchanges 0 and 1 use the padded layout j + 4 (j >> 5) (36 per 32: both sides of both are bank-conflict free with 16-byte
// accesses: writes are served in groups of 8 consecutive lanes, reads in groups of 16); exchange 2 the XOR swizzle
// j ^ ((j >> 3) & 15) inside the wave's own 288-slot region (a stride-8 read side has no conflict-free padding).  Wave w
// owns [288 w, 288 w + 288) in every layout; only exchange 0 crosses waves.
struct XAddr {
    int w, ll, l5;
    int x0a;   // c * 288 + [w * 36 + ll]           exchange 0, far side
    int x0b;   // [w * 288 + ll] + c * 36          exchange 0 near side = exchange 1 far side
    int x1;    // [w * 288 + hi3 * 36 + lo2] + 4 c  exchange 1 near side
    int x2a;   // w * 288 + ([hi3 * 32 | (hi3 & 3) << 2 | lo2] ^ C(c))     exchange 2 far side
    int x2b;   // w * 288 + ([ll * 8 ^ (ll & 15)] ^ c)                      exchange 2 near side
};
__device__ __forceinline__ XAddr xaddr(int tid) {
    XAddr a;
    a.w = (tid >> 5) & 7; a.ll = tid & 31; a.l5 = tid >> 8;
    const int hi3 = a.ll >> 2, lo2 = a.ll & 3;
    a.x0a = a.w * 36 + a.ll;
    a.x0b = a.w * 288 + a.ll;
    a.x1 = a.w * 288 + hi3 * 36 + lo2;
    a.x2a = (hi3 * 32) | ((hi3 & 3) << 2) | lo2;
    a.x2b = (a.ll * 8) ^ (a.ll & 15);
    return a;
}
constexpr int x2c(int c) { return ((c & 3) << 2) | ((c >> 2) << 4) | (c >> 1); }

// ---- forward transform of a pair --------------------------------------------------------------------------------------------
// in : a[k], b[k] = coefficient tid + T k of polynomial A / B (|.| < 2^20), tid = vt(threadIdx.x)
// out: a[2m], a[2m+1] = (re, im) of A's transform at this thread's point m (m < 4) — the order prepared operands are stored
//      in (k_prepare) and the inverse transform expects; likewise b.
// PAIR = false: B is a zero polynomial (b is ignored and left alone) and only bufA is touched.
// Starts with a workgroup barrier in front of its first LDS write (every earlier LDS read of the workgroup has been issued
// and waited for by then), ends with reads of the wave's own region.
template <bool PAIR>
__device__ __forceinline__ void fft_fwd2(double (&a)[E], double (&b)[E], const double* tw_, double* bufA, double* bufB, int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    const XAddr xa = xaddr(tid);
    const bool on = PAIR || xa.l5 == 0;
    d2* buf = reinterpret_cast<d2*>((PAIR && xa.l5) ? bufB : bufA);
    d2 y[8];
    TwPass<3> t;
    load_tw<3>(t, tw, 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d2 p, q;
        p.x = a[k]; p.y = a[k + 4];
        if constexpr (PAIR) { q.x = b[k]; q.y = b[k + 4]; } else { q.x = 0.0; q.y = 0.0; }
        swap32(p, q);
        y[2 * k] = p; y[2 * k + 1] = q;
    }
    fwd_pass<3>(y, t);
    load_tw<3>(t, tw, 8 + xa.w);
    lds_barrier();
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[c * 288 + xa.x0a] = y[c];
    }
    lds_barrier();
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[xa.x0b + c * 36];
    }
    fwd_pass<3>(y, t);
    load_tw<3>(t, tw, 64 + xa.w * 8 + (xa.ll >> 2));
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[xa.x0b + c * 36] = y[c];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[xa.x1 + 4 * c];
    }
    fwd_pass<3>(y, t);
    TwPass<2> t0, t1;
    load_tw<2>(t0, tw, 512 + (xa.w * 32 + xa.ll) * 2);
    load_tw<2>(t1, tw, 512 + (xa.w * 32 + xa.ll) * 2 + 1);
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[xa.w * 288 + (xa.x2a ^ x2c(c))] = y[c];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[xa.w * 288 + (xa.x2b ^ c)];
    }
    fwd_pass<2>(y, t0);
    fwd_pass<2>(y + 4, t1);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        d2 p = y[m], q = y[4 + m];
        swap32(p, q);
        a[2 * m] = p.x; a[2 * m + 1] = p.y;
        if constexpr (PAIR) { b[2 * m] = q.x; b[2 * m + 1] = q.y; }
    }
}

// ---- inverse transform of a pair ---------------------------------------------------------------------------------------------
// in : a / b as fft_fwd2 leaves them (accumulated products against prepared operands, which carry the 1/n)
// out: a[k], b[k] = coefficient tid + T k, ROUNDED to the nearest integer (an exact integer-valued double).
// FENCE: workgroup barrier in front of the first LDS write.  Needed when another wave may still be reading this buffer
// across waves (the far side of exchange 0 of the previous inverse transform in the SAME buffer, or a kernel's own gathers).
template <bool PAIR, bool FENCE, bool ROUND = true>
__device__ __forceinline__ void fft_inv2(double (&a)[E], double (&b)[E], const double* tw_, double* bufA, double* bufB, int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    const XAddr xa = xaddr(tid);
    const bool on = PAIR || xa.l5 == 0;
    d2* buf = reinterpret_cast<d2*>((PAIR && xa.l5) ? bufB : bufA);
    d2 y[8];
    TwPass<2> t0, t1;
    load_tw<2>(t0, tw, 512 + (xa.w * 32 + xa.ll) * 2);
    load_tw<2>(t1, tw, 512 + (xa.w * 32 + xa.ll) * 2 + 1);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        d2 p, q;
        p.x = a[2 * m]; p.y = a[2 * m + 1];
        if constexpr (PAIR) { q.x = b[2 * m]; q.y = b[2 * m + 1]; } else { q.x = 0.0; q.y = 0.0; }
        swap32(p, q);
        y[m] = p; y[4 + m] = q;
    }
    inv_pass<2>(y, t0);
    inv_pass<2>(y + 4, t1);
    TwPass<3> t;
    load_tw<3>(t, tw, 64 + xa.w * 8 + (xa.ll >> 2));
    if constexpr (FENCE) lds_barrier();
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[xa.w * 288 + (xa.x2b ^ c)] = y[c];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[xa.w * 288 + (xa.x2a ^ x2c(c))];
    }
    inv_pass<3>(y, t);
    load_tw<3>(t, tw, 8 + xa.w);
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[xa.x1 + 4 * c] = y[c];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[xa.x0b + c * 36];
    }
    inv_pass<3>(y, t);
    load_tw<3>(t, tw, 1);
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) buf[xa.x0b + c * 36] = y[c];
    }
    lds_barrier();
    if (on) {
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = buf[c * 288 + xa.x0a];
    }
    inv_pass<3>(y, t);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d2 p = y[2 * k], q = y[2 * k + 1];
        swap32(p, q);
        if constexpr (ROUND) {
            a[k] = __builtin_rint(p.x); a[k + 4] = __builtin_rint(p.y);
            if constexpr (PAIR) { b[k] = __builtin_rint(q.x); b[k + 4] = __builtin_rint(q.y); }
        } else {   // (round-off measurements only)
            a[k] = p.x; a[k + 4] = p.y;
            if constexpr (PAIR) { b[k] = q.x; b[k + 4] = q.y; }
        }
    }
}

// ---- a single polynomial on all 64 lanes -------------------------------------------------------------------------------------
// Four complex points per thread; every three-stage pass is two stages in registers, one v_permlane32_swap round that trades
// the lower register bit for lane bit 5, and the third stage — so the exchanges, their layouts and the order of the results are
// those of the pair transform (a polynomial may go forward in a pair and come back alone), at half a pair's work.
//   natural  : y[2 b10 + b9], lane bit 5 = b8     -> stages b10, b9, swap, stage b8 ->   exchange 0 (across waves)
//   then     : y[2 b7 + b6],  lane bit 5 = b5     -> stages b7, b6,  swap, stage b5 ->   exchange 1 (in the wave)
//   then     : y[2 b4 + b3],  lane bit 5 = b2     -> stages b4, b3,  swap, stage b2 ->   exchange 2 (in the wave)
//   then     : y[2 b1 + b0],  lane bit 5 = b2     -> stages b1, b0
struct XAddr1 {
    int w, l5;
    int x0a;   // l5 * 576 + w * 36 + ll                     + b10 * 1152 + b8 * 288
    int x0b;   // w * 288 + l5 * 36 + ll                     + b7 * 144 + b6 * 72
    int x1a;   // w * 288 + l5 * 72 + ll                     + b7 * 144 + b5 * 36
    int x1b;   // w * 288 + hi3 * 36 + l5 * 4 + lo2          + b4 * 16 + b3 * 8
    int x2a;   // (hi3 * 32 | l5 * 8 | lo2) ^ ((hi3 & 3) * 4 | l5)       ^ (b4 * 18 | b2 * 4)
    int x2b;   // (ll * 8 | l5 * 4) ^ (ll & 15)                           ^ c
    int h2, h3;   // heap nodes of passes 2 and 3: 64 + 8 w + hi3,  512 + 64 w + 2 ll + l5
};
__device__ __forceinline__ XAddr1 xaddr1(int tid) {
    XAddr1 a;
    const int w = (tid >> 5) & 7, ll = tid & 31, l5 = tid >> 8, hi3 = ll >> 2, lo2 = ll & 3;
    a.w = w; a.l5 = l5;
    a.x0a = l5 * 576 + w * 36 + ll;
    a.x0b = w * 288 + l5 * 36 + ll;
    a.x1a = w * 288 + l5 * 72 + ll;
    a.x1b = w * 288 + hi3 * 36 + l5 * 4 + lo2;
    a.x2a = w * 288 + (((hi3 * 32) | (l5 * 8) | lo2) ^ (((hi3 & 3) * 4) | l5));
    a.x2b = w * 288 + (((ll * 8) | (l5 * 4)) ^ (ll & 15));
    a.h2 = 64 + 8 * w + hi3;
    a.h3 = 512 + 64 * w + 2 * ll + l5;
    return a;
}
// in / out as fft_fwd2 (one polynomial)
__device__ __forceinline__ void fft_fwd1(double (&a)[E], const double* tw_, double* buf_, int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    d2* buf = reinterpret_cast<d2*>(buf_);
    const XAddr1 xa = xaddr1(tid);
    d2 y[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { y[k].x = a[k]; y[k].y = a[k + 4]; }
    {
        const d2 w1 = tw[1], w2 = tw[2], w3 = tw[3], w4 = tw[4 + xa.l5], w6 = tw[6 + xa.l5];
        bf(y[0], y[2], w1); bf(y[1], y[3], w1);
        bf(y[0], y[1], w2); bf(y[2], y[3], w3);
        swap32(y[0], y[1]); swap32(y[2], y[3]);
        bf(y[0], y[1], w4); bf(y[2], y[3], w6);
    }
    const d2 v1 = tw[8 + xa.w], v2 = tw[16 + 2 * xa.w], v3 = tw[17 + 2 * xa.w], v4 = tw[32 + 4 * xa.w + xa.l5], v6 = tw[34 + 4 * xa.w + xa.l5];
    lds_barrier();
    buf[xa.x0a] = y[0]; buf[xa.x0a + 288] = y[1]; buf[xa.x0a + 1152] = y[2]; buf[xa.x0a + 1440] = y[3];
    lds_barrier();
    y[0] = buf[xa.x0b]; y[1] = buf[xa.x0b + 72]; y[2] = buf[xa.x0b + 144]; y[3] = buf[xa.x0b + 216];
    bf(y[0], y[2], v1); bf(y[1], y[3], v1);
    bf(y[0], y[1], v2); bf(y[2], y[3], v3);
    swap32(y[0], y[1]); swap32(y[2], y[3]);
    bf(y[0], y[1], v4); bf(y[2], y[3], v6);
    const d2 u1 = tw[xa.h2], u2 = tw[2 * xa.h2], u3 = tw[2 * xa.h2 + 1], u4 = tw[4 * xa.h2 + xa.l5], u6 = tw[4 * xa.h2 + 2 + xa.l5];
    buf[xa.x1a] = y[0]; buf[xa.x1a + 36] = y[1]; buf[xa.x1a + 144] = y[2]; buf[xa.x1a + 180] = y[3];
    wave_lds_fence();
    y[0] = buf[xa.x1b]; y[1] = buf[xa.x1b + 8]; y[2] = buf[xa.x1b + 16]; y[3] = buf[xa.x1b + 24];
    bf(y[0], y[2], u1); bf(y[1], y[3], u1);
    bf(y[0], y[1], u2); bf(y[2], y[3], u3);
    swap32(y[0], y[1]); swap32(y[2], y[3]);
    bf(y[0], y[1], u4); bf(y[2], y[3], u6);
    const d2 t1 = tw[xa.h3], t2 = tw[2 * xa.h3], t3 = tw[2 * xa.h3 + 1];
    buf[xa.x2a] = y[0]; buf[xa.x2a ^ 4] = y[1]; buf[xa.x2a ^ 18] = y[2]; buf[xa.x2a ^ 22] = y[3];
    wave_lds_fence();
    y[0] = buf[xa.x2b]; y[1] = buf[xa.x2b ^ 1]; y[2] = buf[xa.x2b ^ 2]; y[3] = buf[xa.x2b ^ 3];
    bf(y[0], y[2], t1); bf(y[1], y[3], t1);
    bf(y[0], y[1], t2); bf(y[2], y[3], t3);
#pragma unroll
    for (int m = 0; m < 4; m++) { a[2 * m] = y[m].x; a[2 * m + 1] = y[m].y; }
}
// in / out / FENCE as fft_inv2 (one polynomial)
template <bool FENCE, bool ROUND = true>
__device__ __forceinline__ void fft_inv1(double (&a)[E], const double* tw_, double* buf_, int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    d2* buf = reinterpret_cast<d2*>(buf_);
    const XAddr1 xa = xaddr1(tid);
    d2 y[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { y[m].x = a[2 * m]; y[m].y = a[2 * m + 1]; }
    {
        const d2 t1 = tw[xa.h3], t2 = tw[2 * xa.h3], t3 = tw[2 * xa.h3 + 1];
        gs(y[0], y[1], t2); gs(y[2], y[3], t3);
        gs(y[0], y[2], t1); gs(y[1], y[3], t1);
    }
    const d2 u1 = tw[xa.h2], u2 = tw[2 * xa.h2], u3 = tw[2 * xa.h2 + 1], u4 = tw[4 * xa.h2 + xa.l5], u6 = tw[4 * xa.h2 + 2 + xa.l5];
    if constexpr (FENCE) lds_barrier();
    buf[xa.x2b] = y[0]; buf[xa.x2b ^ 1] = y[1]; buf[xa.x2b ^ 2] = y[2]; buf[xa.x2b ^ 3] = y[3];
    wave_lds_fence();
    y[0] = buf[xa.x2a]; y[1] = buf[xa.x2a ^ 4]; y[2] = buf[xa.x2a ^ 18]; y[3] = buf[xa.x2a ^ 22];
    gs(y[0], y[1], u4); gs(y[2], y[3], u6);
    swap32(y[0], y[1]); swap32(y[2], y[3]);
    gs(y[0], y[1], u2); gs(y[2], y[3], u3);
    gs(y[0], y[2], u1); gs(y[1], y[3], u1);
    const d2 v1 = tw[8 + xa.w], v2 = tw[16 + 2 * xa.w], v3 = tw[17 + 2 * xa.w], v4 = tw[32 + 4 * xa.w + xa.l5], v6 = tw[34 + 4 * xa.w + xa.l5];
    buf[xa.x1b] = y[0]; buf[xa.x1b + 8] = y[1]; buf[xa.x1b + 16] = y[2]; buf[xa.x1b + 24] = y[3];
    wave_lds_fence();
    y[0] = buf[xa.x1a]; y[1] = buf[xa.x1a + 36]; y[2] = buf[xa.x1a + 144]; y[3] = buf[xa.x1a + 180];
    gs(y[0], y[1], v4); gs(y[2], y[3], v6);
    swap32(y[0], y[1]); swap32(y[2], y[3]);
    gs(y[0], y[1], v2); gs(y[2], y[3], v3);
    gs(y[0], y[2], v1); gs(y[1], y[3], v1);
    const d2 w1 = tw[1], w2 = tw[2], w3 = tw[3], w4 = tw[4 + xa.l5], w6 = tw[6 + xa.l5];
    buf[xa.x0b] = y[0]; buf[xa.x0b + 72] = y[1]; buf[xa.x0b + 144] = y[2]; buf[xa.x0b + 216] = y[3];
    lds_barrier();
    y[0] = buf[xa.x0a]; y[1] = buf[xa.x0a + 288]; y[2] = buf[xa.x0a + 1152]; y[3] = buf[xa.x0a + 1440];
    gs(y[0], y[1], w4); gs(y[2], y[3], w6);
    swap32(y[0], y[1]); swap32(y[2], y[3]);
    gs(y[0], y[1], w2); gs(y[2], y[3], w3);
    gs(y[0], y[2], w1); gs(y[1], y[3], w1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (ROUND) { a[k] = __builtin_rint(y[k].x); a[k + 4] = __builtin_rint(y[k].y); }
        else { a[k] = y[k].x; a[k + 4] = y[k].y; }
    }
}

// ---- the interface the kernels use (B polynomials at a time; data = B consecutive exchange buffers of LDS_DATA doubles) --------
template <int B>
__device__ __forceinline__ void ntt_fwd(double (&x)[B][E], const double* tw, double* data, int tid) {
    static_assert(B >= 1 && B <= BMAX, "one to three polynomials");
    if constexpr (B >= 2) fft_fwd2<true>(x[0], x[1], tw, data, data + LDS_DATA, tid);
    if constexpr (B == 1) fft_fwd1(x[0], tw, data, tid);
    if constexpr (B == 3) fft_fwd1(x[2], tw, data + 2 * LDS_DATA, tid);
}
// PRE is accepted for the call sites' sake (the modular transform had an initial reduction); unused.
template <int B, bool FENCE = true, bool PRE = true>
__device__ __forceinline__ void ntt_inv(double (&x)[B][E], const double* tw, double* data, int tid) {
    static_assert(B >= 1 && B <= 2, "one or two polynomials");
    if constexpr (B == 2) fft_inv2<true, FENCE>(x[0], x[1], tw, data, data + LDS_DATA, tid);
    else fft_inv1<FENCE>(x[0], tw, data, tid);
}
template <bool FENCE = true, bool PRE = true>
__device__ __forceinline__ void ntt_inv2_skew(double (&x)[2][E], const double* tw, double* d0, double* d1, int tid) {
    fft_inv2<true, FENCE>(x[0], x[1], tw, d0, d1, tid);
}
__device__ __forceinline__ void ntt_fwd3_skew(double (&x)[3][E], const double* tw, double* data, int tid) { ntt_fwd<3>(x, tw, data, tid); }

// copy the twiddle table (NC complex values) into LDS
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

// Host side: the twiddle table.  W[h] = exp(i pi a_h) with a_1 = 1/4, a_2h = a_h / 2, a_2h+1 = a_h / 2 + 1/2: node h of
// the recursion splits X^m - r_h into X^(m/2) -+ W[h], r_1 = i.  Stored as (re, im) at doubles [2h, 2h + 1].
#include <cmath>
#include <vector>
namespace fk {
inline std::vector<double> make_fft_twiddles() {
    std::vector<double> tw(N, 0.0);
    std::vector<long double> a(NC, 0.0L);   // angle / pi: dyadic rationals with at most 13 fractional bits, exact
    a[1] = 0.25L;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int h = 1; h < NC; h++) {
        tw[2 * h] = (double)cosl(pi * a[h]);
        tw[2 * h + 1] = (double)sinl(pi * a[h]);
        if (2 * h < NC) { a[2 * h] = a[h] / 2; a[2 * h + 1] = a[h] / 2 + 0.5L; }
    }
    return tw;
}
}  // namespace fk
