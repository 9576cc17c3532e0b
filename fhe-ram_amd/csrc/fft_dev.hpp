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
// backend's own contract.  Measured round-off (tests/test_gpu_fft.py, tools/fft_search.hip): <= 2^-9 on uniformly random
// limbs at any magnitude, 0.125 on the worst coherent pattern (every coefficient -2^16, six terms).  No a-priori bound below
// 1/2 is known for six terms of extreme limbs, so the contract is CHECKED: every rounding of an inverse transform feeds a
// round-off monitor (mon_note / RoMonitor below), whose maximum the host reads back and turns into FHERAM_ERR_PRECISION.
//
// Decomposition: one 512-thread workgroup (8 waves) per polynomial, 4 complex points (= 8 coefficients) per thread, i.e. two
// index bits in registers.  A swap round (v_permlane32_swap / v_permlane16_swap: one instruction per register pair) trades a
// register bit for lane bit 5 or 4, so a pass covers up to FOUR stages without touching LDS: two stages, swap with lane bit 5,
// one stage, swap with lane bit 4, one stage.  The 11 stages are passes of 3 + 4 + 4 with TWO LDS exchanges of 32 KB between
// them — exchange 0 across waves (one workgroup barrier in the inverse, two in the forward), exchange 1 inside the wave's
// own region, XOR-swizzled; both sides of both are bank-conflict free with 16-byte accesses.
// 6 (forward) / 8 (inverse) FP64 instructions per complex butterfly and 4 FMAs per complex multiply-accumulate — against 8 and
// 7 per point for a modular transform on twice as many points, with three exchanges.
// Several polynomials run HALF A PHASE apart in one wave (fft_fwd_skew / fft_inv_skew): the exchange of one is in flight while
// the wave computes the next one's pass; they share twiddles (same thread, same points) and the barriers of exchange 0.
//
// For that the "thread id" of every kernel that calls a transform is the VIRTUAL id vt() below: index bit 8 sits on lane bit
// 5 and bits 7..5 on the wave id, so that the first swap partner (lane ^ 32) holds the points 256 further on.  Consecutive
// lanes 0..31 still hold consecutive coefficients (coalesced 128-byte rows, conflict-free odd-stride LDS gathers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

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
constexpr int LDS_MON = T;       // doubles: one slot per thread for the round-off monitor (behind the exchange buffers)
constexpr int MON_OFF = LDS_TW + BMAX * LDS_DATA;
constexpr size_t LDS_BYTES = (size_t)(LDS_TW + BMAX * LDS_DATA + LDS_MON) * sizeof(double);
// The twiddle table in global memory is N doubles followed by the monitor's two words (one table per context):
//   [0]      slot of W[0], which the recursion never reads: its FIRST DWORD is the monitor mode (0: nothing is reported, 1: one
//            coefficient per thread and transform, 2: every coefficient);
//            [1] stays 0.0 (its upper dword is the split barrier's counter in LDS)
//   [N]      max |x - rint(x)| over every monitored rounding so far, as the bits of a non-negative double (unsigned max)
//   [N + 1]  device-visible address of a pinned host word that is set to 1 once a round-off above MON_LIMIT was seen
constexpr int TW_GLOBAL = N + 2;
constexpr double MON_LIMIT = 0.375;   // halfway between the largest round-off any search has produced (0.25: tools/fft_search.hip) and failure (0.5)

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

// ---- a split barrier in LDS for the fences of a limb loop ------------------------------------------------------------------------
// A workgroup barrier costs a chain step ~0.35 us (the waves wait for the slowest).  The fence in front of an inverse batch only
// says "every wave has finished reading the previous batch's buffers across waves" — and the wave that gets there first has a fold,
// operand loads and six polynomial products to do before it stores into those buffers again.  So: ARRIVE (one LDS add by lane 0)
// behind the wave's last cross-wave load of a batch, WAIT (poll one LDS word) in front of the next batch's first store.  The
// counter is dword 3 of twiddle slot 0 (never a twiddle: twpos >= 1; zeroed explicitly by both table copies below — load_twiddles,
// twiddles_commit — whose __syncthreads publishes it; every kernel runs one of them before its first FENCE == 2 transform).
// LDS operations of a wave execute in order, so the arrival is performed behind the loads it stands for.  The counter only grows,
// and a wave arrives for batch m + 1 behind that batch's rendezvous barrier, which every wave reaches behind its own arrival for
// batch m: arrivals come in rounds of eight, and "a multiple of eight" means "every wave has arrived for every batch so far" — no
// state is carried between calls.  (The rendezvous inside a transform stays the hardware barrier: polling it was measured slower.)
static_assert(T / 64 == 8, "sb_wait_free counts arrivals in rounds of T / 64 waves with a power-of-two mask");
constexpr unsigned SB_WAVES = T / 64;
__device__ __forceinline__ unsigned sb_addr(const double* tw) { return (unsigned)(size_t)tw + 12u; }
__device__ __forceinline__ bool sb_lane0() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u; }
__device__ __forceinline__ void sb_arrive_free(unsigned a) {
    if (sb_lane0()) asm volatile("ds_add_u32 %0, %1" : : "v"(a), "v"(1u) : "memory");
}
__device__ __forceinline__ void sb_wait_free(unsigned a) {
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if ((__builtin_amdgcn_readfirstlane(v) & (SB_WAVES - 1u)) == 0u) break;
        __builtin_amdgcn_s_sleep(1);
    }
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

// the same two with the twiddle i w: the twiddles of the two blocks a stage splits one block into are W[2h] and W[2h + 1] =
// i W[2h] (their angles differ by pi / 2), so only the first is read
__device__ __forceinline__ void bf_i(d2& u, d2& v, const d2 w) {
    double lr = __builtin_fma(-w.y, v.x, u.x);
    lr = __builtin_fma(-w.x, v.y, lr);
    double li = __builtin_fma(-w.y, v.y, u.y);
    li = __builtin_fma(w.x, v.x, li);
    v.x = __builtin_fma(2.0, u.x, -lr);
    v.y = __builtin_fma(2.0, u.y, -li);
    u.x = lr;
    u.y = li;
}
__device__ __forceinline__ void gs_i(d2& a, d2& b, const d2 w) {
    const double dr = a.x - b.x, di = a.y - b.y;
    a.x = a.x + b.x;
    a.y = a.y + b.y;
    b.x = __builtin_fma(di, w.x, dr * -w.y);
    b.y = __builtin_fma(-dr, w.x, di * -w.y);
}

// ---- register <-> lane transpositions ------------------------------------------------------------------------------------------
// lane_swap<L>(a, b): lanes l and l ^ 2^L exchange so that afterwards lanes with bit L clear hold (a, a of the partner) and lanes
// with it set (b of the partner, b): the register index and lane bit L have traded places.
template <int L>
__device__ __forceinline__ void lane_swap_u32(unsigned& a, unsigned& b) {
    if constexpr (L == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0]; b = r[1];
    } else {
        static_assert(L == 4, "lane bits 5 and 4");
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        a = r[0]; b = r[1];
    }
}
template <int L>
__device__ __forceinline__ void lane_swap(double& a, double& b) {
    unsigned al = (unsigned)__double2loint(a), ah = (unsigned)__double2hiint(a);
    unsigned bl = (unsigned)__double2loint(b), bh = (unsigned)__double2hiint(b);
    lane_swap_u32<L>(al, bl);
    lane_swap_u32<L>(ah, bh);
    a = __hiloint2double((int)ah, (int)al);
    b = __hiloint2double((int)bh, (int)bl);
}
template <int L>
__device__ __forceinline__ void lane_swap(d2& a, d2& b) {
    double ar = a.x, ai = a.y, br = b.x, bi = b.y;
    lane_swap<L>(ar, br);
    lane_swap<L>(ai, bi);
    a.x = ar; a.y = ai; b.x = br; b.y = bi;
}

// ---- geometry -------------------------------------------------------------------------------------------------------------------
// Index bits b10..b0 of a complex point; a hardware thread is (wave W, lane bits L5..L0); y[2 hi + lo] are its four points.
//   natural  : y (b10, b9)  L5 = b8            W = (b7 b6 b5)   L4..L0 = b4..b0
//              pass 0: stage b10 (hi pairs), b9 (lo pairs), swap lo <-> L5, b8 (lo pairs)
//   exchange 0 (LDS, across waves)
//   then     : y (b7, b6)   L5 = b5  L4 = b4   W = (b10 b9 b8)  L3..L0 = b3..b0
//              pass 1: b7 (hi), b6 (lo), swap lo <-> L5, b5 (lo), swap hi <-> L4, b4 (hi)           now y (b4, b5), L5 = b6, L4 = b7
//   exchange 1 (LDS, inside the wave's own 256 slots, swizzled j ^ (j >> 4))
//   then     : y (b3, b2)   L5 = b1  L4 = b0   L3..L0 = b7 b6 b5 b4
//              pass 2: b3 (hi), b2 (lo), swap lo <-> L5, b1 (lo), swap hi <-> L4, b0 (hi)           now y (b0, b1), L5 = b2, L4 = b3
// The transform-domain order (what a thread holds after pass 2) is what prepared operands are stored in and the inverse expects.
// Twiddle W[node] of the recursion (node 2^s + J: stage s, block J = the s leading index bits) is stored at twpos(node): the
// natural position up to stage 7, and for the last three stages "transposed" (the bits a lane does not share with its
// neighbours select a block of 128, the seven leading bits the slot inside it) so that the lanes of a wave read consecutive slots.
__host__ __device__ constexpr int twpos(int node) {
    int s = 0;
    while ((2 << s) <= node) s++;
    if (s <= 7) return node;
    const int J = node - (1 << s), low = s - 7;
    return (1 << s) + (J & ((1 << low) - 1)) * 128 + (J >> low);
}
struct XAddr {
    int x0a;   // exchange 0, natural side : L5 * 512 + W * 32 + ll                        + hi * 1024 + lo * 256
    int x0b;   // exchange 0, far side     : W * 256 + L5 * 32 + ll                        + hi * 128 + lo * 64
    int x1a;   // exchange 1, after pass 1 : W * 256 + swz(L4 * 128 + L5 * 64 + (ll & 15))   ^ (hi * 17 | lo * 34)
    int x1b;   // exchange 1, before pass 2: W * 256 + swz((ll & 15) * 16 + L5 * 2 + L4)     ^ (hi * 8 | lo * 4)
    int l5, w;
    int t4;    // 64 + 8 W + 4 L4 + 2 L5                 stage b4: tw[t4 + lo]
    int p;     // 16 W + (ll & 15): the seven leading bits in pass 2;  stage b3: tw[128 + p], b2: tw[256 + hi * 128 + p],
               // b1: tw[512 + (2 hi + L5) * 128 + p], b0: tw[1024 + (4 L4 + 2 L5 + lo) * 128 + p]
    int t1, t0;   // 512 + L5 * 128 + p,  1024 + (4 L4 + 2 L5) * 128 + p
};
__device__ __forceinline__ int swz(int j) { return j ^ (j >> 4); }
// tid = vt(threadIdx.x): W = (tid >> 5) & 7, L5 = tid >> 8, L4..L0 = tid & 31
__device__ __forceinline__ XAddr xaddr(int tid) {
    XAddr a;
    const int W = (tid >> 5) & 7, ll = tid & 31, L5 = tid >> 8, L4 = ll >> 4, m4 = ll & 15;
    a.x0a = L5 * 512 + W * 32 + ll;
    a.x0b = W * 256 + L5 * 32 + ll;
    a.x1a = W * 256 + swz(L4 * 128 + L5 * 64 + m4);
    a.x1b = W * 256 + swz(m4 * 16 + L5 * 2 + L4);
    a.l5 = L5; a.w = W;
    a.t4 = 64 + 8 * W + 4 * L4 + 2 * L5;
    a.p = 16 * W + m4;
    a.t1 = 512 + L5 * 128 + a.p;
    a.t0 = 1024 + (4 * L4 + 2 * L5) * 128 + a.p;
    return a;
}
// twiddles of a pass: a: first stage (hi pairs); b: second stage (lo pairs of hi = 0; hi = 1 uses i b); d e: third stage (lo pairs,
// by hi); f: fourth stage (hi pairs of lo = 0; lo = 1 uses i f).  Pass 0's a and b are the constants W[1], W[2].
struct Tw7 { d2 a, b, d, e, f; };
struct Tw5 { d2 d, e; };
constexpr double W1_RE = 0.70710678118654752440, W1_IM = 0.70710678118654752440;   // exp(i pi / 4)
constexpr double W2_RE = 0.92387953251128675613, W2_IM = 0.38268343236508977173;   // exp(i pi / 8)
__device__ __forceinline__ void tw_p0(Tw5& t, const d2* tw, const XAddr& xa) { t.d = tw[4 + xa.l5]; t.e = tw[6 + xa.l5]; }
__device__ __forceinline__ void tw_p1(Tw7& t, const d2* tw, const XAddr& xa) {
    t.a = tw[8 + xa.w]; t.b = tw[16 + 2 * xa.w]; t.d = tw[32 + 4 * xa.w + xa.l5]; t.e = tw[34 + 4 * xa.w + xa.l5];
    t.f = tw[xa.t4];
}
__device__ __forceinline__ void tw_p2(Tw7& t, const d2* tw, const XAddr& xa) {
    t.a = tw[128 + xa.p]; t.b = tw[256 + xa.p]; t.d = tw[xa.t1]; t.e = tw[xa.t1 + 256];
    t.f = tw[xa.t0];
}
__device__ __forceinline__ void f_pass3(d2 (&y)[4], const Tw5& t) {
    d2 w1, w2;
    w1.x = W1_RE; w1.y = W1_IM; w2.x = W2_RE; w2.y = W2_IM;
    bf(y[0], y[2], w1); bf(y[1], y[3], w1);
    bf(y[0], y[1], w2); bf_i(y[2], y[3], w2);
    lane_swap<5>(y[0], y[1]); lane_swap<5>(y[2], y[3]);
    bf(y[0], y[1], t.d); bf(y[2], y[3], t.e);
}
__device__ __forceinline__ void f_pass4(d2 (&y)[4], const Tw7& t) {
    bf(y[0], y[2], t.a); bf(y[1], y[3], t.a);
    bf(y[0], y[1], t.b); bf_i(y[2], y[3], t.b);
    lane_swap<5>(y[0], y[1]); lane_swap<5>(y[2], y[3]);
    bf(y[0], y[1], t.d); bf(y[2], y[3], t.e);
    lane_swap<4>(y[0], y[2]); lane_swap<4>(y[1], y[3]);
    bf(y[0], y[2], t.f); bf_i(y[1], y[3], t.f);
}
__device__ __forceinline__ void i_pass4(d2 (&y)[4], const Tw7& t) {
    gs(y[0], y[2], t.f); gs_i(y[1], y[3], t.f);
    lane_swap<4>(y[0], y[2]); lane_swap<4>(y[1], y[3]);
    gs(y[0], y[1], t.d); gs(y[2], y[3], t.e);
    lane_swap<5>(y[0], y[1]); lane_swap<5>(y[2], y[3]);
    gs(y[0], y[1], t.b); gs_i(y[2], y[3], t.b);
    gs(y[0], y[2], t.a); gs(y[1], y[3], t.a);
}
__device__ __forceinline__ void i_pass3(d2 (&y)[4], const Tw5& t) {
    d2 w1, w2;
    w1.x = W1_RE; w1.y = W1_IM; w2.x = W2_RE; w2.y = W2_IM;
    gs(y[0], y[1], t.d); gs(y[2], y[3], t.e);
    lane_swap<5>(y[0], y[1]); lane_swap<5>(y[2], y[3]);
    gs(y[0], y[1], w2); gs_i(y[2], y[3], w2);
    gs(y[0], y[2], w1); gs(y[1], y[3], w1);
}
// the two sides of the LDS exchanges
__device__ __forceinline__ void x0a_w(const d2 (&y)[4], d2* buf, const XAddr& xa) { buf[xa.x0a] = y[0]; buf[xa.x0a + 256] = y[1]; buf[xa.x0a + 1024] = y[2]; buf[xa.x0a + 1280] = y[3]; }
__device__ __forceinline__ void x0a_r(d2 (&y)[4], const d2* buf, const XAddr& xa) { y[0] = buf[xa.x0a]; y[1] = buf[xa.x0a + 256]; y[2] = buf[xa.x0a + 1024]; y[3] = buf[xa.x0a + 1280]; }
__device__ __forceinline__ void x0b_w(const d2 (&y)[4], d2* buf, const XAddr& xa) { buf[xa.x0b] = y[0]; buf[xa.x0b + 64] = y[1]; buf[xa.x0b + 128] = y[2]; buf[xa.x0b + 192] = y[3]; }
__device__ __forceinline__ void x0b_r(d2 (&y)[4], const d2* buf, const XAddr& xa) { y[0] = buf[xa.x0b]; y[1] = buf[xa.x0b + 64]; y[2] = buf[xa.x0b + 128]; y[3] = buf[xa.x0b + 192]; }
__device__ __forceinline__ void x1a_w(const d2 (&y)[4], d2* buf, const XAddr& xa) { buf[xa.x1a] = y[0]; buf[xa.x1a ^ 34] = y[1]; buf[xa.x1a ^ 17] = y[2]; buf[xa.x1a ^ 51] = y[3]; }
__device__ __forceinline__ void x1a_r(d2 (&y)[4], const d2* buf, const XAddr& xa) { y[0] = buf[xa.x1a]; y[1] = buf[xa.x1a ^ 34]; y[2] = buf[xa.x1a ^ 17]; y[3] = buf[xa.x1a ^ 51]; }
__device__ __forceinline__ void x1b_w(const d2 (&y)[4], d2* buf, const XAddr& xa) { buf[xa.x1b] = y[0]; buf[xa.x1b ^ 4] = y[1]; buf[xa.x1b ^ 8] = y[2]; buf[xa.x1b ^ 12] = y[3]; }
__device__ __forceinline__ void x1b_r(d2 (&y)[4], const d2* buf, const XAddr& xa) { y[0] = buf[xa.x1b]; y[1] = buf[xa.x1b ^ 4]; y[2] = buf[xa.x1b ^ 8]; y[3] = buf[xa.x1b ^ 12]; }
// natural order: a[k] = coefficient tid + T k, point k' = (a[k'], a[k' + 4]); transform domain: point m = (a[2m], a[2m + 1])
__device__ __forceinline__ void nat_in(d2 (&y)[4], const double (&a)[E]) {
#pragma unroll
    for (int k = 0; k < 4; k++) { y[k].x = a[k]; y[k].y = a[k + 4]; }
}
// ---- round-off monitor ------------------------------------------------------------------------------------------------------------
// The rounded output of an inverse transform is the exact integer only while the accumulated FP64 round-off stays below 1/2.
// Every rounding on the path reports |x - rint(x)| into the thread's own LDS slot (ds_max on the bits of a non-negative
// double: no return value, nothing waits for it); RoMonitor, constructed at the top of every kernel that rounds, folds the
// slots into the context's maximum when the kernel ends — one comparison per wave against the value the kernel started from (a
// scalar load at kernel start), one global atomic only from a wave that raises it.
// Mode (first dword of the table's slot 0, copied to LDS with the table; read at the START of an inverse transform, with its
// first twiddles, so that the rounding itself waits for nothing): 0 nothing is reported, 1 ONE coefficient per thread and
// transform — which of the thread's eight is a compile-time choice of the call site (SEL), so the sites of a kernel cover
// different classes tid + 512 k —, 2 all eight (fheram_config.monitor = 2, what `safe` selects: +2.9 % of a step, the
// sampled form +0.x %: profiles/r06_experiments.txt).
#ifndef FK_MONITOR
#define FK_MONITOR 1     // 0: compiled out (A/B measurements only)
#endif
__device__ __forceinline__ unsigned mon_mode(const double* tw_lds) {
#if FK_MONITOR
    return __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(tw_lds)[0]);
#else
    return 0u;
#endif
}
template <int SEL>
__device__ __forceinline__ void mon_note(const double* tw_lds, int tid, const d2 (&y)[4], const double (&a)[E], unsigned mode) {
#if FK_MONITOR
    static_assert(SEL >= 0 && SEL < E, "one of the thread's eight coefficients");
    double m = __builtin_fabs((SEL < 4 ? y[SEL & 3].x : y[SEL & 3].y) - a[SEL]);
    if (mode > 1u) {   // (wave uniform)
#pragma unroll
        for (int k = 0; k < 4; k++) m = __builtin_fmax(m, __builtin_fmax(__builtin_fabs(y[k].x - a[k]), __builtin_fabs(y[k].y - a[k + 4])));
    }
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(const_cast<double*>(tw_lds) + MON_OFF + tid);
    __hip_atomic_fetch_max(slot, (unsigned long long)__double_as_longlong(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}
struct RoMonitor {
    double* lds_;
    const double* twg_;
    unsigned long long seen_;   // the context's maximum when this kernel started (wave uniform: scalar registers)
    bool on_;
    // lds: the kernel's dynamic LDS (LDS_BYTES); tw_g: the context's table in global memory; on = false: a kernel variant without LDS
    __device__ __forceinline__ RoMonitor(double* lds, const double* tw_g, bool on = true) : lds_(lds), twg_(tw_g), seen_(~0ull), on_(on) {
#if FK_MONITOR
        if (on_) {
            lds_[MON_OFF + vt((int)threadIdx.x)] = 0.0;     // thread-private: no barrier between this, mon_note and the flush
            // mode 0 (first dword of the table): nothing is ever reported.  Both words are wave uniform: scalar loads, in flight
            // while the kernel starts up.
            const unsigned mode = reinterpret_cast<const unsigned*>(tw_g)[0];
            const unsigned long long seen = reinterpret_cast<const unsigned long long*>(tw_g)[N];
            seen_ = mode ? seen : ~0ull;
        }
#endif
    }
    __device__ __forceinline__ ~RoMonitor() {
#if FK_MONITOR
        if (!on_) return;
        unsigned long long b = (unsigned long long)__double_as_longlong(lds_[MON_OFF + vt((int)threadIdx.x)]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)b, off, 64);
            b = o > b ? o : b;
        }
        if (b > seen_ && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) {
            unsigned long long* gmax = reinterpret_cast<unsigned long long*>(const_cast<double*>(twg_) + N);
            __hip_atomic_fetch_max(gmax, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__longlong_as_double((long long)b) > MON_LIMIT) {
                unsigned* hflag = reinterpret_cast<unsigned*>((size_t)__double_as_longlong(twg_[N + 1]));
                if (hflag) __hip_atomic_store(hflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
#endif
    }
};
template <bool ROUND = true, int SEL = 0>
__device__ __forceinline__ void nat_out(const d2 (&y)[4], double (&a)[E], const double* tw_lds, int tid, unsigned mode) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (ROUND) { a[k] = __builtin_rint(y[k].x); a[k + 4] = __builtin_rint(y[k].y); }
        else { a[k] = y[k].x; a[k + 4] = y[k].y; }   // (round-off measurements only)
    }
    if constexpr (ROUND) mon_note<SEL>(tw_lds, tid, y, a, mode);
}
__device__ __forceinline__ void dom_in(d2 (&y)[4], const double (&a)[E]) {
#pragma unroll
    for (int m = 0; m < 4; m++) { y[m].x = a[2 * m]; y[m].y = a[2 * m + 1]; }
}
__device__ __forceinline__ void dom_out(const d2 (&y)[4], double (&a)[E]) {
#pragma unroll
    for (int m = 0; m < 4; m++) { a[2 * m] = y[m].x; a[2 * m + 1] = y[m].y; }
}

// ---- B forward transforms (B <= 3), half a phase apart ---------------------------------------------------------------------------
// in : x[b][k] = coefficient tid + T k of polynomial b (|.| < 2^20), tid = vt(threadIdx.x);  d[b]: its exchange buffer (NC complex slots)
// out: x[b][2m], x[b][2m+1] = (re, im) of the transform at this thread's point m (m < 4)
// Starts with a workgroup barrier in front of its first LDS write (every earlier LDS read of the workgroup has been issued and
// waited for by then); ends with reads of the wave's own region.
template <int B>
__device__ __forceinline__ void fft_fwd_skew(double (&x)[B][E], const double* tw_, double* const (&d)[B], int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    const XAddr xa = xaddr(tid);
    d2 y[B][4];
    Tw5 t0; Tw7 t1, t2;
    tw_p0(t0, tw, xa);
    nat_in(y[0], x[0]); f_pass3(y[0], t0);
    tw_p1(t1, tw, xa);
    lds_barrier();                       // every earlier access of the workgroup to these buffers is done: stores may cross waves now
#pragma unroll
    for (int b = 0; b < B; b++) {
        if (b > 0) { nat_in(y[b], x[b]); f_pass3(y[b], t0); }
        x0a_w(y[b], reinterpret_cast<d2*>(d[b]), xa);
    }
    lds_barrier();
#pragma unroll
    for (int b = 0; b < B; b++) x0b_r(y[b], reinterpret_cast<d2*>(d[b]), xa);
#pragma unroll
    for (int b = 0; b < B; b++) {
        f_pass4(y[b], t1);
        x1a_w(y[b], reinterpret_cast<d2*>(d[b]), xa); wave_lds_fence(); x1b_r(y[b], reinterpret_cast<d2*>(d[b]), xa);
        if (b == 0) tw_p2(t2, tw, xa);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < B; b++) { f_pass4(y[b], t2); dom_out(y[b], x[b]); }
}
// ---- B inverse transforms (B <= 3), half a phase apart ----------------------------------------------------------------------------
// in : x[b] as fft_fwd_skew leaves it (accumulated products against prepared operands, which carry the 1/n)
// out: x[b][k] = coefficient tid + T k, ROUNDED to the nearest integer (an exact integer-valued double)
// FENCE: workgroup barrier in front of the first LDS write.  Needed when another wave may still be reading this buffer
// across waves (the natural side of exchange 0 of the previous inverse transform in the SAME buffer, or a kernel's own gathers).
// FENCE 0: nothing; 1: workgroup barrier; 2: wait until every wave has finished the cross-wave reads of the previous FENCE-2 batch
// (the batches of one limb loop: same buffers, nothing but register work and operand loads in between), and say so at the end.
template <int B, int FENCE, bool ROUND = true>
__device__ __forceinline__ void fft_inv_skew(double (&x)[B][E], const double* tw_, double* const (&d)[B], int tid) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    const XAddr xa = xaddr(tid);
    d2 y[B][4];
    Tw7 t2, t1; Tw5 t0;
    const unsigned mon = ROUND ? mon_mode(tw_) : 0u;
    tw_p2(t2, tw, xa);
#pragma unroll
    for (int b = 0; b < B; b++) dom_in(y[b], x[b]);
    if constexpr (FENCE == 1) lds_barrier();
    if constexpr (FENCE == 2) sb_wait_free(sb_addr(tw_));
#pragma unroll
    for (int b = 0; b < B; b++) {
        i_pass4(y[b], t2);
        x1b_w(y[b], reinterpret_cast<d2*>(d[b]), xa); wave_lds_fence(); x1a_r(y[b], reinterpret_cast<d2*>(d[b]), xa);
        if (b == 0) tw_p1(t1, tw, xa);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < B; b++) {
        i_pass4(y[b], t1);
        x0b_w(y[b], reinterpret_cast<d2*>(d[b]), xa);
        if (b == 0) tw_p0(t0, tw, xa);
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
#pragma unroll
    for (int b = 0; b < B; b++) x0a_r(y[b], reinterpret_cast<d2*>(d[b]), xa);
#pragma unroll
    for (int b = 0; b < B; b++) {
        i_pass3(y[b], t0);
        if (b == 0) nat_out<ROUND, 0>(y[b], x[b], tw_, tid, mon); else if (b == 1) nat_out<ROUND, 3>(y[b], x[b], tw_, tid, mon); else nat_out<ROUND, 6>(y[b], x[b], tw_, tid, mon);
    }
    if constexpr (FENCE == 2) sb_arrive_free(sb_addr(tw_));
}

// ---- ONE inverse transform with work of the caller's between its phases -------------------------------------------------------------
// hook(std::integral_constant<int, s>) is called at four points: s = 0 behind the issue of exchange 1, s = 1 behind the stores of
// exchange 0 (in front of the rendezvous), s = 2 behind the issue of its loads, s = 3 behind the last pass (in front of the rounding).
// The callers stream prepared operands through there (global loads requested, multiply-accumulates of the NEXT output limb
// taken): the operand stream of a chain step — 0.75 to 1.5 MB per step through the CU's ~57 B/clk load path — then runs under the
// transforms instead of beside them (tools/fft_bench.hip, k_stream: a transform pair with 192 KB requested in front of it takes
// 3.19 us against 3.04 without the loads and 1.60 for the loads alone).
template <int FENCE, bool ROUND = true, int MSEL = 5, class Hook>
__device__ __forceinline__ void fft_inv1_hooked(double (&x)[1][E], const double* tw_, double* d0, int tid, Hook&& hook) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    d2* buf = reinterpret_cast<d2*>(d0);
    const XAddr xa = xaddr(tid);
    d2 y[4];
    Tw7 t2, t1; Tw5 t0;
    const unsigned mon = ROUND ? mon_mode(tw_) : 0u;
    tw_p2(t2, tw, xa);
    dom_in(y, x[0]);
    if constexpr (FENCE == 1) lds_barrier();
    if constexpr (FENCE == 2) sb_wait_free(sb_addr(tw_));
    i_pass4(y, t2);
    x1b_w(y, buf, xa); wave_lds_fence(); x1a_r(y, buf, xa);
    __builtin_amdgcn_sched_barrier(0);
    hook(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    tw_p1(t1, tw, xa);   // (behind the hook: 20 registers fewer are live across it)
    i_pass4(y, t1);
    x0b_w(y, buf, xa);
    __builtin_amdgcn_sched_barrier(0);
    hook(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    x0a_r(y, buf, xa);
    tw_p0(t0, tw, xa);
    __builtin_amdgcn_sched_barrier(0);
    hook(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    i_pass3(y, t0);
    if constexpr (FENCE == 2) sb_arrive_free(sb_addr(tw_));   // (behind the last cross-wave load, whose data pass 3 has consumed)
    __builtin_amdgcn_sched_barrier(0);
    hook(std::integral_constant<int, 3>{});
    __builtin_amdgcn_sched_barrier(0);
    nat_out<ROUND, MSEL>(y, x[0], tw_, tid, mon);
}

// ---- TWO inverse transforms half a phase apart with work of the caller's between their phases (six places) ------------------------------
template <int FENCE, int MSEL = 1, class Hook>
__device__ __forceinline__ void fft_inv2_hooked(double (&x)[2][E], const double* tw_, double* d0, double* d1, int tid, Hook&& hook) {
    const d2* tw = reinterpret_cast<const d2*>(tw_);
    d2* const buf[2] = {reinterpret_cast<d2*>(d0), reinterpret_cast<d2*>(d1)};
    const XAddr xa = xaddr(tid);
    d2 y[2][4];
    Tw7 t2, t1; Tw5 t0;
    const unsigned mon = mon_mode(tw_);
    tw_p2(t2, tw, xa);
    dom_in(y[0], x[0]); dom_in(y[1], x[1]);
    if constexpr (FENCE == 1) lds_barrier();
    if constexpr (FENCE == 2) sb_wait_free(sb_addr(tw_));
#define FK_HOOK(s) __builtin_amdgcn_sched_barrier(0); hook(std::integral_constant<int, s>{}); __builtin_amdgcn_sched_barrier(0)
    i_pass4(y[0], t2);
    x1b_w(y[0], buf[0], xa); wave_lds_fence(); x1a_r(y[0], buf[0], xa);
    tw_p1(t1, tw, xa);
    FK_HOOK(0);
    i_pass4(y[1], t2);
    x1b_w(y[1], buf[1], xa); wave_lds_fence(); x1a_r(y[1], buf[1], xa);
    FK_HOOK(1);
    i_pass4(y[0], t1);
    x0b_w(y[0], buf[0], xa);
    tw_p0(t0, tw, xa);
    FK_HOOK(2);
    i_pass4(y[1], t1);
    x0b_w(y[1], buf[1], xa);
    FK_HOOK(3);
    lds_barrier();
    x0a_r(y[0], buf[0], xa);
    x0a_r(y[1], buf[1], xa);
    FK_HOOK(4);
    i_pass3(y[0], t0); nat_out<true, MSEL>(y[0], x[0], tw_, tid, mon);
    FK_HOOK(5);
    i_pass3(y[1], t0); nat_out<true, (MSEL + 3) % E>(y[1], x[1], tw_, tid, mon);
#undef FK_HOOK
    if constexpr (FENCE == 2) sb_arrive_free(sb_addr(tw_));
}

// ---- the interface the kernels use (B polynomials at a time; data = B consecutive exchange buffers of LDS_DATA doubles) --------
template <int B>
__device__ __forceinline__ void ntt_fwd(double (&x)[B][E], const double* tw, double* data, int tid) {
    static_assert(B >= 1 && B <= BMAX, "one to three polynomials");
    if constexpr (B == 1) { double* const d[1] = {data}; fft_fwd_skew<1>(x, tw, d, tid); }
    if constexpr (B == 2) { double* const d[2] = {data, data + LDS_DATA}; fft_fwd_skew<2>(x, tw, d, tid); }
    if constexpr (B == 3) { double* const d[3] = {data, data + LDS_DATA, data + 2 * LDS_DATA}; fft_fwd_skew<3>(x, tw, d, tid); }
}
template <int B, bool FENCE = true>
__device__ __forceinline__ void ntt_inv(double (&x)[B][E], const double* tw, double* data, int tid) {
    static_assert(B >= 1 && B <= 2, "one or two polynomials");
    if constexpr (B == 1) { double* const d[1] = {data}; fft_inv_skew<1, FENCE ? 1 : 0>(x, tw, d, tid); }
    else { double* const d[2] = {data, data + LDS_DATA}; fft_inv_skew<2, FENCE ? 1 : 0>(x, tw, d, tid); }
}
template <bool FENCE = true>
__device__ __forceinline__ void ntt_inv2_skew(double (&x)[2][E], const double* tw, double* d0, double* d1, int tid) {
    double* const d[2] = {d0, d1};
    fft_inv_skew<2, FENCE ? 1 : 0>(x, tw, d, tid);
}
// the pairs of one limb loop (same two buffers every time, register work and operand loads in between): fenced by the free counter
__device__ __forceinline__ void ntt_inv2_loop(double (&x)[2][E], const double* tw, double* d0, double* d1, int tid) {
    double* const d[2] = {d0, d1};
    fft_inv_skew<2, 2>(x, tw, d, tid);
}
__device__ __forceinline__ void ntt_inv1_loop(double (&x)[1][E], const double* tw, double* d0, int tid) {
    double* const d[1] = {d0};
    fft_inv_skew<1, 2>(x, tw, d, tid);
}
__device__ __forceinline__ void ntt_fwd3_skew(double (&x)[3][E], const double* tw, double* data, int tid) { ntt_fwd<3>(x, tw, data, tid); }

// copy the twiddle table (NC complex values) into LDS
__device__ __forceinline__ void load_twiddles(double* tw_lds, const double* __restrict__ tw_g, int tid) {
#pragma unroll
    for (int k = 0; k < E; k++) tw_lds[tid + T * k] = (k == 0 && tid == 1) ? 0.0 : tw_g[tid + T * k];   // (slot 0's second double holds the split barrier's counter: zero, whatever the table says)
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
    for (int k = 0; k < E; k++) tw_lds[tid + T * k] = (k == 0 && tid == 1) ? 0.0 : r.v[k];   // (see load_twiddles)
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
inline std::vector<double> make_fft_twiddles() {   // in the device layout: node h at complex position twpos(h); [N], [N + 1]: the monitor's words (zero)
    std::vector<double> tw(TW_GLOBAL, 0.0);
    std::vector<long double> a(NC, 0.0L);   // angle / pi: dyadic rationals with at most 13 fractional bits, exact
    a[1] = 0.25L;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int h = 1; h < NC; h++) {
        tw[2 * twpos(h)] = (double)cosl(pi * a[h]);
        tw[2 * twpos(h) + 1] = (double)sinl(pi * a[h]);
        if (2 * h < NC) { a[2 * h] = a[h] / 2; a[2 * h + 1] = a[h] / 2 + 0.5L; }
    }
    return tw;
}
}  // namespace fk
