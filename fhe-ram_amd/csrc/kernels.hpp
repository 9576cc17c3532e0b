// Fused GLWE-level kernels of the FHE-RAM path for gfx950.  One or two workgroups (T = N/E threads)
// per ciphertext operation; grids: blockIdx.x = row inside a sub-RAM, blockIdx.y = sub-RAM,
// blockIdx.z = output column when the operation is split by column (NCO = 1).
//
//   k_prepare      GGSWPrepared::prepare / key prepare      (coordinate_prepared.rs:104-116, keys.rs:57-71)
//   k_ext_product  glwe_external_product(_inplace)           (coordinate_prepared.rs:147-177)
//   k_keyswitch<>  glwe_automorphism family, trace step, packer combine, GGSW inversion
//                  (ram.rs:435,457,540,572,616,621; coordinate_prepared.rs:138)
//   k_sub_add_norm / k_rotate   write-path elementwise steps (ram.rs:574-576,617-629)
//   k_read_chain / k_write_chain (round 4)  a row's product chain and trace chain (ram.rs:429-435,502-514 / 612-646) as one
//                  launch each: ep_step_r + ks_trace_l, the normalisation in closed form, steps handed over in LDS / registers
//   k_keyswitch_chain / k_ext_product_chain(_r)  the same chains on their own; k_pair_z  the column-split packer combine
//   k_trace_tail / k_chain_mid  dependent chains on few ciphertexts with in-kernel hand-offs between workgroups of one XCD (k_trace_tail: coordinate 1's
//                  products in front of the trace, round 6)
//   chain_kernels.inc           k_keyswitch_chain / k_read_chain in two register budgets (included twice: 240 registers beside the gate wave, 256 otherwise)
//
// Device GLWE layout: int32 [limb][col][N] (the host's int64 layout narrowed; limbs are
// normalised to 17 bits so nothing is lost).  Prepared operands: double, transform domain,
// scaled by 1/n (n = N/2 complex points: the inverse transform's factor), stored so that thread t's
// elements (2kk, 2kk+1) are one 16-byte word at [kk*T + t] (coalesced 16 B/lane loads).
// Every kernel that rounds the output of an inverse transform constructs a RoMonitor (fft_dev.hpp) first.
#pragma once
#include "fft_dev.hpp"
#include <type_traits>

namespace fk {

// Diagnostic build only (-DFK_STAMP): s_memtime stamps of workgroup (0,0,0), lane 0, written to a
// buffer nothing else reads.  Never compiled into the shipped library.
#ifdef FK_STAMP
__device__ unsigned long long g_stamps[192];
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {           \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            g_stamps[i] = t_;                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define STAMPZ(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (blockIdx.z == 0 || blockIdx.z == 3)) { \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            g_stamps[(blockIdx.z == 0 ? 128 : 144) + (i)] = t_;                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#ifndef FK_STAMP_STEP
#define FK_STAMP_STEP 2
#endif
#ifndef FK_STAMP_STEP_Y
#define FK_STAMP_STEP_Y 3
#endif
// k_trace_tail: every member of group 0, step FK_STAMP_STEP: g_stamps[stamp * 24 + member], 100 MHz real-time clock (the same on every CU)
#define TSTAMP(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && s == FK_STAMP_STEP && g == 0) {                                    \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            g_stamps[(i) * 24 + m] = t_;                                                           \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
// k_chain_mid: every member of ciphertext 0, step FK_STAMP_STEP: g_stamps[stamp * 24 + member]
#define MSTAMP(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (threadIdx.x == 0 && s == FK_STAMP_STEP && ctg == 0) {                                  \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            g_stamps[(i) * 24 + m] = t_;                                                           \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
// ks_trace_y inside a chain launch: wave 0 of workgroup (0,0,0), step FK_STAMP_STEP of the chain
#define YSTAMP(i)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (stamp_on && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                  \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            g_stamps[i] = t_;                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define YSTAMP_STEP(i) ((i) == FK_STAMP_STEP_Y)
#else
#define YSTAMP_STEP(i) false
#define YSTAMP(i) do { } while (0)
#define STAMP(i) do { } while (0)
#define STAMPZ(i) do { } while (0)
#define TSTAMP(i) do { } while (0)
#define MSTAMP(i) do { } while (0)
#endif

struct GlweRef {   // p + y*sy + x*sx  (int32 elements)
    int32_t* p;
    long sy, sx;
};
__device__ __forceinline__ int32_t* at(const GlweRef& r) { return r.p + (long)blockIdx.y * r.sy + (long)blockIdx.x * r.sx; }
__device__ __forceinline__ long glwe_off(int limb, int col) { return (long)(limb * 2 + col) * N; }

// Global accesses as (wave-uniform 64-bit base in scalar registers) + (32-bit lane offset): `global_load_dword v, v_off,
// s[base]`.  The limb / column / coefficient-block part of an address is the same for every lane; left to itself the
// compiler adds it to the lane's 64-bit address with two VALU instructions per access (~100 per key-switch input, ~130 per
// output).  The empty asm pins the base to SGPRs and keeps the zero extension of the offset next to the access, which is
// what instruction selection needs to pick the scalar-base form.  `base` MUST be wave uniform.
#ifndef FK_SADDR
#define FK_SADDR 1
#endif
typedef const __attribute__((address_space(1))) char* gbytes_t;
typedef __attribute__((address_space(1))) char* gbytes_w_t;
// a pointer every lane holds the same value of, moved to scalar registers (folds away where the compiler already knows it
// is uniform; two v_readfirstlane where it does not)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int gload_i32(const int32_t* base, unsigned byte_off) {
#if FK_SADDR
    gbytes_t b = (gbytes_t)uniform_u64((unsigned long long)base);
    asm("" : "+s"(b));
    return *(const __attribute__((address_space(1))) int*)(b + byte_off);
#else
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + byte_off);
#endif
}
__device__ __forceinline__ double gload_f64(const double* base, unsigned byte_off) {
#if FK_SADDR
    gbytes_t b = (gbytes_t)uniform_u64((unsigned long long)base);
    asm("" : "+s"(b));
    return *(const __attribute__((address_space(1))) double*)(b + byte_off);
#else
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
#endif
}
__device__ __forceinline__ void gstore_f64(double* base, unsigned byte_off, double v) {
#if FK_SADDR
    gbytes_w_t b = (gbytes_w_t)uniform_u64((unsigned long long)base);
    asm("" : "+s"(b));
    *(__attribute__((address_space(1))) double*)(b + byte_off) = v;
#else
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v;
#endif
}
__device__ __forceinline__ void gstore_i32(int32_t* base, unsigned byte_off, int v) {
#if FK_SADDR
    gbytes_w_t b = (gbytes_w_t)uniform_u64((unsigned long long)base);
    asm("" : "+s"(b));
    *(__attribute__((address_space(1))) int*)(b + byte_off) = v;
#else
    *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(base) + byte_off) = v;
#endif
}

// ---------------------------------------------------------------------------------------
// k_prepare: forward transform of `npoly` small polynomials into prepared form, TWO per workgroup (the unit of the transforms
// is a pair, fft_dev.hpp; the last workgroup of an odd count transforms one): the twiddle table and two exchange buffers of
// LDS.  The 288 polynomials of an address (6 digits at 2^18) are 144 workgroups: one round on 256 CUs.
// The prepared operand carries the inverse transform's 1/n (a power of two: exact).
// ginv != 0: the polynomial is first mapped through phi_g (g = ginv^-1 mod 2N), i.e. the prepared
// operand is FFT(phi_g(K)).  Automorphism keys are stored this way so that the key-switch can apply
// phi_g to its INPUT instead of to every output limb:  phi_g(sum_r x_r * K_r) = sum_r phi_g(x_r) * phi_g(K_r).
// ---------------------------------------------------------------------------------------
constexpr size_t LDS_PREPARE_BYTES = (size_t)(LDS_TW + 2 * LDS_DATA) * sizeof(double);
__global__ __launch_bounds__(T) void k_prepare(const int32_t* __restrict__ in, double* __restrict__ out,
                                               const double* __restrict__ tw_g, double ninv, int ginv, int npoly) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    TwRegs twr;
    twiddles_issue(twr, tw_g, tid);            // the table and the polynomials in ONE round trip (the commit's barrier comes after both)
    const int p0 = 2 * (int)blockIdx.x;
    const bool two = p0 + 1 < npoly;           // (workgroup uniform)
    int v[2][E];
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int32_t* src = in + (long)(p0 + ((b == 1 && two) ? 1 : 0)) * N;
        if (ginv == 0) {
#pragma unroll
            for (int k = 0; k < E; k++) v[b][k] = src[tid + T * k];
        } else {   // destination i' takes +-source i, i = i' * ginv mod 2N (one-off gather at key load)
#pragma unroll
            for (int k = 0; k < E; k++) {
                const int s = ((tid + T * k) * ginv) & (2 * N - 1);
                const int w = src[s & (N - 1)];
                v[b][k] = s >= N ? -w : w;
            }
        }
    }
    twiddles_commit(twr, tw, tid);
    double x[2][E];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k = 0; k < E; k++) x[b][k] = (double)v[b][k];
    ntt_fwd<2>(x, tw, data, tid);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        if (b == 1 && !two) break;
        double2* o = reinterpret_cast<double2*>(out + (long)(p0 + b) * N);
#pragma unroll
        for (int kk = 0; kk < E / 2; kk++) {
            double2 w;
            w.x = x[b][2 * kk] * ninv;
            w.y = x[b][2 * kk + 1] * ninv;
            o[kk * T + tid] = w;
        }
    }
}

// acc += x (.) g for one prepared polynomial: 4 complex points per thread, (re, im) = elements (2kk, 2kk+1); 4 FMAs per point
__device__ __forceinline__ void cmac(double& ar, double& ai, double xr, double xi, double gr, double gi) {
    ar = __builtin_fma(xr, gr, ar);
    ar = __builtin_fma(-xi, gi, ar);
    ai = __builtin_fma(xr, gi, ai);
    ai = __builtin_fma(xi, gr, ai);
}
__device__ __forceinline__ void mac_poly(double (&acc)[E], const double (&x)[E], const double* __restrict__ g, int tid) {
    const double2* gp = reinterpret_cast<const double2*>(g);
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) {
#ifdef FK_NO_OPERANDS   // measurement only (wrong results): what the step costs without its operand stream
        double2 v;
        asm volatile("" : "=v"(v.x), "=v"(v.y));
#else
        const double2 v = gp[kk * T + tid];
#endif
        cmac(acc[2 * kk], acc[2 * kk + 1], x[2 * kk], x[2 * kk + 1], v.x, v.y);
    }
}

// Prepared-operand registers of one polynomial (E/2 16-byte words per thread).
struct OpRegs { double2 v[E / 2]; };
__device__ __forceinline__ void load_ops(OpRegs& o, const double* __restrict__ g, int tid) {
#ifdef FK_NO_OPERANDS
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) asm volatile("" : "=v"(o.v[kk].x), "=v"(o.v[kk].y));
    return;
#endif
#if FK_SADDR
    // Every operand polynomial of a launch is addressed as (wave-uniform base) + (lane offset): the base — key / GGSW
    // pointer, limb, column, kk * 8 KiB — lives in scalar registers and is advanced by scalar adds, the lane offset is ONE
    // 32-bit register (tid * 16) for all loads: `global_load_dwordx4 v, v_off, s[base]`.  Left to itself the compiler folds
    // the kk * 8 KiB into the vector address and spends a 64-bit VALU add per load (24 VALU instructions per output limb).
    // The empty asm makes the base opaque (and pins it to SGPRs), so the constant cannot migrate.
    typedef double d2v __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) char* gbytes;       // global address space, spelled out: the asm below
    typedef const __attribute__((address_space(1))) d2v* gwords;        // would otherwise leave a generic (flat) pointer
    unsigned off = (unsigned)tid * 16u;
    asm("" : "+v"(off));   // opaque per call: the zero extension of the offset must stay next to the loads (not hoisted out of the limb loop) for the scalar-base form to be selected
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) {
#ifdef FK_HALF_OPERANDS   // measurement only (wrong results): half the operand bytes
        if (kk >= E / 4) { asm volatile("" : "=v"(o.v[kk].x), "=v"(o.v[kk].y)); continue; }
#endif
        gbytes base = (gbytes)uniform_u64((unsigned long long)g + (unsigned long long)kk * T * 16);
        asm("" : "+s"(base));
        const d2v w = *(gwords)(base + off);
        o.v[kk].x = w.x;
        o.v[kk].y = w.y;
    }
#else
    const double2* gp = reinterpret_cast<const double2*>(g);
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) o.v[kk] = gp[kk * T + tid];
#endif
}
// Makes the compiler materialise x here (no instruction): code motion passes sink a multiply-accumulate towards its first use, which
// may lie behind the NEXT operands' requests — and then its wait covers those requests as well.
__device__ __forceinline__ void pin_regs(double (&x)[E]) {
#pragma unroll
    for (int k = 0; k < E; k++) asm volatile("" : "+v"(x[k]));
}
// (the first products of a step, which have no transform to run under: measured apart with -DFK_NO_PROLOGUE_OPS)
__device__ __forceinline__ void load_ops_p(OpRegs& o, const double* __restrict__ g, int tid) {
#ifdef FK_NO_PROLOGUE_OPS
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) asm volatile("" : "=v"(o.v[kk].x), "=v"(o.v[kk].y));
#else
    load_ops(o, g, tid);
#endif
}
__device__ __forceinline__ void mac_regs(double (&acc)[E], const double (&x)[E], const OpRegs& o) {
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) cmac(acc[2 * kk], acc[2 * kk + 1], x[2 * kk], x[2 * kk + 1], o.v[kk].x, o.v[kk].y);
}

// ---------------------------------------------------------------------------------------
// k_ext_product: res = a (x) G   (SURVEY.md A.4), SA = limbs of a and res, SG = limbs of G.
// Phase 1: the 2*SA limb polynomials of a are transformed (SA at a time) and stay in registers.
// Phase 2: for each output column, the SG output limbs are produced from the least significant
// one upwards, two at a time: pointwise MAC of the 2*SA inputs against G, a paired inverse
// transform, then the base-2^17 normalisation steps whose carry is the only state that survives
// to the next limbs (vec_znx_big_normalize walks the limbs in exactly this order).
// NCO = 1: blockIdx.z selects the output column (two workgroups per ciphertext, phase 1 done by
// both).  res must not alias a.
// ---------------------------------------------------------------------------------------
// Batch sizes of the transforms (tuning knobs): BF polynomials per forward NTT call, BI per inverse.
#ifndef FK_BF
#define FK_BF 3
#endif
#ifndef FK_BI
#define FK_BI 1
#endif
constexpr int BF = FK_BF, BI = FK_BI;
#ifndef FK_EARLY_FETCH
#define FK_EARLY_FETCH 0
#endif
#ifndef FK_SPREAD_FETCH
#define FK_SPREAD_FETCH 1      // ks_run: the next limb's operand loads spread over the post-step
#endif
#ifndef FK_WARM_OPERANDS
#define FK_WARM_OPERANDS 0     // ep_run: the first output limb's operands touched in front of the forward transforms (every line: 52.1 against 51.3 us per product; one stripe per polynomial: neutral): off
#endif
#ifndef FK_EP_PREFETCH
#define FK_EP_PREFETCH 0       // ep_run: the first output limb's operands of a column requested ahead of the column loop (75 spilled registers: off)
#endif
#ifndef FK_SPREAD_FETCH_EP
#define FK_SPREAD_FETCH_EP 1   // ep_run: the next limb's twelve operand loads in two bursts around the normalisation step (8 + 4: 50.7 against 51.3 us per product; interleaved with its parts: 247 registers, over the cap, six spilled, 51.9)
#endif
// Register cap of the chain kernels (see k_keyswitch_chain): one workgroup per CU, two waves per SIMD; above 240 registers the
// two waves leave no room for the one-wave gate launch of read_prepare_write and the workgroup stays off that CU.
#ifndef FK_CHAIN_VGPRS
#define FK_CHAIN_VGPRS 120   // the attribute counts in units of two on gfx90a+ (unified VGPR + AGPR file): 120 -> 240 registers: two waves leave 32 registers of a SIMD for a small third one
#endif
// ... and the same kernels with the whole register file (128 -> 256 registers), for the launches that can never meet the gate wave: it is
// parked by read_prepare_write only, so Ram::read and Ram::write run the wide variants (k_read_chain<4,4>: 89 -> 19 spilled registers,
// k_keyswitch_chain<3,4,3,3>: 29 -> 13; +2.6 % RAM ops/s, +3.6 % on the README block: profiles/r06_experiments.txt)
// (the attribute wants a literal: the wide variants are second kernels around the same body: k_read_chain_w, k_keyswitch_chain_w)
#define FK_WIDE_VGPRS 128
static_assert(BF <= BMAX && BI <= BMAX, "LDS holds BMAX exchange buffers");

// forward transform of S polynomials, BF at a time
#ifndef FK_KS_PROLOGUE_EXTRA
#define FK_KS_PROLOGUE_EXTRA 2   // ks_trace_l: operand polynomials of a step's first pair requested beside the window, behind the forward transforms
#endif
#ifndef FK_KS_WINDOW
#define FK_KS_WINDOW 2    // ks_trace_l: operand polynomials in flight under the transforms (16 registers each; 3: 69 spilled registers, slower)
#endif
#ifndef FK_EP_WINDOW
#define FK_EP_WINDOW 2    // ep_step_r: the same (3: 75 spilled registers, slower)
#endif
#ifndef FK_FWD_SKEW
#define FK_FWD_SKEW 1   // ks_trace_l: the three forward transforms half a phase apart (ntt_fwd3_skew): trace step 35.8 -> 35.6 us, step 2.198 -> 2.18 ms; the same in the products: +1 % per product, not used there
#endif
template <int S, int R = 0, bool SKEW = false>
__device__ __forceinline__ void fwd_all(double (&x)[S][E], const double* tw, double* data, int tid) {
    if constexpr (SKEW && FK_FWD_SKEW && S == 3 && R == 0 && BF == 3) { ntt_fwd3_skew(x, tw, data, tid); return; }
    if constexpr (R < S) {
        constexpr int C = (S - R < BF) ? (S - R) : BF;
        ntt_fwd<C>(*reinterpret_cast<double(*)[C][E]>(&x[R]), tw, data, tid);
        fwd_all<S, R + C, false>(x, tw, data, tid);
    }
}

template <int SA, int SG>
__device__ __forceinline__ void ep_mac(double (&acc)[E], const double (&x0)[SA][E], const double (&x1)[SA][E], OpRegs (&g)[SA],
                                       const double* __restrict__ ggsw, int j, int co, int jnext, int tid) {
    // g holds the column_in 0 operands of limb j on entry (requested by ep_fetch0 after the previous
    // inverse transform)
    // each operand register set is refilled with the column_in 1 operands as soon as its column_in 0 product
    // has been taken, so the refills are in flight during the remaining products of the first half
#pragma unroll
    for (int r = 0; r < SA; r++) {
        mac_regs(acc, x0[r], g[r]);
        __builtin_amdgcn_sched_barrier(0);
        load_ops(g[r], ggsw + (long)(((2 * r + 1) * SG + j) * 2 + co) * N, tid);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < SA; r++) mac_regs(acc, x1[r], g[r]);
    (void)jnext;
    __builtin_amdgcn_sched_barrier(0);
}
template <int SA, int SG>
__device__ __forceinline__ void ep_fetch0(OpRegs (&g)[SA], const double* __restrict__ ggsw, int j, int co, int tid) {
#pragma unroll
    for (int r = 0; r < SA; r++) load_ops(g[r], ggsw + (long)(((2 * r) * SG + j) * 2 + co) * N, tid);
}

// STAGE selects how a ciphertext operation is spread over workgroups:
//   0  fused: one workgroup (NCO = 2) or one per output column (NCO = 1) does everything;
//   1  limb-parallel, used when the batch is far too small to fill the chip (the dependent chain
//      at the end of every RAM op has only word_size ciphertexts per round): blockIdx.z = (column,
//      limb); the workgroup produces ONE un-normalised output limb polynomial into `big`;
//   2  the normalisation / post-step pass over those limbs (blockIdx.z = column).
// Stages 1+2 compute exactly the values stage 0 computes, in the same order per coefficient.
constexpr int BIG_STRIDE = 2 * 5 * N;   // doubles of `big` per ciphertext: [column][limb <= 5][N]
__device__ __forceinline__ long big_ct() { return ((long)blockIdx.y * gridDim.x + blockIdx.x) * BIG_STRIDE; }

// The body is a device function so that a chain of products on the same ciphertext can run inside ONE launch
// (k_ext_product_chain): load_tw = false skips the twiddle table (already in LDS from the previous step).
template <int SA, int SG, int NCO, int STAGE = 0>
__device__ __forceinline__ void ep_run(GlweRef a, GlweRef res, const double* __restrict__ ggsw,
                                       const double* __restrict__ tw_g, double* __restrict__ big, double* lds, bool load_tw, const int tid,
                                       const bool stamp_on = false) {
    YSTAMP(0);
    if constexpr (STAGE == 2) {
        const int co = (int)blockIdx.z;
        int32_t* rp = at(res);
        const double* bp = big + big_ct() + (long)co * SG * N;
        double carry[E];
#pragma unroll
        for (int k = 0; k < E; k++) carry[k] = 0.0;
#pragma unroll
        for (int j = SG - 1; j >= 0; j--) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double v = bp[(long)j * N + tid + T * k] + carry[k];
                const double cy = carry_of(v);
                carry[k] = cy;
                if (j < SA) rp[glwe_off(j, co) + tid + T * k] = (int)digit_of(v, cy);
            }
        }
        return;
    }
    double* tw = lds;
    double* data = lds + LDS_TW;
    TwRegs twr;
    if (load_tw) twiddles_issue(twr, tw_g, tid);
    const int32_t* ap = at(a);
    int32_t* rp = at(res);
    const int co0 = (NCO == 1) ? (int)blockIdx.z : 0;

    double x0[SA][E], x1[SA][E];   // limbs of column 0 / column 1 of a
    OpRegs gpre[SA];               // operands requested ahead of the column loop (FK_EP_PREFETCH)
    // The operands of a product are touched for the first time by every workgroup at once, when the first output limb of the
    // first column asks for them: the stamps (tools/stamp_chain.py) show 3 us more for that limb than for any other (cold
    // lines, cold translations).  The registers to request them earlier are not there (FK_EP_PREFETCH spills), but ONE 8-byte
    // load per thread and operand polynomial, 64 bytes apart, touches every line of the six polynomials of that limb: issued
    // here, in front of the forward transforms, summed into one register that nothing reads.
    [[maybe_unused]] double warm = 0.0;
    if constexpr (STAGE == 0 && FK_WARM_OPERANDS) {
#pragma unroll
        for (int w = 0; w < 2 * SA; w++) warm += ggsw[(long)((w * SG + (SG - 1)) * 2 + co0) * N + tid];   // (one coalesced 4 KB stripe per polynomial: translations and the first lines)
    }
    {
        int xi[SA][E];
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) xi[r][k] = gload_i32(ap + glwe_off(r, 0), (unsigned)(tid + T * k) * 4u);
        if (load_tw) twiddles_commit(twr, tw, tid);
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) x0[r][k] = (double)xi[r][k];
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) xi[r][k] = gload_i32(ap + glwe_off(r, 1), (unsigned)(tid + T * k) * 4u);
        YSTAMP(1);
        fwd_all<SA>(x0, tw, data, tid);
        YSTAMP(2);
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) x1[r][k] = (double)xi[r][k];
        if constexpr (STAGE == 0 && FK_EP_PREFETCH) {
            // the column_in 0 operands of the first output limb of the first column: requested in front of the second batch of
            // forward transforms (accumulator and carries are not live yet), so that no column starts with an exposed round trip
#pragma unroll
            for (int r = 0; r < SA; r++) load_ops(gpre[r], ggsw + (long)(((2 * r) * SG + (SG - 1)) * 2 + co0) * N, tid);
        }
        fwd_all<SA>(x1, tw, data, tid);
        YSTAMP(3);
    }
    if constexpr (STAGE == 0 && FK_WARM_OPERANDS) asm volatile("" ::"v"(warm));   // the loads are kept; their values end here
    if constexpr (STAGE == 1) {
        const int co = (int)blockIdx.z / SG, j = SG - 1 - (int)blockIdx.z % SG;
        OpRegs g[SA];
#pragma unroll
        for (int r = 0; r < SA; r++) load_ops(g[r], ggsw + (long)(((2 * r) * SG + j) * 2 + co) * N, tid);
        double acc[1][E];
#pragma unroll
        for (int k = 0; k < E; k++) acc[0][k] = 0.0;
        ep_mac<SA, SG>(acc[0], x0, x1, g, ggsw, j, co, -1, tid);
        ntt_inv<1, false>(acc, tw, data, tid);   // the only inverse transform of this workgroup
        double* bp = big + big_ct() + (long)(co * SG + j) * N;
#pragma unroll
        for (int k = 0; k < E; k++) bp[tid + T * k] = acc[0][k];
        return;
    }

    // Consecutive inverse transforms alternate between two exchange buffers: the cross-wave reads of one
    // are then always fenced from the next writes into the same buffer by the exchange-0 barrier of the
    // transform in between, and no barrier is needed at the start of a transform.
    constexpr bool DB = (2 * BI <= BMAX);
    int it = 0;
#pragma unroll 1
    for (int c = 0; c < NCO; c++) {
        const int co = co0 + c;
        double carry[E];
#pragma unroll
        for (int k = 0; k < E; k++) carry[k] = 0.0;
        OpRegs g[SA];
        if constexpr (STAGE == 0 && FK_EP_PREFETCH) {
#pragma unroll
            for (int r = 0; r < SA; r++) g[r] = gpre[r];
        } else {
#pragma unroll
            for (int r = 0; r < SA; r++) load_ops(g[r], ggsw + (long)(((2 * r) * SG + (SG - 1)) * 2 + co) * N, tid);
        }

        auto emit = [&](const double (&v_)[E], int j) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double v = v_[k] + carry[k];
                const double cy = carry_of(v);
                carry[k] = cy;
                if (j < SA) gstore_i32(rp + glwe_off(j, co), (unsigned)(tid + T * k) * 4u, (int)digit_of(v, cy));
            }
        };
        constexpr int REM = SG % BI;   // limbs left over for a final narrower batch
#pragma unroll 1
        for (int j = SG - 1; j >= BI - 1 + REM; j -= BI) {
            double acc[BI][E];
#pragma unroll
            for (int b = 0; b < BI; b++) {
#pragma unroll
                for (int k = 0; k < E; k++) acc[b][k] = 0.0;
                if (b > 0) ep_fetch0<SA, SG>(g, ggsw, j - b, co, tid);
                ep_mac<SA, SG>(acc[b], x0, x1, g, ggsw, j - b, co, j - b - 1, tid);
            }
            YSTAMP(8 + (c * SG + (SG - 1 - j)) * 4);
            if constexpr (FK_EARLY_FETCH == 1) {
                if (j - BI >= 0) ep_fetch0<SA, SG>(g, ggsw, j - BI, co, tid);
                __builtin_amdgcn_sched_barrier(0);
            }
            ntt_inv<BI, !DB>(acc, tw, data + (DB ? (it++ & 1) * BI * LDS_DATA : 0), tid);
            YSTAMP(9 + (c * SG + (SG - 1 - j)) * 4);
            if constexpr (STAGE == 0 && FK_EP_PREFETCH) {
                if (j - BI < 0 && c + 1 < NCO) {   // last limb of this column: the next column's first operands arrive during its normalisation step
#pragma unroll
                    for (int r = 0; r < SA; r++) load_ops(gpre[r], ggsw + (long)(((2 * r) * SG + (SG - 1)) * 2 + co + 1) * N, tid);
                }
            }
            if constexpr (FK_EARLY_FETCH == 0 && FK_SPREAD_FETCH_EP && BI == 1) {
                // the next limb's column_in 0 operands, one polynomial at a time between the parts of the normalisation step (see
                // ks_trace_y: twelve loads per thread from all waves at once wait for the address unit)
                const bool more = j - BI >= 0;
                const int jn = j - BI;
                auto fetch1 = [&](int r) { if (more) load_ops(g[r], ggsw + (long)(((2 * r) * SG + jn) * 2 + co) * N, tid); };
                fetch1(0);
                if (SA > 1) fetch1(1);
                __builtin_amdgcn_sched_barrier(0);
                emit(acc[0], j);
                __builtin_amdgcn_sched_barrier(0);
                if (SA > 2) fetch1(2);
                static_assert(SA <= 3, "three operand polynomials per half");
            } else {
                if constexpr (FK_EARLY_FETCH == 0) { if (j - BI >= 0) ep_fetch0<SA, SG>(g, ggsw, j - BI, co, tid); }   // overlaps the normalisation step
#pragma unroll
                for (int b = 0; b < BI; b++) emit(acc[b], j - b);
            }
            YSTAMP(10 + (c * SG + (SG - 1 - j)) * 4);
        }
        if constexpr (REM == 1) {
            double acc[1][E];
#pragma unroll
            for (int k = 0; k < E; k++) acc[0][k] = 0.0;
            ep_mac<SA, SG>(acc[0], x0, x1, g, ggsw, 0, co, -1, tid);
            ntt_inv<1, !DB>(acc, tw, data + (DB ? (it++ & 1) * BI * LDS_DATA : 0), tid);
            emit(acc[0], 0);
        }
    }
    YSTAMP(5);
}

template <int SA, int SG, int NCO, int STAGE = 0>
__global__ __launch_bounds__(T, T / 256) void k_ext_product(GlweRef a, GlweRef res, const double* __restrict__ ggsw,
                                                            const double* __restrict__ tw_g, double* __restrict__ big) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, tw_g, STAGE != 2);
    ep_run<SA, SG, NCO, STAGE>(a, res, ggsw, tw_g, big, lds, true, vt((int)threadIdx.x));
}
// CoordinatePrepared::product(_inplace) (coordinate_prepared.rs:147-177) as ONE launch: the n external products
// of a coordinate's digits on the same ciphertext, one workgroup per ciphertext.  Step i reads what step
// i - 1 wrote (the workgroup's own slots of the two ping-pong buffers: no other workgroup touches them), so
// the steps need no device-wide synchronisation — what a sequence of launches pays per step (kernel drain and
// ramp, kernel arguments, the twiddle table, a cold first load: 5-6 us of 55) is paid once.
constexpr int CHAIN_MAX = 12;
constexpr int MID_GROUPS_MAX = 64;   // ciphertexts of a k_chain_mid launch (at most eight per XCD)
struct EpChainArgs {
    GlweRef src, buf[2];             // step i writes buf[i & 1] (buf[0] must not be src)
    const double* ggsw[CHAIN_MAX];   // prepared digits
    const double* tw;
    int n;
    unsigned* done = nullptr;        // fallback launch behind k_chain_mid: a ciphertext is redone unless done[ct * 32 + 3] == done_seq
    unsigned done_seq = 0;
    unsigned* host_count = nullptr;  // pinned host word that mirrors the number of ciphertexts redone (read by the host without a sync)
};
template <int SA, int SG>
__global__ __launch_bounds__(T, T / 256) __attribute__((amdgpu_num_vgpr(FK_CHAIN_VGPRS))) void k_ext_product_chain(EpChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ca.tw);
    if (ca.done) {
        if (__hip_atomic_load(ca.done + (blockIdx.y * gridDim.x + blockIdx.x) * 32 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ca.done_seq) return;
        if (threadIdx.x == 0) {
            const unsigned taken = atomicAdd(ca.done + MID_GROUPS_MAX * 32 + 1, 1u) + 1u;   // ciphertexts redone (fheram_mid_stats)
            if (ca.host_count) __hip_atomic_store(ca.host_count, taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    GlweRef in = ca.src;
#pragma unroll 1
    for (int i = 0; i < ca.n; i++) {
        const GlweRef out = ca.buf[i & 1];
        int tid = vt((int)threadIdx.x);
        asm volatile("" : "+v"(tid));   // per-step copy the optimiser cannot see through: keeps it from hoisting every
                                        // thread-index-derived address out of the step loop (56 spilled registers)
        __builtin_assume(tid >= 0 && tid < T);   // ... but it may still use the range (index patterns fold to immediates)
        ep_run<SA, SG, 2, 0>(in, out, ca.ggsw[i], ca.tw, nullptr, lds, i == 0, tid, YSTAMP_STEP(i + 1));
        __syncthreads();   // the step's stores have completed (vmcnt 0) and its LDS traffic is over
        in = out;
    }
}

// ---------------------------------------------------------------------------------------
// k_keyswitch: key-switch core with fused pre/post steps.
//   big[co][j] = sum_r NTT(x.mask limb r) (.) K[r][j][co]      (SX forward, SX*2*SK MAC, 2*SK inverse)
//   big[body_col][j] += x.body limb j
//   big <- phi_g(big)                                           (LDS permutation, odd stride)
//   post-step + normalise SK -> SO limbs
// ---------------------------------------------------------------------------------------
enum KsMode {
    KS_AUTO = 0,    // x = a;                          out = norm(phi(big))                       glwe_automorphism
    KS_TRACE = 1,   // x = rsh1(rot(a, rho));          out = norm(phi(big) + x)                   rsh + automorphism_add_inplace
    KS_PAIR = 2,    // x = rsh1(rot(a,-t) - b);        out = rot(norm(rsh1(rot(a,-t)+b) - norm(phi(big))), +t)   packer combine
    KS_ADD = 3,     // x = a;                          out = norm(phi(big) + a)                   automorphism_add_inplace
    KS_SUBNEG = 4,  // x = a;                          out = norm(a - phi(big))                   automorphism_sub_negate
    KS_TENSOR = 5   // x = a; body goes to column 1;   out = norm(big), g = 1                     ggsw_expand_row
};
struct KsArgs {
    GlweRef a, b, out;
    const double* key;   // prepared [row SX][limb SK][col_out 2]
    const double* tw;
    int g;               // Galois element mod 2N (odd, in [1, 2N))
    int ginv;            // g^-1 mod 2N
    int t;               // KS_PAIR: rotation amount (N >> (level+1))
    int rot_mul;         // KS_TRACE: rho = -(rot_base + blockIdx.x * rot_mul)   (write path: ct_lo * X^-row;
    int rot_base;        //           row = shard + x * n_shards when the RAM is sharded by rows)
    double* big;         // STAGE 1/2: un-normalised limb polynomials, BIG_STRIDE doubles per ciphertext
};

// sign * a[(limb, col)][src] for the coefficient at position i of rot(a, rho)
__device__ __forceinline__ void rot_src(int i, int rho, int& src, bool& neg) {
    int s = (i - rho) & (2 * N - 1);
    neg = s >= N;
    src = neg ? s - N : s;
}
__device__ __forceinline__ int cneg(int v, bool neg) { return neg ? -v : v; }

// limbs of column `col` of the key-switch input x at coefficient i (pre-step applied)
template <int MODE, int SX>
__device__ __forceinline__ void load_x(const KsArgs& ka, const int32_t* ap, const int32_t* bp, int col, int i, int (&x)[SX]) {
    if constexpr (MODE == KS_TRACE) {
        int src; bool sgn;
        rot_src(i, -(ka.rot_base + (int)blockIdx.x * ka.rot_mul), src, sgn);
        int v[SX];
#pragma unroll
        for (int j = 0; j < SX; j++) v[j] = cneg(ap[glwe_off(j, col) + src], sgn);
        rsh1_coeff<SX>(v, x);
    } else if constexpr (MODE == KS_PAIR) {
        int src; bool sgn;
        rot_src(i, -ka.t, src, sgn);
        int v[SX];
#pragma unroll
        for (int j = 0; j < SX; j++) v[j] = cneg(ap[glwe_off(j, col) + src], sgn) - bp[glwe_off(j, col) + i];
        rsh1_coeff<SX>(v, x);
    } else {
#pragma unroll
        for (int j = 0; j < SX; j++) x[j] = ap[glwe_off(j, col) + i];
    }
}
// KS_PAIR: rsh1(rot(a,-t) + b) at coefficient i
template <int SX>
__device__ __forceinline__ void load_pair_sum(const KsArgs& ka, const int32_t* ap, const int32_t* bp, int col, int i, int (&x)[SX]) {
    int src; bool sgn;
    rot_src(i, -ka.t, src, sgn);
    int v[SX];
#pragma unroll
    for (int j = 0; j < SX; j++) v[j] = cneg(ap[glwe_off(j, col) + src], sgn) + bp[glwe_off(j, col) + i];
    rsh1_coeff<SX>(v, x);
}

// load_x in two halves, so that a kernel can issue the loads of several coefficients / columns before it
// waits for any of them: raw limbs of a (and b) at the source coefficient, then the pre-step in registers
template <int MODE, int SX>
struct RawX { int a[SX]; int b[SX]; bool neg; };
template <int MODE, int SX>
__device__ __forceinline__ void load_raw(const KsArgs& ka, const int32_t* ap, const int32_t* bp, int col, int i, RawX<MODE, SX>& r) {
    int src = i;
    r.neg = false;
    if constexpr (MODE == KS_TRACE) rot_src(i, -(ka.rot_base + (int)blockIdx.x * ka.rot_mul), src, r.neg);
    if constexpr (MODE == KS_PAIR) rot_src(i, -ka.t, src, r.neg);
#pragma unroll
    for (int j = 0; j < SX; j++) {
        r.a[j] = gload_i32(ap + glwe_off(j, col), (unsigned)src * 4u);
        if constexpr (MODE == KS_PAIR) r.b[j] = gload_i32(bp + glwe_off(j, col), (unsigned)i * 4u);
    }
}
template <int MODE, int SX>
__device__ __forceinline__ void pre_step(const RawX<MODE, SX>& r, int (&x)[SX]) {
    if constexpr (MODE == KS_TRACE) {
        int v[SX];
#pragma unroll
        for (int j = 0; j < SX; j++) v[j] = cneg(r.a[j], r.neg);
        rsh1_coeff<SX>(v, x);
    } else if constexpr (MODE == KS_PAIR) {
        int v[SX];
#pragma unroll
        for (int j = 0; j < SX; j++) v[j] = cneg(r.a[j], r.neg) - r.b[j];
        rsh1_coeff<SX>(v, x);
    } else {
#pragma unroll
        for (int j = 0; j < SX; j++) x[j] = r.a[j];
    }
}

// KS_PAIR: rsh1(rot(a,-t) + b) from the raw limbs
template <int MODE, int SX>
__device__ __forceinline__ void pair_sum_raw(const RawX<MODE, SX>& r, int (&x)[SX]) {
    int v[SX];
#pragma unroll
    for (int j = 0; j < SX; j++) v[j] = cneg(r.a[j], r.neg) + r.b[j];
    rsh1_coeff<SX>(v, x);
}
// three limbs in [-2^16, 2^16] as 18-bit fields of one 64-bit word
__device__ __forceinline__ unsigned long long pack18(const int (&v)[3]) {
    return (unsigned long long)(unsigned)(v[0] + (1 << 17)) | ((unsigned long long)(unsigned)(v[1] + (1 << 17)) << 18) |
           ((unsigned long long)(unsigned)(v[2] + (1 << 17)) << 36);
}
__device__ __forceinline__ int unpack18(unsigned long long w, int j) { return (int)((w >> (18 * j)) & 0x3FFFF) - (1 << 17); }

__device__ __forceinline__ int sel_limb(const int (&x)[3], int j) { return j == 0 ? x[0] : (j == 1 ? x[1] : x[2]); }
__device__ __forceinline__ int sel_limb(const int (&x)[4], int j) { return j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : x[3])); }

// Structure as k_ext_product: the SX mask limbs are transformed together and stay in registers; the
// SK output limbs of each column are then streamed from the least significant one upwards (MAC,
// inverse transform, body add, post-step and one normalisation step each), so the live state
// between limbs is just the carries.
// phi_g is applied on the INPUT side: phi_g(sum_r x_r * K_r + body) = sum_r phi_g(x_r) * phi_g(K_r) + phi_g(body),
// the key being prepared as NTT(phi_g(K)) (k_prepare).  The pre-stepped mask limbs are staged in LDS
// as int32 and gathered with the odd stride g^-1 (conflict free) before the forward transforms, and
// the body limbs are staged once per ciphertext and gathered when they are added: two LDS gathers
// of small integers per ciphertext instead of one permutation of every output limb.
// NCO as in k_ext_product.  out must not alias a or b.
template <int MODE, int SX, int SK, int SO, int NCO, int STAGE = 0>
__device__ __forceinline__ void ks_run(const KsArgs& ka, double* lds, bool load_tw, const int tid) {
    double* tw = lds;
    double* data = lds + LDS_TW;
    STAMP(0);
    TwRegs twr;
    if (load_tw) twiddles_issue(twr, ka.tw, tid);
    STAMP(1);
    const int32_t* ap = at(ka.a);
    const int32_t* bp = (MODE == KS_PAIR) ? at(ka.b) : nullptr;
    int32_t* op = at(ka.out);
    constexpr int BODY_COL = (MODE == KS_TENSOR) ? 1 : 0;
    constexpr bool PHI = (MODE != KS_TENSOR);   // KS_TENSOR has g = 1
    const int co0 = (STAGE == 1) ? (int)blockIdx.z / SK : ((NCO == 1 || STAGE == 2) ? (int)blockIdx.z : 0);
    constexpr int NCOL = (STAGE == 0) ? NCO : 1;
    // Staging areas in the exchange buffers: the mask limbs (int32) use the start of the area, before any
    // transform; the body limbs, three 18-bit fields per 64-bit word, the buffers the inverse transforms
    // (KBI at a time) leave alone.
    constexpr int KBI = (SX <= 3) ? BI : 1;
    constexpr int WPC = (SX + 2) / 3;   // packed words per coefficient
    // double-buffered inverse transforms (see k_ext_product) when the body staging leaves two buffer sets free
    constexpr bool DB = ((size_t)WPC * N * 8 <= (size_t)(BMAX - 2 * KBI) * LDS_DATA * sizeof(double));
    constexpr int NBUF = DB ? 2 * KBI : KBI;
    static_assert((size_t)WPC * N * 8 <= (size_t)(BMAX - NBUF) * LDS_DATA * sizeof(double), "body staging does not fit");
    int* mstage = reinterpret_cast<int*>(data);
    unsigned long long* bstage = reinterpret_cast<unsigned long long*>(data + NBUF * LDS_DATA);
    int it = 0;
    auto stage_body = [&](const int (&v)[SX], int i) {
#pragma unroll
        for (int w = 0; w < WPC; w++) {
            unsigned long long word = 0;
#pragma unroll
            for (int r = 3 * w; r < SX && r < 3 * w + 3; r++) word |= (unsigned long long)(unsigned)(v[r] + (1 << 17)) << (18 * (r - 3 * w));
            bstage[w * N + i] = word;
        }
    };
    // phi_g: destination i' = tid + T*k takes +-source i, i = i' * ginv mod 2N
    const int sidx0 = (tid * ka.ginv) & (2 * N - 1);
    const int sstep = (T * ka.ginv) & (2 * N - 1);

    // FAST path (the automorphism family on 3-limb ciphertexts, fused stage): BOTH columns of the input are
    // requested at the very start, pre-stepped once and kept as packed words (three 18-bit fields, 2 VGPRs per
    // coefficient): pa0 / pa1 = what the post-step adds to column 0 / 1 (x itself; KS_PAIR: the pair sum),
    // pbody = x column 0 for the body staging.  Without this every column began with a global round trip of
    // its own (4.3 + 2.1 us of a 50 us kernel, profiles/r02_stamps_keyswitch_unchained.txt).
    // (not for the fused two-column KS_PAIR: three packed operands per coefficient do not fit its registers)
    constexpr bool FAST = PHI && SX == 3 && STAGE == 0 && !(MODE == KS_PAIR && NCO == 2);
    constexpr bool HAS_XA = (MODE == KS_TRACE || MODE == KS_ADD || MODE == KS_SUBNEG || MODE == KS_PAIR);
    unsigned long long pa0[E], pa1[E], pbody[E];
#pragma unroll
    for (int k = 0; k < E; k++) pa0[k] = pa1[k] = pbody[k] = 0;

    // Phase 1: mask column of x, all limbs, mapped through phi_g and transformed
    double xh[SX][E];
    if constexpr (FAST) {
        constexpr int SXF = 3;
        const bool need0 = (NCO == 2) || (co0 == 0);     // column 0: the body, and the post-step operand of column 0
        RawX<MODE, SXF> rm[E], rb[E];
#pragma unroll
        for (int k = 0; k < E; k++) load_raw<MODE, SXF>(ka, ap, bp, 1, tid + T * k, rm[k]);
        if (need0) {
#pragma unroll
            for (int k = 0; k < E; k++) load_raw<MODE, SXF>(ka, ap, bp, 0, tid + T * k, rb[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < E; k++) {
            int xm[SXF];
            pre_step<MODE, SXF>(rm[k], xm);
#pragma unroll
            for (int r = 0; r < SXF; r++) mstage[r * N + tid + T * k] = xm[r];
            if constexpr (MODE == KS_PAIR) {
                int sm[SXF];
                pair_sum_raw<MODE, SXF>(rm[k], sm);
                pa1[k] = pack18(sm);
            } else if constexpr (HAS_XA) {
                pa1[k] = pack18(xm);
            }
        }
        if (need0) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                int x0[SXF];
                pre_step<MODE, SXF>(rb[k], x0);
                pbody[k] = pack18(x0);
                if constexpr (MODE == KS_PAIR) {
                    int sm[SXF];
                    pair_sum_raw<MODE, SXF>(rb[k], sm);
                    pa0[k] = pack18(sm);
                } else if constexpr (HAS_XA) {
                    pa0[k] = pbody[k];
                }
            }
        }
        if (load_tw) twiddles_commit(twr, tw, tid); else __syncthreads();   // its barrier also publishes the staged limbs
        int sidx = sidx0;
#pragma unroll
        for (int k = 0; k < E; k++) {
            const bool ng = sidx >= N;
            const int s_ = sidx & (N - 1);
#pragma unroll
            for (int r = 0; r < SXF; r++) xh[r][k] = (double)cneg(mstage[r * N + s_], ng);
            sidx = (sidx + sstep) & (2 * N - 1);
        }
        STAMP(2);
        fwd_all<SX>(xh, tw, data, tid);
        STAMP(3);
    } else if constexpr (STAGE != 2) {
        if constexpr (PHI) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                int xm[SX];
                load_x<MODE, SX>(ka, ap, bp, 1, tid + T * k, xm);
#pragma unroll
                for (int r = 0; r < SX; r++) mstage[r * N + tid + T * k] = xm[r];
            }
            if (load_tw) twiddles_commit(twr, tw, tid); else __syncthreads();   // its barrier also publishes the staged limbs
            int sidx = sidx0;
#pragma unroll
            for (int k = 0; k < E; k++) {
                const bool ng = sidx >= N;
                const int s = sidx & (N - 1);
#pragma unroll
                for (int r = 0; r < SX; r++) xh[r][k] = (double)cneg(mstage[r * N + s], ng);
                sidx = (sidx + sstep) & (2 * N - 1);
            }
            // the first exchange of the forward transform starts with a barrier: every gather is done
            // before the staging area is overwritten
        } else {
#pragma unroll
            for (int k = 0; k < E; k++) {
                int xm[SX];
                load_x<MODE, SX>(ka, ap, bp, 1, tid + T * k, xm);
#pragma unroll
                for (int r = 0; r < SX; r++) xh[r][k] = (double)xm[r];
            }
            if (load_tw) twiddles_commit(twr, tw, tid); else __syncthreads();
        }
        STAMP(2);
        fwd_all<SX>(xh, tw, data, tid);
        STAMP(3);
    }

#pragma unroll 1
    for (int c = 0; c < NCOL; c++) {
        const int co = co0 + c;
        STAMP(6 + 24 * c);
        // per-coefficient limbs needed by the post-step of this column
        int xa[E][SX];   // KS_TRACE/ADD/SUBNEG: x column co;  KS_PAIR: rsh1(rot(a,-t)+b) column co
        int xb[E][SX];   // KS_TENSOR: body limbs of x (column 0), added to column 1
        unsigned long long px[E];   // FAST: packed post-step operand of this column
        if constexpr (FAST) {
#pragma unroll
            for (int k = 0; k < E; k++) px[k] = (co == 0) ? pa0[k] : pa1[k];
            if (co == BODY_COL) {   // stage the body limbs of x (natural order) for the gathers of add_body
                lds_barrier();      // slower waves may still be inside the wave-local exchanges of the forward transforms
#pragma unroll
                for (int k = 0; k < E; k++) bstage[tid + T * k] = pbody[k];
                // published by the barriers of the first inverse transform, which precede every gather
            }
        } else {
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int i = tid + T * k;
            if constexpr (MODE == KS_PAIR) load_pair_sum<SX>(ka, ap, bp, co, i, xa[k]);
            else if constexpr (MODE == KS_TRACE || MODE == KS_ADD || MODE == KS_SUBNEG) load_x<MODE, SX>(ka, ap, bp, co, i, xa[k]);
            else if constexpr (MODE == KS_TENSOR) load_x<MODE, SX>(ka, ap, bp, 0, i, xb[k]);
        }
        }
        if constexpr (PHI && !FAST) {
            if (co == BODY_COL) {   // stage the body limbs of x (natural order) for the gathers of add_body
                lds_barrier();      // slower waves may still be inside the wave-local exchanges of the forward transforms
#pragma unroll
                for (int k = 0; k < E; k++) {
                    const int i = tid + T * k;
                    if constexpr (MODE == KS_TRACE || MODE == KS_ADD || MODE == KS_SUBNEG) {
                        stage_body(xa[k], i);   // column 0 of x is the body
                    } else {
                        int xm[SX];
                        load_x<MODE, SX>(ka, ap, bp, 0, i, xm);
                        stage_body(xm, i);
                    }
                }
                // published by the barriers of the first inverse transform, which precede every gather
            }
        }
        double carry[E], carry2[E];
#pragma unroll
        for (int k = 0; k < E; k++) { carry[k] = 0.0; carry2[k] = 0.0; }

        // key operands of the next limb are requested right after the inverse transform of the current
        // one, so their latency overlaps the post-step but they do not occupy 48 VGPRs during the
        // transform (holding them across it cost more than it hid: measured)
        OpRegs g[SX];
        auto fetch = [&](int j) {
#pragma unroll
            for (int r = 0; r < SX; r++) load_ops(g[r], ka.key + (long)((r * SK + j) * 2 + co) * N, tid);
        };
        auto mac = [&](double (&acc)[E], int jnext) {
#pragma unroll
            for (int k = 0; k < E; k++) acc[k] = 0.0;
            (void)jnext;
#pragma unroll
            for (int r = 0; r < SX; r++) mac_regs(acc, xh[r], g[r]);
            __builtin_amdgcn_sched_barrier(0);
        };
        // vec_znx_big_add_small_inplace of the body limbs, seen through phi_g
        auto add_body = [&](double (&acc)[E], int j) {
            if (co == BODY_COL && j < SX) {
                if constexpr (PHI) {
                    const int w = j >= 3 ? 1 : 0, sh = 18 * (j - 3 * w);
                    int sidx = sidx0;
#pragma unroll
                    for (int k = 0; k < E; k++) {
                        const int v = (int)((bstage[w * N + (sidx & (N - 1))] >> sh) & 0x3FFFF) - (1 << 17);
                        acc[k] += (double)cneg(v, sidx >= N);
                        sidx = (sidx + sstep) & (2 * N - 1);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < E; k++) acc[k] += (double)sel_limb(xb[k], j);
                }
            }
        };
        // post-step + one normalisation step at destination coefficients i' = tid + T*k
        auto emit = [&](const double (&acc)[E], int j) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                const int i = tid + T * k;
                double v = acc[k];
                int xl = 0;   // limb j of the post-step operand at this coefficient
                if constexpr (HAS_XA) {
                    if constexpr (FAST) xl = unpack18(px[k], j < SX ? j : 0);
                    else xl = sel_limb(xa[k], j);
                }
                if constexpr (MODE == KS_TRACE || MODE == KS_ADD) { if (j < SX) v += (double)xl; }
                if constexpr (MODE == KS_SUBNEG) v = (j < SX ? (double)xl : 0.0) - v;
                v += carry[k];
                const double cy = carry_of(v);
                carry[k] = cy;
                if (j < SO) {
                    const double d = digit_of(v, cy);
                    if constexpr (MODE == KS_PAIR) {
                        // a <- normalize(rsh1(a*X^-t + b) - tmp); a <- a * X^t
                        const double v2 = (double)xl - d + carry2[k];
                        const double cy2 = carry_of(v2);
                        carry2[k] = cy2;
                        int dst = i + ka.t;
                        const bool ng = dst >= N;
                        if (ng) dst -= N;
                        gstore_i32(op + glwe_off(j, co), (unsigned)dst * 4u, cneg((int)digit_of(v2, cy2), ng));
                    } else {
                        gstore_i32(op + glwe_off(j, co), (unsigned)i * 4u, (int)d);
                    }
                }
            }
        };

        if constexpr (STAGE == 1) {   // one un-normalised limb polynomial per workgroup
            const int j = SK - 1 - (int)blockIdx.z % SK;
            fetch(j);
            double acc[1][E];
            mac(acc[0], -1);
            ntt_inv<1, false>(acc, tw, data, tid);   // the only inverse transform of this workgroup
            add_body(acc[0], j);
            double* bgp = ka.big + big_ct() + (long)(co * SK + j) * N;
#pragma unroll
            for (int k = 0; k < E; k++) bgp[tid + T * k] = acc[0][k];
            return;
        }
        constexpr int REM = SK % KBI;
        STAMP(4 + 24 * c);
        fetch(SK - 1);
#pragma unroll 1
        for (int j = SK - 1; j >= KBI - 1 + REM; j -= KBI) {
            double acc[KBI][E];
            STAMP(8 + 4 * (SK - 1 - j) + 24 * c);
#pragma unroll
            for (int b = 0; b < KBI; b++) {
                if (b > 0) fetch(j - b);
                mac(acc[b], j - b - 1);
            }
            STAMP(9 + 4 * (SK - 1 - j) + 24 * c);
            if constexpr (FK_EARLY_FETCH == 1) {   // operands of the next limb requested BEFORE the transform: 48 more live registers across it
                if (j - KBI >= 0) fetch(j - KBI);
                __builtin_amdgcn_sched_barrier(0);
            }
            ntt_inv<KBI, !DB>(acc, tw, data + (DB ? (it++ & 1) * KBI * LDS_DATA : 0), tid);
            if constexpr (FK_EARLY_FETCH == 0 && FK_SPREAD_FETCH && KBI == 1 && SX == 3) {
                // next limb's operands: their latency overlaps the post-step; one operand polynomial at a time around its parts (see
                // ks_trace_y: twelve loads per thread from all waves at once queue at the address unit)
                const bool more = j - KBI >= 0;
                auto fetch1 = [&](int r) { if (more) load_ops(g[r], ka.key + (long)((r * SK + (j - KBI)) * 2 + co) * N, tid); };
                fetch1(0);
                __builtin_amdgcn_sched_barrier(0);
                STAMP(10 + 4 * (SK - 1 - j) + 24 * c);
                add_body(acc[0], j);
                __builtin_amdgcn_sched_barrier(0);
                fetch1(1);
                __builtin_amdgcn_sched_barrier(0);
                STAMP(11 + 4 * (SK - 1 - j) + 24 * c);
                emit(acc[0], j);
                __builtin_amdgcn_sched_barrier(0);
                fetch1(2);
            } else {
            if constexpr (FK_EARLY_FETCH == 0) { if (j - KBI >= 0) fetch(j - KBI); }   // (FK_EARLY_FETCH == 2: timing diagnostic, operands never refetched, results wrong)   // next limb's operands: their latency overlaps the post-step
            STAMP(10 + 4 * (SK - 1 - j) + 24 * c);
#pragma unroll
            for (int b = 0; b < KBI; b++) add_body(acc[b], j - b);
            STAMP(11 + 4 * (SK - 1 - j) + 24 * c);
#pragma unroll
            for (int b = 0; b < KBI; b++) emit(acc[b], j - b);
            }
        }
        STAMP(5 + 24 * c);
        if constexpr (REM == 1) {
            double acc[1][E];
            mac(acc[0], -1);
            ntt_inv<1, !DB>(acc, tw, data + (DB ? (it++ & 1) * KBI * LDS_DATA : 0), tid);
            add_body(acc[0], 0);
            emit(acc[0], 0);
        }
    }
}

template <int MODE, int SX, int SK, int SO, int NCO, int STAGE = 0>
__global__ __launch_bounds__(T, T / 256) void k_keyswitch(KsArgs ka) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ka.tw, STAGE != 2);
    ks_run<MODE, SX, SK, SO, NCO, STAGE>(ka, lds, true, vt((int)threadIdx.x));
}
// ---------------------------------------------------------------------------------------
// One fused trace step (a <- rsh1(a); a <- a + phi_g(KS(a)): GLWE::trace, ram.rs:457,540,572,616,621, and the packer
// levels in which a leaf is alone, ram.rs:435,514) for the steps INSIDE a chain launch, where producer and consumer of an
// intermediate ciphertext are the same workgroup and its form is ours to choose.
//
// vec_znx_rsh(1) has a closed form: with A = a_0*2^34 + a_1*2^17 + a_2 the integer the three limbs of a coefficient
// stand for, rsh1(a) is the unique centred base-2^17 digit vector of Y = ceil(A / 2) (each odd limb sends -2^16 to the
// next limb and rounds itself up, the last one only rounds up: oracle/znx.hpp rsh_inplace; checked against it digit for
// digit incl. un-normalised +2^16 limbs, tests/test_oracle.py::test_rsh1_closed_form).  |A| < 2^51, so A and Y are exact
// doubles and every operation below (scaling by powers of two, +0.5, floor, fma with an exact result) is exact.
//
// So the chain hands Y over — ONE double per coefficient and column, [col][N] in the ciphertext's slot — instead of the
// normalised limbs: the producer folds its output digits into A with one fma each (instead of a conversion and a store
// per limb) and stores floor(A/2 + 1/2); the consumer takes the digits of Y where it needs them, three FP64 instructions
// per digit, as doubles (no integer pre-step, no packing, no int -> double conversions, a third of the loads, stores and
// staging traffic).  Digits are the same integers either way: results are bit-identical to the limb form.
//   IN_Y  : the input is in that form (else: an int32 GLWE, read rotated by X^-rho, write path ram.rs:621,629)
//   OUT_Y : the output is written in that form (else: an int32 GLWE)
// ---------------------------------------------------------------------------------------
constexpr double TWO_2B = 17179869184.0;   // 2^34
// c = q * 2^17 + d with d in [-2^16, 2^16): returns d, leaves q in c
__device__ __forceinline__ double take_digit(double& c) {
    const double q = carry_of(c);
    const double d = digit_of(c, q);
    c = q;
    return d;
}

// ---------------------------------------------------------------------------------------
// ks_trace_z (round 4): the same trace step with the normalisation in CLOSED FORM — one accumulator per coefficient and
// column instead of a carry chain over the limbs.
//
// vec_znx_big_normalize of the SK un-normalised limbs v_j = acc_j + x_j (x_j: limb j of rsh1(a), zero for j >= 3) walks
// them from the least significant one, carry = floor(v/2^17 + 1/2), and drops the carry out of limb 0.  With balanced
// digits that is: the extra limbs j >= 3 only contribute their rounded carry  e = carry(v_3 + carry(v_4 ...)),  and the
// three output digits are THE balanced base-2^17 digits of
//        V = v_0 * 2^34 + v_1 * 2^17 + v_2 + e        taken modulo 2^51
// (digits in [-2^16, 2^16): the window [-C, 2^51 - C), C = 2^16 + 2^33 + 2^50).  Modulo 2^51 only v_1 mod 2^34 and
// v_0 mod 2^17 matter, and sum_j x_j * 2^(17 (2 - j)) is Y itself (its top "digit" is the unwrapped quotient), so
//        V = Y  [+- Y_body(src) for the body column: vec_znx_big_add_small of phi_g(rsh1(a).body)]
//            + e + acc_2 + cmod(acc_1, 2^34) * 2^17 + cmod(acc_0, 2^17) * 2^34          (cmod: centred remainder)
// with every term an exact integer below 2^50 in magnitude (|acc_j| < 2^47, |Y| < 2^50 + 2^33): |V| < 2^52.2, exact in
// FP64.  The output is A = V - 2^51 * floor((V + C) / 2^51), handed over as Y' = ceil(A / 2) as in ks_trace_y.
// Same integers as the limb-by-limb walk (tests: every digest and oracle comparison of the Y form also runs in this form).
//
// What it buys: (i) the post-step of an output limb is 1-4 FP64 instructions per coefficient instead of ~9 (+ ~8 for the
// body column's three digit gathers: the body is gathered ONCE, as Y); (ii) no carry, no running quotient of Y: 32 registers
// fewer, which pay for (iii) TWO inverse transforms at a time (limbs are independent now: no carry chain orders them), which share
// barriers and LDS round trips as the three forward transforms always did, and (iv, round 5) a second accumulator pair and two operand
// register sets: the operand stream runs under the transforms (ks_trace_l).
// ---------------------------------------------------------------------------------------
constexpr double TWO_51 = 2251799813685248.0;          // 2^51
constexpr double INV_TWO_51 = 1.0 / 2251799813685248.0;
constexpr double INV_TWO_2B = 1.0 / 17179869184.0;      // 2^-34
constexpr double WIN_C = 65536.0 + 8589934592.0 + 1125899906842624.0;   // 2^16 + 2^33 + 2^50
// centred remainder of x modulo 2^17 / 2^34 (x an exact integer)
__device__ __forceinline__ double cmod17(double x) { return __builtin_fma(-__builtin_floor(__builtin_fma(x, INV_TWO_B, 0.5)), TWO_B, x); }
__device__ __forceinline__ double cmod34(double x) { return __builtin_fma(-__builtin_floor(__builtin_fma(x, INV_TWO_2B, 0.5)), TWO_2B, x); }
// V -> the integer whose balanced base-2^17 digits are the three output limbs (top carry dropped)
__device__ __forceinline__ double window51(double v) { return __builtin_fma(-__builtin_floor(__builtin_fma(v, INV_TWO_51, WIN_C * INV_TWO_51)), TWO_51, v); }
// adds output limb j's contribution to V (SO = 3 output limbs; limbs j >= 3 go through the extra-carry e first)
template <int SK>
__device__ __forceinline__ void fold_limb(double (&od)[E], double (&ec)[E], const double (&acc)[E], int j) {
    // (j is wave uniform: one scalar branch per limb, not per coefficient)
    if (j >= 3) {
        if (SK >= 5 && j < SK - 1) {
#pragma unroll
            for (int k = 0; k < E; k++) ec[k] = carry_of(acc[k] + ec[k]);
        } else {
#pragma unroll
            for (int k = 0; k < E; k++) ec[k] = carry_of(acc[k]);
        }
        if (j == 3) {
#pragma unroll
            for (int k = 0; k < E; k++) od[k] += ec[k];
        }
    } else if (j == 2) {
#pragma unroll
        for (int k = 0; k < E; k++) od[k] += acc[k];
    } else if (j == 1) {
#pragma unroll
        for (int k = 0; k < E; k++) od[k] = __builtin_fma(cmod34(acc[k]), TWO_B, od[k]);
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) od[k] = __builtin_fma(cmod17(acc[k]), TWO_2B, od[k]);
    }
}
// for 4 limbs the extra carry lives inside the call for limb 3: no state between limbs (16 registers that stay free)
template <int SK>
__device__ __forceinline__ void fold_limb4(double (&od)[E], const double (&acc)[E], int j) {
    static_assert(SK == 4, "four limbs");
    double ec[E];
    fold_limb<SK>(od, ec, acc, j);
}

// ---------------------------------------------------------------------------------------
// ks_trace_l (round 4): ks_trace_z with the hand-over between the steps of a chain through LDS and registers.  Producer and
// consumer of an intermediate ciphertext are the same workgroup; what the consumer needs of it is (i) the mask column's Y in
// natural order somewhere every wave can gather from (its digits, seen through phi_g, are the inputs of the forward
// transforms), (ii) each thread's own coefficients of both columns (the accumulators start from them), (iii) the body
// column's Y in natural order for ONE gather (phi_g of the body, added to the body column's accumulator).  So: column 1's
// output goes to exchange buffer 2 once the body column's gather of this step has been taken from there (one barrier), the
// body column's output stays in registers (vc) and is staged into buffer 2 by the NEXT step behind the first fence of its
// inverse transforms.  No global loads or stores between the steps (round 3's register hand-over kept the global round trip
// of the gathers' staging; here it is gone), no store drain at a step's end, and one workgroup barrier fewer per step.
//   IN_Y  : the input comes that way (else: an int32 GLWE, first step)    OUT_Y : the output leaves that way (else int32, last step)
// Round 5: the 2 * SK * SX operand polynomials of a step (768 KB at SK = 4) stream under the pairs of inverse transforms
// (fft_inv2_hooked): the products of the next pair of output limbs are taken between the phases of this pair's transforms.  Round 6: the
// same for five limbs (pairs + one: the lone limb's products under the previous pair's transforms, the next column's first under its own).
// ---------------------------------------------------------------------------------------
//   OUT2: (k_write_chain) the last trace step of write_mid_step hands normalize(ct_hi - trace(ct_hi) + trace(ct_lo X^-row)) (ram.rs:617,625-626)
//   to write_last_step's products: ka.b = the row (ct_hi), ka.out = trace(ct_hi); the result leaves as A (not Y), the mask column in
//   buffer 2, the body column in vc (ep_step_r IN = 2)
template <int SK, bool IN_Y, int OUT>
__device__ __forceinline__ void ks_trace_l(const KsArgs& ka, double* lds, bool load_tw, const int tid, double (&vc)[E], const bool stamp_on = false) {
    constexpr bool OUT_Y = OUT != 0;
    YSTAMP(0);
    constexpr int SX = 3;
    double* tw = lds;
    double* data = lds + LDS_TW;
    TwRegs twr;
    if (load_tw) twiddles_issue(twr, ka.tw, tid);
    const int32_t* ap = at(ka.a);
    int32_t* op = at(ka.out);
    const int sidx0 = (tid * ka.ginv) & (2 * N - 1);      // phi_g: destination i' = tid + T*k takes +-source i = i' * ginv mod 2N
    const int sstep = (T * ka.ginv) & (2 * N - 1);

    auto y_of = [](const RawX<KS_TRACE, SX>& r) {
        double a_ = __builtin_fma(__builtin_fma((double)r.a[0], TWO_B, (double)r.a[1]), TWO_B, (double)r.a[2]);
        a_ = r.neg ? -a_ : a_;                                       // the rotation's sign comes before the shift
        return __builtin_floor(__builtin_fma(a_, 0.5, 0.5));        // ceil(A / 2)
    };
    auto load_column = [&](int col, double (&y)[E]) {
        if constexpr (IN_Y) {
            const double* yp = reinterpret_cast<const double*>(ap);
#pragma unroll
            for (int k = 0; k < E; k++) y[k] = gload_f64(yp + (long)col * N, (unsigned)(tid + T * k) * 8u);
        } else {
            RawX<KS_TRACE, SX> rw[E];
#pragma unroll
            for (int k = 0; k < E; k++) load_raw<KS_TRACE, SX>(ka, ap, nullptr, col, tid + T * k, rw[k]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < E; k++) y[k] = y_of(rw[k]);
        }
    };
    // V of both columns starts as the column's own Y (natural order).  Hand-over through LDS and registers (IN_Y): the mask
    // column's Y was staged in exchange buffer 2 by the previous step (published by that step's later barriers), the body
    // column's Y comes in registers (vc).  First step of a chain: both from the int32 source.
    double* stage2 = data + 2 * LDS_DATA;
    double v1[E], v0[E];
    if constexpr (IN_Y) {
#pragma unroll
        for (int k = 0; k < E; k++) { v0[k] = vc[k]; v1[k] = stage2[tid + T * k]; }
    } else {
        load_column(1, v1);
        __builtin_amdgcn_sched_barrier(0);   // (one column's 24 raw limbs at a time)
        load_column(0, v0);
#pragma unroll
        for (int k = 0; k < E; k++) stage2[tid + T * k] = v1[k];
        if (load_tw) twiddles_commit(twr, tw, tid); else __syncthreads();   // its barrier also publishes the staged column
    }
    YSTAMP(1);
    // the digits of the mask column seen through phi_g (to be transformed)
    double xh[SX][E];
    {
        int sidx = sidx0;
#pragma unroll
        for (int k = 0; k < E; k++) {
            const bool ng = sidx >= N;
            double c = stage2[sidx & (N - 1)];
            const double d2 = take_digit(c);
            const double d1 = take_digit(c);
            xh[2][k] = ng ? -d2 : d2;
            xh[1][k] = ng ? -d1 : d1;
            xh[0][k] = ng ? -c : c;
            sidx = (sidx + sstep) & (2 * N - 1);
        }
    }
    constexpr bool STREAM = true;        // (round 6: odd limb counts — the README block's 5-limb keys — stream too: pairs + one, see below)
    static_assert(!(SK & 1) || SK == 5, "odd limb counts: the pairs-plus-one schedule below is written out for five limbs");
    constexpr int KW = FK_KS_WINDOW, NQ = 2 * SX;
    [[maybe_unused]] OpRegs w[STREAM ? KW : 1];
    [[maybe_unused]] double accn[2][E];
    // operand q of the pair of output limbs (j, j - 1) of column co_: limb j - q / SX, digit row q % SX
    [[maybe_unused]] auto kopnd = [&](int co_, int j_, int q) { return ka.key + (long)(((q % SX) * SK + (j_ - q / SX)) * 2 + co_) * N; };
    if constexpr (STREAM) {
#pragma unroll
        for (int i = 0; i < KW; i++) load_ops_p(w[i], kopnd(1, SK - 1, i), tid);
        __builtin_amdgcn_sched_barrier(0);
    }
    YSTAMP(2);
    fwd_all<SX, 0, true>(xh, tw, data, tid);   // its first exchange starts with a barrier: every gather above is done before the buffers are overwritten
    YSTAMP(3);
    // the body column's Y goes to buffer 2 (natural order), for the gather of column 0 (own coefficient +- phi_g's source
    // coefficient), once every wave is through the forward transforms (their wave-local exchanges use that buffer too)
    lds_barrier();
#pragma unroll
    for (int k = 0; k < E; k++) stage2[tid + T * k] = v0[k];
    double y1n[E];   // column 1's output Y, on its way to buffer 2 (the next step's mask staging)
#pragma unroll
    for (int k = 0; k < E; k++) y1n[k] = 0.0;
    if constexpr (STREAM) {
        // The first pair's products have nothing to run under.  Their first KW operands arrived during the forward transforms; the next KP go into
        // registers that are free here (no pair is being transformed yet: the second accumulator pair's) and are requested at once, the last
        // NQ - KW - KP into the window as it drains: the wait is one round trip instead of two.
        constexpr int KP = FK_KS_PROLOGUE_EXTRA;
        static_assert(KW + KP <= NQ, "prologue window");
        [[maybe_unused]] OpRegs pw[KP > 0 ? KP : 1];
#pragma unroll
        for (int i = 0; i < KP; i++) load_ops_p(pw[i], kopnd(1, SK - 1, KW + i), tid);
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < E; k++) accn[b][k] = 0.0;
        // order of use: w[0..KW-1] (polynomials 0..KW-1), pw (KW..KW+KP-1), then the window again (KW+KP..NQ-1)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            constexpr int dummy = 0; (void)dummy;
            const bool from_pw = (q >= KW && q < KW + KP);
            const int slot = (q < KW) ? q : (q - KW - KP) % KW;          // window slot of a polynomial that goes through the window
            if (from_pw) mac_regs(accn[q / SX], xh[q % SX], pw[(q - KW) < KP && q >= KW ? q - KW : 0]);
            else mac_regs(accn[q / SX], xh[q % SX], w[slot]);
            pin_regs(accn[q / SX]);
            __builtin_amdgcn_sched_barrier(0);
            if (!from_pw) {
                // the slot just drained takes the next polynomial that has no register yet: first the rest of this pair, then the next pair's first KW
                const int nxt = (q < KW) ? KW + KP + q : q + KW;          // position in the stream (>= NQ: the next pair)
                if (nxt < NQ) load_ops_p(w[slot], kopnd(1, SK - 1, nxt), tid);
                else if (nxt - NQ < KW) load_ops(w[(nxt - NQ)], kopnd(SK >= 4 ? 1 : 0, SK >= 4 ? SK - 3 : SK - 1, nxt - NQ), tid);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int ci = 0; ci < 2; ci++) {   // (unrolled: what is live across a column differs between the two)
        const int co = 1 - ci;
        double od[E], ec[E];
        if (co == 1) {
#pragma unroll
            for (int k = 0; k < E; k++) od[k] = v1[k];
        } else {
            int sidx = sidx0;
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double b = stage2[sidx & (N - 1)];
                od[k] = stage2[tid + T * k] + ((sidx >= N) ? -b : b);      // Y_body + phi_g(Y_body)
                sidx = (sidx + sstep) & (2 * N - 1);
            }
            if constexpr (OUT_Y) {      // every wave has gathered: buffer 2 takes the next step's mask column
                lds_barrier();
#pragma unroll
                for (int k = 0; k < E; k++) stage2[tid + T * k] = y1n[k];
            }
        }
#pragma unroll
        for (int k = 0; k < E; k++) ec[k] = 0.0;
        if constexpr (SK & 1) {
        // Five limbs: pairs (4,3), (2,1) and limb 0 on its own, per column: six units per step.  The operand stream is ONE sequence of
        // 2 * 15 polynomials (column 1 first; per column limb 4 digit rows 0..2, limb 3, ... limb 0): position p -> column p < 15 ? 1 : 0,
        // limb 4 - (p % 15) / 3, digit row p % 3, register set p % KW.  The products of positions 0..5 (column 1's first pair) were taken
        // in the prologue above; every later product is taken between the phases of an EARLIER unit's transforms: a pair's transforms
        // (fft_inv2_hooked, six places) host the next six positions, the lone limb's (fft_inv1_hooked) the next three — so while the
        // lone limb of column 1 is summed into accn[0], accn[1] already takes limb 4 of column 0, and limb 3 goes to accn[0] under the
        // lone limb's own transform (the pair then arrives swapped: SWAP).
        constexpr int NTOT = 2 * SK * SX;
        [[maybe_unused]] auto kop = [&](int p_) { return kopnd(p_ < SK * SX ? 1 : 0, SK - 1, p_ % (SK * SX)); };
        auto unit = [&](auto nb_tag, auto p_tag, auto nh_tag, auto swap_tag, auto j_tag) {
            constexpr int NB = decltype(nb_tag)::value, P = decltype(p_tag)::value, NH = decltype(nh_tag)::value, J = decltype(j_tag)::value;
            constexpr bool SWAP = decltype(swap_tag)::value;
            int tid_u = tid;
            asm volatile("" : "+v"(tid_u));   // per-unit copy (see k_ext_product_chain): six unrolled units would otherwise share — and keep alive — every address derived from it
            __builtin_assume(tid_u >= 0 && tid_u < T);
            double acc[NB][E];
            if constexpr (NB == 2) {
#pragma unroll
                for (int k = 0; k < E; k++) { acc[0][k] = accn[SWAP ? 1 : 0][k]; acc[1][k] = accn[SWAP ? 0 : 1][k]; accn[0][k] = 0.0; accn[1][k] = 0.0; }
            } else {
#pragma unroll
                for (int k = 0; k < E; k++) { acc[0][k] = accn[0][k]; accn[0][k] = 0.0; }      // (accn[1] may hold the next column's first limb)
            }
            auto hook = [&](auto stag) {
                constexpr int sl = decltype(stag)::value;
                if constexpr (sl < NH) {
                    constexpr int pp = P + sl;
                    mac_regs(accn[sl / SX], xh[pp % SX], w[pp % KW]);
                    pin_regs(accn[sl / SX]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (pp + KW < NTOT) load_ops(w[pp % KW], kop(pp + KW), tid_u);
                }
            };
            if constexpr (NB == 2) {
                fft_inv2_hooked<2>(acc, tw, data, data + LDS_DATA, tid_u, hook);
                fold_limb<SK>(od, ec, acc[0], J);
                fold_limb<SK>(od, ec, acc[1], J - 1);
            } else {
                fft_inv1_hooked<2>(acc, tw, data, tid_u, hook);
                fold_limb<SK>(od, ec, acc[0], J);
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        if (ci == 0) {
            unit(integral_constant<int, 2>{}, integral_constant<int, 6>{}, integral_constant<int, 6>{}, false_type{}, integral_constant<int, 4>{});
            unit(integral_constant<int, 2>{}, integral_constant<int, 12>{}, integral_constant<int, 6>{}, false_type{}, integral_constant<int, 2>{});
            unit(integral_constant<int, 1>{}, integral_constant<int, 18>{}, integral_constant<int, 3>{}, false_type{}, integral_constant<int, 0>{});
        } else {
            unit(integral_constant<int, 2>{}, integral_constant<int, 21>{}, integral_constant<int, 6>{}, true_type{}, integral_constant<int, 4>{});
            unit(integral_constant<int, 2>{}, integral_constant<int, 27>{}, integral_constant<int, 3>{}, false_type{}, integral_constant<int, 2>{});
            unit(integral_constant<int, 1>{}, integral_constant<int, 30>{}, integral_constant<int, 0>{}, false_type{}, integral_constant<int, 0>{});
        }
        } else {
        // The operand stream under the inverse transforms: the six products of the NEXT pair of output limbs are taken between the phases of
        // this pair's transforms (fft_inv2_hooked), each from a register set that is refilled at once with the polynomial W places further on in
        // the step's stream of 2 * SK * SX operand polynomials.  Two accumulator pairs: the one being transformed, the one being summed.
        auto batch = [&](auto hn_tag, auto hnn_tag, const int j) {
            constexpr bool HN = decltype(hn_tag)::value, HNN = decltype(hnn_tag)::value;
            const int tid_u = tid;   // (a per-batch copy the optimiser cannot see through, as in the five-limb units above, removes two thirds of this kernel's spill code and is 0.7 % slower: profiles/r06_experiments.txt)
            double acc[2][E];
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int k = 0; k < E; k++) { acc[b][k] = accn[b][k]; accn[b][k] = 0.0; }
            const int nco = (j >= 3) ? co : 1 - co, nj = (j >= 3) ? j - 2 : SK - 1;
            const int nnco = (nj >= 3) ? nco : 1 - nco, nnj = (nj >= 3) ? nj - 2 : SK - 1;
            auto hook = [&](auto stag) {
                constexpr int q = decltype(stag)::value;
                if constexpr (HN) {
                    mac_regs(accn[q / SX], xh[q % SX], w[q % KW]);
                    pin_regs(accn[q / SX]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q + KW < NQ) load_ops(w[q % KW], kopnd(nco, nj, q + KW), tid_u);
                    else if constexpr (HNN) load_ops(w[q % KW], kopnd(nnco, nnj, q + KW - NQ), tid_u);
                }
            };
            YSTAMP(8 + (ci * SK + (SK - 1 - j)) * 4);
            fft_inv2_hooked<2>(acc, tw, data, data + LDS_DATA, tid_u, hook);
            YSTAMP(9 + (ci * SK + (SK - 1 - j)) * 4);
            if constexpr (SK == 4) { fold_limb4<SK>(od, acc[0], j); fold_limb4<SK>(od, acc[1], j - 1); }
            else { fold_limb<SK>(od, ec, acc[0], j); fold_limb<SK>(od, ec, acc[1], j - 1); }
            YSTAMP(11 + (ci * SK + (SK - 1 - j)) * 4);
        };
        using std::true_type;
        using std::false_type;
        if (ci == 0) {
#pragma unroll 1
            for (int j = SK - 1; j >= 1; j -= 2) batch(true_type{}, true_type{}, j);
        } else {
#pragma unroll 1
            for (int j = SK - 1; j >= 5; j -= 2) batch(true_type{}, true_type{}, j);
            batch(true_type{}, false_type{}, 3);
            batch(false_type{}, false_type{}, 1);
        }
        }
#pragma unroll
        for (int k = 0; k < E; k++) {
            double a_ = window51(od[k]);
            if constexpr (OUT == 2) {
                // normalize(ct_hi - trace(ct_hi) + this) in closed form: the three operands as the integers their limbs stand for
                const int32_t* hp = at(ka.b);
                const int32_t* tp = op;
                auto whole = [&](const int32_t* q) {
                    const double l0 = (double)gload_i32(q + glwe_off(0, co), (unsigned)(tid + T * k) * 4u);
                    const double l1 = (double)gload_i32(q + glwe_off(1, co), (unsigned)(tid + T * k) * 4u);
                    const double l2 = (double)gload_i32(q + glwe_off(2, co), (unsigned)(tid + T * k) * 4u);
                    return __builtin_fma(__builtin_fma(l0, TWO_B, l1), TWO_B, l2);
                };
                const double an = window51(whole(hp) - whole(tp) + a_);
                if (co == 1) y1n[k] = an; else vc[k] = an;
            } else if constexpr (OUT_Y) {
                const double yn = __builtin_floor(__builtin_fma(a_, 0.5, 0.5));
                if (co == 1) y1n[k] = yn; else vc[k] = yn;      // nothing goes to global memory between the steps of a chain
            } else {
                const double d2 = take_digit(a_);
                const double d1 = take_digit(a_);
                gstore_i32(op + glwe_off(2, co), (unsigned)(tid + T * k) * 4u, (int)d2);
                gstore_i32(op + glwe_off(1, co), (unsigned)(tid + T * k) * 4u, (int)d1);
                gstore_i32(op + glwe_off(0, co), (unsigned)(tid + T * k) * 4u, (int)a_);
            }
        }
    }
    YSTAMP(5);
}

// ---------------------------------------------------------------------------------------
// ep_step_r (round 4): one product INSIDE a product chain, handing its result to the next product of the same workgroup in
// registers and LDS instead of through global memory.  A product's input is consumed at the natural coefficients of each
// thread only (no gather: the external product has no automorphism), so the hand-over needs no barrier at all: the result
// travels as A = the integer whose balanced base-2^17 digits are the three output limbs (closed-form normalisation, see
// ks_trace_z: one accumulator per coefficient, no carry chain), column 1 in registers (ac), column 0 — finished first — in
// thread-private slots of the third exchange buffer (`park`), which the inverse transforms leave alone.  The consumer takes its digits
// with take_digit.  The operand stream runs under the transforms (round 5: see the comment at the first request below).
//   IN_R : the input comes that way (else an int32 GLWE: first product)     OUT_R : the output leaves that way (else int32: last)
// ---------------------------------------------------------------------------------------
//   IN : 0 an int32 GLWE (first product), 1 from the previous product (column 0 parked in LDS, column 1 in ac), 2 from a trace step
//        (k_write_chain: column 0 in ac, column 1 in exchange buffer 2 in natural order)
//   OUT: 0 an int32 GLWE (last product), 1 to the next product, 3 to a trace step (k_read_chain: the columns are produced in the
//        other order and leave as Y = ceil(A/2), the mask column staged in buffer 2, the body column in ac; store_i32: the int32
//        limbs are stored as well — read_prepare_write's products are in place, ram.rs:502-504)
template <int SG, int IN, int OUT>
__device__ __forceinline__ void ep_step_r(GlweRef a, GlweRef res, const double* __restrict__ ggsw, const double* __restrict__ tw_g,
                                          double* lds, bool load_tw, const int tid, double (&ac)[E], const bool stamp_on = false, const bool store_i32 = false) {
    YSTAMP(0);
    constexpr int SA = 3;
    double* tw = lds;
    double* data = lds + LDS_TW;
    TwRegs twr;
    if (load_tw) twiddles_issue(twr, tw_g, tid);
    const int32_t* ap = at(a);
    int32_t* rp = at(res);
    double* park = data + 2 * LDS_DATA + (tid >> 6) * (64 * (E + 1)) + (tid & 63);   // thread-private slots of buffer 2 (tid is the VIRTUAL id: a (tid >> 6) group spans two hardware waves, which is harmless because no slot is shared)
    double x0[SA][E], x1[SA][E];
    auto digits_of = [&](const double (&av)[E], double (&x)[SA][E]) {
#pragma unroll
        for (int k = 0; k < E; k++) {
            double c = av[k];
            x[2][k] = take_digit(c);
            x[1][k] = take_digit(c);
            x[0][k] = c;            // A lies in the digit window: the second quotient IS the top digit
        }
    };
    auto load_limbs = [&](int col, double (&x)[SA][E]) {
        int xi[SA][E];
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) xi[r][k] = gload_i32(ap + glwe_off(r, col), (unsigned)(tid + T * k) * 4u);
#pragma unroll
        for (int r = 0; r < SA; r++)
#pragma unroll
            for (int k = 0; k < E; k++) x[r][k] = (double)xi[r][k];
    };
    [[maybe_unused]] double a1s[E];
    if constexpr (IN == 1) {
        double a0[E];
#pragma unroll
        for (int k = 0; k < E; k++) a0[k] = park[64 * k];      // read before the forward transforms overwrite the buffer
        digits_of(a0, x0);
    } else if constexpr (IN == 2) {
#pragma unroll
        for (int k = 0; k < E; k++) a1s[k] = data[2 * LDS_DATA + tid + T * k];   // column 1, staged by the trace step (natural order): read before the forward transforms overwrite the buffer
        digits_of(ac, x0);
    } else {
        load_limbs(0, x0);
        if (load_tw) twiddles_commit(twr, tw, tid);
    }
    YSTAMP(1);
    // The operand stream (2 * SA * 2 * SG prepared polynomials of 32 KB: 1.5 MB per product at SG = 4) runs UNDER the transforms:
    // the products of output limb u + 1 are taken between the phases of limb u's inverse transform and around its fold, each from
    // a register set that is refilled at once with the polynomial W places further on in the stream (across limbs and columns:
    // the stream is one sequence).  Two accumulators: the one being transformed, the one being summed.  The first limb's products
    // have no inverse transform to run under: its first W operands arrive during the first three forward transforms, the next W
    // during the other three.
    constexpr int W = FK_EP_WINDOW;
    constexpr int NQ = 2 * SA;
    static_assert(W >= 1 && W <= SA, "window");
    OpRegs w[W];
    double accn[E];
    auto opnd = [&](int co_, int j_, int q) { return ggsw + (long)(((2 * (q % SA) + q / SA) * SG + j_) * 2 + co_) * N; };
    const int co0 = (OUT == 3) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < W; i++) load_ops_p(w[i], opnd(co0, SG - 1, i), tid);
    __builtin_amdgcn_sched_barrier(0);
    fwd_all<SA>(x0, tw, data, tid);
    YSTAMP(2);
#pragma unroll
    for (int k = 0; k < E; k++) accn[k] = 0.0;
#pragma unroll
    for (int q = 0; q < W; q++) {
        mac_regs(accn, x0[q], w[q]);
        pin_regs(accn);
        __builtin_amdgcn_sched_barrier(0);
        load_ops_p(w[q], opnd(co0, SG - 1, q + W), tid);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (IN == 1) digits_of(ac, x1); else if constexpr (IN == 2) digits_of(a1s, x1); else load_limbs(1, x1);
    fwd_all<SA>(x1, tw, data, tid);
    YSTAMP(3);
#pragma unroll
    for (int q = W; q < NQ; q++) {
        mac_regs(accn, q < SA ? x0[q % SA] : x1[q % SA], w[q % W]);
        pin_regs(accn);
        __builtin_amdgcn_sched_barrier(0);
        if (q + W < NQ) load_ops_p(w[q % W], opnd(co0, SG - 1, q + W), tid);
        else load_ops(w[q % W], opnd(co0, SG - 2, q + W - NQ), tid);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {   // (unrolled: the carried column is written by the second pass only — a rolled loop keeps its old value alive throughout)
        const int co = (OUT == 3) ? 1 - cc : cc;
        double od[E], ec[E];
#pragma unroll
        for (int k = 0; k < E; k++) { od[k] = 0.0; ec[k] = 0.0; }
        // one output limb: its inverse transform with the NEXT limb's products between the phases.  HN: there is a next limb in the
        // stream; HNN: and one behind that (whose first W polynomials are requested here).  Only the last two limbs of a product
        // differ: they are peeled, so that the steady-state body has no branch.
        auto unit = [&](auto hn_tag, auto hnn_tag, const int j) {
            constexpr bool HN = decltype(hn_tag)::value, HNN = decltype(hnn_tag)::value;
            const int tid_u = tid;
            double acc[1][E];
#pragma unroll
            for (int k = 0; k < E; k++) { acc[0][k] = accn[k]; accn[k] = 0.0; }
            const int nco = (j > 0) ? co : 1 - co, nj = (j > 0) ? j - 1 : SG - 1;
            const int nnco = (nj > 0) ? nco : 1 - nco, nnj = (nj > 0) ? nj - 1 : SG - 1;
            auto step = [&](auto qtag) {
                constexpr int q = decltype(qtag)::value;
                if constexpr (HN) {
                    mac_regs(accn, q < SA ? x0[q % SA] : x1[q % SA], w[q % W]);
                    pin_regs(accn);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (q + W < NQ) load_ops(w[q % W], opnd(nco, nj, q + W), tid_u);
                    else if constexpr (HNN) load_ops(w[q % W], opnd(nnco, nnj, q + W - NQ), tid_u);
                }
            };
            auto hook = [&](auto stag) {
                constexpr int sl = decltype(stag)::value;
                step(std::integral_constant<int, sl>{});   // one product per place: evenly spaced requests
            };
            YSTAMP(8 + (co * SG + (SG - 1 - j)) * 4);
            fft_inv1_hooked<2>(acc, tw, data, tid_u, hook);   // one exchange buffer, fenced by the free counter; buffer 1 holds V (below)
            YSTAMP(9 + (co * SG + (SG - 1 - j)) * 4);
            __builtin_amdgcn_sched_barrier(0);
            step(std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            {   // V of this column lives in this thread's own slots of exchange buffer 1 between the folds (16 registers fewer across the transform)
                double2* odp = reinterpret_cast<double2*>(data + LDS_DATA) + tid_u;
                if (j != SG - 1) {
#pragma unroll
                    for (int kk = 0; kk < E / 2; kk++) { const double2 v = odp[kk * T]; od[2 * kk] = v.x; od[2 * kk + 1] = v.y; }
                }
                if constexpr (SG == 4) fold_limb4<SG>(od, acc[0], j); else fold_limb<SG>(od, ec, acc[0], j);
                if (j != 0) {
#pragma unroll
                    for (int kk = 0; kk < E / 2; kk++) { double2 v; v.x = od[2 * kk]; v.y = od[2 * kk + 1]; odp[kk * T] = v; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            step(std::integral_constant<int, 5>{});
            __builtin_amdgcn_sched_barrier(0);
            YSTAMP(10 + (co * SG + (SG - 1 - j)) * 4);
        };
        using std::true_type;
        using std::false_type;
        if (cc == 0) {
#pragma unroll 1
            for (int j = SG - 1; j >= 0; j--) unit(true_type{}, true_type{}, j);
        } else {
#pragma unroll 1
            for (int j = SG - 1; j >= 2; j--) unit(true_type{}, true_type{}, j);
            unit(true_type{}, false_type{}, 1);
            unit(false_type{}, false_type{}, 0);
        }
#pragma unroll
        for (int k = 0; k < E; k++) {
            double a_ = window51(od[k]);
            if constexpr (OUT == 1) {
                if (co == 0) park[64 * k] = a_; else ac[k] = a_;
            } else if constexpr (OUT == 3) {
                // to a trace step: Y = ceil(A / 2); the mask column (first) into buffer 2 in natural order — the inverse transforms
                // leave that buffer alone, every wave is through the forward transforms, nobody parks in this step, and the barriers
                // of the other column's transforms publish it —, the body column (last) in registers
                const double yn = __builtin_floor(__builtin_fma(a_, 0.5, 0.5));
                if (co == 1) data[2 * LDS_DATA + tid + T * k] = yn; else ac[k] = yn;
                if (store_i32) {
                    const double d2 = take_digit(a_);
                    const double d1 = take_digit(a_);
                    gstore_i32(rp + glwe_off(2, co), (unsigned)(tid + T * k) * 4u, (int)d2);
                    gstore_i32(rp + glwe_off(1, co), (unsigned)(tid + T * k) * 4u, (int)d1);
                    gstore_i32(rp + glwe_off(0, co), (unsigned)(tid + T * k) * 4u, (int)a_);
                }
            } else {
                const double d2 = take_digit(a_);
                const double d1 = take_digit(a_);
                gstore_i32(rp + glwe_off(2, co), (unsigned)(tid + T * k) * 4u, (int)d2);
                gstore_i32(rp + glwe_off(1, co), (unsigned)(tid + T * k) * 4u, (int)d1);
                gstore_i32(rp + glwe_off(0, co), (unsigned)(tid + T * k) * 4u, (int)a_);
            }
        }
    }
    YSTAMP(5);
}
// the product chain with that hand-over (n >= 2)
template <int SG>
__global__ __launch_bounds__(T, T / 256) __attribute__((amdgpu_num_vgpr(FK_CHAIN_VGPRS))) void k_ext_product_chain_r(EpChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ca.tw);
    GlweRef in = ca.src;
    double ac[E];
#pragma unroll
    for (int k = 0; k < E; k++) ac[k] = 0.0;
#pragma unroll 1
    for (int i = 0; i < ca.n; i++) {
        const GlweRef out = ca.buf[i & 1];
        int tid = vt((int)threadIdx.x);
        asm volatile("" : "+v"(tid));   // see k_ext_product_chain
        __builtin_assume(tid >= 0 && tid < T);
        if (i == 0) ep_step_r<SG, 0, 1>(in, out, ca.ggsw[i], ca.tw, lds, true, tid, ac);
        else if (i + 1 < ca.n) ep_step_r<SG, 1, 1>(in, out, ca.ggsw[i], ca.tw, lds, false, tid, ac, YSTAMP_STEP(i + 1));
        else ep_step_r<SG, 1, 0>(in, out, ca.ggsw[i], ca.tw, lds, false, tid, ac);
        in = out;
    }
}

// GLWE::trace(start, start + n) (ram.rs:457,540,572,616,621 and the packer levels in which every
// leaf is alone) as ONE launch: n trace steps on the same ciphertext, one workgroup per ciphertext, ping-pong
// between the workgroup's own slots of two buffers (see k_ext_product_chain).  Only the first step may read its
// input rotated (write path).
struct KsChainArgs {
    KsArgs base;                     // a = source, rot_mul / rot_base of the first step, tw, big unused
    GlweRef buf[2];                  // step i writes buf[i & 1] (buf[0] must not be the source)
    const double* key[CHAIN_MAX];
    int ginv[CHAIN_MAX];
    int n;
    const unsigned* pred = nullptr;  // fallback launch behind k_trace_tail: runs only if *pred == pred_seq (that launch gave up)
    unsigned pred_seq = 0;
    unsigned* host_count = nullptr;  // pinned host word that mirrors the number of fallbacks taken (read by the host without a sync)
    unsigned* done = nullptr;        // fallback launch behind k_chain_mid: a ciphertext is redone unless done[ct * 32 + 3] == done_seq
    unsigned done_seq = 0;
};
// YF: the intermediates of the chain are handed over as Y = ceil(A/2) (ks_trace_y).  YF = false is the limb-form chain; it
// is also what runs as the predicated fallback behind k_trace_tail: that launch normally has nothing to do, but it needs
// its registers and LDS granted before it can say so, and it must slip in next to the side-stream work that
// read_prepare_write starts beside the trace chain — with the leaner limb-form kernel (<= 232 VGPRs: two waves leave room
// on a SIMD) it does; the Y-form kernel, which takes the whole register file, waited for that work to drain (+0.2 ms per
// read_prepare_write, measured).
// Register budget: one workgroup per CU, two waves per SIMD.  At 249+ registers the two waves take a SIMD's whole register
// file, and ANY other wave resident on the CU — the one-wave gate launch that read_prepare_write parks on the side stream is
// enough — keeps the workgroup off that CU: a 256-workgroup launch on 256 CUs then runs in two rounds (+0.24 ms per
// read_prepare_write, measured when the Y-form kernel first compiled to 250).  Capped so that a small wave still fits.
// ---------------------------------------------------------------------------------------
// k_pair_z (round 4): the packer combine (GLWEPacker, ram.rs:435,514; KS_PAIR of ks_run) split by output column, in the closed
// form of ks_trace_z.  With A(v) = v_0 2^34 + v_1 2^17 + v_2 the integer a coefficient's limbs stand for and a' = rot(a, -t):
//   x   = rsh1(a' - b)            its limbs are the balanced digits of  Yx = ceil((A(a') - A(b)) / 2),  top digit wrapped
//   u   = rsh1(a' + b)            ... of Yu = ceil((A(a') + A(b)) / 2)        (rsh1 is exact on un-normalised limbs: the half a
//                                 limb loses goes to the next one as -2^16, only the last limb rounds, up)
//   tmp = normalize(phi_g(KS(x.mask) + (x.body, 0)))      =  window51(W),  W = e + big_2 + cmod34(big_1) 2^17 + cmod17(big_0) 2^34
//                                                            (+- window51(Yx.body) at phi_g's source coefficient, body column)
//   out = rot(normalize(u - tmp), +t)                     =  the digits of window51(window51(Yu) - window51(W)), moved by t with
//                                                            the rotation's sign
// One workgroup per (pair, output column): 3 forward transforms, 2 x 2 inverse transforms in skewed pairs, no carry chains, no
// second normalisation walk; both inputs and the output are int32 GLWEs (the levels of the packing tree are launches).
// ---------------------------------------------------------------------------------------
template <int SK>
__global__ __launch_bounds__(T, T / 256) __attribute__((amdgpu_num_vgpr(FK_CHAIN_VGPRS))) void k_pair_z(KsArgs ka) {   // (capped as the chain kernels: the first pair level of read_prepare_write is 256 workgroups beside the gate wave)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ka.tw);
    static_assert((SK & 1) == 0 || SK == 5, "pairs of output limbs (+ one)");
    constexpr int SX = 3;
    const int tid = vt((int)threadIdx.x);
    const int co = (int)blockIdx.z;
    double* tw = lds;
    double* data = lds + LDS_TW;
    TwRegs twr;
    twiddles_issue(twr, ka.tw, tid);
    const int32_t* ap = at(ka.a);
    const int32_t* bp = at(ka.b);
    int32_t* op = at(ka.out);
    double* mstage = data;                 // exchange buffer 0: Yx of the mask column, natural order
    double* bstage = data + LDS_DATA;      // exchange buffer 1: window51(Yx) of the body column (column 0 only)
    const int sidx0 = (tid * ka.ginv) & (2 * N - 1);
    const int sstep = (T * ka.ginv) & (2 * N - 1);
    // A(a') and A(b) of column `col` at the natural coefficients tid + T*k
    auto load_ab = [&](int col, double (&aa)[E], double (&ab)[E]) {
        RawX<KS_PAIR, SX> rw[E];
#pragma unroll
        for (int k = 0; k < E; k++) load_raw<KS_PAIR, SX>(ka, ap, bp, col, tid + T * k, rw[k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < E; k++) {
            const double va = __builtin_fma(__builtin_fma((double)rw[k].a[0], TWO_B, (double)rw[k].a[1]), TWO_B, (double)rw[k].a[2]);
            aa[k] = rw[k].neg ? -va : va;
            ab[k] = __builtin_fma(__builtin_fma((double)rw[k].b[0], TWO_B, (double)rw[k].b[1]), TWO_B, (double)rw[k].b[2]);
        }
    };
    double u[E];        // window51(Yu) of this column
    {
        double aa[E], ab[E];
        load_ab(1, aa, ab);
#pragma unroll
        for (int k = 0; k < E; k++) {
            mstage[tid + T * k] = __builtin_floor(__builtin_fma(aa[k] - ab[k], 0.5, 0.5));
            if (co == 1) u[k] = window51(__builtin_floor(__builtin_fma(aa[k] + ab[k], 0.5, 0.5)));
        }
        if (co == 0) {
            load_ab(0, aa, ab);
#pragma unroll
            for (int k = 0; k < E; k++) {
                bstage[tid + T * k] = window51(__builtin_floor(__builtin_fma(aa[k] - ab[k], 0.5, 0.5)));
                u[k] = window51(__builtin_floor(__builtin_fma(aa[k] + ab[k], 0.5, 0.5)));
            }
        }
    }
    twiddles_commit(twr, tw, tid);      // its barrier also publishes the staged columns
    double xh[SX][E];
    double od[E], ec[E];
    {
        int sidx = sidx0;
#pragma unroll
        for (int k = 0; k < E; k++) {
            const bool ng = sidx >= N;
            double c = mstage[sidx & (N - 1)];
            const double d2 = take_digit(c);
            const double d1 = take_digit(c);
            const double d0 = cmod17(c);         // the limbs' top digit is wrapped (|Yx| reaches 2^50: the quotient can be +-2^16)
            xh[2][k] = ng ? -d2 : d2;
            xh[1][k] = ng ? -d1 : d1;
            xh[0][k] = ng ? -d0 : d0;
            double w = 0.0;
            if (co == 0) { const double b = bstage[sidx & (N - 1)]; w = ng ? -b : b; }
            od[k] = w;
            ec[k] = 0.0;
            sidx = (sidx + sstep) & (2 * N - 1);
        }
    }
    OpRegs g[SX];
#pragma unroll
    for (int r = 0; r < SX; r++) load_ops(g[r], ka.key + (long)((r * SK + (SK - 1)) * 2 + co) * N, tid);
    fwd_all<SX>(xh, tw, data, tid);   // its first exchange starts with a barrier: every gather above is done
    auto pair = [&](int j) {
        double acc[2][E];
#pragma unroll
        for (int k = 0; k < E; k++) { acc[0][k] = 0.0; acc[1][k] = 0.0; }
#pragma unroll
        for (int r = 0; r < SX; r++) {
            mac_regs(acc[0], xh[r], g[r]);
            __builtin_amdgcn_sched_barrier(0);
            load_ops(g[r], ka.key + (long)((r * SK + (j - 1)) * 2 + co) * N, tid);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < SX; r++) mac_regs(acc[1], xh[r], g[r]);
        __builtin_amdgcn_sched_barrier(0);
        ntt_inv2_loop(acc, tw, data, data + LDS_DATA, tid);
        fold_limb<SK>(od, ec, acc[0], j);
        fold_limb<SK>(od, ec, acc[1], j - 1);
        if (j >= 2) {   // (requested in front of this pair's transforms instead: 48 registers live across them, read 0.4659 -> 0.4670 ms: not kept)
#pragma unroll
            for (int r = 0; r < SX; r++) load_ops(g[r], ka.key + (long)((r * SK + (j - 2)) * 2 + co) * N, tid);
        }
    };
#pragma unroll 1
    for (int j = SK - 1; j >= 1; j -= 2) pair(j);
    if constexpr (SK & 1) {
        double acc[1][E];
#pragma unroll
        for (int k = 0; k < E; k++) acc[0][k] = 0.0;
#pragma unroll
        for (int r = 0; r < SX; r++) mac_regs(acc[0], xh[r], g[r]);
        ntt_inv1_loop(acc, tw, data, tid);
        fold_limb<SK>(od, ec, acc[0], 0);
    }
#pragma unroll
    for (int k = 0; k < E; k++) {
        double r_ = window51(u[k] - window51(od[k]));
        const double d2 = take_digit(r_);
        const double d1 = take_digit(r_);
        int dst = tid + T * k + ka.t;
        const bool ng = dst >= N;
        if (ng) dst -= N;
        gstore_i32(op + glwe_off(2, co), (unsigned)dst * 4u, cneg((int)d2, ng));
        gstore_i32(op + glwe_off(1, co), (unsigned)dst * 4u, cneg((int)d1, ng));
        gstore_i32(op + glwe_off(0, co), (unsigned)dst * 4u, cneg((int)r_, ng));
    }
}

// ---------------------------------------------------------------------------------------
// k_read_chain / k_write_chain (round 4): the two dependent chains that a row goes through back to back — one workgroup per
// ciphertext in both — as ONE launch, the ciphertext handed from the last step of one to the first step of the other in
// registers and LDS like between the steps of either (ep_step_r, ks_trace_l).
//   read / read_prepare_write (ram.rs:429-435,502-514): the products of coordinate 0's digits, then the packer levels in which
//       every row is alone.  store_ep: read_prepare_write's products are in place (ram.rs:502-504): their result is ALSO written
//       to the rows (ep.buf[(n_ep - 1) & 1]).
//   write (ram.rs:612-646): write_mid_step's trace(ct_lo X^-row), the elementwise normalize(ct_hi - trace(ct_hi) + that), then
//       write_last_step's products with the inverse digits of coordinate 0, in place on the rows.
// What a launch boundary costs between them — the launch gap, the twiddle table, an int32 round trip through L2 with its
// conversions, the first / last step variants of both chains, and for the write a whole elementwise launch — is paid once.
// ---------------------------------------------------------------------------------------
struct RowChainArgs {
    EpChainArgs ep;
    KsChainArgs ks;
    GlweRef hi, trhi;      // k_write_chain: the rows (ct_hi) and trace(ct_hi)
    int store_ep = 0;      // k_read_chain: the products' result is also stored (in-place products of read_prepare_write)
};
// (defined in chain_kernels.inc, which is included twice: k_keyswitch_chain / k_read_chain capped at FK_CHAIN_VGPRS registers for the launches that may
// meet the gate wave, k_keyswitch_chain_w / k_read_chain_w with the whole register file for those that cannot — the attribute wants a literal,
// and a shared body function would take the kernel's argument struct by reference, i.e. copy it to scratch)
#define FK_KS_CHAIN_NAME k_keyswitch_chain
#define FK_READ_CHAIN_NAME k_read_chain
#define FK_VG FK_CHAIN_VGPRS
#include "chain_kernels.inc"
#undef FK_KS_CHAIN_NAME
#undef FK_READ_CHAIN_NAME
#undef FK_VG
#define FK_KS_CHAIN_NAME k_keyswitch_chain_w
#define FK_READ_CHAIN_NAME k_read_chain_w
#define FK_VG FK_WIDE_VGPRS
#include "chain_kernels.inc"
#undef FK_KS_CHAIN_NAME
#undef FK_READ_CHAIN_NAME
#undef FK_VG

template <int SK, int SG>   // (only ever launched by Ram::write: never beside the gate wave)
__global__ __launch_bounds__(T, T / 256) __attribute__((amdgpu_num_vgpr(FK_WIDE_VGPRS))) void k_write_chain(RowChainArgs ra) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ra.ep.tw);
    double vc[E];   // (written by the first step before anything reads it)
    KsArgs ka = ra.ks.base;
#pragma unroll 1
    for (int i = 0; i < ra.ks.n; i++) {          // n >= 2
        ka.out = ra.ks.buf[i & 1];
        ka.key = ra.ks.key[i];
        ka.ginv = ra.ks.ginv[i];
        int tid = vt((int)threadIdx.x);
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < T);
        if (i == 0) ks_trace_l<SK, false, 1>(ka, lds, true, tid, vc);
        else if (i + 1 < ra.ks.n) ks_trace_l<SK, true, 1>(ka, lds, false, tid, vc);
        else { ka.b = ra.hi; ka.out = ra.trhi; ks_trace_l<SK, true, 2>(ka, lds, false, tid, vc); }
        ka.rot_mul = 0;
        ka.rot_base = 0;
    }
    GlweRef in = ra.ep.src;
#pragma unroll 1
    for (int i = 0; i < ra.ep.n; i++) {          // n >= 2
        const GlweRef out = ra.ep.buf[i & 1];
        int tid = vt((int)threadIdx.x);
        asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < T);
        if (i == 0) ep_step_r<SG, 2, 1>(in, out, ra.ep.ggsw[i], ra.ep.tw, lds, false, tid, vc);
        else if (i + 1 < ra.ep.n) ep_step_r<SG, 1, 1>(in, out, ra.ep.ggsw[i], ra.ep.tw, lds, false, tid, vc);
        else ep_step_r<SG, 1, 0>(in, out, ra.ep.ggsw[i], ra.ep.tw, lds, false, tid, vc);
        in = out;
    }
}

// ---------------------------------------------------------------------------------------
// k_trace_tail: the dependent trace chain at the end of a read (ram.rs:457,540: n steps on word_size ciphertexts)
// as ONE launch that keeps the fine limb split: 2*SK*SX workgroups per ciphertext, one forward and one inverse
// transform each per step, then a normalisation phase, with TWO IN-KERNEL HAND-OFFS per step instead of two kernel
// boundaries.  What makes the hand-off cheap (tools/xcd_barrier.hip: 1.1 us against 3 us for a kernel boundary and
// 9-18 us for agent-scope fences): the workgroups of a ciphertext all sit on ONE XCD — block b runs on XCD b % 8
// (round-robin placement), so group g = the blocks with b % 8 == g — and an XCD has one L2: a producer only has to
// drain its stores (vmcnt 0; the L1 is write-through) and a consumer only has to bypass its own L1 (sc1 loads);
// no L2 write-back or invalidate.  The placement is CHECKED, not assumed: every workgroup publishes its XCC id and
// the group gives up unless they agree.  Every wait is bounded: a group that cannot meet (its workgroups are not
// co-resident because another context holds the CUs, wrong placement) raises the abort word to this launch's
// generation and leaves; the fused chain launch enqueued right behind (k_keyswitch_chain, predicated on that word)
// then redoes the chain from the untouched source.  Results are the same either way (same per-coefficient arithmetic).
// Round 6: coordinate 1's external products (ram.rs:454 / 525-527: the steps in front of that trace, two digits or more) run in the SAME launch
// as product steps — fine split as k_ext_product_fine (member (co, j, r) of the first 24: digit r of both columns, two operand polynomials, one
// inverse transform) and the products' own normalisation phase; the launch enqueued behind is then the predicated fused row chain (k_read_chain:
// products + trace steps on one workgroup per ciphertext).  Measured and removed the same round (profiles/r06_experiments.txt 3, 7): ONE hand-off per
// step with the closed-form contributions added by L2 atomics, and the last pair levels of the packing tree as steps of this launch.
//   grid (8 * 2*SK*SX): group g = (b - xoff) % 8 = ciphertext (x = g % gx, y = g / gx), member m = b / 8 = ((co*SK + (SK-1-j))*SX + r)
//   sync: [group][32] words: arrivals, leavers, XCC mask;  sync[8*32] = abort generation
// ---------------------------------------------------------------------------------------
constexpr int TAIL_GROUPS = 8;
// One wave on the side stream that holds back what is enqueued behind it until the single-launch trace chain of
// generation `seq` says its workgroups are placed (sync[8*32 + 2]), or GATE_SPIN_MAX polls have passed (a few ms: only the
// timing of the work behind it depends on this, never a result).  An event recorded on the main stream for the same
// purpose delays the launch behind it by 7-13 us; a stream wait-value makes the command processor poll (slower still).
constexpr int GATE_SPIN_MAX = 1 << 13;
__global__ __launch_bounds__(64) void k_tail_gate(const unsigned* gate, unsigned seq) {
    if (threadIdx.x != 0) return;
    for (int spin = 0; spin < GATE_SPIN_MAX; spin++) {
        if ((int)(__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0) break;
        __builtin_amdgcn_s_sleep(8);
    }
}
constexpr int TAIL_SPIN_MAX = 1 << 14;   // x one L2 round trip (>= 0.3 us) >= 5 ms
struct TailArgs {
    GlweRef src, buf[2];             // step i writes buf[i & 1] (buf[0] must not be the source)
    const double* key[CHAIN_MAX];
    int ginv[CHAIN_MAX];
    const double* tw;
    double* big;                     // 2*SK*SX partial polynomials per ciphertext (BIG_STRIDE * SX doubles apart)
    unsigned* sync;
    unsigned seq;                    // generation of this launch (never 0)
    int n, n_ct, gx;
    int xoff;                        // group g sits on the blocks with (b + 8 - xoff) % 8 == g: contexts that share a GPU start on different XCDs
    int give_up_at;                  // test hook: member 5 of group 0 gives up at this step (-1: never)
    // round 6: the external products of coordinate 1 (ram.rs:454 / 525-527) in front of the trace chain, in the same launch: steps
    // 0 .. n_ep-1 are products (fine split: member (co, j, r) of the first 24 transforms digit r of both columns, multiplies with its two
    // operand polynomials and transforms back once; the normalisation phase is the products' own), steps n_ep .. n_ep+n-1 the trace
    int n_ep = 0;
    const double* ggsw[4];           // prepared digits of the coordinate
    GlweRef ep_out;                  // where the last product's result goes = the trace chain's source (read_prepare_write: tree[0], ram.rs:526); not src, not buf[]
};
constexpr int TAIL_EP_MAX = 4;
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ int ld_l2(const int32_t* p) { return __hip_atomic_load(const_cast<int32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_l2(const double* p) { return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// All G workgroups of a group meet; false = the group gave up (this workgroup must leave).  `flag` is an LDS word.
__device__ __forceinline__ bool tail_barrier(unsigned* ctr, unsigned* abortp, unsigned seq, unsigned want, int* flag, bool check_xcc, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the L2
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        for (int spin = 0; spin < TAIL_SPIN_MAX; spin++) {
            if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) { ok = 1; break; }
            if ((spin & 15) == 15 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) break;
        }
        if (ok && check_xcc) ok = (__builtin_popcount(__hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1);
        if (ok && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) ok = 0;
        if (!ok) __hip_atomic_store(abortp, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}
// The same in two halves (see k_chain_mid): announce the arrival, go on with work nobody waits for, wait later.
__device__ __forceinline__ void tail_arrive(unsigned* ctr, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool tail_wait(unsigned* ctr, unsigned* abortp, unsigned seq, unsigned want, int* flag, int tid) {
    if (tid == 0) {
        int ok = 0;
        for (int spin = 0; spin < TAIL_SPIN_MAX; spin++) {
            if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) { ok = 1; break; }
            if ((spin & 15) == 15 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) break;
        }
        if (ok && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) ok = 0;
        if (!ok) __hip_atomic_store(abortp, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}
// Two int32 per 8-byte L1-bypassing load (coefficients 2q, 2q+1 of one limb polynomial).
__device__ __forceinline__ void ld_l2_pair(const int32_t* p, int& a, int& b) {
    const long long v = __hip_atomic_load(reinterpret_cast<long long*>(const_cast<int32_t*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = (int)(v & 0xffffffffll);
    b = (int)(v >> 32);
}
// Every consumer of an intermediate ciphertext of the chain wants x = rsh1(a) (the pre-step of the NEXT trace step),
// never a itself: the normalisation phase of step s therefore writes rsh1 of its result (it holds all limbs of the
// coefficient anyway) for s < n-1, and a workgroup of step s+1 loads ONE limb polynomial (16 KB) instead of three.
template <int SX, int SK, int SO>
__global__ __launch_bounds__(T, T / 256) void k_trace_tail(TailArgs ta) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int G = 2 * SK * SX;
    constexpr int SG = 4, GE = 2 * SG * SX;        // a product: 4 output limbs per column, 24 active members
    static_assert(GE <= G && SX == 3 && SO == 3, "the products' members are a subset of the trace steps'");
    const int g = ((int)blockIdx.x + TAIL_GROUPS - ta.xoff) % TAIL_GROUPS, m = (int)blockIdx.x / TAIL_GROUPS;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)   // the last block: every block of the launch has been placed (k_tail_gate)
        __hip_atomic_store(ta.sync + TAIL_GROUPS * 32 + 2, ta.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g >= ta.n_ct) return;
    RoMonitor ro_mon(lds, ta.tw);
    const int tid = vt((int)threadIdx.x);
    double* tw = lds;
    double* data = lds + LDS_TW;
    int* mstage = reinterpret_cast<int*>(data);
    int* flag = reinterpret_cast<int*>(data + 2 * LDS_DATA);   // the third exchange buffer is not used here
    unsigned* ctr = ta.sync + g * 32;
    unsigned* abortp = ta.sync + TAIL_GROUPS * 32;
    const int r = m % SX, zz = m / SX;
    const int j = SK - 1 - zz % SK, co = zz / SK;                 // this member in a trace step
    const int je = SG - 1 - zz % SG, coe = zz / SG;               // ... and in a product (m < GE)
    const bool ep_member = m < GE;
    const long ct = (long)(g / ta.gx);
    const long cx = (long)(g % ta.gx);
    double* const big0 = ta.big + (long)g * BIG_STRIDE * SX;   // + step parity * TAIL_GROUPS * BIG_STRIDE * SX (see the normalisation phase)
    const int n_ep = ta.n_ep, n_all = ta.n_ep + ta.n;
    OpRegs kop, kop1;                 // the operand(s) of the coming step: a trace key polynomial, or the two GGSW polynomials of a product
    if (n_ep > 0) {
        if (ep_member) {
            load_ops(kop, ta.ggsw[0] + (long)(((2 * r) * SG + je) * 2 + coe) * N, tid);
            load_ops(kop1, ta.ggsw[0] + (long)(((2 * r + 1) * SG + je) * 2 + coe) * N, tid);
        }
    } else {
        load_ops(kop, ta.key[0] + (long)((r * SK + j) * 2 + co) * N, tid);
    }
    if (tid == 0) {
        __hip_atomic_fetch_or(ctr + 2, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // performed before this workgroup's first arrival is counted
    }
    load_twiddles(tw, ta.tw, tid);
    unsigned epoch = 0;
    // step s_ writes: a product the ping-pong buffers (the last one ep_out), trace step t = s_ - n_ep buf[t & 1]
    auto out_of = [&](int s_) -> GlweRef { return s_ < n_ep ? (s_ == n_ep - 1 ? ta.ep_out : ta.buf[s_ & 1]) : ta.buf[(s_ - n_ep) & 1]; };
#pragma unroll 1
    for (int s = 0; s < n_all; s++) {
        const GlweRef rin = (s == 0) ? ta.src : out_of(s - 1);
        const GlweRef rout = out_of(s);
        const int32_t* ap = rin.p + ct * rin.sy + cx * rin.sx;
        int32_t* op = rout.p + ct * rout.sy + cx * rout.sx;
        const bool is_ep = s < n_ep;
        const int t = s - n_ep;                // trace step index (is_ep: negative)
        const int ginv = is_ep ? 1 : ta.ginv[t];
        const bool stepped = (t > 0);          // the input already is rsh1(a): written by the previous TRACE step of this launch
        const bool fresh = (s == 0);           // the input was written by an earlier launch: ordinary loads
        const bool last = (s + 1 == n_all);
        double* const bigg = big0 + (long)(s & 1) * TAIL_GROUPS * BIG_STRIDE * SX;
        TSTAMP(0);
        if (is_ep) {
            // ---- fine phase of a product: partial[co][j][r] = INTT(NTT(a.col0 limb r) . G[2r][j][co] + NTT(a.col1 limb r) . G[2r+1][j][co])
            if (ep_member) {
                double x[2][E];
                if (fresh) {
#pragma unroll
                    for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                        for (int k = 0; k < E; k++) x[c2][k] = (double)gload_i32(ap + glwe_off(r, c2), (unsigned)(tid + T * k) * 4u);
                } else {
#pragma unroll
                    for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                        for (int k = 0; k < E; k++) x[c2][k] = (double)ld_l2(ap + glwe_off(r, c2) + tid + T * k);
                }
                ntt_fwd<2>(x, tw, data, tid);
                double acc[1][E];
#pragma unroll
                for (int k = 0; k < E; k++) acc[0][k] = 0.0;
                mac_regs(acc[0], x[0], kop);
                mac_regs(acc[0], x[1], kop1);
                if (s + 1 < n_ep) {
                    load_ops(kop, ta.ggsw[s + 1] + (long)(((2 * r) * SG + je) * 2 + coe) * N, tid);
                    load_ops(kop1, ta.ggsw[s + 1] + (long)(((2 * r + 1) * SG + je) * 2 + coe) * N, tid);
                }
                ntt_inv<1, false>(acc, tw, data, tid);
                double* bgp = bigg + (long)((coe * SG + je) * SX + r) * N;
#pragma unroll
                for (int k = 0; k < E; k++) bgp[tid + T * k] = acc[0][k];
            }
            if (s + 1 == n_ep && !last) load_ops(kop, ta.key[0] + (long)((r * SK + j) * 2 + co) * N, tid);   // the first trace step's operand (every member)
        } else {
        // ---- fine phase: x = rsh1(a); partial[co][j][r] = INTT(NTT(phi_g(x.mask limb r)) . K[r][j][co]) (+ phi_g(x.body limb j))
        // staging: thread t brings coefficients 8t .. 8t+7 (natural order) of the limb polynomials it needs
        if (stepped) {
            int v[E];
#pragma unroll
            for (int q = 0; q < E / 2; q++) ld_l2_pair(ap + glwe_off(r, 1) + E * tid + 2 * q, v[2 * q], v[2 * q + 1]);
#pragma unroll
            for (int k = 0; k < E; k++) mstage[E * tid + k] = v[k];
        } else {
            // the source: written by an earlier launch (ordinary 16-byte loads) or by the last product of this one (past the L1)
            int rm[SX][E];
            if (fresh) {
#pragma unroll
                for (int q = 0; q < SX; q++)
#pragma unroll
                    for (int h = 0; h < E / 4; h++) {
                        const int4 v4 = *reinterpret_cast<const int4*>(ap + glwe_off(q, 1) + E * tid + 4 * h);
                        rm[q][4 * h] = v4.x; rm[q][4 * h + 1] = v4.y; rm[q][4 * h + 2] = v4.z; rm[q][4 * h + 3] = v4.w;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < SX; q++)
#pragma unroll
                    for (int h = 0; h < E / 2; h++) ld_l2_pair(ap + glwe_off(q, 1) + E * tid + 2 * h, rm[q][2 * h], rm[q][2 * h + 1]);
            }
#pragma unroll
            for (int k = 0; k < E; k++) {
                int xi[SX], xm[SX];
#pragma unroll
                for (int q = 0; q < SX; q++) xi[q] = rm[q][k];
                rsh1_coeff<SX>(xi, xm);
                mstage[E * tid + k] = sel_limb(xm, r);
            }
        }
        __syncthreads();
        TSTAMP(1);
        double x[1][E];
        {
            int sidx = (tid * ginv) & (2 * N - 1);
            const int sstep = (T * ginv) & (2 * N - 1);
#pragma unroll
            for (int k = 0; k < E; k++) {
                x[0][k] = (double)cneg(mstage[sidx & (N - 1)], sidx >= N);
                sidx = (sidx + sstep) & (2 * N - 1);
            }
        }
        ntt_fwd<1>(x, tw, data, tid);        // starts with a barrier: every gather of the staged limbs is done
        TSTAMP(2);
        double acc[1][E];
#pragma unroll
        for (int k = 0; k < E; k++) acc[0][k] = 0.0;
        mac_regs(acc[0], x[0], kop);
        if (!last) load_ops(kop, ta.key[t + 1] + (long)((r * SK + j) * 2 + co) * N, tid);   // arrives during the rest of the step
        ntt_inv<1, false>(acc, tw, data, tid);
        TSTAMP(3);
        {
            double* bgp = bigg + (long)((co * SK + j) * SX + r) * N;
#pragma unroll
            for (int k = 0; k < E; k++) bgp[tid + T * k] = acc[0][k];
        }
        }
        if (ta.give_up_at == s && g == 0 && m == 5) {
            if (tid == 0) __hip_atomic_store(abortp, ta.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        TSTAMP(4);
        if (!tail_barrier(ctr, abortp, ta.seq, (++epoch) * G, flag, s == 0, tid)) break;
        TSTAMP(5);
        // ---- normalisation phase: one thread per (column, coefficient); same arithmetic as k_keyswitch_norm<KS_TRACE> / k_ext_product_fine_norm.
        // Every member takes an equal share of each column (CH consecutive coefficients: the phase is as long as its busiest
        // member's L2 reads; with T per member a third of the members had none).  The next fine phase of a TRACE step needs the mask
        // column only: the members announce themselves when their share of it is stored and do the body column under the hand-off's
        // latency (nobody reads it before the next normalisation phase; the partials are double buffered by step parity for
        // that, see k_chain_mid).  A product's next fine phase reads both columns.
        constexpr int CH = (N + G - 1) / G;
        static_assert(CH <= T, "one coefficient per thread and column");
        auto norm_share = [&](const int nco) {
            const int i = m * CH + tid;
            if (tid < CH && i < N) {
                const double* bgp = bigg + (long)nco * SK * SX * N + i;
                double v_[SK];
#pragma unroll
                for (int q = 0; q < SK; q++) {
                    v_[q] = ld_l2(bgp + (long)(q * SX) * N);
#pragma unroll
                    for (int w = 1; w < SX; w++) v_[q] += ld_l2(bgp + (long)(q * SX + w) * N);   // exact: integers below 2^47
                }
                int raw[SX], xa[SX];
#pragma unroll
                for (int q = 0; q < SX; q++) raw[q] = ld_l2(ap + glwe_off(q, nco) + i);
                // vec_znx_big_add_small_inplace of the body column seen through phi_g (column 0 only): it joins the sums here,
                // where every member has the same share of it, instead of lengthening the fine phase of the three members that owned it
                const int si = (i * ginv) & (2 * N - 1);
                int braw[SX], xb[SX];
#pragma unroll
                for (int q = 0; q < SX; q++) braw[q] = (nco == 0) ? ld_l2(ap + glwe_off(q, 0) + (si & (N - 1))) : 0;
                if (stepped) {
#pragma unroll
                    for (int q = 0; q < SX; q++) { xa[q] = raw[q]; xb[q] = braw[q]; }
                } else {
                    rsh1_coeff<SX>(raw, xa);
                    rsh1_coeff<SX>(braw, xb);
                }
                double carry = 0.0;
                int d[SO], y[SO];
#pragma unroll
                for (int q = SK - 1; q >= 0; q--) {
                    double v = v_[q];
                    if (q < SX) v += (double)xa[q < SX ? q : 0] + (double)cneg(xb[q < SX ? q : 0], si >= N);
                    v += carry;
                    const double cy = carry_of(v);
                    carry = cy;
                    if (q < SO) d[q < SO ? q : 0] = (int)digit_of(v, cy);
                }
                if (last) {
#pragma unroll
                    for (int q = 0; q < SO; q++) y[q] = d[q];
                } else {
                    rsh1_coeff<SO>(d, y);
                }
#pragma unroll
                for (int q = 0; q < SO; q++) op[glwe_off(q, nco) + i] = y[q];
            }
        };
        // a product's: the sums over the three digits' partials, the limb walk, the normalised limbs as they are (they ARE the next product's digits)
        auto norm_share_ep = [&](const int nco) {
            const int i = m * CH + tid;
            if (tid < CH && i < N) {
                const double* bgp = bigg + (long)nco * SG * SX * N + i;
                double v_[SG];
#pragma unroll
                for (int q = 0; q < SG; q++) {
                    v_[q] = ld_l2(bgp + (long)(q * SX) * N);
#pragma unroll
                    for (int w = 1; w < SX; w++) v_[q] += ld_l2(bgp + (long)(q * SX + w) * N);   // exact: integers below 2^47
                }
                double carry = 0.0;
#pragma unroll
                for (int q = SG - 1; q >= 0; q--) {
                    const double v = v_[q] + carry;
                    const double cy = carry_of(v);
                    carry = cy;
                    if (q < SO) op[glwe_off(q, nco) + i] = (int)digit_of(v, cy);
                }
            }
        };
        if (is_ep) {
            norm_share_ep(1);
            norm_share_ep(0);
            TSTAMP(6);
            if (!last) { if (!tail_barrier(ctr, abortp, ta.seq, (++epoch) * G, flag, false, tid)) break; }
        } else {
        norm_share(1);
        if (last) {
            norm_share(0);
            TSTAMP(6);
        } else {
            tail_arrive(ctr, tid);
            norm_share(0);
            TSTAMP(6);
            ++epoch;
            if (!tail_wait(ctr, abortp, ta.seq, epoch * G, flag, tid)) break;
        }
        }
        TSTAMP(7);
    }
    // the last workgroup of the group to leave (every one passes here exactly once, given up or not) rewinds the
    // group's words for the next launch: nobody can still be waiting on them
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(G - 1)) {
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#ifdef FK_STAMP
// Diagnostic: one inverse + one forward pair transform with stamps around them.
__global__ __launch_bounds__(T, T / 256) void k_ntt_probe(const double* __restrict__ tw_g, double* sink) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double x[2][E];
#pragma unroll
    for (int k = 0; k < E; k++) { x[0][k] = (double)(tid * 8 + k); x[1][k] = (double)(tid + k); }
    for (int rep = 0; rep < 2; rep++) {
        STAMP(0);
        ntt_inv<2, true>(x, tw, data, tid);
        STAMP(1);
        ntt_fwd<2>(x, tw, data, tid);
        STAMP(2);
#pragma unroll
        for (int k = 0; k < E; k++) { x[0][k] *= 0x1p-11; x[1][k] *= 0x1p-11; }
    }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < E; k++) acc += x[0][k] + x[1][k];
    sink[blockIdx.x * T + tid] = acc;
}
#endif

// ---------------------------------------------------------------------------------------
// Fine limb split, for the dependent chain at the end of every RAM op (word_size ciphertexts per round on a
// 256-CU chip): one workgroup per (ciphertext, output column, output limb, INPUT limb).  The inverse
// transform is linear, so  INTT(sum_r x^_r . K_r) = sum_r INTT(x^_r . K_r)  exactly (every partial product
// is an integer below N*2^16*2^17 < p/2 and lifts to itself): each workgroup does ONE forward and ONE
// inverse transform instead of SX + 1, and the normalisation pass adds the SX partial polynomials before it
// walks the limbs.  Per coefficient the sums and the carry chain are the ones the fused kernel computes.
//   grid (x, y, 2*SK*SX): z = ((co*SK + (SK-1-j))*SX + r);  partials at big + ct*BIG_STRIDE*SX + ((co*SK + j)*SX + r)*N
// ---------------------------------------------------------------------------------------
template <int MODE, int SX, int SK>
__global__ __launch_bounds__(T, T / 256) void k_keyswitch_fine(KsArgs ka) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, ka.tw);
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    STAMPZ(0);
    TwRegs twr;
    twiddles_issue(twr, ka.tw, tid);
    const int32_t* ap = at(ka.a);
    const int32_t* bp = (MODE == KS_PAIR) ? at(ka.b) : nullptr;
    constexpr int BODY_COL = (MODE == KS_TENSOR) ? 1 : 0;
    constexpr bool PHI = (MODE != KS_TENSOR);
    const int r = (int)blockIdx.z % SX, zz = (int)blockIdx.z / SX;
    const int j = SK - 1 - zz % SK, co = zz / SK;
    OpRegs g;
    load_ops(g, ka.key + (long)((r * SK + j) * 2 + co) * N, tid);
    const int sidx0 = (tid * ka.ginv) & (2 * N - 1);
    const int sstep = (T * ka.ginv) & (2 * N - 1);
    int* mstage = reinterpret_cast<int*>(data);
    // vec_znx_big_add_small_inplace of body limb j, seen through phi_g: ONE workgroup per (column, limb) adds it.
    // Like the mask limb it is loaded coalesced, staged in LDS and gathered with the odd stride g^-1.  ALL global
    // loads of the workgroup are issued before the first one is waited for: the inputs were written by the
    // previous kernel on other XCDs and every dependent round trip costs a fabric latency (2-3 us).
    const bool adds_body = (r == 0 && co == BODY_COL && j < SX);
    int* bstage = mstage + N;
    int bodyv[E];
    double x[1][E];
    RawX<MODE, SX> rm[E], rb[E];
#pragma unroll
    for (int k = 0; k < E; k++) load_raw<MODE, SX>(ka, ap, bp, 1, tid + T * k, rm[k]);
    if (adds_body) {
#pragma unroll
        for (int k = 0; k < E; k++) load_raw<MODE, SX>(ka, ap, bp, 0, tid + T * k, rb[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PHI) {
#pragma unroll
        for (int k = 0; k < E; k++) {
            int xm[SX];
            pre_step<MODE, SX>(rm[k], xm);
            mstage[tid + T * k] = sel_limb(xm, r);
        }
        if (adds_body) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                int xb[SX];
                pre_step<MODE, SX>(rb[k], xb);
                bstage[tid + T * k] = sel_limb(xb, j);
            }
        }
        STAMPZ(1);
        twiddles_commit(twr, tw, tid);   // its barrier also publishes the staged limbs
        STAMPZ(2);
        int sidx = sidx0;
#pragma unroll
        for (int k = 0; k < E; k++) {
            x[0][k] = (double)cneg(mstage[sidx & (N - 1)], sidx >= N);
            bodyv[k] = adds_body ? cneg(bstage[sidx & (N - 1)], sidx >= N) : 0;
            sidx = (sidx + sstep) & (2 * N - 1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) {
            int xm[SX];
            pre_step<MODE, SX>(rm[k], xm);
            x[0][k] = (double)sel_limb(xm, r);
            bodyv[k] = 0;
            if (adds_body) {
                int xb[SX];
                pre_step<MODE, SX>(rb[k], xb);
                bodyv[k] = sel_limb(xb, j);
            }
        }
        twiddles_commit(twr, tw, tid);
    }
    STAMPZ(3);
    ntt_fwd<1>(x, tw, data, tid);        // starts with a barrier: every gather of the staged limb is done
    STAMPZ(4);
    double acc[1][E];
#pragma unroll
    for (int k = 0; k < E; k++) acc[0][k] = 0.0;
    mac_regs(acc[0], x[0], g);
    STAMPZ(5);
    ntt_inv<1, false>(acc, tw, data, tid);
    STAMPZ(6);
#pragma unroll
    for (int k = 0; k < E; k++) acc[0][k] += (double)bodyv[k];
    double* bgp = ka.big + big_ct() * SX + (long)((co * SK + j) * SX + r) * N;
#pragma unroll
    for (int k = 0; k < E; k++) bgp[tid + T * k] = acc[0][k];
    STAMPZ(7);
}

// grid (x, y, 2*SG*2*SA): z = ((co*SG + (SG-1-j))*2 + cin)*SA + r;  partials at big + ct*BIG_STRIDE*2*SA + ((co*SG + j)*2*SA + cin*SA + r)*N
template <int SA, int SG>
__global__ __launch_bounds__(T, T / 256) void k_ext_product_fine(GlweRef a, const double* __restrict__ ggsw,
                                                                 const double* __restrict__ tw_g, double* __restrict__ big) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, tw_g);
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    TwRegs twr;
    twiddles_issue(twr, tw_g, tid);
    const int32_t* ap = at(a);
    const int r = (int)blockIdx.z % SA, z1 = (int)blockIdx.z / SA;
    const int cin = z1 % 2, z2 = z1 / 2;
    const int j = SG - 1 - z2 % SG, co = z2 / SG;
    OpRegs g;
    load_ops(g, ggsw + (long)(((2 * r + cin) * SG + j) * 2 + co) * N, tid);
    double x[1][E];
#pragma unroll
    for (int k = 0; k < E; k++) x[0][k] = (double)ap[glwe_off(r, cin) + tid + T * k];
    twiddles_commit(twr, tw, tid);
    ntt_fwd<1>(x, tw, data, tid);
    double acc[1][E];
#pragma unroll
    for (int k = 0; k < E; k++) acc[0][k] = 0.0;
    mac_regs(acc[0], x[0], g);
    ntt_inv<1, false>(acc, tw, data, tid);
    double* bp = big + big_ct() * (2 * SA) + (long)((co * SG + j) * 2 * SA + cin * SA + r) * N;
#pragma unroll
    for (int k = 0; k < E; k++) bp[tid + T * k] = acc[0][k];
}
// normalisation pass of k_ext_product_fine: one thread per coefficient, grid (x, y, 2 columns * N/256 slices)
template <int SA, int SG>
__global__ __launch_bounds__(256) void k_ext_product_fine_norm(GlweRef res, const double* __restrict__ big) {
    constexpr int SLICES = N / 256, NP = 2 * SA;
    const int co = (int)blockIdx.z / SLICES;
    const int i = ((int)blockIdx.z % SLICES) * 256 + (int)threadIdx.x;
    int32_t* rp = at(res);
    const double* bp = big + big_ct() * NP + (long)co * SG * NP * N + i;
    double carry = 0.0;
#pragma unroll
    for (int j = SG - 1; j >= 0; j--) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NP; q++) v += bp[(long)(j * NP + q) * N];   // exact: integers below 2^47
        v += carry;
        const double cy = carry_of(v);
        carry = cy;
        if (j < SA) rp[glwe_off(j, co) + i] = (int)digit_of(v, cy);
    }
}

// Normalisation / post-step pass of the limb-parallel path (STAGE 2 of k_keyswitch): one thread per
// coefficient, grid (x, y, 2 columns * N/256 slices).  Same per-coefficient arithmetic as the
// `emit` step of the fused kernel.
// NP > 1: the limbs arrive as NP partial polynomials each (k_keyswitch_fine) and are added first.
template <int MODE, int SX, int SK, int SO, int NP = 1>
__global__ __launch_bounds__(256) void k_keyswitch_norm(KsArgs ka) {
    constexpr int SLICES = N / 256;
    const int co = (int)blockIdx.z / SLICES;
    const int i = ((int)blockIdx.z % SLICES) * 256 + (int)threadIdx.x;
    GlweRef ra = ka.a, rb = ka.b, ro = ka.out;
    const int32_t* ap = ra.p + (long)blockIdx.y * ra.sy + (long)blockIdx.x * ra.sx;
    const int32_t* bp = (MODE == KS_PAIR) ? rb.p + (long)blockIdx.y * rb.sy + (long)blockIdx.x * rb.sx : nullptr;
    int32_t* op = ro.p + (long)blockIdx.y * ro.sy + (long)blockIdx.x * ro.sx;
    const double* bgp = ka.big + big_ct() * NP + (long)co * SK * NP * N + i;
    double v_[SK];
#pragma unroll
    for (int j = 0; j < SK; j++) {
        v_[j] = bgp[(long)(j * NP) * N];
#pragma unroll
        for (int q = 1; q < NP; q++) v_[j] += bgp[(long)(j * NP + q) * N];   // exact: integers below 2^47
    }
    int xa[SX];
    if constexpr (MODE == KS_PAIR) load_pair_sum<SX>(ka, ap, bp, co, i, xa);
    else if constexpr (MODE == KS_TRACE || MODE == KS_ADD || MODE == KS_SUBNEG) load_x<MODE, SX>(ka, ap, bp, co, i, xa);
    double carry = 0.0, carry2 = 0.0;
#pragma unroll
    for (int j = SK - 1; j >= 0; j--) {
        double v = v_[j];
        if constexpr (MODE == KS_TRACE || MODE == KS_ADD) { if (j < SX) v += (double)xa[j < SX ? j : 0]; }
        if constexpr (MODE == KS_SUBNEG) v = (j < SX ? (double)xa[j < SX ? j : 0] : 0.0) - v;
        v += carry;
        const double cy = carry_of(v);
        carry = cy;
        if (j < SO) {
            const double d = digit_of(v, cy);
            if constexpr (MODE == KS_PAIR) {
                const double v2 = (double)xa[j < SX ? j : 0] - d + carry2;
                const double cy2 = carry_of(v2);
                carry2 = cy2;
                int dst = i + ka.t;
                const bool ng = dst >= N;
                if (ng) dst -= N;
                op[glwe_off(j, co) + dst] = cneg((int)digit_of(v2, cy2), ng);
            } else {
                op[glwe_off(j, co) + i] = (int)d;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Write-path elementwise steps.
// ---------------------------------------------------------------------------------------
// out = normalize(a - b + c)      (ram.rs:574-576 with b = trace(a), c = w;  ram.rs:617,625-626)
template <int S>
__global__ __launch_bounds__(256) void k_sub_add_norm(GlweRef a, GlweRef b, GlweRef c, GlweRef out) {
    const int32_t* ap = at(a);
    const int32_t* bp = at(b);
    const int32_t* cp = at(c);
    int32_t* op = at(out);
    for (int idx = blockIdx.z * blockDim.x + threadIdx.x; idx < 2 * N; idx += blockDim.x * gridDim.z) {
        const int col = idx >> LOGN, i = idx & (N - 1);
        double in_l[S], out_l[S];
#pragma unroll
        for (int j = 0; j < S; j++) {
            const long o = glwe_off(j, col) + i;
            in_l[j] = (double)(ap[o] - bp[o] + cp[o]);
        }
        normalize_coeff<S, S>(in_l, out_l);
#pragma unroll
        for (int j = 0; j < S; j++) op[glwe_off(j, col) + i] = (int)out_l[j];
    }
}
// ---------------------------------------------------------------------------------------
// k_chain_mid (round 3): a dependent chain of n steps on 9..64 ciphertexts — too many for k_trace_tail's one group per XCD,
// too few for one workgroup per ciphertext (MAX_ADDR = 2^14, the source default, has 16 ciphertexts per round, 2^15 32, 2^16
// 64: the alone packer levels and the products of coordinate 0, each a pair of launches per step before) — as ONE launch
// with in-kernel hand-offs, by the mechanism of k_trace_tail: the MEMBERS workgroups of a ciphertext sit on one XCD (block b
// runs on XCD b % 8; checked, not assumed), 32 / MEMBERS ciphertexts per XCD, meet at a counter in that XCD's L2 and read
// each other's data past their L1.  Three splits, by how many ciphertexts must share the chip:
//   <RS = 3, LPM = 2>  <= 16 ciphertexts: member (r, h): digit r of the input (TRACE: of the mask column, seen through phi_g;
//     EP: of both columns) -> one / two forward transforms; output limb polynomials 2h, 2h+1 of the 2*SK: MAC with the
//     operands of digit r, one inverse transform each -> a PARTIAL (the sum over r is taken by the normalisation phase);
//   <RS = 1, LPM = 1>  <= 32 (TRACE): member h: all three digits (three forward transforms), output limb polynomial h;
//   <RS = 1, LPM = 2>  <= 64 (TRACE): the same with two output limb polynomials per member;                  hand-off A
//   normalisation phase: one thread per (column, coefficient), MEMBERS * T threads per ciphertext — the body column of a
//     trace step (vec_znx_big_add_small) and its `+ x` join the sums here;                                    hand-off B
// Intermediate ciphertexts live in a scratch of their own, in the one-double form of ks_trace_y: TRACE: Y = ceil(A/2) (the
// consumer wants rsh1 of the previous output: its digits ARE the digits of Y); EP: A itself.  Same sums, same carry chain per
// coefficient as the fused kernels: bit-identical results.  The destination is written by the LAST step only, after the
// group's last hand-off — so the chain may run in place (read_prepare_write's products on the stored rows).
// Giving up.  Every wait is bounded.  Each ciphertext's group stands alone: a member that cannot meet the others (not
// co-resident, wrong placement) POISONS the group's counter (a compare-and-swap from a value below the awaited one: once a
// barrier has been reached by all, nobody can poison it any more, so the last hand-off is all-or-nothing and the destination
// is written by every member or by none) and leaves; the last member to leave records whether the group completed, and the
// fused chain launch enqueued behind redoes exactly the ciphertexts that did not, from the untouched source.
//   grid: 8 * (32 / MEMBERS) * MEMBERS blocks; XCD x = b % 8, i = b / 8; slot = i / MEMBERS, member = i % MEMBERS; ciphertext = slot * 8 + x
//   sync: [ciphertext][32] words: arrivals (| poison), leavers, XCC mask, completed generation;  sync[64 * 32 + 1] = ciphertexts redone
// ---------------------------------------------------------------------------------------
constexpr unsigned MID_POISON = 0x80000000u;
struct MidArgs {
    GlweRef src, dst;                // dst may be src
    const double* opnd[CHAIN_MAX];   // TRACE: prepared trace key of step i;  EP: prepared GGSW digit i
    int ginv[CHAIN_MAX];             // TRACE: g_i^-1 mod 2N
    const double* tw;
    double* big;                     // [ciphertext] x (RS * BIG_STRIDE) doubles: partials [col][limb][r][N]
    double* y;                       // [2][MID_GROUPS_MAX][2][N] doubles: the intermediates, ping-pong by step parity
    unsigned* sync;
    unsigned seq;                    // generation of this launch (never 0)
    int n, n_ct, gx;                 // steps, ciphertexts, ciphertexts per row of the (x, y) grid the GlweRefs are indexed by
    int rot_mul, rot_base;           // TRACE: the first step reads its input rotated by X^-(rot_base + x * rot_mul)  (write path)
    int give_up_at;                  // test hook: member 1 of ciphertext 0 gives up at this step (-1: never)
};
// All members of a group meet at `want` arrivals; false = the group has given up (this workgroup must leave).
__device__ __forceinline__ bool mid_barrier(unsigned* ctr, unsigned want, int* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the L2
    __syncthreads();
    if (tid == 0) {
        unsigned v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        int ok = -1;
        for (int spin = 0; spin < TAIL_SPIN_MAX && ok < 0; spin++) {
            if (v & MID_POISON) ok = 0;
            else if (v >= want) ok = 1;
            else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (ok < 0) {   // waited long enough: give up — unless everybody has arrived meanwhile (then nobody may)
            if (v & MID_POISON) ok = 0;
            else if (v >= want) ok = 1;
            else if (__hip_atomic_compare_exchange_strong(ctr, &v, v | MID_POISON, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}
// The same in two halves: a member announces its arrival (its stores so far drained) and goes on with work nobody waits for;
// the wait comes when it needs what the others announced.
__device__ __forceinline__ void mid_arrive(unsigned* ctr, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool mid_wait(unsigned* ctr, unsigned want, int* flag, int tid) {
    if (tid == 0) {
        unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = -1;
        for (int spin = 0; spin < TAIL_SPIN_MAX && ok < 0; spin++) {
            if (v & MID_POISON) ok = 0;
            else if (v >= want) ok = 1;
            else v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (ok < 0) {
            if (v & MID_POISON) ok = 0;
            else if (v >= want) ok = 1;
            else if (__hip_atomic_compare_exchange_strong(ctr, &v, v | MID_POISON, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}
template <bool EP, int SK, int RS, int LPM>
__global__ __launch_bounds__(T, T / 256) void k_chain_mid(MidArgs ma) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int SX = 3, SO = 3, NP = 2 * SK;
    static_assert((RS == 1 || RS == SX) && NP % LPM == 0 && (!EP || RS == SX), "member split");
    constexpr int NX = EP ? 2 : (RS == 1 ? SX : 1);   // forward transforms per member
    constexpr int MEMBERS = RS * NP / LPM;     // <3,2>: 12 / 15 (4- / 5-limb operands);  <1,1>: 8 / 10;  <1,2>: 4 / 5
    constexpr int GPX = 32 / MEMBERS;          // ciphertexts per XCD: 2, 4 / 3, 8 / 6
    static_assert(GPX * MEMBERS <= 32 && 8 * GPX <= MID_GROUPS_MAX, "members of an XCD's ciphertexts must be co-resident on its 32 CUs");
    const int xcd = (int)blockIdx.x % 8, bi = (int)blockIdx.x / 8;
    const int slot = bi / MEMBERS, m = bi % MEMBERS;
    const int ctg = slot * 8 + xcd;            // ciphertext (group) of this workgroup
    if (slot >= GPX || ctg >= ma.n_ct) return;
    RoMonitor ro_mon(lds, ma.tw);
    const int tid0 = vt((int)threadIdx.x);
    int tid = tid0;
    const int r = m % RS, h = m / RS;          // input digit (RS == 1: all three); output limb polynomials h * LPM + l
    double* tw = lds;
    double* data = lds + LDS_TW;
    double* stage0 = data;                     // exchange buffer 0, before the forward transform: the mask column's digit r (TRACE)
    int* flag = reinterpret_cast<int*>(lds + LDS_TW + BMAX * LDS_DATA - 2);   // the padding at the very end of the LDS allocation
    unsigned* ctr = ma.sync + ctg * 32;
    const long cty = (long)(ctg / ma.gx), ctx_ = (long)(ctg % ma.gx);
    double* const big0 = ma.big + (long)ctg * BIG_STRIDE * RS;   // + step parity * n_ct * BIG_STRIDE * RS (see the normalisation phase)
    if (tid == 0) {
        __hip_atomic_fetch_or(ctr + 2, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // performed before this workgroup's first arrival is counted
    }
    load_twiddles(tw, ma.tw, tid);
    const int32_t* ap = ma.src.p + cty * ma.src.sy + ctx_ * ma.src.sx;
    int32_t* op = ma.dst.p + cty * ma.dst.sy + ctx_ * ma.dst.sx;
    const int rho = -(ma.rot_base + (int)ctx_ * ma.rot_mul);         // first TRACE step only
    unsigned epoch = 0;
    OpRegs g[NX];
#pragma unroll 1
    for (int s = 0; s < ma.n; s++) {
        const double* yin = ma.y + ((long)((s + 1) & 1) * MID_GROUPS_MAX + ctg) * (2 * N);   // [col][N], written by step s - 1
        double* yout = ma.y + ((long)(s & 1) * MID_GROUPS_MAX + ctg) * (2 * N);
        const bool first = (s == 0), last = (s + 1 == ma.n);
        double* const bigg = big0 + (long)(s & 1) * ma.n_ct * BIG_STRIDE * RS;
        tid = tid0;
        asm volatile("" : "+v"(tid));   // per-step copy the optimiser cannot see through (see k_ext_product_chain)
        __builtin_assume(tid >= 0 && tid < T);
        const int ginv = EP ? 1 : ma.ginv[s];
        const int sidx0 = (tid * ginv) & (2 * N - 1), sstep = (T * ginv) & (2 * N - 1);
        // the one-double form of column `col` at natural coefficient i: A (EP) or Y = ceil(A/2) (TRACE).  Inner steps read it
        // from the scratch the other members of the group wrote (past the L1); the first step makes it from the int32 source.
        // The two cases are kept apart at every use (`if (first)` around whole load loops, never inside one): a branch per
        // load put every L2 round trip behind the previous one (8 in a row: 3.2 us of a 15 us step, tools/stamp_mid.py)
        auto raw_double = [&](int col, int i) -> double {
            int src = i; bool neg = false;
            if constexpr (!EP) rot_src(i, rho, src, neg);
            double a = __builtin_fma(__builtin_fma((double)ap[glwe_off(0, col) + src], TWO_B, (double)ap[glwe_off(1, col) + src]), TWO_B,
                                     (double)ap[glwe_off(2, col) + src]);
            if constexpr (EP) return a;
            a = neg ? -a : a;
            return __builtin_floor(__builtin_fma(a, 0.5, 0.5));
        };
        // digit r of c (c = the one-double form): digits leave from the least significant end
        auto digit_r = [&](double c) -> double {
            double d = take_digit(c);
            if (r < 2) d = take_digit(c);
            if (r < 1) d = c;
            return d;
        };
        auto fetch = [&](int step, int l) {
            const int pi = h * LPM + l, co = pi / SK, j = SK - 1 - pi % SK;
#pragma unroll
            for (int cin = 0; cin < NX; cin++)   // operand row: EP: digit r of column cin;  TRACE: digit r (RS == 1: digit cin)
                load_ops(g[cin], ma.opnd[step] + (long)(((EP ? (2 * r + cin) : (RS == 1 ? cin : r)) * SK + j) * 2 + co) * N, tid);
        };
        MSTAMP(0);
        // (the operands of this step's first polynomial were requested during the previous step — or, for the first step, are
        // requested below, BEHIND the input loads: loads return in order, and the inputs are what phase 1 waits for)
        // ---- phase 1: digit r of the input (TRACE: of the mask column through phi_g), forward transform(s)
        double x[NX][E];
        if constexpr (EP) {
            if (first) {                       // the limbs as they are stored (they need not be normalised: a rotation's negation leaves +2^16)
                int xi[2][E];
#pragma unroll
                for (int col = 0; col < 2; col++)
#pragma unroll
                    for (int k = 0; k < E; k++) xi[col][k] = ap[glwe_off(r, col) + tid + T * k];
                __builtin_amdgcn_sched_barrier(0);
                fetch(0, 0);
#pragma unroll
                for (int col = 0; col < 2; col++)
#pragma unroll
                    for (int k = 0; k < E; k++) x[col][k] = (double)xi[col][k];
            } else {
                double yv[2][E];
#pragma unroll
                for (int col = 0; col < 2; col++)
#pragma unroll
                    for (int k = 0; k < E; k++) yv[col][k] = ld_l2(yin + (long)col * N + tid + T * k);
#pragma unroll
                for (int col = 0; col < 2; col++)
#pragma unroll
                    for (int k = 0; k < E; k++) x[col][k] = digit_r(yv[col][k]);
            }
        } else {
            double yv[E];
            if (first) {
#pragma unroll
                for (int k = 0; k < E; k++) yv[k] = raw_double(1, tid + T * k);
                __builtin_amdgcn_sched_barrier(0);
                fetch(0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < E; k++) yv[k] = ld_l2(yin + (long)N + tid + T * k);
            }
#pragma unroll
            for (int k = 0; k < E; k++) stage0[tid + T * k] = (RS == 1) ? yv[k] : digit_r(yv[k]);
            __syncthreads();
            int sidx = sidx0;
#pragma unroll
            for (int k = 0; k < E; k++) {
                double d = stage0[sidx & (N - 1)];
                const bool ng = sidx >= N;
                if constexpr (RS == 1) {       // all three digits of the mask column
                    const double d2 = take_digit(d), d1 = take_digit(d);
                    x[NX - 1][k] = ng ? -d2 : d2; x[NX > 1 ? 1 : 0][k] = ng ? -d1 : d1;
                }
                x[0][k] = ng ? -d : d;
                sidx = (sidx + sstep) & (2 * N - 1);
            }
        }
        MSTAMP(1);
        fwd_all<NX>(x, tw, data, tid);         // starts with a barrier: every gather of the staged digits is done
        MSTAMP(2);
        // ---- phase 2: this member's partials of its two output limb polynomials
#pragma unroll
        for (int l = 0; l < LPM; l++) {
            const int pi = h * LPM + l, co = pi / SK, j = SK - 1 - pi % SK;
            double acc[1][E];
#pragma unroll
            for (int k = 0; k < E; k++) acc[0][k] = 0.0;
#pragma unroll
            for (int cin = 0; cin < NX; cin++) mac_regs(acc[0], x[cin], g[cin]);
            __builtin_amdgcn_sched_barrier(0);
            if (l + 1 < LPM) fetch(s, l + 1);  // arrives during the inverse transform
            else if (!last) fetch(s + 1, 0);   // the next step's first operands: they arrive during the hand-offs
            // buffer l % 2: alternating, and the previous transform in buffers 0 and 1 was a forward one (fenced inside) or none
            ntt_inv<1, false>(acc, tw, data + (l & 1) * LDS_DATA, tid);
            double* bgp = bigg + (long)((co * SK + j) * RS + r) * N;
#pragma unroll
            for (int k = 0; k < E; k++) bgp[tid + T * k] = acc[0][k];
        }
        MSTAMP(3);
        if (ma.give_up_at == s && ctg == 0 && m == 1) {
            if (tid == 0) __hip_atomic_fetch_or(ctr, MID_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (!mid_barrier(ctr, (++epoch) * MEMBERS, flag, tid)) break;
        MSTAMP(4);
        if (first) {                           // placement: all members of the group on one XCD (they all read the same mask: all stay or all leave)
            if (__builtin_popcount(__hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 1) {
                if (tid == 0) __hip_atomic_fetch_or(ctr, MID_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        // ---- normalisation phase: one thread per (column, coefficient): same arithmetic as the emit step of the fused kernels.
        // Member m owns CH consecutive coefficients of each column (an equal share for every member: the phase is as long as its
        // busiest member's L2 reads; consecutive coefficients on consecutive lanes); all loads of a share are issued before any
        // is used (one L2 round trip).  A trace step's NEXT fine phase needs the mask column only: the members announce
        // themselves when their share of it is stored and do the body column — which nobody reads before the next step's
        // normalisation phase, i.e. behind the next hand-off A — under the hand-off's latency.  (The partials are double
        // buffered by step parity for that: a fast member's next fine phase writes partials while a slow one still reads the
        // body column's.)
        constexpr int CH = (N + MEMBERS - 1) / MEMBERS;
        constexpr int NIT = (CH + T - 1) / T;
        static_assert(NIT <= 2, "one or two coefficients per thread and column");
        const int ibase = m * CH, icount = (N - ibase) < CH ? (N - ibase > 0 ? N - ibase : 0) : CH;
        auto norm_share = [&](const int nco) {
            double v_[NIT][SK], cqv[NIT], cbv[NIT];
            // the step's input at these coefficients first (loads only, nothing used yet: the sums below wait for their own
            // loads, and a load issued behind that wait would be a second round trip)
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                cqv[u] = 0.0;                  // TRACE: Y of the step's input at this coefficient (the `+ x` of the trace step)
                cbv[u] = 0.0;                  // TRACE, body column: Y of the input's body where phi_g takes this coefficient from
            }
            if constexpr (!EP) {
                if (first) {
#pragma unroll
                    for (int u = 0; u < NIT; u++) {
                        const int i = (tid + u * T < icount) ? ibase + tid + u * T : 0;
                        cqv[u] = raw_double(nco, i);
                        if (nco == 0) cbv[u] = raw_double(0, (i * ginv) & (N - 1));
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < NIT; u++) {
                        const int i = (tid + u * T < icount) ? ibase + tid + u * T : 0;
                        cqv[u] = ld_l2(yin + (long)nco * N + i);
                        if (nco == 0) cbv[u] = ld_l2(yin + ((i * ginv) & (N - 1)));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                const int i = (tid + u * T < icount) ? ibase + tid + u * T : 0;
                const double* bgp = bigg + (long)nco * SK * RS * N + i;
#pragma unroll
                for (int q = 0; q < SK; q++) {
                    v_[u][q] = ld_l2(bgp + (long)(q * RS) * N);
#pragma unroll
                    for (int w = 1; w < RS; w++) v_[u][q] += ld_l2(bgp + (long)(q * RS + w) * N);   // exact: integers below 2^50
                }
            }
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                if (tid + u * T >= icount) break;
                const int i = ibase + tid + u * T;
                double cq = cqv[u], cb = cbv[u];
                const bool ngb = ((i * ginv) & (2 * N - 1)) >= N;   // phi_g: X^N = -1
                double carry = 0.0, ad = 0.0;
                int dig[SO];
#pragma unroll
                for (int q = SK - 1; q >= 0; q--) {
                    double v = v_[u][q];
                    if constexpr (!EP) {
                        if (q < SX) {
                            v += (q > 0) ? take_digit(cq) : cq;
                            // vec_znx_big_add_small_inplace of body limb q, seen through phi_g (column 0 only: cb = 0 otherwise)
                            const double db = (q > 0) ? take_digit(cb) : cb;
                            v += ngb ? -db : db;
                        }
                    }
                    v += carry;
                    const double cy = carry_of(v);
                    carry = cy;
                    if (q < SO) {
                        const double d = digit_of(v, cy);
                        dig[q < SO ? q : 0] = (int)d;
                        ad = __builtin_fma(d, (q == 2) ? 1.0 : ((q == 1) ? TWO_B : TWO_2B), ad);
                    }
                }
                if (last) {
#pragma unroll
                    for (int q = 0; q < SO; q++) op[glwe_off(q, nco) + i] = dig[q];
                } else {
                    yout[(long)nco * N + i] = EP ? ad : __builtin_floor(__builtin_fma(ad, 0.5, 0.5));
                }
            }
        };
        norm_share(1);
        if (last) {
            norm_share(0);
            MSTAMP(5);
        } else if constexpr (EP) {             // a product's next fine phase reads both columns
            norm_share(0);
            MSTAMP(5);
            if (!mid_barrier(ctr, (++epoch) * MEMBERS, flag, tid)) break;
        } else {
            mid_arrive(ctr, tid);
            norm_share(0);
            MSTAMP(5);
            if (!mid_wait(ctr, (++epoch) * MEMBERS, flag, tid)) break;
        }
        MSTAMP(6);
    }
    // the last workgroup of the group to leave (every one passes here exactly once, given up or not) records whether the
    // group completed and rewinds the group's words for the next launch: nobody can still be waiting on them
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(MEMBERS - 1)) {
            const unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 3, (v & MID_POISON) ? 0u : ma.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// int32 device limbs -> int64 host layout, written straight into pinned host memory (the result of a read)
// mon: the round-off monitor's maximum (fft_dev.hpp), copied behind the result: the host sees it with the result, for free
__global__ __launch_bounds__(256) void k_export_i64(const int32_t* __restrict__ src, long long* __restrict__ dst, int n4, const long long* mon) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) dst[4 * (long)n4] = __hip_atomic_load(mon, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i >= n4) return;
    const int4 v = reinterpret_cast<const int4*>(src)[i];
    longlong2 a, b;
    a.x = v.x; a.y = v.y; b.x = v.z; b.y = v.w;
    reinterpret_cast<longlong2*>(dst)[2 * i] = a;
    reinterpret_cast<longlong2*>(dst)[2 * i + 1] = b;
}
// out = a   (glwe_copy, ram.rs:526,535,537)
template <int S>
__global__ __launch_bounds__(256) void k_copy(GlweRef a, GlweRef out) {
    const int4* ap = reinterpret_cast<const int4*>(at(a));
    int4* op = reinterpret_cast<int4*>(at(out));
    for (int idx = blockIdx.z * blockDim.x + threadIdx.x; idx < S * 2 * N / 4; idx += blockDim.x * gridDim.z) op[idx] = ap[idx];
}
// out = a * X^rho   (glwe_rotate, ram.rs:629); out must not alias a
template <int S>
__global__ __launch_bounds__(256) void k_rotate(GlweRef a, GlweRef out, int rho) {
    const int32_t* ap = at(a);
    int32_t* op = at(out);
    for (int idx = blockIdx.z * blockDim.x + threadIdx.x; idx < 2 * N; idx += blockDim.x * gridDim.z) {
        const int col = idx >> LOGN, i = idx & (N - 1);
        int src; bool sgn;
        rot_src(i, rho, src, sgn);
#pragma unroll
        for (int j = 0; j < S; j++) op[glwe_off(j, col) + i] = cneg(ap[glwe_off(j, col) + src], sgn);
    }
}

// CMux steps of Address::set_from_fheuint (conversion.rs:41-65; see fheram_address_set_from_fheuint):
// out = normalize(a * X^rho - a)
template <int S>
__global__ __launch_bounds__(256) void k_cmux_pre(GlweRef a, GlweRef out, int rho) {
    const int32_t* ap = at(a);
    int32_t* op = at(out);
    for (int idx = blockIdx.z * blockDim.x + threadIdx.x; idx < 2 * N; idx += blockDim.x * gridDim.z) {
        const int col = idx >> LOGN, i = idx & (N - 1);
        int src; bool sgn;
        rot_src(i, rho, src, sgn);
        double in_l[S], out_l[S];
#pragma unroll
        for (int j = 0; j < S; j++) in_l[j] = (double)(cneg(ap[glwe_off(j, col) + src], sgn) - ap[glwe_off(j, col) + i]);
        normalize_coeff<S, S>(in_l, out_l);
#pragma unroll
        for (int j = 0; j < S; j++) op[glwe_off(j, col) + i] = (int)out_l[j];
    }
}
// out = normalize(a + b)   (out may be a)
template <int S>
__global__ __launch_bounds__(256) void k_add_norm(GlweRef a, GlweRef b, GlweRef out) {
    const int32_t* ap = at(a);
    const int32_t* bp = at(b);
    int32_t* op = at(out);
    for (int idx = blockIdx.z * blockDim.x + threadIdx.x; idx < 2 * N; idx += blockDim.x * gridDim.z) {
        const int col = idx >> LOGN, i = idx & (N - 1);
        double in_l[S], out_l[S];
#pragma unroll
        for (int j = 0; j < S; j++) in_l[j] = (double)(ap[glwe_off(j, col) + i] + bp[glwe_off(j, col) + i]);
        normalize_coeff<S, S>(in_l, out_l);
#pragma unroll
        for (int j = 0; j < S; j++) op[glwe_off(j, col) + i] = (int)out_l[j];
    }
}

// ---------------------------------------------------------------------------------------
// Setup side: GLWE::encrypt_sk / decrypt with HOST-sampled randomness.
// The host lays down a pre-ciphertext (body = plaintext limbs + the noise polynomial on its
// limb, mask = the uniform limbs it drew); the workgroup replaces the body by
//   DEC = 0:  normalise(body - mask * s)        GLWE::encrypt_sk        (ram.rs:369-376, coordinate.rs:161-168, keys.rs:158-173)
//   DEC = 1:  normalise(body + mask * s)        GLWE::decrypt (phase)   (examples/fhe-ram.rs:217-222)
// limb by limb from the least significant one (one forward + one inverse transform per limb,
// only the carry survives).  pt1 != null (GGSW rows with col_in = 1, DEC = 0): the plaintext is
// added to the MASK column after the product with s was taken, and the mask re-normalised.
// s_hat: prepared secret (k_prepare).  One workgroup per ciphertext, in place.
// ---------------------------------------------------------------------------------------
template <int S, int DEC>
__global__ __launch_bounds__(T, T / 256) void k_encrypt_sk(int32_t* __restrict__ cts, const double* __restrict__ s_hat,
                                                           const double* __restrict__ tw_g, const int32_t* __restrict__ pt1) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, tw_g);
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    int32_t* cp = cts + (long)blockIdx.x * (S * 2 * N);
    const int32_t* pp = pt1 ? pt1 + (long)blockIdx.x * (S * N) : nullptr;
    OpRegs sh;
    load_ops(sh, s_hat, tid);
    double carry[E], carry_m[E];
#pragma unroll
    for (int k = 0; k < E; k++) carry[k] = carry_m[k] = 0.0;
#pragma unroll 1
    for (int j = S - 1; j >= 0; j--) {
        int32_t* body = cp + glwe_off(j, 0);
        int32_t* mask = cp + glwe_off(j, 1);
        int mi[E], bi[E];
#pragma unroll
        for (int k = 0; k < E; k++) { mi[k] = mask[tid + T * k]; bi[k] = body[tid + T * k]; }
        double x[1][E], acc[1][E];
#pragma unroll
        for (int k = 0; k < E; k++) { x[0][k] = (double)mi[k]; acc[0][k] = 0.0; }
        ntt_fwd<1>(x, tw, data, tid);
        mac_regs(acc[0], x[0], sh);
        ntt_inv<1, false>(acc, tw, data, tid);   // follows a forward transform, whose cross-wave reads are fenced
#pragma unroll
        for (int k = 0; k < E; k++) {
            const double v = (DEC ? (double)bi[k] + acc[0][k] : (double)bi[k] - acc[0][k]) + carry[k];
            const double cy = carry_of(v);
            carry[k] = cy;
            body[tid + T * k] = (int)digit_of(v, cy);
        }
        if (pp) {
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double v = (double)(mi[k] + pp[(long)j * N + tid + T * k]) + carry_m[k];
                const double cy = carry_of(v);
                carry_m[k] = cy;
                mask[tid + T * k] = (int)digit_of(v, cy);
            }
        }
    }
}

}  // namespace fk
