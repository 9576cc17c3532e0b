// Fused GLWE-level kernels of the FHE-RAM path for gfx950.  One workgroup (512 threads) per
// ciphertext operation; 2-D grids: blockIdx.x = row inside a sub-RAM, blockIdx.y = sub-RAM.
//
//   k_prepare      GGSWPrepared::prepare / key prepare      (coordinate_prepared.rs:104-116, keys.rs:57-71)
//   k_ext_product  glwe_external_product(_inplace)           (coordinate_prepared.rs:147-177)
//   k_keyswitch<>  glwe_automorphism family, trace step, packer combine, GGSW inversion
//                  (ram.rs:435,457,540,572,616,621; coordinate_prepared.rs:138)
//   k_sub_add_norm / k_rotate   write-path elementwise steps (ram.rs:574-576,617-629)
//
// Device GLWE layout: int32 [limb][col][N] (the host's int64 layout narrowed; limbs are
// normalised to 17 bits so nothing is lost).  Prepared operands: double, transform domain,
// scaled by 1/N, stored so that thread t's elements (2kk, 2kk+1) are one 16-byte word at
// [kk*512 + t] (coalesced 16 B/lane loads).
#pragma once
#include "ntt_dev.hpp"

namespace fk {

struct GlweRef {   // p + y*sy + x*sx  (int32 elements)
    int32_t* p;
    long sy, sx;
};
__device__ __forceinline__ int32_t* at(const GlweRef& r) { return r.p + (long)blockIdx.y * r.sy + (long)blockIdx.x * r.sx; }
__device__ __forceinline__ long glwe_off(int limb, int col) { return (long)(limb * 2 + col) * N; }

// ---------------------------------------------------------------------------------------
// k_prepare: forward transform of `npoly` small polynomials into prepared form.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void k_prepare(const int32_t* __restrict__ in, double* __restrict__ out,
                                               const double* __restrict__ tw_g, double ninv) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    const int32_t* src = in + (long)blockIdx.x * N;
    double x[E];
#pragma unroll
    for (int k = 0; k < E; k++) x[k] = (double)src[tid + T * k];
    ntt_fwd(x, tw, data, tid);
    double2* o = reinterpret_cast<double2*>(out + (long)blockIdx.x * N);
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) {
        double2 v;
        v.x = reduce(mulmod(reduce(x[2 * kk]), ninv));
        v.y = reduce(mulmod(reduce(x[2 * kk + 1]), ninv));
        o[kk * T + tid] = v;
    }
}

// acc[k] += x[k] (.) g[k]   for one prepared polynomial
__device__ __forceinline__ void mac_poly(double (&acc)[E], const double (&x)[E], const double* __restrict__ g, int tid) {
    const double2* gp = reinterpret_cast<const double2*>(g);
#pragma unroll
    for (int kk = 0; kk < E / 2; kk++) {
        const double2 v = gp[kk * T + tid];
        acc[2 * kk] = macmod(acc[2 * kk], x[2 * kk], v.x);
        acc[2 * kk + 1] = macmod(acc[2 * kk + 1], x[2 * kk + 1], v.y);
    }
}

// ---------------------------------------------------------------------------------------
// k_ext_product: res = a (x) G   (SURVEY.md A.4), SA = limbs of a and res, SG = limbs of G.
// 2*SA forward transforms, 2*SA*2*SG pointwise MACs, 2*SG inverse transforms, SG->SA normalise.
// In-place (res == a) is safe: a thread reads and writes only its own coefficients.
// ---------------------------------------------------------------------------------------
template <int SA, int SG>
__global__ __launch_bounds__(T) void k_ext_product(GlweRef a, GlweRef res, const double* __restrict__ ggsw,
                                                   const double* __restrict__ tw_g) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    const int32_t* ap = at(a);
    int32_t* rp = at(res);

    double acc[2 * SG][E];   // [col_out * SG + limb]
#pragma unroll
    for (int q = 0; q < 2 * SG; q++)
#pragma unroll
        for (int k = 0; k < E; k++) acc[q][k] = 0.0;

#pragma unroll 1
    for (int in = 0; in < 2 * SA; in++) {   // in = row*2 + col_in  (GGSW row = limb of a)
        const int row = in >> 1, cin = in & 1;
        double x[E];
        const int32_t* src = ap + glwe_off(row, cin);
#pragma unroll
        for (int k = 0; k < E; k++) x[k] = (double)src[tid + T * k];
        ntt_fwd(x, tw, data, tid);
        const double* g = ggsw + (long)in * (SG * 2) * N;   // [limb][col_out] polys of this (row, cin)
#pragma unroll
        for (int j = 0; j < SG; j++)
#pragma unroll
            for (int co = 0; co < 2; co++) mac_poly(acc[co * SG + j], x, g + (long)(j * 2 + co) * N, tid);
    }
#pragma unroll
    for (int co = 0; co < 2; co++) {
#pragma unroll
        for (int j = 0; j < SG; j++) ntt_inv(acc[co * SG + j], tw, data, tid);
#pragma unroll
        for (int k = 0; k < E; k++) {
            double in_l[SG], out_l[SA];
#pragma unroll
            for (int j = 0; j < SG; j++) in_l[j] = acc[co * SG + j][k];
            normalize_coeff<SG, SA>(in_l, out_l);
#pragma unroll
            for (int j = 0; j < SA; j++) rp[glwe_off(j, co) + tid + T * k] = (int)out_l[j];
        }
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
    int rot_mul;         // KS_TRACE: rho = -(blockIdx.x * rot_mul)   (write path: ct_lo * X^-row)
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
        rot_src(i, -(int)blockIdx.x * ka.rot_mul, src, sgn);
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

template <int MODE, int SX, int SK, int SO>
__global__ __launch_bounds__(T) void k_keyswitch(KsArgs ka) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, ka.tw, tid);
    const int32_t* ap = at(ka.a);
    const int32_t* bp = (MODE == KS_PAIR) ? at(ka.b) : nullptr;
    int32_t* op = at(ka.out);
    constexpr int BODY_COL = (MODE == KS_TENSOR) ? 1 : 0;

    double acc[2 * SK][E];
#pragma unroll
    for (int q = 0; q < 2 * SK; q++)
#pragma unroll
        for (int k = 0; k < E; k++) acc[q][k] = 0.0;

    {
        // mask column of x at this thread's coefficients, all limbs (int32)
        int xm[E][SX];
#pragma unroll
        for (int k = 0; k < E; k++) load_x<MODE, SX>(ka, ap, bp, 1, tid + T * k, xm[k]);
#pragma unroll
        for (int r = 0; r < SX; r++) {
            double x[E];
#pragma unroll
            for (int k = 0; k < E; k++) x[k] = (double)xm[k][r];
            ntt_fwd(x, tw, data, tid);
            const double* g = ka.key + (long)r * (SK * 2) * N;
#pragma unroll
            for (int j = 0; j < SK; j++)
#pragma unroll
                for (int co = 0; co < 2; co++) mac_poly(acc[co * SK + j], x, g + (long)(j * 2 + co) * N, tid);
        }
    }

#pragma unroll
    for (int co = 0; co < 2; co++) {
        // inverse transforms of this output column; body add; automorphism through LDS
        int xb[E][SX];
        if (co == BODY_COL) {
#pragma unroll
            for (int k = 0; k < E; k++) load_x<MODE, SX>(ka, ap, bp, 0, tid + T * k, xb[k]);
        }
#pragma unroll
        for (int j = 0; j < SK; j++) {
            double(&v)[E] = acc[co * SK + j];
            ntt_inv(v, tw, data, tid);
            if (co == BODY_COL && j < SX) {
#pragma unroll
                for (int k = 0; k < E; k++) v[k] += (double)xb[k][j];
            }
            if constexpr (MODE != KS_TENSOR) {
                // phi_g: destination i' takes +-source i, i = i' * ginv mod 2N
                __syncthreads();
#pragma unroll
                for (int k = 0; k < E; k++) data[tid + T * k] = v[k];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < E; k++) {
                    int s = ((tid + T * k) * ka.ginv) & (2 * N - 1);
                    double sg = 1.0;
                    if (s >= N) { s -= N; sg = -1.0; }
                    v[k] = sg * data[s];
                }
            }
        }
        // post-step at destination coefficients i' = tid + 512k
#pragma unroll
        for (int k = 0; k < E; k++) {
            const int i = tid + T * k;
            double in_l[SK], out_l[SO];
#pragma unroll
            for (int j = 0; j < SK; j++) in_l[j] = acc[co * SK + j][k];
            if constexpr (MODE == KS_TRACE || MODE == KS_ADD) {
                int xa[SX];
                load_x<MODE, SX>(ka, ap, bp, co, i, xa);
#pragma unroll
                for (int j = 0; j < SX; j++) in_l[j] += (double)xa[j];
            } else if constexpr (MODE == KS_SUBNEG) {
                int xa[SX];
                load_x<MODE, SX>(ka, ap, bp, co, i, xa);
#pragma unroll
                for (int j = 0; j < SK; j++) in_l[j] = (j < SX ? (double)xa[j] : 0.0) - in_l[j];
            }
            normalize_coeff<SK, SO>(in_l, out_l);
            if constexpr (MODE == KS_PAIR) {
                static_assert(MODE != KS_PAIR || SO == SX, "pair needs SO == SX");
                int a2[SX];
                load_pair_sum<SX>(ka, ap, bp, co, i, a2);
                double d_l[SO], r_l[SO];
#pragma unroll
                for (int j = 0; j < SO; j++) d_l[j] = (double)a2[j] - out_l[j];
                normalize_coeff<SO, SO>(d_l, r_l);
                int dst = i + ka.t;
                const bool ng = dst >= N;
                if (ng) dst -= N;
#pragma unroll
                for (int j = 0; j < SO; j++) op[glwe_off(j, co) + dst] = cneg((int)r_l[j], ng);
            } else {
#pragma unroll
                for (int j = 0; j < SO; j++) op[glwe_off(j, co) + i] = (int)out_l[j];
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
    for (int idx = threadIdx.x; idx < 2 * N; idx += blockDim.x) {
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
// out = a * X^rho   (glwe_rotate, ram.rs:629); out must not alias a
template <int S>
__global__ __launch_bounds__(256) void k_rotate(GlweRef a, GlweRef out, int rho) {
    const int32_t* ap = at(a);
    int32_t* op = at(out);
    for (int idx = threadIdx.x; idx < 2 * N; idx += blockDim.x) {
        const int col = idx >> LOGN, i = idx & (N - 1);
        int src; bool sgn;
        rot_src(i, rho, src, sgn);
#pragma unroll
        for (int j = 0; j < S; j++) op[glwe_off(j, col) + i] = cneg(ap[glwe_off(j, col) + src], sgn);
    }
}

}  // namespace fk
