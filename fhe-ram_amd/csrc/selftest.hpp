// Included by fheram.hip (same translation unit): directed self-tests of the FP64 modular arithmetic the kernels
// are built on (ntt_dev.hpp: mulmod / macmod / reduce and the lazy-reduction bounds of the transforms).  The
// operands come from the host (tests/test_gpu_modarith.py builds the adversarial ones: |a| up to 16p, |b| <= p/2,
// products at 2^100..2^101, quotients on rounding ties, worst-case magnitudes for the 12-stage drift) and the
// results go back as doubles, which the host compares with exact integer arithmetic.  Nothing on the RAM path calls
// these entry points.
#pragma once
#include "ctx.hpp"

namespace fk {

__global__ void k_selftest_modarith(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ acc,
                                    double* __restrict__ out_mul, double* __restrict__ out_mac, double* __restrict__ out_red, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_mul[i] = mulmod(a[i], b[i]);
    out_mac[i] = macmod(acc[i], a[i], b[i]);
    out_red[i] = reduce(a[i]);
}

// One polynomial per workgroup through the transforms exactly as the fused kernels call them.
//   dir 0: forward transform of in (coefficients, natural order) -> out[E*tid + k] UNREDUCED (the lazy values the MACs
//          consume; the host checks residues and the magnitude bound)
//   dir 1: inverse transform (x N) of in (transform-domain values at E*tid + k, any magnitude the caller chooses
//          below 16p) -> out[tid + T*k], centred
//   dir 2: as dir 1 but without the initial reduce() (the key-switch kernels skip it: three MAC terms stay below 3p)
__global__ __launch_bounds__(T) void k_selftest_ntt(const double* __restrict__ in, double* __restrict__ out,
                                                    const double* __restrict__ tw_g, int dir) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = threadIdx.x;
    load_twiddles(tw, tw_g, tid);
    const double* src = in + (long)blockIdx.x * N;
    double* dst = out + (long)blockIdx.x * N;
    double x[1][E];
    if (dir == 0) {
#pragma unroll
        for (int k = 0; k < E; k++) x[0][k] = src[tid + T * k];
        ntt_fwd<1>(x, tw, data, tid);
#pragma unroll
        for (int k = 0; k < E; k++) dst[E * tid + k] = x[0][k];
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) x[0][k] = src[E * tid + k];
        if (dir == 1) ntt_inv<1, true, true>(x, tw, data, tid);
        else ntt_inv<1, true, false>(x, tw, data, tid);
#pragma unroll
        for (int k = 0; k < E; k++) dst[tid + T * k] = x[0][k];
    }
}

}  // namespace fk

extern "C" {

int fheram_selftest_modarith(fheram_ctx* c, int n, const double* a, const double* b, const double* acc, double* out_mul,
                             double* out_mac, double* out_red) {
    if (!c || n <= 0 || !a || !b || !acc || !out_mul || !out_mac || !out_red) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    const size_t nb = (size_t)n * sizeof(double);
    HIPCHK(c, hipMalloc(&d, 6 * nb));
    hipError_t e = hipMemcpy(d, a, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)n, acc, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fk::k_selftest_modarith, dim3((n + 255) / 256), dim3(256), 0, c->stream, d, d + n, d + 2 * (size_t)n,
                           d + 3 * (size_t)n, d + 4 * (size_t)n, d + 5 * (size_t)n, n);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out_mul, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_mac, d + 4 * (size_t)n, nb, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_red, d + 5 * (size_t)n, nb, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(c, FHERAM_ERR_DEVICE, std::string("selftest: ") + hipGetErrorString(e));
    return FHERAM_OK;
}

int fheram_selftest_ntt(fheram_ctx* c, int dir, int n_poly, const double* in, double* out) {
    if (!c || n_poly <= 0 || !in || !out || dir < 0 || dir > 2) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    const size_t nb = (size_t)n_poly * fk::N * sizeof(double);
    HIPCHK(c, hipMalloc(&d, 2 * nb));
    hipError_t e = hipMemcpy(d, in, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fk::k_selftest_ntt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fk::LDS_BYTES);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fk::k_selftest_ntt, dim3(n_poly), dim3(fk::T), fk::LDS_BYTES, c->stream, d, d + (size_t)n_poly * fk::N, c->d_tw, dir);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + (size_t)n_poly * fk::N, nb, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(c, FHERAM_ERR_DEVICE, std::string("selftest: ") + hipGetErrorString(e));
    return FHERAM_OK;
}

/* the modulus and the primitive 2N-th root the transforms use (for the host-side exact reference of the tests) */
int fheram_selftest_constants(uint64_t* p, uint64_t* psi) {
    if (p) *p = fk::P_U64;
    if (psi) *psi = fk::PSI_8192;
    return FHERAM_OK;
}

}  // extern "C"
