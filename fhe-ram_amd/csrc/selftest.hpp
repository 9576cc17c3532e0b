// Included by fheram.hip (same translation unit): directed self-test of the FP64 arithmetic the kernels are built on
// (fft_dev.hpp: the pair transforms, the prepared-operand scaling and the complex multiply-accumulate, called exactly as the
// fused kernels call them).  The operands come from the host (tests/test_gpu_fft.py builds random and adversarial ones up to
// the path's worst case: six terms of limbs at +-2^16) and the sums go back as RAW doubles — before the rounding the
// inverse transform ends with on the path — so the host can compare them with exact integer arithmetic and report the
// round-off that the rounding has to absorb (< 1/2 is correctness; the measured margin is what the test pins).
// Nothing on the RAM path calls this entry point.
#pragma once
#include "ctx.hpp"

namespace fk {

// out[0] = sum_r a_r * g_r,  out[1] = sum_r a_r * g_{r ^ 1}   (negacyclic, R terms, R even), raw doubles.
// SINGLE: every transform runs on its own instead of two at a time, half a phase apart.
// ROUNDED: the inverse transforms end with the path's rounding (nat_out<true>), i.e. they report to the round-off monitor.
template <bool SINGLE, bool ROUNDED = false>
__global__ __launch_bounds__(T) void k_selftest_convolve(const int32_t* __restrict__ a, const int32_t* __restrict__ g, double* __restrict__ out,
                                                         const double* __restrict__ tw_g, double ninv, int R) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    RoMonitor ro_mon(lds, tw_g, ROUNDED);
    double* tw = lds;
    double* data = lds + LDS_TW;
    const int tid = vt((int)threadIdx.x);
    load_twiddles(tw, tw_g, tid);
    double acc[2][E];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k = 0; k < E; k++) acc[b][k] = 0.0;
#pragma unroll 1
    for (int r = 0; r < R; r += 2) {
        double x[2][E], gg[2][E];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int k = 0; k < E; k++) {
                x[b][k] = (double)a[(long)(r + b) * N + tid + T * k];
                gg[b][k] = (double)g[(long)(r + b) * N + tid + T * k];
            }
        if constexpr (SINGLE) {
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&x[0]), tw, data, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&x[1]), tw, data + LDS_DATA, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&gg[0]), tw, data, tid);
            ntt_fwd<1>(*reinterpret_cast<double(*)[1][E]>(&gg[1]), tw, data + LDS_DATA, tid);
        } else {
            ntt_fwd<2>(x, tw, data, tid);
            ntt_fwd<2>(gg, tw, data, tid);
        }
        OpRegs o[2];
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int kk = 0; kk < E / 2; kk++) { o[b].v[kk].x = gg[b][2 * kk] * ninv; o[b].v[kk].y = gg[b][2 * kk + 1] * ninv; }
        mac_regs(acc[0], x[0], o[0]); mac_regs(acc[0], x[1], o[1]);
        mac_regs(acc[1], x[0], o[1]); mac_regs(acc[1], x[1], o[0]);
    }
    if constexpr (SINGLE) {
        double* const d0[1] = {data};
        double* const d1[1] = {data + LDS_DATA};
        fft_inv_skew<1, 1, ROUNDED>(*reinterpret_cast<double(*)[1][E]>(&acc[0]), tw, d0, tid);
        fft_inv_skew<1, 1, ROUNDED>(*reinterpret_cast<double(*)[1][E]>(&acc[1]), tw, d1, tid);
    } else {
        double* const d[2] = {data, data + LDS_DATA};
        fft_inv_skew<2, 1, ROUNDED>(acc, tw, d, tid);
    }
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int k = 0; k < E; k++) out[(long)b * N + tid + T * k] = acc[b][k];
}

}  // namespace fk

extern "C" {

namespace {
int selftest_run(fheram_ctx* c, int n_terms, const int32_t* a, const int32_t* g, double* out, int singles, bool rounded, double scale);
}
int fheram_selftest_convolve(fheram_ctx* c, int n_terms, const int32_t* a, const int32_t* g, double* out, int singles) {
    return selftest_run(c, n_terms, a, g, out, singles, false, 1.0);
}
int fheram_selftest_convolve_rounded(fheram_ctx* c, int n_terms, const int32_t* a, const int32_t* g, double* out, double operand_scale) {
    return selftest_run(c, n_terms, a, g, out, 0, true, operand_scale);
}
namespace {
int selftest_run(fheram_ctx* c, int n_terms, const int32_t* a, const int32_t* g, double* out, int singles, bool rounded, double scale) {
    if (!c || n_terms <= 0 || n_terms > 8 || (n_terms & 1) || !a || !g || !out) return fail(c, FHERAM_ERR_INVALID_ARG, "bad argument (n_terms: even, <= 8)");
    HIPCHK(c, hipSetDevice(c->device));
    int32_t* d = nullptr;
    double* dout = nullptr;
    const size_t nb = (size_t)n_terms * fk::N * sizeof(int32_t);
    HIPCHK(c, hipMalloc(&d, 2 * nb));
    hipError_t e = hipMalloc(&dout, 2 * fk::N * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d, a, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + (size_t)n_terms * fk::N, g, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fk::k_selftest_convolve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fk::LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fk::k_selftest_convolve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fk::LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fk::k_selftest_convolve<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fk::LDS_BYTES);
    if (e == hipSuccess) {
        if (rounded) hipLaunchKernelGGL((fk::k_selftest_convolve<false, true>), dim3(1), dim3(fk::T), fk::LDS_BYTES, c->stream, d, d + (size_t)n_terms * fk::N, dout, c->d_tw, c->ninv * scale, n_terms);
        else if (singles) hipLaunchKernelGGL(fk::k_selftest_convolve<true>, dim3(1), dim3(fk::T), fk::LDS_BYTES, c->stream, d, d + (size_t)n_terms * fk::N, dout, c->d_tw, c->ninv, n_terms);
        else hipLaunchKernelGGL(fk::k_selftest_convolve<false>, dim3(1), dim3(fk::T), fk::LDS_BYTES, c->stream, d, d + (size_t)n_terms * fk::N, dout, c->d_tw, c->ninv, n_terms);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, 2 * fk::N * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d);
    if (dout) hipFree(dout);
    if (e != hipSuccess) return fail(c, FHERAM_ERR_DEVICE, std::string("selftest: ") + hipGetErrorString(e));
    return rounded ? check_precision(c) : FHERAM_OK;
}
}  // namespace

}  // extern "C"
