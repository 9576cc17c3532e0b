// Measurement tool (not part of the product): what does ONE step of the dependent trace chain pay for its exchanges between the
// workgroups of a ciphertext (micro-benchmark in isolation before rebuilding k_trace_tail)?  No transforms, only
// the traffic and the hand-offs, G members per ciphertext on one XCD (block b runs on XCD b % 8), N_CT ciphertexts side by side.
//   scheme A (k_trace_tail today): every member stores a partial polynomial (32 KB), hand-off 1, a normalisation phase in which a
//            third of the threads read all G partials of a coefficient and store the digits, hand-off 2, every member loads one limb
//            polynomial (16 KB).
//   scheme B (one hand-off): every member ADDS its closed-form contribution (one int64 per coefficient: 32 KB of L2 atomics) into the
//            step's accumulator, hand-off, every member loads two accumulator polynomials (S and S3: 64 KB) at its own coefficients.
//   hipcc -O3 --offload-arch=gfx950 -o tools/tail_exchange_bench tools/tail_exchange_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int T = 512, E = 8, N = 4096, GROUPS = 8, SPIN_MAX = 1 << 14;

__device__ __forceinline__ double ld_l2(const double* p) { return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long ld_l2(const long long* p) { return __hip_atomic_load(const_cast<long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_l2(const int* p) { return __hip_atomic_load(const_cast<int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bounded: returns false (and raises *abortp) when the others do not show up
__device__ __forceinline__ bool handoff(unsigned* ctr, unsigned* abortp, unsigned want, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        for (int spin = 0; spin < SPIN_MAX; spin++) {
            if ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) { ok = 1; break; }
            if ((spin & 15) == 15 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        if (!ok) __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}
__device__ __forceinline__ void spin_work(int n, double& x) {   // stand-in for the transforms: n dependent FMAs per thread
    for (int i = 0; i < n; i++) x = __builtin_fma(x, 1.0000001, 1e-9);
}

template <int G>
__global__ __launch_bounds__(T) void k_scheme_a(double* big, int* ct, unsigned* sync, int n_ct, int iters, int work, double* sink) {
    __shared__ int flag;
    const int g = (int)blockIdx.x % GROUPS, m = (int)blockIdx.x / GROUPS;
    if (g >= n_ct) return;
    unsigned* ctr = sync + g * 32;
    unsigned* abortp = sync + GROUPS * 32;
    double* bigg = big + (size_t)g * G * N;
    int* c = ct + (size_t)g * 6 * N;
    const int tid = (int)threadIdx.x;
    double x = (double)tid;
    unsigned epoch = 0;
    constexpr int CH = (N + G - 1) / G;
    for (int it = 0; it < iters; it++) {
        // input: one limb polynomial (int32, 16 KB) past the L1
        int v[E];
#pragma unroll
        for (int k = 0; k < E; k++) v[k] = ld_l2(c + N + tid + T * k);
#pragma unroll
        for (int k = 0; k < E; k++) x += (double)v[k];
        spin_work(work, x);
        double* p = bigg + (size_t)m * N;
#pragma unroll
        for (int k = 0; k < E; k++) p[tid + T * k] = x + k;
        if (!handoff(ctr, abortp, (++epoch) * G, &flag)) break;
        for (int col = 1; col >= 0; col--) {
            const int i = m * CH + tid;
            if (tid < CH && i < N) {
                double s = 0;
                for (int q = 0; q < G / 2; q++) s += ld_l2(bigg + (size_t)(col * (G / 2) + q) * N + i);
                int r0 = ld_l2(c + (col * 3 + 0) * N + i), r1 = ld_l2(c + (col * 3 + 1) * N + i), r2 = ld_l2(c + (col * 3 + 2) * N + i);
                const int d = (int)s + r0 + r1 + r2;
                c[(col * 3 + 0) * N + i] = d; c[(col * 3 + 1) * N + i] = d + 1; c[(col * 3 + 2) * N + i] = d + 2;
            }
            if (col == 1) {   // (the real kernel announces itself after the mask column and does the body column under the hand-off)
            }
        }
        if (!handoff(ctr, abortp, (++epoch) * G, &flag)) break;
    }
    sink[blockIdx.x * T + tid] = x;
}

template <int G>
__global__ __launch_bounds__(T) void k_scheme_b(long long* acc, unsigned* sync, int n_ct, int iters, int work, double* sink) {
    __shared__ int flag;
    const int g = (int)blockIdx.x % GROUPS, m = (int)blockIdx.x / GROUPS;
    if (g >= n_ct) return;
    unsigned* ctr = sync + g * 32;
    unsigned* abortp = sync + GROUPS * 32;
    const int tid = (int)threadIdx.x;
    double x = (double)tid;
    unsigned epoch = 0;
    const int co = m / (G / 2), jr = m % (G / 2), j = jr / 3;     // member (column, limb, digit row)
    for (int it = 0; it < iters; it++) {
        // accumulators of this step: [step % 12][ciphertext][column][S, S3][N]
        long long* s_prev = acc + ((size_t)((it + 11) % 12) * GROUPS + g) * 4 * N;
        long long* s_cur = acc + ((size_t)(it % 12) * GROUPS + g) * 4 * N;
        long long a[E], b[E];
#pragma unroll
        for (int k = 0; k < E; k++) { a[k] = ld_l2(s_prev + (size_t)(2 + 0) * N + tid + T * k); b[k] = ld_l2(s_prev + (size_t)(2 + 1) * N + tid + T * k); }   // the mask column's S and S3
#pragma unroll
        for (int k = 0; k < E; k++) x += (double)(a[k] + b[k]);
        spin_work(work, x);
        long long* dst = s_cur + (size_t)(co * 2 + (j >= 3 ? 1 : 0)) * N;
#pragma unroll
        for (int k = 0; k < E; k++) __hip_atomic_fetch_add(dst + tid + T * k, (long long)x + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!handoff(ctr, abortp, (++epoch) * G, &flag)) break;
    }
    sink[blockIdx.x * T + tid] = x;
}

template <typename F>
static float run(F launch, unsigned* sync) {
    hipMemset(sync, 0, (GROUPS + 1) * 32 * sizeof(unsigned));
    launch(20);
    hipDeviceSynchronize();
    hipMemset(sync, 0, (GROUPS + 1) * 32 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(600);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned ab = 0;
    hipMemcpy(&ab, sync + GROUPS * 32, 4, hipMemcpyDeviceToHost);
    if (ab) printf("   (ABORTED: a hand-off did not complete)\n");
    return ms * 1e3f / 600;
}

int main() {
    constexpr int G = 24;
    double *big, *sink; int* ct; long long* acc; unsigned* sync;
    hipMalloc(&big, (size_t)GROUPS * G * N * 8); hipMalloc(&sink, (size_t)GROUPS * 32 * T * 8);
    hipMalloc(&ct, (size_t)GROUPS * 6 * N * 4); hipMemset(ct, 0, (size_t)GROUPS * 6 * N * 4);
    hipMalloc(&acc, (size_t)12 * GROUPS * 4 * N * 8); hipMemset(acc, 0, (size_t)12 * GROUPS * 4 * N * 8);
    hipMalloc(&sync, (GROUPS + 1) * 32 * sizeof(unsigned));
    for (int n_ct : {1, 4, 8}) {
        for (int work : {0, 1500}) {    // 1500 dependent FP64 FMAs ~ 2.8 us: a forward and an inverse transform's worth of time
            const float a = run([&](int it) { hipLaunchKernelGGL(k_scheme_a<G>, dim3(GROUPS * G), dim3(T), 0, 0, big, ct, sync, n_ct, it, work, sink); }, sync);
            const float b = run([&](int it) { hipLaunchKernelGGL(k_scheme_b<G>, dim3(GROUPS * G), dim3(T), 0, 0, acc, sync, n_ct, it, work, sink); }, sync);
            printf("%d ciphertexts x %d members, stand-in work %4d FMAs: scheme A (two hand-offs, partials) %6.2f us per step   scheme B (one hand-off, L2 atomics) %6.2f us per step\n",
                   n_ct, G, work, a, b);
        }
    }
    return 0;
}
