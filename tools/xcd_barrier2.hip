// Measurement tool (not part of the product): variants of the in-kernel hand-off of tools/xcd_barrier.hip (MODE 0 there:
// one XCD per group, drained stores, relaxed counter at the L2, sc1 loads), to see what the 1.0-1.2 us per hand-off are made of.
//   V 0: the product's barrier: fetch_add, then tid 0 polls the counter (no sleep)
//   V 1: the poll sleeps one tick between loads (s_sleep 1)
//   V 2: arrivals go to the counter, the LAST arriver (it sees it in the fetch_add's return value) writes a go word in ANOTHER
//        128-byte line, everybody polls that line: polls and adds do not meet at one L2 line
//   V 3: as 0, but every WAVE drains its own stores and arrives by itself (8 arrivals per workgroup, no workgroup barrier
//        in front of the arrival)
//   V 4: no data hand-off at all (barriers only): the cost of the two meetings alone
//   V 5: no meeting at all: the data carry a 2-bit round tag (v * 4 + tag, exact for integers below 2^50), the producer just
//        stores, the consumer polls its 8 values until every tag is the round's; the second meeting of a round (consumer done
//        before the producer overwrites) is replaced by double buffering on the round's parity
//   hipcc -O3 --offload-arch=gfx950 -o tools/xcd_barrier2 tools/xcd_barrier2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int T = 512, E = 8, POLY = 4096, NXCD = 8;
constexpr size_t LDS = 140 * 1024;

template <int V>
__device__ __forceinline__ void meet(unsigned* ctr, unsigned* go, unsigned want_round, int G, int* flag) {
    if (V == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            const unsigned want = want_round * G * (T / 64);
            for (int spin = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spin < (1 << 22); spin++) { }
        }
        __syncthreads();
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned want = want_round * G;
        const unsigned v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (V == 2) {
            if (v == want) __hip_atomic_store(go, want_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else for (int spin = 0; __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want_round && spin < (1 << 22); spin++) { }
        } else if (v < want) {
            for (int spin = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spin < (1 << 22); spin++) { if (V == 1) __builtin_amdgcn_s_sleep(1); }
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(T) void k_tagged(double* buf, unsigned* errors, int G, int groups, int rounds) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    const int g = b % NXCD, m = b / NXCD;
    if (g >= groups || m >= G) return;
    lds[threadIdx.x] = 0.0;
    unsigned bad = 0;
    for (int r = 0; r < rounds; r++) {
        double* mine = buf + (((size_t)(r & 1) * NXCD + g) * G + m) * POLY;
        const double* theirs = buf + (((size_t)(r & 1) * NXCD + g) * G + (m + 1) % G) * POLY;
        const double tag = (double)((r + 1) & 3);
#pragma unroll
        for (int k = 0; k < E; k++) mine[threadIdx.x + T * k] = ((double)(r * 7 + m * 3 + k) + lds[threadIdx.x]) * 4.0 + tag;
        const double expect0 = (double)(r * 7 + ((m + 1) % G) * 3);
        double v[E];
        for (int spin = 0; spin < (1 << 20); spin++) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < E; k++) v[k] = __hip_atomic_load(theirs + threadIdx.x + T * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < E; k++) { const double q = __builtin_floor(v[k] * 0.25); ok = ok && (v[k] - 4.0 * q == tag); v[k] = q; }
            if (__all(ok)) break;     // wave-uniform exit: the whole wave polls again while any lane is stale
        }
#pragma unroll
        for (int k = 0; k < E; k++) bad += (v[k] != expect0 + k);
        __syncthreads();   // stands for the workgroup-level dependency of the real kernel (transform across waves)
    }
    if (bad) atomicAdd(errors, bad);
}
template <int V>
__global__ __launch_bounds__(T) void k_rounds(double* buf, unsigned* counters, unsigned* errors, int G, int groups, int rounds) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    const int g = b % NXCD, m = b / NXCD;
    if (g >= groups || m >= G) return;
    lds[threadIdx.x] = 0.0;
    double* mine = buf + ((size_t)g * G + m) * POLY;
    const double* theirs = buf + ((size_t)g * G + (m + 1) % G) * POLY;
    unsigned* ctr = counters + g * 128;   // four 128-byte lines per group: counter A, go A, counter B, go B
    unsigned bad = 0;
    for (int r = 0; r < rounds; r++) {
        if (V != 4) {
#pragma unroll
            for (int k = 0; k < E; k++) mine[threadIdx.x + T * k] = (double)(r * 7 + m * 3 + k) + lds[threadIdx.x];
        }
        meet<V>(ctr, ctr + 32, (unsigned)(r + 1), G, nullptr);
        if (V != 4) {
            const double expect0 = (double)(r * 7 + ((m + 1) % G) * 3);
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double v = __hip_atomic_load(theirs + threadIdx.x + T * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad += (v != expect0 + k);
            }
        }
        meet<V>(ctr + 64, ctr + 96, (unsigned)(r + 1), G, nullptr);
    }
    if (bad) atomicAdd(errors, bad);
}
template <int V>
void run(const char* name, int G, int groups, int rounds) {
    const int grid = G * NXCD;
    double* buf; unsigned *ctr, *err;
    hipMalloc(&buf, (size_t)NXCD * G * POLY * 8); hipMalloc(&ctr, NXCD * 512); hipMalloc(&err, 4);
    hipFuncSetAttribute((const void*)k_rounds<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    unsigned herr = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(ctr, 0, NXCD * 512); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_rounds<V>), dim3(grid), dim3(T), LDS, 0, buf, ctr, err, G, groups, rounds);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); herr += e;
    }
    printf("%-66s G=%2d groups=%d: %.2f us per round, errors %u\n", name, G, groups, best * 1e3 / rounds, herr);
    hipFree(buf); hipFree(ctr); hipFree(err);
}
void run_tagged(int G, int groups, int rounds) {
    const int grid = G * NXCD;
    double* buf; unsigned* err;
    hipMalloc(&buf, (size_t)2 * NXCD * G * POLY * 8); hipMalloc(&err, 4);
    hipFuncSetAttribute((const void*)k_tagged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    unsigned herr = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(buf, 0, (size_t)2 * NXCD * G * POLY * 8); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_tagged, dim3(grid), dim3(T), LDS, 0, buf, err, G, groups, rounds);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); herr += e;
    }
    printf("%-66s G=%2d groups=%d: %.2f us per round, errors %u\n", "V5 tagged data, no meeting (ONE hand-off per round, double buffered)", G, groups, best * 1e3 / rounds, herr);
    hipFree(buf); hipFree(err);
}
int main() {
    const int rounds = 400;
    for (int groups : {4, 8})
        for (int G : {12, 24}) {
            run<0>("V0 fetch_add + poll the counter", G, groups, rounds);
            run<1>("V1 ... with s_sleep 1 between polls", G, groups, rounds);
            run<2>("V2 last arriver writes a go word in another line, polls go there", G, groups, rounds);
            run<3>("V3 every wave arrives by itself", G, groups, rounds);
            run<4>("V4 the two meetings alone (no data)", G, groups, rounds);
            run_tagged(G, groups, rounds);
        }
    return 0;
}
