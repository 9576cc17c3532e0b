// Measurement tool (not part of the product): what does an IN-KERNEL hand-off between the workgroups of one group
// cost when the group sits on ONE XCD (shared L2: no L2 write-back / invalidate needed, only the per-CU L1 has to be
// bypassed) against a group spread over all XCDs (agent-scope release / acquire)?
// Shape of the latency-bound tail of a read: GROUPS groups of G workgroups (512 threads, 140 KB of LDS each, so one
// per CU); per round every workgroup writes 32 KB, the group meets at a counter barrier, every workgroup reads the
// 32 KB a neighbour of its group wrote and checks them (stale data = error).
//   hipcc -O3 --offload-arch=gfx950 -o tools/xcd_barrier tools/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int T = 512, E = 8, POLY = 4096, NXCD = 8;
constexpr size_t LDS = 140 * 1024;

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
// MODE 0: group g = the workgroups with blockIdx % 8 == g (one XCD under round-robin placement); stores drained
//         (vmcnt 0), relaxed counter at L2, data read with sc1 loads (L1 bypass).  No L2 maintenance.
// MODE 1: same placement, but agent-scope release / acquire fences (what the memory model asks for when the
//         producer may sit on another XCD).
// MODE 2: group g = G consecutive block ids (spread over all XCDs), agent-scope fences.
template <int MODE>
__global__ __launch_bounds__(T) void k_rounds(double* buf, unsigned* counters, int* xcc_out, unsigned* errors, int G, int groups, int rounds) {
    extern __shared__ double lds[];
    const int b = blockIdx.x;
    int g, m;
    if (MODE == 2) { g = b / G; m = b % G; } else { g = b % NXCD; m = b / NXCD; }
    if (threadIdx.x == 0) xcc_out[b] = xcc_id();
    if (g >= groups || m >= G) return;
    lds[threadIdx.x] = 0.0;
    double* mine = buf + ((size_t)g * G + m) * POLY;
    const double* theirs = buf + ((size_t)g * G + (m + 1) % G) * POLY;
    unsigned* ctr = counters + g * 32;   // one 128-byte line per group
    unsigned bad = 0;
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int k = 0; k < E; k++) mine[threadIdx.x + T * k] = (double)(r * 7 + m * 3 + k) + lds[threadIdx.x];
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * G;
            for (int spin = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spin < (1 << 22); spin++) __builtin_amdgcn_s_sleep(1);   // bounded: a lost peer ends in errors, not in a hang
        }
        __syncthreads();
        if (MODE != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const double expect0 = (double)(r * 7 + ((m + 1) % G) * 3);
#pragma unroll
        for (int k = 0; k < E; k++) {
            double v;
            if (MODE == 0) v = __hip_atomic_load(theirs + threadIdx.x + T * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = theirs[threadIdx.x + T * k];
            bad += (v != expect0 + k);
        }
        // second barrier of a round (the consumer side must be done before the producer overwrites): counted in the cost
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every load of this wave has returned before it says so
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * G;
            for (int spin = 0; __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spin < (1 << 22); spin++) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
void run(const char* name, int G, int groups, int rounds) {
    const int grid = (MODE == 2) ? G * groups : G * NXCD;
    double* buf; unsigned *ctr, *err; int* xcc;
    hipMalloc(&buf, (size_t)NXCD * G * POLY * 8); hipMalloc(&ctr, NXCD * 128); hipMalloc(&err, 4); hipMalloc(&xcc, grid * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    unsigned herr = 0;
    for (int rep = 0; rep < 5; rep++) {
        hipMemset(ctr, 0, NXCD * 128); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_rounds<MODE>), dim3(grid), dim3(T), LDS, 0, buf, ctr, xcc, err, G, groups, rounds);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); herr += e;
    }
    std::vector<int> hx(grid);
    hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
    int misplaced = 0;
    if (MODE != 2) for (int i = 0; i < grid; i++) misplaced += (hx[i] != hx[i % NXCD]);
    printf("%-46s G=%2d groups=%d: %.2f us per round (2 barriers + 32 KB hand-off), errors %u, blocks off their XCD %d\n",
           name, G, groups, best * 1e3 / rounds, herr, misplaced);
    if (MODE != 2 && G == 24 && groups == 4) { printf("  XCC id of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %d", hx[i]); printf("\n"); }
    hipFree(buf); hipFree(ctr); hipFree(err); hipFree(xcc);
}
int main() {
    hipFuncSetAttribute((const void*)k_rounds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void*)k_rounds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute((const void*)k_rounds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    const int rounds = 200;
    for (int groups : {1, 4, 8})
        for (int G : {12, 24}) {
            run<0>("one XCD per group, drained stores + sc1 loads", G, groups, rounds);
            run<1>("one XCD per group, agent-scope fences", G, groups, rounds);
            run<2>("groups spread over XCDs, agent-scope fences", G, groups, rounds);
        }
    return 0;
}
