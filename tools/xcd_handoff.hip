// Measurement tool (not part of the product): does a consumer kernel that runs on the SAME XCD as the producer of
// its input (block ids congruent mod 8 under the observed round-robin placement) read it faster across a
// kernel boundary than one on another XCD?  Producer: each of G workgroups writes 32 KB.  Consumer: workgroup b
// reads the 32 KB of workgroup (b + shift) % G, 12 such buffers in the "norm" shape (12 x 8 B per thread).
//   hipcc -O3 --offload-arch=gfx950 -o tools/xcd_handoff tools/xcd_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int T = 512, E = 8, POLY = 4096;
__global__ __launch_bounds__(T) void k_produce(double* buf, int it) {
    double* p = buf + (size_t)blockIdx.x * POLY;
#pragma unroll
    for (int k = 0; k < E; k++) p[threadIdx.x + T * k] = (double)(threadIdx.x + k + it);
}
__global__ __launch_bounds__(T) void k_consume(const double* buf, double* sink, int shift, int G) {
    const double* p = buf + (size_t)((blockIdx.x + shift) % G) * POLY;
    double acc = 0;
#pragma unroll
    for (int k = 0; k < E; k++) acc += p[threadIdx.x + T * k];
    sink[(size_t)blockIdx.x * T + threadIdx.x] = acc;
}
int main() {
    const int G = 96;
    double *buf, *sink;
    hipMalloc(&buf, (size_t)G * POLY * 8); hipMalloc(&sink, (size_t)G * T * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int shifts[] = {0, 8, 16, 1, 3, 4, 7};
    for (int rep = 0; rep < 2; rep++)
    for (int s : shifts) {
        for (int i = 0; i < 20; i++) { k_produce<<<G, T>>>(buf, i); k_consume<<<G, T>>>(buf, sink, s, G); }
        hipDeviceSynchronize();
        const int n = 200;
        hipEventRecord(a);
        for (int i = 0; i < n; i++) { k_produce<<<G, T>>>(buf, i); k_consume<<<G, T>>>(buf, sink, s, G); }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("shift %2d (%s XCD): %.2f us per produce+consume pair\n", s, s % 8 == 0 ? "same " : "other", ms * 1e3 / n);
    }
    return 0;
}
