// Measurement tool (not part of the product): issue rate of the FP64 / integer VALU instructions the
// modular arithmetic in csrc/ntt_dev.hpp is built from, on gfx950.  Prints wave-instructions per SIMD
// per cycle-ish (normalised to v_fma_f64 = 1.0).   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define CHAINS 8

template <int OP> __global__ void __launch_bounds__(512) k_rate(double* out, double seed, int iters) {
  double v[CHAINS];
  long long w[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { v[i] = seed + threadIdx.x * 1e-3 + i; w[i] = (long long)(seed * 1e6) + threadIdx.x + i; }
  const double c1 = 1.0000001, c2 = 1e-9, MAGIC = 6755399441055744.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == 0) v[i] = __builtin_fma(v[i], c1, c2);
      if (OP == 1) v[i] = v[i] * c1;
      if (OP == 2) v[i] = v[i] + c2;
      if (OP == 3) v[i] = __builtin_rint(v[i]) ;
      if (OP == 4) v[i] = (v[i] + MAGIC) - MAGIC;                    // 2 instructions
      if (OP == 5) v[i] = __builtin_floor(v[i]);
      if (OP == 6) w[i] = (w[i] >> 3) + 12345;                       // ashr i64 + add i64
      if (OP == 7) w[i] = w[i] + 0x123456789LL;                      // add_co + addc
      if (OP == 8) { int t = (int)w[i]; v[i] = (double)t; w[i] += 1; }   // cvt_f64_i32 (+ add)
      if (OP == 9) { v[i] = (double)(int)v[i]; }                     // cvt_i32_f64 + cvt_f64_i32
      if (OP == 10) { asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i])); }
      if (OP == 11) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2)); }
      if (OP == 12) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c1)); }
      if (OP == 13) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c2)); }
      if (OP == 14) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[i]) : "v"((int)w[i])); }
      if (OP == 15) { int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(v[i])); w[i] = t; }
      if (OP == 16) { asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(w[i])); }
      if (OP == 17) { int t = (int)w[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(7)); w[i] = t; }
      if (OP == 18) { int t = (int)w[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(7)); w[i] = t; }
      if (OP == 19) { asm volatile("v_floor_f64 %0, %0" : "+v"(v[i])); }
      if (OP == 20) { int t = (int)w[i]; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t) : "v"(7)); w[i] = t; }
      if (OP == 21) { asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS])); }
    }
  }
  double s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i] + (double)w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// shader clock while FP64-busy: d(s_memtime) / d(s_memrealtime) * 100 MHz (MI355X_MICROARCH.md, "in-kernel clock")
__global__ void __launch_bounds__(512) k_clock(double* out, unsigned long long* stamps, int iters) {
  double v[CHAINS];
  for (int i = 0; i < CHAINS; ++i) v[i] = 1.5 + threadIdx.x * 1e-3 + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0000001), "v"(1e-9));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < CHAINS; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

// dependent-chain behaviour at the evaluator's occupancy: one 512-thread workgroup per CU (2 waves/SIMD),
// CH independent mulmod-like chains per thread (mul, fma, mul, rndne, fma, add: each op depends on the previous)
template <int CH> __global__ void __launch_bounds__(512) k_chain(double* out, int iters) {
  double v[CH];
  for (int i = 0; i < CH; ++i) v[i] = 1.5 + threadIdx.x * 1e-3 + i;
  const double w = 123456789.0, pinv = 1.0 / 281474976768001.0, p = 281474976768001.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const double h = v[i] * w;
      const double l = __builtin_fma(v[i], w, -h);
      const double q = __builtin_rint(h * pinv);
      const double r = __builtin_fma(-q, p, h);
      v[i] = r + l;
    }
  }
  double s = 0; for (int i = 0; i < CH; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH> static void chain_probe(double* d, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096;
  for (int r = 0; r < 200; ++r) k_chain<CH><<<blocks, 512>>>(d, iters);   // settle the clock
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) k_chain<CH><<<blocks, 512>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winst = (double)blocks * 8 / 1024 * iters * CH * 6 * 20;
  printf("mulmod chains: %d waves/SIMD, %d independent chains/thread: %.2f ns per wave-instr per SIMD\n", blocks * 8 / 1024, CH, ms * 1e6 / winst);
}

static void clock_probe(double* d) {
  const int blocks = 512, iters = 1 << 16;
  unsigned long long* st; hipMalloc(&st, blocks * 2 * sizeof(unsigned long long));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float total = 0, ms = 0;
  int launches = 0;
  while (total < 2500.f) {                       // >= 2 s of back-to-back FP64 launches, then read the last one
    hipEventRecord(a);
    k_clock<<<blocks, 512>>>(d, st, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); total += ms; ++launches;
  }
  unsigned long long h[blocks * 2]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < blocks; ++i) sum += (double)h[2 * i] / (double)h[2 * i + 1];
  const double mhz = sum / blocks * 100.0;
  const double winst = (double)blocks * 8 / 1024 * iters * CHAINS;
  printf("sustained FP64 load: %d launches, last %.3f ms, shader clock %.0f MHz, v_fma_f64 = %.2f cycles per wave-instr per SIMD\n",
         launches, ms, mhz, ms * 1e-3 * mhz * 1e6 / winst);
  hipFree(st);
}

template <int OP> double run(const char* name, double* d, int per_iter, double ref) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 2;
  k_rate<OP><<<blocks, 512>>>(d, 1.5, 16);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a);
    k_rate<OP><<<blocks, 512>>>(d, 1.5, ITERS);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  // wave-instructions per SIMD: blocks*8 waves / (256 CUs * 4 SIMDs) * ITERS*CHAINS*per_iter
  double winst = (double)blocks * 8 / (256.0 * 4) * ITERS * CHAINS * per_iter;
  double ns_per = best * 1e6 / winst;
  printf("%-28s %8.3f ms  %6.2f ns per wave-instr per SIMD  (x%.2f of fma)\n", name, best, ns_per, ref > 0 ? ns_per / ref : 1.0);
  return ns_per;
}

int main() {
  double* d; hipMalloc(&d, 256 * 2 * 512 * sizeof(double));
  double f = run<11>("v_fma_f64 (asm)", d, 1, 0);
  run<0>("fma (C)", d, 1, f);
  run<12>("v_mul_f64 (asm)", d, 1, f);
  run<13>("v_add_f64 (asm)", d, 1, f);
  run<10>("v_rndne_f64 (asm)", d, 1, f);
  run<3>("rint (C)", d, 1, f);
  run<4>("magic add/sub (2 instr)", d, 2, f);
  run<19>("v_floor_f64 (asm)", d, 1, f);
  run<14>("v_cvt_f64_i32 (asm)", d, 1, f);
  run<15>("v_cvt_i32_f64 (asm)", d, 1, f);
  run<16>("v_ashrrev_i64 (asm)", d, 1, f);
  run<21>("v_lshl_add_u64 (asm)", d, 1, f);
  run<17>("v_add_u32 (asm)", d, 1, f);
  run<18>("v_mul_lo_u32 (asm)", d, 1, f);
  run<20>("v_cndmask_b32 (asm)", d, 1, f);
  run<7>("add i64 (C: add_co+addc)", d, 2, f);
  clock_probe(d);
  chain_probe<1>(d, 256); chain_probe<2>(d, 256); chain_probe<4>(d, 256); chain_probe<8>(d, 256);
  chain_probe<1>(d, 512); chain_probe<2>(d, 512); chain_probe<4>(d, 512); chain_probe<8>(d, 512);
  hipFree(d);
  return 0;
}
