// tools/ubench.hip -- instruction-throughput probes on gfx950 for the ops that bound the simulate kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/_ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  uint64_t w[8];
  double d[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 17u + i * 1013u;
    w[i] = ((uint64_t)a[i] << 20) | 12345u;
    d[i] = 1.0 + (double)a[i] * 1e-9;
    f[i] = 1.0f + (float)(a[i] & 1023u) * 1e-4f;
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(0xD2511F53u) : "vcc"); }
      if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u)); }
      if (OP == 2) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(0xD2511F53u)); }
      if (OP == 3) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(0x9E3779B9u)); }
      if (OP == 4) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(1.0000001)); }
      if (OP == 5) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0000001)); }
      if (OP == 6) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(1.0000001f)); }
      if (OP == 7) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }
      if (OP == 8) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u)); }
      if (OP == 9) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i+1)&7])); }
      if (OP == 10) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(1.0000001)); }
      if (OP == 11) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed) : "vcc"); }
      if (OP == 12) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i])); }
      if (OP == 13) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(0x511F53u)); }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32) ^ (uint32_t)d[i] ^ (uint32_t)f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
double run(const char* name, int blocks) {
  uint32_t* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * 4 /*waves*/ * ITERS * 8;
  // cycles per wave-instruction per SIMD at 2.4 GHz, assuming all 1024 SIMDs busy
  double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / insts;
  printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz, %d blocks)\n", name, ms, cyc, blocks);
  hipFree(d);
  return cyc;
}

int main() {
  const int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves/SIMD
  run<3>("v_xor_b32", blocks);
  run<6>("v_mul_f32", blocks);
  run<9>("v_pk_mul_f32", blocks);
  run<11>("v_cndmask_b32", blocks);
  run<8>("v_mul_u32_u24", blocks);
  run<13>("v_mul_hi_u32_u24", blocks);
  run<1>("v_mul_lo_u32", blocks);
  run<2>("v_mul_hi_u32", blocks);
  run<0>("v_mad_u64_u32", blocks);
  run<4>("v_fma_f64", blocks);
  run<5>("v_add_f64", blocks);
  run<10>("v_mul_f64", blocks);
  run<7>("v_cvt_f64_f32", blocks);
  run<12>("v_sqrt_f32", blocks);
  return 0;
}
