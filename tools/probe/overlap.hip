// dev probe: does VALU work of a wave overlap with its own / its SIMD partner's MFMAs on gfx950?
// loop body = 1 MFMA (32x32x16 bf16: 32 matrix-pipe cycles, or 2x 16x16x32) + NV independent VALU ops (+ NE v_exp_f32).
// prints cycles per loop iteration per SIMD for waves-per-SIMD = 1, 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NV, int NE, int SHAPE>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  f32x16_t acc0 = {}, acc1 = {};
  f32x4_t q0 = {}, q1 = {}, q2 = {}, q3 = {};
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 32) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
    } else if (SHAPE == 16) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q1) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 15]) : "v"(v[(j + 1) & 15]), "v"(v[(j + 2) & 15]));
#pragma unroll
    for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j + 5) & 15]));
    if (SHAPE == 32) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
    } else if (SHAPE == 16) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(q3) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j + 8) & 15]) : "v"(v[(j + 1) & 15]), "v"(v[(j + 2) & 15]));
#pragma unroll
    for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j + 9) & 15]));
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  for (int i = 0; i < 4; ++i) s += q0[i] + q1[i] + q2[i] + q3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (2.0f * iters);
}

template <int NV, int NE, int SHAPE>
void run(float* d, const char* name) {
  for (int threads : {256, 512}) {   // 1 or 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    probe<NV, NE, SHAPE><<<256, threads>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NV, NE, SHAPE><<<256, threads>>>(d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0, cyc = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %7.1f ns per half-iteration (1 MFMA-slot of 32 pipe cycles + %2d fma + %d exp per wave), s_memtime ticks %6.1f\n",
           name, threads / 256, ms * 1e6 / (2.0 * iters), NV, NE, cyc);
  }
}

int main() {
  float* d;
  hipMalloc(&d, ((1 << 20) + 16) * 4);
  run<0, 0, 32>(d, "32x32x16 only");
  run<0, 0, 16>(d, "2x 16x16x32 only");
  run<4, 0, 32>(d, "32x32x16 + 4 fma");
  run<8, 0, 32>(d, "32x32x16 + 8 fma");
  run<16, 0, 32>(d, "32x32x16 + 16 fma");
  run<8, 0, 16>(d, "2x 16x16x32 + 8 fma");
  run<16, 0, 16>(d, "2x 16x16x32 + 16 fma");
  run<0, 2, 32>(d, "32x32x16 + 2 exp");
  run<4, 2, 32>(d, "32x32x16 + 4 fma + 2 exp");
  run<8, 2, 32>(d, "32x32x16 + 8 fma + 2 exp");
  run<8, 0, 0>(d, "no MFMA, 8 fma");
  run<0, 2, 0>(d, "no MFMA, 2 exp");
  run<8, 2, 0>(d, "no MFMA, 8 fma + 2 exp");
  return 0;
}
