// Dev tool, torch-free (as pp_probe.cpp): times omni_flash_attn_fwd_ex of one or several builds of libomni_cdna4 on the bench's
// attention launch (10 items x 4160 tokens, 24 heads x 128: the joint text + image sequence of a 1024^2 CFG step-batch, q / k / v
// as three [rows, 3072] matrices, K32-blocked output as the out-proj GEMM reads it) and prints a digest of the output per build:
// equal digests = bit-identical builds.  30-45 s of box time per call (the box included) instead of minutes behind `import torch`.
//
//   build:  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probe/attn_bench.cpp -o tools/probe/attn_bench -ldl
//   run:    tools/probe/attn_bench [--iters 20] [--items 10] [--seq 4160] [--rowmajor] lib1.so [lib2.so ...]
//           variants: tools/build_variants.sh attention_w64 name "-DOMNI_W64_...=..."  (or `attention` for the small-grid kernel)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/omni_cdna4.h"

#define HIP_OK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
      exit(2);                                                                                      \
    }                                                                                               \
  } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float v = ((float)(h & 0xffff) / 32768.0f - 1.0f) * scale;
    uint32_t b;
    memcpy(&b, &v, 4);
    p[i] = (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
  }
}

__global__ void digest_u32(const uint32_t* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = (unsigned long long)p[i] * 0x9E3779B97F4A7C15ull + (unsigned long long)i * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    acc += h * 0xBF58476D1CE4E5B9ull;
  }
  atomicAdd(out, acc);
}

typedef int (*attn_fn)(const omni_bf16*, const omni_bf16*, const omni_bf16*, omni_bf16*, int64_t, int64_t, int64_t, int64_t,
                       const int32_t*, int32_t, int32_t, int32_t, int32_t, float, int32_t, omni_stream);

int main(int argc, char** argv) {
  int iters = 20, items = 10, seq = 4160, H = 24, DH = 128;
  bool rowmajor = false;
  std::vector<std::string> libs;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--items") && i + 1 < argc) items = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seq") && i + 1 < argc) seq = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--rowmajor")) rowmajor = true;
    else libs.push_back(argv[i]);
  }
  if (libs.empty()) {
    fprintf(stderr, "usage: attn_bench [--iters n] [--items b] [--seq s] [--rowmajor] lib.so [lib2.so ...]\n");
    return 1;
  }
  const int D = H * DH;
  const size_t rows = (size_t)items * seq, n = rows * D;
  uint16_t *q, *k, *v, *o;
  int32_t* cu;
  unsigned long long* dig;
  HIP_OK(hipMalloc(&q, n * 2));
  HIP_OK(hipMalloc(&k, n * 2));
  HIP_OK(hipMalloc(&v, n * 2));
  HIP_OK(hipMalloc(&o, n * 2));
  HIP_OK(hipMalloc(&cu, (items + 1) * 4));
  HIP_OK(hipMalloc(&dig, 8));
  fill_bf16<<<2048, 256>>>(q, n, 11u, 1.0f);
  fill_bf16<<<2048, 256>>>(k, n, 12u, 1.0f);
  fill_bf16<<<2048, 256>>>(v, n, 13u, 1.0f);
  std::vector<int32_t> hcu(items + 1);
  for (int i = 0; i <= items; ++i) hcu[i] = i * seq;
  HIP_OK(hipMemcpy(cu, hcu.data(), (items + 1) * 4, hipMemcpyHostToDevice));
  HIP_OK(hipDeviceSynchronize());
  const float scale = 1.0f / sqrtf((float)DH);
  const double flop = 4.0 * items * H * (double)seq * seq * DH;
  const int32_t k32 = rowmajor ? 0 : (int32_t)rows;

  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  for (int round = 0; round < 2; ++round) {
    for (const std::string& path : libs) {
      void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 2; }
      attn_fn fn = (attn_fn)dlsym(h, "omni_flash_attn_fwd_ex");
      if (!fn) { fprintf(stderr, "%s has no omni_flash_attn_fwd_ex\n", path.c_str()); return 2; }
      HIP_OK(hipMemset(o, 0, n * 2));
      for (int i = 0; i < 3; ++i) {
        const int st = fn(q, k, v, o, D, D, D, D, cu, items, H, DH, seq, scale, k32, nullptr);
        if (st) { fprintf(stderr, "%s: omni_flash_attn_fwd_ex -> %d\n", path.c_str(), st); return 2; }
      }
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) fn(q, k, v, o, D, D, D, D, cu, items, H, DH, seq, scale, k32, nullptr);
      HIP_OK(hipEventRecord(e1, nullptr));
      HIP_OK(hipDeviceSynchronize());
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      unsigned long long hd = 0;
      HIP_OK(hipMemset(dig, 0, 8));
      digest_u32<<<4096, 256>>>(reinterpret_cast<const uint32_t*>(o), n / 2, dig);
      HIP_OK(hipMemcpy(&hd, dig, 8, hipMemcpyDeviceToHost));
      printf("round %d  %-58s %8.1f us  %7.1f TF/s  out %016llx\n", round, path.c_str(), us, flop / (us * 1e-6) / 1e12, hd);
    }
  }
  return 0;
}
