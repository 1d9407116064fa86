// Dev tool, torch-free: times omni_gemm_bf16 of one or several builds of libomni_cdna4 on the bench's roofline launch
// (gemm_bf16_pp_kernel<GELU>, M = 40960 + 640, N = 12288, K = 3072, all operands K32-blocked) and, for a build compiled with
// -DOMNI_DEV -DOMNI_PP_PROBE=1 (csrc/gemm.hip "Dev-only timing probe"), reads the per-phase s_memtime sums back and prints where a
// wave of each ping-pong group spends the cycles of a phase and how long the matrix pipe of a SIMD waits between the last MFMA
// of one group's cluster and the first MFMA of the partner's.  No Python, no torch: the whole run is a few seconds of box time
// (a fresh box pays 1-2 minutes for `import torch` alone).
//
//   build:  tools/probe/build_pp_probe.sh            (hipcc; also builds the probe variant of the library)
//   run:    tools/probe/pp_probe [--iters 20] [--m 40960] [--n 12288] [--k 3072] [--epi 0|1|2|4] lib1.so [lib2.so ...]
//           (--epi 2: gated residual as out-proj / MLP-down run it; --epi 4 with --n 9216: the QKV projection with q/k norm + RoPE)
//           tools/probe/pp_probe --sweep ref.so other.so ...     (bit-identity of the builds over ragged / small / large shapes)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
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

// bf16 noise in [-1, 1) x scale: a cheap integer hash per element (the clock under load depends on the data: zeros run faster)
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

// order-independent 64-bit digest of a buffer (sum of per-word hashes): equal digests of two builds = bit-identical outputs
__global__ void digest_u32(const uint32_t* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long h = (unsigned long long)p[i] * 0x9E3779B97F4A7C15ull + (unsigned long long)i * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    acc += h * 0xBF58476D1CE4E5B9ull;
  }
  atomicAdd(out, acc);
}

typedef int (*gemm_fn)(const omni_gemm_params*, omni_stream);

int main(int argc, char** argv) {
  int iters = 20, m_img = 40960, m_txt = 640, N = 12288, K = 3072;
  bool sweep = false;
  int epi = OMNI_EPI_BIAS_GELU_TANH;
  std::vector<std::string> libs;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--m") && i + 1 < argc) m_img = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--sweep")) sweep = true;
    else if (!strcmp(argv[i], "--n") && i + 1 < argc) N = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--k") && i + 1 < argc) K = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--epi") && i + 1 < argc) epi = atoi(argv[++i]);
    else libs.push_back(argv[i]);
  }
  if (libs.empty()) {
    fprintf(stderr, "usage: pp_probe [--iters n] [--m rows] lib.so [lib2.so ...]\n");
    return 1;
  }
  uint16_t *a0, *a1, *w0, *w1, *bias, *o0, *o1;
  uint32_t* probe;
  unsigned long long* dig;
  HIP_OK(hipMalloc(&dig, 8));
  const size_t tiles = ((size_t)(m_img + 255) / 256 + (size_t)(m_txt + 255) / 256) * ((N + 255) / 256);
  const size_t probe_words = tiles * 8 * 16;
  HIP_OK(hipMalloc(&a0, (size_t)m_img * K * 2));
  HIP_OK(hipMalloc(&a1, (size_t)m_txt * K * 2));
  HIP_OK(hipMalloc(&w0, (size_t)N * K * 2));
  HIP_OK(hipMalloc(&w1, (size_t)N * K * 2));
  HIP_OK(hipMalloc(&bias, (size_t)N * 2));
  HIP_OK(hipMalloc(&o0, (size_t)m_img * N * 2));
  HIP_OK(hipMalloc(&o1, (size_t)m_txt * N * 2));
  HIP_OK(hipMalloc(&probe, probe_words * 4));
  fill_bf16<<<2048, 256>>>(a0, (size_t)m_img * K, 1u, 1.0f);
  fill_bf16<<<2048, 256>>>(a1, (size_t)m_txt * K, 2u, 1.0f);
  fill_bf16<<<2048, 256>>>(w0, (size_t)N * K, 3u, 0.02f);
  fill_bf16<<<2048, 256>>>(w1, (size_t)N * K, 4u, 0.02f);
  HIP_OK(hipMemset(bias, 0, (size_t)N * 2));
  HIP_OK(hipDeviceSynchronize());

  omni_gemm_params p;
  memset(&p, 0, sizeof(p));
  p.ngroups = 2; p.N = N; p.K = K; p.epilogue = epi; p.w_k32_blocked = 1;
  p.g[0].A = a0; p.g[0].lda = K; p.g[0].M = m_img; p.g[0].W = w0; p.g[0].bias = bias; p.g[0].out = o0; p.g[0].ldo = N;
  p.g[0].a_k32_rows = m_img; p.g[0].out_k32_rows = m_img;
  p.g[1].A = a1; p.g[1].lda = K; p.g[1].M = m_txt; p.g[1].W = w1; p.g[1].bias = bias; p.g[1].out = o1; p.g[1].ldo = N;
  p.g[1].a_k32_rows = m_txt; p.g[1].out_k32_rows = m_txt;
  if (epi == OMNI_EPI_BIAS_GATE_RES) {
    // out-proj / MLP-down as the DiT block runs them: res + gate[item] * (acc + bias), row-major output, one gate row per item
    uint16_t *res0, *res1, *gate;
    HIP_OK(hipMalloc(&res0, (size_t)m_img * N * 2));
    HIP_OK(hipMalloc(&res1, (size_t)m_txt * N * 2));
    HIP_OK(hipMalloc(&gate, (size_t)64 * N * 2));
    fill_bf16<<<2048, 256>>>(res0, (size_t)m_img * N, 5u, 1.0f);
    fill_bf16<<<256, 256>>>(res1, (size_t)m_txt * N, 6u, 1.0f);
    fill_bf16<<<256, 256>>>(gate, (size_t)64 * N, 7u, 1.0f);
    for (int g = 0; g < 2; ++g) {
      p.g[g].out_k32_rows = 0;
      p.g[g].res = g ? res1 : res0; p.g[g].ldres = N; p.g[g].gate = gate; p.g[g].gate_item_stride = N;
      p.g[g].rows_per_item = g ? 64 : 4096;
    }
  }
  if (epi == OMNI_EPI_BIAS_SPLIT3 || epi == OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE) {
    // the fused QKV projection: three row-major outputs of N / 3 columns; q / k additionally per-head RMS-normed and rotated
    if (N % 384) { fprintf(stderr, "--epi 3 / 4 need N = 3 x (a multiple of 128)\n"); return 1; }
    const int sn = N / 3, npos = 4160;
    uint16_t *ob0, *ob1, *nw, *cs;
    int32_t* pos;
    HIP_OK(hipMalloc(&ob0, (size_t)m_img * sn * 2 * 2));            // out1, out2 of group 0
    HIP_OK(hipMalloc(&ob1, (size_t)m_txt * sn * 2 * 2));
    HIP_OK(hipMalloc(&nw, 2 * 128 * 2));
    HIP_OK(hipMalloc(&cs, (size_t)2 * npos * 64 * 2));
    HIP_OK(hipMalloc(&pos, (size_t)(m_img + m_txt) * 4));
    fill_bf16<<<1, 256>>>(nw, 256, 8u, 1.0f);
    fill_bf16<<<256, 256>>>(cs, (size_t)2 * npos * 64, 9u, 1.0f);
    std::vector<int32_t> hp((size_t)m_img + m_txt);
    for (size_t i = 0; i < hp.size(); ++i) hp[i] = (int32_t)(i % npos);
    HIP_OK(hipMemcpy(pos, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    p.split_n = sn;
    for (int g = 0; g < 2; ++g) {
      const size_t mg = g ? m_txt : m_img;
      uint16_t* extra = g ? ob1 : ob0;
      p.g[g].out_k32_rows = 0; p.g[g].ldo = sn;
      p.g[g].out1 = extra; p.g[g].out2 = extra + mg * sn;
      p.g[g].qk_norm_q_w = nw; p.g[g].qk_norm_k_w = nw + 128; p.g[g].qk_rope_cos = cs; p.g[g].qk_rope_sin = cs + (size_t)npos * 64;
      p.g[g].qk_row_pos = pos + (g ? m_img : 0); p.g[g].qk_eps = 1e-6f;
    }
  }
  p.splitk_ws = reinterpret_cast<float*>(probe);      // read by probe builds only (splitk_ws_floats = 0: never a split-K workspace)
  p.splitk_ws_floats = 0;
  const double flop = 2.0 * (m_img + m_txt) * (double)N * K;

  if (sweep) {
    // correctness sweep: every build must produce the digest of the FIRST one on every configuration (ragged M / N, one to many
    // K-tiles, row-major and K32-blocked operands, bias and GELU epilogues, one and two groups)
    struct Cfg { int m0, m1, n, k, epi, blocked; };
    const Cfg cfgs[] = {
        {40960, 640, 12288, 3072, 1, 1}, {40960, 640, 3072, 3072, 0, 1}, {20480, 320, 3072, 12288, 0, 1}, {40960, 640, 9216, 3072, 0, 1},
        {4160, 0, 3072, 3072, 0, 0},     {4099, 77, 3072, 3072, 0, 0},   {1000, 0, 64, 3072, 0, 0},        {640, 0, 3072, 3584, 0, 0},
        {2048, 64, 3072, 64, 0, 0},      {2048, 64, 3072, 128, 0, 0},    {2048, 64, 3072, 192, 0, 0},      {2048, 64, 3072, 256, 0, 0},
        {2048, 64, 3072, 320, 0, 0},     {2048, 64, 3072, 384, 0, 1},    {2048, 64, 12288, 448, 1, 1},     {8192, 0, 4096, 8192, 0, 0},
        {300, 0, 264, 3072, 1, 0},       {256, 256, 256, 512, 0, 1},
    };
    std::vector<gemm_fn> fns;
    for (const std::string& path : libs) {
      void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 2; }
      fns.push_back((gemm_fn)dlsym(h, "omni_gemm_bf16"));
    }
    int bad = 0;
    for (const Cfg& c : cfgs) {
      if ((size_t)c.m0 * c.k > (size_t)m_img * K || (size_t)c.n * c.k > (size_t)N * K || (size_t)c.m0 * c.n > (size_t)m_img * N ||
          c.m1 > m_txt) { printf("skip %d+%d x %d x %d (buffers)\n", c.m0, c.m1, c.n, c.k); continue; }
      omni_gemm_params q;
      memset(&q, 0, sizeof(q));
      q.ngroups = c.m1 ? 2 : 1; q.N = c.n; q.K = c.k; q.epilogue = c.epi; q.w_k32_blocked = c.blocked;
      q.g[0].A = a0; q.g[0].lda = c.k; q.g[0].M = c.m0; q.g[0].W = w0; q.g[0].bias = bias; q.g[0].out = o0; q.g[0].ldo = c.n;
      q.g[1].A = a1; q.g[1].lda = c.k; q.g[1].M = c.m1; q.g[1].W = w1; q.g[1].bias = bias; q.g[1].out = o1; q.g[1].ldo = c.n;
      if (c.blocked) { q.g[0].a_k32_rows = c.m0; q.g[1].a_k32_rows = c.m1; }
      if (c.blocked && c.n % 32 == 0) { q.g[0].out_k32_rows = c.m0; q.g[1].out_k32_rows = c.m1; }
      unsigned long long first = 0;
      printf("%6d+%-4d x %5d x %5d epi %d %s:", c.m0, c.m1, c.n, c.k, c.epi, c.blocked ? "k32-blocked" : "row-major  ");
      for (size_t li = 0; li < fns.size(); ++li) {
        HIP_OK(hipMemset(o0, 0x5a, (size_t)c.m0 * c.n * 2));
        if (c.m1) HIP_OK(hipMemset(o1, 0x5a, (size_t)c.m1 * c.n * 2));
        const int st = fns[li](&q, nullptr);
        unsigned long long hd = 0;
        HIP_OK(hipMemset(dig, 0, 8));
        digest_u32<<<2048, 256>>>(reinterpret_cast<const uint32_t*>(o0), (size_t)c.m0 * c.n / 2, dig);
        if (c.m1) digest_u32<<<64, 256>>>(reinterpret_cast<const uint32_t*>(o1), (size_t)c.m1 * c.n / 2, dig);
        HIP_OK(hipMemcpy(&hd, dig, 8, hipMemcpyDeviceToHost));
        if (li == 0) first = hd;
        const bool ok = st == 0 && hd == first;
        bad += !ok;
        printf("  [%zu] st %d %016llx %s", li, st, hd, ok ? "ok" : "DIFFERS");
      }
      printf("\n");
    }
    printf("sweep: %s\n", bad ? "FAILED" : "all builds bit-identical to the first on every configuration");
    return bad ? 3 : 0;
  }

  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  std::vector<uint32_t> host(probe_words);
  for (int round = 0; round < 2; ++round) {             // two interleaved rounds: the clock drifts while the box warms up
    for (const std::string& path : libs) {
      void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 2; }
      gemm_fn gemm = (gemm_fn)dlsym(h, "omni_gemm_bf16");
      if (!gemm) { fprintf(stderr, "%s has no omni_gemm_bf16\n", path.c_str()); return 2; }
      for (int i = 0; i < 3; ++i) {
        const int st = gemm(&p, nullptr);
        if (st) { fprintf(stderr, "%s: omni_gemm_bf16 -> %d\n", path.c_str(), st); return 2; }
      }
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) gemm(&p, nullptr);
      HIP_OK(hipEventRecord(e1, nullptr));
      HIP_OK(hipDeviceSynchronize());
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      unsigned long long hd = 0;
      HIP_OK(hipMemset(dig, 0, 8));
      digest_u32<<<4096, 256>>>(reinterpret_cast<const uint32_t*>(o0), (size_t)m_img * N / 2, dig);
      digest_u32<<<256, 256>>>(reinterpret_cast<const uint32_t*>(o1), (size_t)m_txt * N / 2, dig);
      HIP_OK(hipMemcpy(&hd, dig, 8, hipMemcpyDeviceToHost));
      printf("round %d  %-58s %8.1f us  %7.1f TF/s  out %016llx\n", round, path.c_str(), us, flop / (us * 1e-6) / 1e12, hd);
      HIP_OK(hipMemset(o0, 0, (size_t)m_img * N * 2));   // the next build must write every element itself
      if (round == 1) {
        HIP_OK(hipMemset(probe, 0, probe_words * 4));
        gemm(&p, nullptr);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(host.data(), probe, probe_words * 4, hipMemcpyDeviceToHost));
        size_t waves = 0;
        double seg[2][6] = {{0}}, cnt[2] = {0, 0}, hand[2] = {0, 0}, handn[2] = {0, 0}, clus[2] = {0, 0};
        for (size_t wg = 0; wg < tiles; ++wg) {
          for (int w = 0; w < 8; ++w) {
            const uint32_t* o = &host[(wg * 8 + w) * 16];
            if (!o[7]) continue;                          // not a probe build (or a skipped tile)
            ++waves;
            const int g = w >> 2;
            const double n = o[7];
            const uint32_t d1 = o[0] - (o[4] - o[6]), d2 = o[1] - o[0], d3 = o[2] - o[1], d4 = (o[3] - o[5]) - o[2], d5 = o[4] - o[3];
            seg[g][0] += d1 / n; seg[g][1] += d2 / n; seg[g][2] += d3 / n; seg[g][3] += d4 / n; seg[g][4] += d5 / n;
            seg[g][5] += (double)(uint32_t)(o[6] - o[5]) / n;
            cnt[g] += 1;
          }
          // hand-offs on SIMD s of this workgroup (waves s and s + 4), middle K-tile: snapshots T3[0..3] at o[8..11], T4[0..2] at o[12..14]
          for (int s = 0; s < 4; ++s) {
            const uint32_t* a = &host[(wg * 8 + s) * 16];
            const uint32_t* b = &host[(wg * 8 + s + 4) * 16];
            if (!a[7] || !b[7] || !a[8] || !b[8]) continue;
            for (int ph = 0; ph < 3; ++ph) {
              hand[0] += (double)(int32_t)(b[8 + ph] - a[12 + ph]);         // group 0's cluster ph ends -> group 1's cluster ph starts
              hand[1] += (double)(int32_t)(a[8 + ph + 1] - b[12 + ph]);     // group 1's cluster ph ends -> group 0's cluster ph + 1 starts
              handn[0] += 1; handn[1] += 1;
              clus[0] += (double)(int32_t)(a[12 + ph] - a[8 + ph]);
              clus[1] += (double)(int32_t)(b[12 + ph] - b[8 + ph]);
            }
          }
        }
        if (waves) {
          printf("  probe: %zu waves; shader cycles per PHASE of a wave (a K-tile = 4 phases; a phase of a wave spans two cluster slots of its SIMD)\n", waves);
          printf("  %-8s %12s %12s %14s %12s %14s %10s\n", "group", "load issue", "DMA wait", "barrier 1+lgkm", "cluster", "barrier 2", "phase");
          for (int g = 0; g < 2; ++g)
            if (cnt[g] > 0)
              printf("  %-8d %12.1f %12.1f %14.1f %12.1f %14.1f %10.1f\n", g, seg[g][0] / cnt[g], seg[g][1] / cnt[g], seg[g][2] / cnt[g],
                     seg[g][3] / cnt[g], seg[g][4] / cnt[g], seg[g][5] / cnt[g]);
          if (handn[0] > 0)
            printf("  middle K-tile, per SIMD: cluster (T3 -> last MFMA issued + probe adds) g0 %.1f  g1 %.1f cycles;  hand-off last-MFMA-issued -> partner's first-MFMA-may-issue:  g0->g1 %.1f  g1->g0 %.1f cycles\n",
                   clus[0] / handn[0], clus[1] / handn[1], hand[0] / handn[0], hand[1] / handn[1]);
        }
      }
      // (the library stays loaded: two builds of one kernel name are distinct code objects under RTLD_LOCAL)
    }
  }
  return 0;
}
