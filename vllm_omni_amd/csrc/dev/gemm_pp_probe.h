// Phase probe of the product ping-pong GEMM kernel (included by ../gemm.hip only with -DOMNI_DEV -DOMNI_PP_PROBE=1).
// Dev-only timing probe (-DOMNI_DEV -DOMNI_PP_PROBE=1; tools/probe/pp_probe.cpp reads it back): every wave stamps s_memtime
// (one tick = one shader cycle) at five points of every phase -
//   T1 load section issued (fragment reads + this phase's two DMA pieces)   T2 counted DMA wait passed   T3 barrier + lgkmcnt(0)
//   passed = first MFMA may issue   T4 last MFMA of the cluster issued   T5 closing barrier passed (= the next phase's start)
// - and keeps the SUMS of each stamp over all phases in SGPRs (differences of the sums = cycles per segment; 32-bit wrap-around
// cancels), plus the T3 / T4 of the four phases of the middle K-tile (the hand-off between the two waves of a SIMD: partner's T3
// minus this wave's T4).  A stamp is consumed one phase later, at a point where the wave's lgkm queue holds nothing else (the end
// of a cluster), so the probe adds no wait to the load sections: ~11 scalar instructions per phase.  Output: 16 uint32 per wave at
// P.splitk_ws (the non-split launch does not use it; the harness passes splitk_ws_floats = 0).  Product builds: all of it is empty.
#define OMNI_PP_STAMP(v) do { (v) = __builtin_amdgcn_s_memtime(); } while (0)
// The additions are pinned between two empty volatile asm statements that "modify" the accumulators: as free C code hipcc sank
// them below the closing barrier (and waited for the T5 stamp at the head of the next load section) or could hoist them to the
// head of the cluster (a wait for T3 in front of the first MFMA).  Stamps and sums are 64-bit so that no half of a stamp's SGPR
// pair is dead while its s_memtime is in flight (the allocator re-used the high half at once: a write-after-write wait).
// Phase index of a quadrant: (mq, nq) = (0,0) (0,1) (1,1) (1,0) -> 0 1 2 3.
#define OMNI_PP_PROBE_ACCUM(nq, mq)                                                                        \
  do {                                                                                                     \
    constexpr int ph_ = (mq) ? 3 - (nq) : (nq);                                                            \
    asm volatile("" : "+s"(pb_s[0]), "+s"(pb_s[1]), "+s"(pb_s[2]), "+s"(pb_s[3]), "+s"(pb_s[4]));          \
    pb_s[0] += pb_t1; pb_s[1] += pb_t2; pb_s[2] += pb_t3; pb_s[3] += pb_t4; pb_s[4] += pb_t5;              \
    pb_snap3[ph_] = pb_snap ? (uint32_t)pb_t3 : pb_snap3[ph_];                                             \
    pb_snap4[(ph_ + 3) & 3] = pb_snap ? (uint32_t)pb_t4 : pb_snap4[(ph_ + 3) & 3];                         \
    asm volatile("" : "+s"(pb_s[0]), "+s"(pb_s[1]), "+s"(pb_s[2]), "+s"(pb_s[3]), "+s"(pb_s[4]),           \
                      "+s"(pb_snap3[ph_]), "+s"(pb_snap4[(ph_ + 3) & 3]));                                  \
    ++pb_ph;                                                                                               \
  } while (0)
// state of the probe, declared in front of the K-loop
#define OMNI_PP_PROBE_DECLS()                                                                              \
  uint64_t pb_t1, pb_t2, pb_t3, pb_t4, pb_t5, pb_s[5] = {0u, 0u, 0u, 0u, 0u};                              \
  uint32_t pb_snap3[4] = {0u, 0u, 0u, 0u}, pb_snap4[4] = {0u, 0u, 0u, 0u};                                 \
  uint32_t pb_ph = 0;                                                                                      \
  bool pb_snap = false;                                                                                    \
  OMNI_PP_STAMP(pb_t5);                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
  const uint64_t pb_init = pb_t5;                                                                          \
  pb_t4 = pb_t5 /* the "previous phase" of phase 0: its T4 / T5 sums start with this stamp */
// behind the K-loop: 16 uint32 per wave at P.splitk_ws (the non-split launch does not use it)
#define OMNI_PP_PROBE_WRITE()                                                                              \
  do {                                                                                                     \
  if (!SPLITK && P.splitk_ws) {                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    pb_s[3] += pb_t4; pb_s[4] += pb_t5;                                                                     \
    if (lane == 0) {                                                                                        \
      uint32_t* const o = reinterpret_cast<uint32_t*>(P.splitk_ws) + ((int64_t)blockIdx.x * 8 + wave) * 16; \
      o[0] = (uint32_t)pb_s[0]; o[1] = (uint32_t)pb_s[1]; o[2] = (uint32_t)pb_s[2]; o[3] = (uint32_t)pb_s[3]; o[4] = (uint32_t)pb_s[4]; \
      o[5] = (uint32_t)pb_init; o[6] = (uint32_t)pb_t5; o[7] = pb_ph;                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) { o[8 + i] = pb_snap3[i]; o[12 + i] = pb_snap4[i]; }                      \
    }                                                                                                       \
  }                                                                                                         \
  } while (0)
