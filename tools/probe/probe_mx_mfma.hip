// probe: operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void k(const v8i* a, const v8i* b, v4f* c, const int* sa, const int* sb, int mode) {
  int l = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  if (mode == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 1, sa[l], 2, sb[l]);
  if (mode == 2) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 3, sa[l], 3, sb[l]);
  c[l] = acc;
}
// e4m3fn encode of small integers / simple values
static uint8_t e4m3(float x) {
  if (x == 0) return 0;
  uint8_t s = x < 0 ? 0x80 : 0; x = fabsf(x);
  int e; float m = frexpf(x, &e);      // x = m * 2^e, m in [0.5,1)
  int E = e - 1 + 7;                    // x = (2m) * 2^(e-1)
  int M = (int)roundf((2 * m - 1) * 8);
  if (M == 8) { M = 0; E++; }
  if (E <= 0) return s;                 // (no subnormals needed here)
  return s | (uint8_t)(E << 3) | (uint8_t)M;
}
int main() {
  const int M = 16, N = 16, K = 128;
  static float A[16][128], B[128][16];
  srand(1);
  for (int i = 0; i < M; i++) for (int kk = 0; kk < K; kk++) A[i][kk] = (float)((rand() % 7) - 3);
  for (int kk = 0; kk < K; kk++) for (int j = 0; j < N; j++) B[kk][j] = (float)((rand() % 5) - 2);
  // scales per (row, 32-block): exponents 126..129 -> x0.5, 1, 2, 4
  static int SA[16][4], SB[16][4];
  for (int i = 0; i < 16; i++) for (int g = 0; g < 4; g++) { SA[i][g] = 126 + (i + g) % 4; SB[i][g] = 126 + (2 * i + g) % 3; }
  for (int hyp = 0; hyp < 2; hyp++) {
    // operand packing under hypothesis hyp.  hyp 0: lane (i = l&15, g = l>>4) holds k = 32g + byte.  hyp 1: bytes 0-15 -> k = 16g + b, bytes 16-31 -> k = 64 + 16g + (b-16)
    uint8_t ha[64][32], hb[64][32];
    int hsa[3][64], hsb[3][64];
    for (int l = 0; l < 64; l++) {
      int i = l & 15, g = l >> 4;
      for (int bb = 0; bb < 32; bb++) {
        int kk = hyp == 0 ? 32 * g + bb : (bb < 16 ? 16 * g + bb : 64 + 16 * g + (bb - 16));
        ha[l][bb] = e4m3(A[i][kk]);
        hb[l][bb] = e4m3(B[kk][i]);
      }
    }
    v8i *da, *db; v4f* dc; int *dsa, *dsb;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, ha, 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++) {
      // scale VGPR: byte `opsel` carries the lane's scale (row l&15, block l>>4); other bytes garbage (0x55)
      int opa = mode == 0 ? 0 : (mode == 1 ? 1 : 3), opb = mode == 0 ? 0 : (mode == 1 ? 2 : 3);
      int ssa[64], ssb[64];
      for (int l = 0; l < 64; l++) {
        int i = l & 15, g = l >> 4;
        ssa[l] = 0x55555555; ssb[l] = 0x55555555;
        ssa[l] = (ssa[l] & ~(0xff << (8 * opa))) | (SA[i][g] << (8 * opa));
        ssb[l] = (ssb[l] & ~(0xff << (8 * opb))) | (SB[i][g] << (8 * opb));
      }
      hipMemcpy(dsa, ssa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, ssb, 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb, mode);
      float hc[64][4];
      hipMemcpy(hc, dc, 64 * 16, hipMemcpyDeviceToHost);
      // expected: D[i][j] = sum_g 2^(SA[i][g]-127) 2^(SB[j][g]-127) sum_{k in block g} A[i][k] B[k][j]; block g = k/32 (scale block hypothesis: 32 consecutive k)
      double err_rc = 0, err_cr = 0, err_noscale = 0;
      for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        int col = l & 15, row = (l >> 4) * 4 + r;
        double e1 = 0, e2 = 0, e3 = 0;
        for (int kk = 0; kk < K; kk++) {
          int g = kk / 32;
          e1 += ldexp(1.0, SA[row][g] - 127) * ldexp(1.0, SB[col][g] - 127) * A[row][kk] * B[kk][col];
          e2 += ldexp(1.0, SA[col][g] - 127) * ldexp(1.0, SB[row][g] - 127) * A[col][kk] * B[kk][row];
          e3 += A[row][kk] * B[kk][col];
        }
        err_rc += fabs(hc[l][r] - e1); err_cr += fabs(hc[l][r] - e2); err_noscale += fabs(hc[l][r] - e3);
      }
      printf("hyp %d mode %d (opsel a %d b %d): |D - expected| rowcol %.3f  colrow %.3f  unscaled %.3f   D[0..3] of lane 0: %.2f %.2f %.2f %.2f\n",
             hyp, mode, opa, opb, err_rc, err_cr, err_noscale, hc[0][0], hc[0][1], hc[0][2], hc[0][3]);
    }
  }
  return 0;
}
