// Hardware probe (not product code): dumps the lane/register layouts this repo's kernels ASSUME for
//   v_mfma_f32_32x32x16_bf16 (C/D map, A/B row ownership) and ds_read_b64_tr_b16 (16-lane transpose),
// so one GPU run confirms or refutes them.   hipcc --offload-arch=gfx950 -O2 tools/probe_isa.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;

__global__ void mfma_probe(float* out /*[32*32]*/, int mode) {
  // mode 0: A[i][k] = i (rows), B = 1/16 -> C[i][j] = i      => reveals which C element a (lane, reg) holds: row
  // mode 1: A = 1/16, B[k][j] = j        -> C[i][j] = j      => column
  const int lane = threadIdx.x;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(mode == 0 ? (float)(lane & 31) : 0.0625f);
    b[e] = (__bf16)(mode == 1 ? (float)(lane & 31) : 0.0625f);
  }
  f32x16_t c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

__global__ void kgroup_probe(float* out) {
  // A[i][k]: lane group hi = lane>>5 holds k-slots; put A = 1 only in hi==0 lanes' element 0, B[k][j] = 1 only in
  // hi==0 lanes' element e: the product is non-zero only if (hi=0, elem 0) of A pairs with (hi=0, elem e) of B.
  const int lane = threadIdx.x;
  for (int e = 0; e < 8; ++e) {
    bf16x8_t a, b;
    for (int x = 0; x < 8; ++x) { a[x] = (__bf16)0.f; b[x] = (__bf16)0.f; }
    if ((lane >> 5) == 0) { a[0] = (__bf16)1.f; b[e] = (__bf16)1.f; }
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (lane == 0) out[e] = c[0];
  }
  for (int e = 0; e < 8; ++e) {  // cross-group: A in hi=0 elem 0, B in hi=1 elem e -> must be 0 everywhere
    bf16x8_t a, b;
    for (int x = 0; x < 8; ++x) { a[x] = (__bf16)0.f; b[x] = (__bf16)0.f; }
    if ((lane >> 5) == 0) a[0] = (__bf16)1.f;
    if ((lane >> 5) == 1) b[e] = (__bf16)1.f;
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    if (lane == 0) out[8 + e] = c[0];
  }
}

__global__ void tr_probe(int* out /*[64*4]*/) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = (short)i;   // element value = its index
  __syncthreads();
  // each lane supplies the address of 4 contiguous elements: row (m>>2) of a [4][ROWSTRIDE] block, cols 4*(m&3)
  const int m = lane & 15, g = lane >> 4;
  const int ROW = 64;  // elements
  const int elem = g * 1024 + (m >> 2) * ROW + 4 * (m & 3);
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(uint32_t)(uintptr_t)(lds + elem));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

int main() {
  float* d; int* di;
  hipMalloc(&d, 64 * 16 * 4); hipMalloc(&di, 64 * 4 * 4);
  float h[64 * 16]; int hi_[256];
  int bad = 0;
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(mfma_probe, 1, 64, 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
      const float want = mode == 0 ? (float)row : (float)col;
      if (h[lane * 16 + r] != want) { if (bad < 8) printf("MFMA C-map mismatch mode %d lane %d reg %d: got %g want %g\n", mode, lane, r, h[lane*16+r], want); ++bad; }
    }
  }
  printf("mfma_32x32x16 C/D map (row=(r&3)+8(r>>2)+4(lane>>5), col=lane&31; A row=lane&31; B col=lane&31): %s\n", bad ? "MISMATCH" : "OK");
  hipLaunchKernelGGL(kgroup_probe, 1, 64, 0, 0, d);
  hipMemcpy(h, d, 16 * 4, hipMemcpyDeviceToHost);
  printf("k-slot pairing A(hi0,e0) x B(hi0,e): "); for (int e = 0; e < 8; ++e) printf("%g ", h[e]);
  printf(" | A(hi0,e0) x B(hi1,e): "); for (int e = 0; e < 8; ++e) printf("%g ", h[8 + e]); printf("\n  (expect 1 0 0 0 0 0 0 0 | all 0)\n");
  hipLaunchKernelGGL(tr_probe, 1, 64, 0, 0, di);
  hipMemcpy(hi_, di, sizeof(hi_), hipMemcpyDeviceToHost);
  int tbad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
    const int i = lane & 15, g = lane >> 4;
    const int want = g * 1024 + j * 64 + i;   // row j, column i of the group's [4][.] block
    if (hi_[lane * 4 + j] != want) { if (tbad < 8) printf("tr16 mismatch lane %d j %d: got %d want %d\n", lane, j, hi_[lane*4+j], want); ++tbad; }
  }
  printf("ds_read_b64_tr_b16 (lane i of a 16-group gets [row j][col i], rows addressed by lanes 4j..4j+3): %s\n", tbad ? "MISMATCH" : "OK");
  if (tbad) { for (int lane = 0; lane < 20; ++lane) printf("lane %d: %d %d %d %d\n", lane, hi_[lane*4], hi_[lane*4+1], hi_[lane*4+2], hi_[lane*4+3]); }
  return (bad || tbad) ? 1 : 0;
}
