// probe: operand / result lane layout of v_mfma_f32_32x32x2_f32 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* A /*32x2*/, const float* B /*2x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x;
  const float a = A[(l & 31) * 2 + (l >> 5)];   // A[i = l&31][k = l>>5]
  const float b = B[(l >> 5) * 32 + (l & 31)];  // B[k = l>>5][j = l&31]
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = acc[r];
  }
}
int main() {
  float hA[64], hB[64], hD[1024], ref[1024];
  for (int i = 0; i < 64; ++i) { hA[i] = 1.0f + 0.37f * i; hB[i] = 2.0f - 0.11f * i * i * 0.01f + (i % 7); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) ref[i * 32 + j] = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
  float *dA, *dB, *dD;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double md = 0; int bad = 0;
  for (int i = 0; i < 1024; ++i) { double d = fabs(hD[i] - ref[i]); if (d > md) md = d; if (d > 1e-3) ++bad; }
  printf("max diff %g, bad %d / 1024\n", md, bad);
  if (bad) { // try transposed interpretation
    int badT = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (fabs(hD[j * 32 + i] - ref[i * 32 + j]) > 1e-3) ++badT;
    printf("transposed bad %d\n", badT);
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 6; ++j) printf("%9.3f/%9.3f ", hD[i * 32 + j], ref[i * 32 + j]); printf("\n"); }
  }
  return 0;
}
