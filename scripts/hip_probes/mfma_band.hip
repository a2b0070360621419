// probe: banded Toeplitz product with chained v_mfma_f32_32x32x2_f32 (the V-pass of blur_mfma_kernel in isolation)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 61, KS = (32 + K) / 2;
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__global__ void probe(const float* w, const float* In /*92x32*/, float* D /*32x32*/) {
  const int l = threadIdx.x, li = l & 31, lk = l >> 5;
  float wl[KS];
  static_for<0, KS>([&](auto kc) { constexpr int kk = decltype(kc)::value; const int t = 2 * kk + lk - li; wl[kk] = (t >= 0 && t < K) ? w[t] : 0.0f; });
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  static_for<0, KS>([&](auto kc) {
    constexpr int kk = decltype(kc)::value;
    const float b = In[(2 * kk + lk) * 32 + li];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[kk], b, acc, 0, 0, 0);
  });
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[r];
}
int main() {
  static float hw[K], hIn[92 * 32], hD[1024];
  double s = 0; for (int i = 0; i < K; ++i) { hw[i] = expf(-0.5f * (i - 30) * (i - 30) / (15.25f * 15.25f)); s += hw[i]; }
  for (int i = 0; i < K; ++i) hw[i] /= (float)s;
  for (int i = 0; i < 92 * 32; ++i) hIn[i] = 0.f;
  hIn[34 * 32 + 22] = -1.0f;  // impulse like the failing case
  for (int i = 0; i < 92 * 32; ++i) if (i % 7 == 3) hIn[i] += 0.01f * (i % 13);
  float *dw, *dIn, *dD; (void)hipMalloc(&dw, sizeof(hw)); (void)hipMalloc(&dIn, sizeof(hIn)); (void)hipMalloc(&dD, sizeof(hD));
  (void)hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice); (void)hipMemcpy(dIn, hIn, sizeof(hIn), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dw, dIn, dD);
  (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  double md = 0; int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0; for (int k = 0; k < 92; ++k) { int t = k - i; if (t >= 0 && t < K) ref += (double)hw[t] * hIn[k * 32 + j]; }
    double d = fabs(hD[i * 32 + j] - ref); if (d > md) md = d; if (d > 1e-5) ++bad;
  }
  printf("band probe: max diff %g, bad %d / 1024; D[4][22]=%g D[20][22]=%g\n", md, bad, hD[4 * 32 + 22], hD[20 * 32 + 22]);
  return 0;
}
