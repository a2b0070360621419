// Band-kernel argument block and the matrix-core dispatch shared by taxim_kernels.hip and taxim_mfma.hip
// (the MFMA instantiations are a translation unit of their own: they dominate the build time).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tacex {

struct BlurArgs {
  const float* src;     // (B,H,W) previous level; unused when FIRST
  const float* hm;      // (B,H,W) height map (mm)
  const float* gel;     // (H,W)
  const float* shift_a; // (B,) S = (hm - shift_a) - shift_b
  const float* shift_b; // (B,)
  const float* pdepth;  // (B,) P
  float* dst;           // (B,H,W)
  uint8_t* mask_out;    // (B,H,W) nullable
  const float* taps;    // (K,)
  int H, W, B;
  int pitch;            // LDS row pitch in float2 units (even, == 2 mod 32)
  int padx;             // left padding in float2 units (even, >= (K-1)/2)
  float contact_scale;
  int restore;          // apply Z[M] = J[M]
  int row0, nbands;     // MFMA kernel: first row and band count of this launch
  const int* rows_ext;  // (B,4) first / last frame row | first / last frame column with a non-zero level-0 input, nullable (MFMA kernel:
                        // zero-band and zero-block skipping)
  int ext_grow;         // rows by which the non-zero range of THIS level's input has grown (sum of the previous levels' radii)
  int ext_grow_x;       // columns, likewise
};

bool mfma_supported(int k, bool first, int H, int W);
hipError_t dispatch_mfma(int k, bool first, const BlurArgs& a, hipStream_t st);

}  // namespace tacex
