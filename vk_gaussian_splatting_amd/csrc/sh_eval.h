// sh_eval.h — view-dependent radiance from the stored SH record (shared by the kernels that shade).
#pragma once
#include "kernels_common.h"

namespace mgs {

// SH storage: one record per splat, [coef][rgb] exactly as the reference lays it out
// (threedgs_particle_buffers.h.slang:112-207), padded to 48 elements: 192 B fp32 / 96 B fp16 / 48 B uint8.
// Shading is DEFERRED to the compositor (only splats that reach an unsaturated screen region are shaded),
// so the record is fetched by whoever stages the splat — a random 192-byte read of three full 64-byte
// sectors — instead of being streamed for every frustum survivor.
// Register-lean evaluation: the 15 basis values are computed first, then the record is consumed in groups of
// four 16-byte vectors (scheduling fences keep the compiler from hoisting all 12 loads, which cost the
// compositor 70 VGPRs and half its occupancy); the first group's miss brings the whole record's sectors in.
// b / 255.0f for an integer-valued b in [0, 255], correctly rounded like the IEEE division the shader compiles to, in
// three instructions instead of the ~10 of a division: one Newton step on b * (1/255).  Equal to b / 255.0f for all 256
// inputs (exhaustive check: tests/test_host_cpu.py::test_div255_refinement_is_exact).
__device__ __forceinline__ float div255(float b)
{
  constexpr float r   = 1.0f / 255.0f;
  const float     q   = __fmul_rn(b, r);
  const float     rem = __fmaf_rn(-q, 255.0f, b);
  return __fmaf_rn(rem, r, q);
}

template <int FMT>
__device__ __forceinline__ void decodeShVector(const uint4& x, float (&e)[FMT == 0 ? 4 : (FMT == 1 ? 8 : 16)])
{
  const uint32_t w[4] = {x.x, x.y, x.z, x.w};
  if constexpr(FMT == 0)
  {
#pragma unroll
    for(int q = 0; q < 4; ++q)
      e[q] = __uint_as_float(w[q]);
  }
  else if constexpr(FMT == 1)
  {
#pragma unroll
    for(int q = 0; q < 4; ++q)
    {
      const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&w[q]));
      e[2 * q]       = f.x;
      e[2 * q + 1]   = f.y;
    }
  }
  else
  {
#pragma unroll
    for(int q = 0; q < 4; ++q)
#pragma unroll
      for(int k = 0; k < 4; ++k)  // threedgs_particle_buffers.h.slang:128-131: v/255*2-1
        e[4 * q + k] = __fmaf_rn(div255((float)((w[q] >> (8 * k)) & 255u)), 2.0f, -1.0f);  // x*2 is exact: == (x*2)-1
  }
}

// basis[k] multiplies coefficient k (k = 0..14: degree 1, 2, 3 bands); constants and signs of
// threedgs_particle_storage.h.slang:48-52,121-155
__device__ __forceinline__ void shBasis(float x, float y, float z, float (&bs)[15])
{
  const float C1 = 0.4886025119029199f;
  bs[0] = -C1 * y;
  bs[1] = C1 * z;
  bs[2] = -C1 * x;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  bs[3] = 1.0925484f * xy;
  bs[4] = -1.0925484f * yz;
  bs[5] = 0.3153916f * (2.0f * zz - xx - yy);
  bs[6] = -1.0925484f * xz;
  bs[7] = 0.5462742f * (xx - yy);
  bs[8]  = -0.5900435899266435f * (3.0f * xx - yy) * y;
  bs[9]  = 2.890611442640554f * xy * z;
  bs[10] = -0.4570457994644658f * (4.0f * zz - xx - yy) * y;
  bs[11] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
  bs[12] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
  bs[13] = 1.445305721320277f * (xx - yy) * z;
  bs[14] = -0.5900435899266435f * x * (xx - 3.0f * yy);
}

// rgb += SH(deg, dir) for one splat
template <int FMT>
__device__ __forceinline__ void addShRadiance(const void* sh, uint32_t li, int degree, float dx, float dy, float dz, float& r,
                                              float& g, float& b)
{
  constexpr int PER = FMT == 0 ? 4 : (FMT == 1 ? 8 : 16);
  constexpr int REC = 48 / PER;  // 16-byte vectors per record
  const uint4*  p   = reinterpret_cast<const uint4*>(sh) + (size_t)li * REC;
  float         bs[15];
  shBasis(dx, dy, dz, bs);
  const int nEl = degree >= 3 ? 45 : (degree == 2 ? 24 : (degree == 1 ? 9 : 0));
  float     acc[3] = {0.f, 0.f, 0.f};
#ifndef MGS_SH_GROUP
#define MGS_SH_GROUP 3
#endif
  constexpr int GROUP = FMT == 0 ? MGS_SH_GROUP : (FMT == 1 ? 3 : 1);  // 16-byte vectors per round trip, chosen by the compositor's spill count at 6 waves per SIMD (fp32 3 / fp16 3 / uint8 1)
#pragma unroll
  for(int v0 = 0; v0 < REC; v0 += GROUP)
  {
    if(v0 * PER < nEl)
    {
      uint4 x[GROUP];
#pragma unroll
      for(int v = 0; v < GROUP; ++v)
        x[v] = p[v0 + v];
#pragma unroll
      for(int v = 0; v < GROUP; ++v)
      {
        float e[PER];
        decodeShVector<FMT>(x[v], e);
#pragma unroll
        for(int q = 0; q < PER; ++q)
        {
          const int el = (v0 + v) * PER + q;  // compile-time: element el = coefficient el/3, channel el%3
          if(el < 45)
            acc[el % 3] += (el < nEl) ? bs[el / 3] * e[q] : 0.0f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  r += acc[0];
  g += acc[1];
  b += acc[2];
}

}  // namespace mgs
