// kernels_common.h — wave64 helpers for the gfx950 kernels.  CDNA4 only: 64-lane wavefronts are
// hard-coded on purpose (no 32-wide fallbacks, no portability macros).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "device_types.h"

namespace mgs {

constexpr int kWave = 64;

__device__ __forceinline__ int laneId() { return (int)__lane_id(); }

// bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t lanesBelow(uint64_t mask)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v)
{
  const int lane = laneId();
#pragma unroll
  for(int o = 1; o < 64; o <<= 1)
  {
    const uint32_t t = __shfl_up(v, o, 64);
    if(lane >= o)
      v += t;
  }
  return v;
}

// Exclusive scan of one value per thread over a 256-thread block.  `s_tmp` needs 4 entries.
// Returns the exclusive prefix; *total receives the block sum.
__device__ __forceinline__ uint32_t blockExclusiveScan256(uint32_t v, uint32_t* s_tmp, uint32_t* total)
{
  const int      lane = laneId();
  const int      w    = threadIdx.x >> 6;
  const uint32_t inc  = waveInclusiveScan(v);
  if(lane == 63)
    s_tmp[w] = inc;
  __syncthreads();
  const uint32_t w0 = s_tmp[0], w1 = s_tmp[1], w2 = s_tmp[2], w3 = s_tmp[3];
  uint32_t       base = 0;
  if(w > 0) base += w0;
  if(w > 1) base += w1;
  if(w > 2) base += w2;
  *total = w0 + w1 + w2 + w3;
  __syncthreads();
  return base + inc - v;
}

// The bin rectangle of a splat rides through the frame's key sort in the bits of the id word that the ids do not need, so that
// the binning stage finds it in sorted order instead of gathering 4.2 M rectangles by id (35 us of random 4-byte loads).
// Most splats cover one bin or two: the code enumerates (shape, x0, y0) for the shapes 1x1, 2x1, 1x2, 2x2 — as many of them
// (`shapes`, in that order) as fit the spare bits —; every other rectangle is the escape code (all ones), looked up by id.
//   code = base[shape] + y0 * (binsX - dx) + x0,  base = 0, nb, nb + (bx-1) by, nb + (bx-1) by + bx (by-1)
// 1080p (15 x 9 bins, 5.83 M splats = 23 id bits): 493 codes in 9 bits.
__device__ __forceinline__ uint32_t rideBase(int shape, int binsX, int binsY)
{
  const int nb = binsX * binsY;
  return (uint32_t)(shape == 0 ? 0 : (shape == 1 ? nb : (shape == 2 ? nb + (binsX - 1) * binsY : nb + (binsX - 1) * binsY + binsX * (binsY - 1))));
}
__device__ __forceinline__ uint32_t rideEncode(uint32_t rect, int binsX, int binsY, int shapes, uint32_t escape)
{
  const uint32_t x0 = rect & 255u, y0 = (rect >> 8) & 255u, dx = ((rect >> 16) & 255u) - x0, dy = (rect >> 24) - y0;
  const int      shape = (int)(dx | (dy << 1));
  if(dx > 1u || dy > 1u || shape >= shapes)
    return escape;
  return rideBase(shape, binsX, binsY) + y0 * (uint32_t)(binsX - (int)dx) + x0;
}
// branch-free, no division (round 5, third session: hipcc expands __fdividef to the full IEEE sequence and the shape selection
// to exec-mask branches — 36 vector instructions per decode, four decodes per wave in k_dbin_count and in k_dbin_emit, both bound
// by VALU issue): the thresholds are nested, so the base and the shape follow from three compares; y0 = r / wS by a 16-bit
// reciprocal, exact for r < 1024 and wS <= 32 (error r / 65536 < 1 / wS; tests/test_ride_code.py walks every grid).  The
// wave-uniform part is made once per kernel (RideDecoder).
struct RideDecoder
{
  uint32_t b1, b2, b3, w0, w1, i0, i1;
};
__device__ __forceinline__ RideDecoder rideDecoder(int binsX, int binsY)
{
  RideDecoder D;
  D.b1 = rideBase(1, binsX, binsY);
  D.b2 = rideBase(2, binsX, binsY);
  D.b3 = rideBase(3, binsX, binsY);
  D.w0 = (uint32_t)binsX;
  D.w1 = (uint32_t)max(binsX - 1, 1);
  D.i0 = (65536u + D.w0 - 1u) / D.w0;
  D.i1 = (65536u + D.w1 - 1u) / D.w1;
  return D;
}
__device__ __forceinline__ uint32_t rideDecode(uint32_t code, const RideDecoder& D)
{
  const bool     c1 = code >= D.b1, c2 = code >= D.b2, c3 = code >= D.b3;
  const uint32_t r  = code - (c3 ? D.b3 : (c2 ? D.b2 : (c1 ? D.b1 : 0u)));
  const uint32_t dy = c2 ? 1u : 0u, dx = (c3 || (c1 && !c2)) ? 1u : 0u;
  const uint32_t wS = dx ? D.w1 : D.w0;
  const uint32_t y0 = (r * (dx ? D.i1 : D.i0)) >> 16;
  const uint32_t x0 = r - y0 * wS;
  return x0 | (y0 << 8) | ((x0 + dx) << 16) | ((y0 + dy) << 24);
}

// The codes of a thread's (up to) eight splats, collected over the rounds of a project kernel's front end and consumed by its
// hand-over to the sort: a 128-bit shift register in four VGPRs.  Every round shifts it by 16 bits and inserts that round's
// code, whether the round did anything or not, so after exactly eight rounds round r sits in slot 7 - r: static indexing on
// both sides, although the front end's loop is rolled.
struct RideCodes
{
  uint32_t c[4] = {0u, 0u, 0u, 0u};
  __device__ __forceinline__ void push(uint32_t code)
  {
    c[3] = (c[3] << 16) | (c[2] >> 16);
    c[2] = (c[2] << 16) | (c[1] >> 16);
    c[1] = (c[1] << 16) | (c[0] >> 16);
    c[0] = (c[0] << 16) | (code & 0xFFFFu);
  }
  __device__ __forceinline__ uint32_t get(int round) const { return (c[(7 - round) >> 1] >> (16 * ((7 - round) & 1))) & 0xFFFFu; }
};

// encodeMinMaxFp32 (shaders/dist.comp.slang:33-38): order-preserving fp32 -> u32
__device__ __forceinline__ uint32_t encodeKey(float v)
{
  uint32_t bits = __float_as_uint(v);
  bits ^= (uint32_t)((int32_t)bits >> 31) | 0x80000000u;
  return bits;
}

// out = M * v with the exact unfused evaluation order of the oracle / host:
// ((x*c0 + y*c1) + z*c2) + w*c3, every product and sum rounded separately.  The pragma only
// affects operators written lexically inside the block (intrinsics like __fmul_rn are inlined with
// the translation unit's default contract(fast) and WOULD be fused), hence plain * and +.
__device__ __forceinline__ void mulMat4Exact(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma clang fp contract(off)
#pragma unroll
  for(int r = 0; r < 4; ++r)
  {
    const float a = x * m[r];
    const float b = y * m[4 + r];
    const float c = z * m[8 + r];
    const float d = w * m[12 + r];
    out[r]        = ((a + b) + c) + d;
  }
}

// Shortcuts of mulMat4Exact that are BIT-IDENTICAL to it under conditions the host and the partition test establish (round 5;
// a wave64 VALU instruction holds its SIMD for four cycles, tools/micro/valu_rate.hip, and phase 1 of the project kernel is a
// quarter of its instructions):
//  * mulMat4ExactAffineW1: the matrix's last row is (+0, +0, +0, 1) and w == 1.0f exactly.  Then 1 * m[12 + r] == m[12 + r] (the
//    product is exact) and out[3] == ((x*0 + y*0) + z*0) + 1 == 1.0f for finite x, y, z (a sum of zeros of either sign plus one).
//  * mulPerspExactW1: P has the perspective pattern P[1] = P[2] = P[3] = P[4] = P[6] = P[7] = P[12] = P[13] = P[15] = +0,
//    P[14] != 0, and w == 1.0f.  Every dropped product is a zero of either sign; adding such a zero to a nonzero term returns
//    the term, and where all terms of a row vanish the full evaluation ends in "+ (1 * +0)", which yields +0 — the explicit
//    "+ 0.0f" below does the same (it is not an identity: -0 + 0 = +0).  Row 2 ends in "+ P[14]", which absorbs any zero.
//    Finite inputs only (Inf * 0 would be NaN in the full product): FrameConst::perspAffine / modelIsAffine bound the
//    matrices' entries by 2^24 and the partition test (bit 3) the coordinates by 2^40, so that every intermediate of the chain stays
//    below 2^118 (round 6, ADVICE r5: the earlier bounds 2^40 / 2^60 allowed 2^142).
__device__ __forceinline__ void mulMat4ExactAffineW1(const float* m, float x, float y, float z, float out[3])
{
#pragma clang fp contract(off)
#pragma unroll
  for(int r = 0; r < 3; ++r)
  {
    const float a = x * m[r];
    const float b = y * m[4 + r];
    const float c = z * m[8 + r];
    out[r]        = ((a + b) + c) + m[12 + r];
  }
}
__device__ __forceinline__ void mulPerspExactW1XY(const float* P, float x, float y, float z, float& ox, float& oy)
{
#pragma clang fp contract(off)
  {
    const float a = x * P[0], c = z * P[8];
    ox            = (a + c) + 0.0f;
  }
  {
    const float b = y * P[5], c = z * P[9];
    oy            = (b + c) + 0.0f;
  }
}
__device__ __forceinline__ void mulPerspExactW1ZW(const float* P, float z, float& oz, float& ow)
{
#pragma clang fp contract(off)
  {
    const float c = z * P[10];
    oz            = c + P[14];
  }
  {
    const float c = z * P[11];
    ow            = c + 0.0f;
  }
}

// rows 2 and 3 of mulMat4Exact only (clip z and w): what the depth key needs when the frustum test is known to pass
__device__ __forceinline__ void mulMat4ExactZW(const float* m, float x, float y, float z, float w, float& oz, float& ow)
{
#pragma clang fp contract(off)
  {
    const float a = x * m[2], b = y * m[6], c = z * m[10], d = w * m[14];
    oz            = ((a + b) + c) + d;
  }
  {
    const float a = x * m[3], b = y * m[7], c = z * m[11], d = w * m[15];
    ow            = ((a + b) + c) + d;
  }
}

// size culling of dist.comp.slang:93-134 with the oracle's exact unfused operation order
__device__ __forceinline__ bool sizeCulled(float maxExpScale, float splatScale, float axisMax, float viewZ, float maxFocal,
                                           float minPixels)
{
#pragma clang fp contract(off)
  const float radius = maxExpScale * splatScale;
  float       extent = radius * 2.8284271247f * 2.0f;
  extent             = extent * axisMax;
  const float viewDist = fabsf(viewZ);
  if(viewDist > 0.0001f)
  {
    const float projectedPixels = (extent * maxFocal) / viewDist;
    return projectedPixels < minPixels;
  }
  return false;
}

// Ordered compaction of up to 8 rounds x 256 flags without a barrier per round: every wave posts its
// per-round popcounts, ONE barrier, 32 lanes scan the 8x4 table, second barrier, then every thread knows
// the base of its (round, wave).  Returns the total; bases land in s_base[round*4 + wave].
__device__ __forceinline__ uint32_t scanRoundWaveCounts(uint32_t* s_cnt /*32*/, uint32_t* s_base /*33*/)
{
  __syncthreads();
  if(threadIdx.x < 64)
  {
    const uint32_t v   = threadIdx.x < 32 ? s_cnt[threadIdx.x] : 0u;
    const uint32_t inc = waveInclusiveScan(v);
    if(threadIdx.x < 32)
      s_base[threadIdx.x] = inc - v;
    if(threadIdx.x == 31)
      s_base[32] = inc;
  }
  __syncthreads();
  return s_base[32];
}

// IEEE-correct fp32 division (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
__device__ __forceinline__ float divExact(float a, float b)
{
#pragma clang fp contract(off)
  return a / b;
}

// ---- fisheye dist-stage cull (CAMERA_TYPE == CAMERA_FISHEYE) -------------------------------------------------------
// shaders/dist.comp.slang:75-90 culls with projectPointFisheye (threedgut_camera_projections.h.slang:149-171) of the perfect
// fisheye model (initPerfectFisheyeCamera, threedgut_camera_models.h.slang:120-136: zero radial coefficients, principal point
// at the viewport centre, maxAngle of computeMaxAngle) on (1,1,-1) * viewPos, margin GUT_IN_IMAGE_MARGIN_FACTOR = 0.1, and
// then the z test of the pinhole branch.  The decision selects the sorted set, so it must equal the oracle's bit for bit:
// the arithmetic is unfused, divisions and roots are IEEE, and atan2 — whose result is implementation-defined in the
// reference (SPIR-V Atan2) — is the fixed polynomial below, restated identically in oracle/mgs_oracle.cpp.
// Non-finite positions are culled (every comparison on a NaN is written so that it fails the validity test).
__device__ __forceinline__ float atan2Det(float y /* > 0 */, float x)
{
#pragma clang fp contract(off)
  const float ax = fabsf(x);
  const float t  = y / ax;  // [0, +inf]
  float base = 0.0f, u = t;
  if(t > 2.414213562373095f)
  {
    base = 1.5707963267948966f;
    u    = -1.0f / t;
  }
  else if(t > 0.4142135623730950f)
  {
    base = 0.7853981633974483f;
    u    = (t - 1.0f) / (t + 1.0f);
  }
  const float z = u * u;
  const float p = ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z) * u + u;
  const float a = base + p;
  return (x < 0.0f) ? (3.14159274101257f - a) : a;
}

__device__ __forceinline__ bool fisheyeCullValid(const FrameConst& F, float vx, float vy, float vz)
{
#pragma clang fp contract(off)
  const float px = vx, py = vy, pz = -vz;  // float3(1,1,-1) * viewPos.xyz, dist.comp.slang:83
  // stableNorm2, threedgut_camera_projections.h.slang:32-44
  const float ax = fabsf(px), ay = fabsf(py);
  const float mn = (ax < ay) ? ax : ay, mx = (ax < ay) ? ay : ax;
  float       nrm = 0.0f;
  if(mx > 0.0f)
  {
    const float r = mn / mx;
    nrm           = mx * sqrtf(1.0f + r * r);  // IEEE (hipcc default: correctly rounded divide / sqrt)
  }
  const float rho       = (nrm > 1e-7f) ? nrm : 1e-7f;                              // max(., eps)            :152
  const float thetaFull = atan2Det(rho, pz);                                        //                        :153
  const float theta     = (thetaFull < F.gutMaxAngle) ? thetaFull : F.gutMaxAngle;  // min(., maxAngle)       :160
  const float theta2    = theta * theta;
  const float delta     = (theta * (0.0f * theta2 + 1.0f)) / rho;                   // radial coefficients 0  :164
  const float resx = (float)F.width, resy = (float)F.height;
  const float ox = (F.gutFocal[0] * px) * delta + resx / 2.0f;                       //                        :165
  const float oy = (F.gutFocal[1] * py) * delta + resy / 2.0f;
  const float tx = resx * 0.1f, ty = resy * 0.1f;                                    // withinResolution       :78-83
  return (theta < F.gutMaxAngle) && (ox > -tx) && (oy > -ty) && (ox < resx + tx) && (oy < resy + ty);
}

// the dist-stage frustum test (dist.comp.slang:64-91) on ndc = clip / w and the view-space position
__device__ __forceinline__ bool distStageCulled(const FrameConst& F, float nx, float ny, float nz, float vx, float vy, float vz)
{
  if(F.cameraModel == 1)
  {
    if(!fisheyeCullValid(F, vx, vy, vz))
      return true;
    return nz < 0.f - F.frustumDilation || nz > 1.0f;  // :88
  }
  const float c = 1.0f + F.frustumDilation;  // :71-73 (NaN compares false everywhere, as in the shader)
  return fabsf(nx) > c || fabsf(ny) > c || nz < 0.f - F.frustumDilation || nz > 1.0f;
}

// ---- random numbers of the stochastic paths --------------------------------------------------------------------
// The reference takes xxhash32 / pcg / rand from nvshaders/random.h.slang of nvpro_core2, a dependency that is not in
// the reference tree (CMake fetches it).  Restated from the published file: xxhash32 over a uint3 (Jarzynski & Olano,
// "Hash Functions for GPU Rendering", the shadertoy XlGcRh variant), the PCG output function (pcg-random.org, RXS-M-XS
// 32), and rand() = the top 23 bits of pcg as the mantissa of a float in [1,2) minus 1.  Unpinned: there are no vectors
// of that file here; the oracle restates the same three functions and the tests check oracle == device bit for bit.
__device__ __host__ __forceinline__ uint32_t rngXxhash32(uint32_t px, uint32_t py, uint32_t pz)
{
  const uint32_t p0 = 2246822519u, p1 = 3266489917u, p2 = 668265263u, p3 = 374761393u;
  uint32_t       h  = pz + p3 + px * p1;
  h                 = p2 * ((h << 17) | (h >> 15));
  h += py * p1;
  h = p2 * ((h << 17) | (h >> 15));
  h = p0 * (h ^ (h >> 15));
  h = p1 * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
__device__ __host__ __forceinline__ uint32_t rngPcg(uint32_t& state)
{
  const uint32_t prev = state * 747796405u + 2891336453u;
  const uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state               = prev;
  return (word >> 22u) ^ word;
}
__device__ __forceinline__ float rngRand(uint32_t& seed)
{
  const uint32_t r = rngPcg(seed);
  return __uint_as_float(0x3f800000u | (r >> 9)) - 1.0f;
}

}  // namespace mgs
