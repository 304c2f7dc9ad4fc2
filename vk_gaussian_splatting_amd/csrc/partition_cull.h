// partition_cull.h — the per-partition test of the project kernels (k_project.hip, k_gut.hip): can any splat of this workgroup's
// 2048-splat partition survive the dist-stage cull / reach this device's strip?
//
// Storage order is Morton order, so a partition is a compact cell of space; its 8 AABB corners go through the same P*V*M.  The
// per-splat test of dist.comp.slang (:71-73) culls on ndc = clip/w; for points with w > 0 every one of its conditions is LINEAR
// in the point (x > c*w, -x > c*w, ..., z > w, z < -d*w), so if all 8 corners satisfy the same condition every centre inside
// the box does too.  Margins make the test conservative against fp32 rounding: a partition is skipped only if every splat in it
// would certainly be culled.  For strips the footprint of a splat of the partition is bounded by
// R = s*(k*fmax*S*rmax/zmin + 3.2) + 1 px (derivation in DESIGN.md §3.1), and the partition is skipped when [ymin-R, ymax+R]
// misses the strip's pixel rows.
//
// Every wave of the workgroup evaluates it for itself: lane l takes corner l & 7 (eight copies per wave), the "all corners"
// conditions are wave votes, the extrema three xor-shuffles — ~100 vector instructions per wave against ~2 800 for the partition's
// projection, no LDS, no barrier, and no kernel of its own in front of the frame (rounds 1-3 ran it as one thread per partition:
// 8.6 us of launch and latency for 2 847 threads of work).
#pragma once
#include "kernels_common.h"

namespace mgs {

__device__ __forceinline__ float oct8Min(float v)
{
  v = fminf(v, __shfl_xor(v, 1, 64));
  v = fminf(v, __shfl_xor(v, 2, 64));
  return fminf(v, __shfl_xor(v, 4, 64));
}
__device__ __forceinline__ float oct8Max(float v)
{
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  return fmaxf(v, __shfl_xor(v, 4, 64));
}

// The partition's box, fetched FIRST in the kernel: loads return in order, so issued ahead of the 24 centre loads of phase 1
// the decision is ready while those are still in flight.
struct PartitionBox
{
  float x, y, z;     // this lane's corner
  float lo[3], hi[3], rmax, bad;
};
__device__ __forceinline__ PartitionBox partitionLoad(const InstanceConst& I, uint32_t localPart)
{
  const float* bx = I.partBox + 8 * (size_t)localPart;
  const int    q  = (int)(threadIdx.x & 7u);
  PartitionBox b;
  b.x = bx[(q & 1) ? 3 : 0];
  b.y = bx[(q & 2) ? 4 : 1];
  b.z = bx[(q & 4) ? 5 : 2];
#pragma unroll
  for(int a = 0; a < 3; ++a)
  {
    b.lo[a] = bx[a];
    b.hi[a] = bx[3 + a];
  }
  b.rmax = bx[6];
  b.bad  = bx[7];
  return b;
}

// returns the partition's flags: bit 0 skip, bit 1 every centre passes the frustum test, bit 2 all centres finite, bit 3 all
// coordinates below 2^40 in magnitude (with bit 2 and every entry of M, V, P below 2^24 — modelIsAffine / perspAffine — the largest
// intermediate of P*(V*(M*p)) stays below 2^40 * (4 * 2^24)^3 = 2^118: nothing overflows, so the full evaluation never meets Inf * 0); R: footprint
// bound of the partition's splats in pixels (strips only; 3.0e38 = unknown).  Wave-uniform result; call with all lanes active.
__device__ __forceinline__ uint32_t partitionTest(const FrameArgs& A, const InstanceConst& I, const PartitionBox& B, float& Rout)
{
  uint32_t     skip = 0, inside = 0;
  const float    mag    = fmaxf(fmaxf(fmaxf(fabsf(B.lo[0]), fabsf(B.lo[1])), fmaxf(fabsf(B.lo[2]), fabsf(B.hi[0]))), fmaxf(fabsf(B.hi[1]), fabsf(B.hi[2])));
  const uint32_t finite = (B.bad == 0.0f) ? (mag < 1.099511627776e12f ? 12u : 4u) : 0u;
  Rout = 3.0e38f;
  if(B.bad != 0.0f)
    return 0u;  // non-finite data: never cull this partition
  const float c = 1.0f + A.f.frustumDilation, dl = A.f.frustumDilation;
  const float m = 1.0e-3f;  // relative safety margin
  const float x = B.x, y = B.y, z = B.z;
  const float* MV = I.modelView;
  const float  tx = MV[0] * x + MV[4] * y + MV[8] * z + MV[12];
  const float  ty = MV[1] * x + MV[5] * y + MV[9] * z + MV[13];
  const float  tz = MV[2] * x + MV[6] * y + MV[10] * z + MV[14];
  const float  tw = MV[3] * x + MV[7] * y + MV[11] * z + MV[15];
  const float* P  = A.f.proj;
  const float  cx = P[0] * tx + P[4] * ty + P[8] * tz + P[12] * tw;
  const float  cy = P[1] * tx + P[5] * ty + P[9] * tz + P[13] * tw;
  const float  cz = P[2] * tx + P[6] * ty + P[10] * tz + P[14] * tw;
  const float  cw = P[3] * tx + P[7] * ty + P[11] * tz + P[15] * tw;
  const float  aw = fabsf(cw), tol = m * (aw + fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-6f;
  const bool allWpos = __all(cw > tol);
  const bool outXp = __all(cx > c * cw + tol), outXn = __all(-cx > c * cw + tol);
  const bool outYp = __all(cy > c * cw + tol), outYn = __all(-cy > c * cw + tol);
  const bool outZf = __all(cz > cw + tol), outZn = __all(cz < -dl * cw - tol);
  // every corner passes the per-splat test with margin -> so does every centre inside the box
  const bool allIn = __all((cw > tol) && (fabsf(cx) < c * cw - tol) && (fabsf(cy) < c * cw - tol) && (cz > -dl * cw + tol) && (cz < cw - tol));
  const float yp    = (cy / cw + 1.0f) * 0.5f * (float)A.f.height;
  const float ymin  = oct8Min(yp), ymax = oct8Max(yp);
  const float zvmin = oct8Min(-tz);  // view depth (camera looks down -z)
  const float rx = oct8Max(fabsf(tx / tz)), ry = oct8Max(fabsf(ty / tz));
  // CAMERA_FISHEYE (dist.comp.slang:75-90): the x/y box is replaced by the fisheye validity test (cone of maxAngle around the
  // view axis + the image rectangle); the z test stays.  Conservative form: the z conditions as above, and the bounding
  // sphere of the box against the cone.  "Every centre passes" is never claimed for fisheye frames.
  const bool fisheye = A.f.cameraModel == 1;
  if(allWpos && (outZf || outZn || (!fisheye && (outXp || outXn || outYp || outYn))))
    skip = 1;
  if(fisheye && !skip)
  {
    const float  mx = 0.5f * (B.lo[0] + B.hi[0]), my = 0.5f * (B.lo[1] + B.hi[1]), mz = 0.5f * (B.lo[2] + B.hi[2]);
    const float  hx = 0.5f * (B.hi[0] - B.lo[0]), hy = 0.5f * (B.hi[1] - B.lo[1]), hz = 0.5f * (B.hi[2] - B.lo[2]);
    const float  sx = MV[0] * mx + MV[4] * my + MV[8] * mz + MV[12];
    const float  sy = MV[1] * mx + MV[5] * my + MV[9] * mz + MV[13];
    const float  sz = MV[2] * mx + MV[6] * my + MV[10] * mz + MV[14];
    const float  rad  = sqrtf(hx * hx + hy * hy + hz * hz) * I.modelScale * 1.001f + 1e-6f;
    const float  dist = sqrtf(sx * sx + sy * sy + sz * sz);
    if(dist > rad * 1.001f)
    {
      const float thetaC = atan2f(sqrtf(sx * sx + sy * sy), -sz);
      if(thetaC - asinf(rad / dist) > A.f.gutMaxAngle * 1.001f + 1e-3f)
        skip = 1;
    }
  }
  inside = (allIn && !skip && !fisheye) ? 2u : 0u;
  // the strip bound R is the 3DGS (pinhole EWA) footprint: not valid for a 3DGUT fisheye frame
  const bool strip = (A.f.stripRow1 - A.f.stripRow0) < A.f.tilesY && !(fisheye && A.f.pipeline == 1);
  if(!skip && strip && allWpos && zvmin > 1e-4f)
  {
    const float S    = I.modelScale;  // largest singular value of the model 3x3 (host, per frame)
    const float fmx  = fmaxf(fabsf(A.f.focal[0]), fabsf(A.f.focal[1]));
    const float kk   = sqrtf(2.0f + rx * rx + ry * ry);
    float       R    = A.f.splatScale * (kk * fmx * S * B.rmax / zvmin + 3.2f);
    R                = fminf(R, 2897.0f * A.f.splatScale) * 1.01f + 2.0f;  // both bases are clamped at 2048 px
    const float y0   = (float)(A.f.stripRow0 * kTilePx), y1 = (float)(min(A.f.stripRow1 * kTilePx, A.f.height));
    if(ymax + R < y0 || ymin - R > y1)
      skip = 1;
    Rout = R;
  }
  return skip | inside | finite;
}

}  // namespace mgs
