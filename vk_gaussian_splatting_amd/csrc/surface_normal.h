// surface_normal.h — the per-splat world normal of NEED_SURFACE_INFO, shared by the 3DGS (k_raster.hip) and 3DGUT (k_gut.hip)
// compositors.
#pragma once
#include "kernels_common.h"

namespace mgs {

// World-space normal of one splat as the mesh shader emits it with NEED_SURFACE_INFO (threedgs_raster.mesh.slang:209-235):
// particle = (centre, exp(scale), normalised quaternion) (threedgrt.h.slang:42-48); max-density-plane normal
// n = Sigma^-1 (camera - centre) with the thin-particle cases (threedgrt.h.slang:358-419); to world space by the
// instance's 3x3; optionally through the 2x16-bit octahedral code (octahedral_normal.h.slang:27-87).
__device__ inline float4 splatWorldNormal(const FrameConst& F, const InstanceConst& I, uint32_t li, bool quantize)
{
  const float  px = I.centers[3 * (size_t)li], py = I.centers[3 * (size_t)li + 1], pz = I.centers[3 * (size_t)li + 2];
  const float  s0 = expf(I.scales[3 * (size_t)li]), s1 = expf(I.scales[3 * (size_t)li + 1]),
              s2 = expf(I.scales[3 * (size_t)li + 2]);
  const float4 rq = *reinterpret_cast<const float4*>(I.rotations + 4 * (size_t)li);  // (w,x,y,z)
  const float  ql = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
  const float  w = rq.x / ql, x = rq.y / ql, y = rq.z / ql, z = rq.w / ql;
  const float  xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  // quatToMat3Transpose (quaternions.h.slang:56-73), rows as written; mul(v, M) = row vector times matrix
  const float m[9] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy),
                      2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx),
                      2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy)};
  const float l0 = I.camModel[0] - px, l1 = I.camModel[1] - py, l2 = I.camModel[2] - pz;  // modelRayOrigin - position
  const float th = F.thinParticleThreshold;
  const bool  t0 = s0 < th, t1 = s1 < th, t2 = s2 < th;
  const int   smallCount = (t0 ? 1 : 0) + (t1 ? 1 : 0) + (t2 ? 1 : 0);
  float       n0, n1, n2;
  if(smallCount == 0)
  {
    const float c0 = (l0 * m[0] + l1 * m[3] + l2 * m[6]) * (1.0f / (s0 * s0));
    const float c1 = (l0 * m[1] + l1 * m[4] + l2 * m[7]) * (1.0f / (s1 * s1));
    const float c2 = (l0 * m[2] + l1 * m[5] + l2 * m[8]) * (1.0f / (s2 * s2));
    const float g0 = c0 * m[0] + c1 * m[1] + c2 * m[2];
    const float g1 = c0 * m[3] + c1 * m[4] + c2 * m[5];
    const float g2 = c0 * m[6] + c1 * m[7] + c2 * m[8];
    const float rl = 1.0f / sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
    n0 = g0 * rl; n1 = g1 * rl; n2 = g2 * rl;
    if(n0 * l0 + n1 * l1 + n2 * l2 < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
  }
  else if(smallCount == 1)
  {
    const int a = t0 ? 0 : (t1 ? 1 : 2);
    n0 = m[a]; n1 = m[3 + a]; n2 = m[6 + a];
    if(n0 * l0 + n1 * l1 + n2 * l2 < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
  }
  else
  {
    const float l = sqrtf(l0 * l0 + l1 * l1 + l2 * l2);
    n0 = l0 / l; n1 = l1 / l; n2 = l2 / l;
  }
  const float* M = I.model;  // glm column-major
  float        w0 = M[0] * n0 + M[4] * n1 + M[8] * n2;
  float        w1 = M[1] * n0 + M[5] * n1 + M[9] * n2;
  float        w2 = M[2] * n0 + M[6] * n1 + M[10] * n2;
  const float  wl = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
  w0 /= wl; w1 /= wl; w2 /= wl;
  if(quantize)
  {
    const float inv = 1.0f / (fabsf(w0) + fabsf(w1) + fabsf(w2));
    float       e0 = w0 * inv, e1 = w1 * inv;
    if(w2 < 0.0f)
    {
      const float a0 = (1.0f - fabsf(e1)) * (e0 >= 0.0f ? 1.0f : -1.0f);
      const float a1 = (1.0f - fabsf(e0)) * (e1 >= 0.0f ? 1.0f : -1.0f);
      e0 = a0; e1 = a1;
    }
    const uint32_t qx = (uint32_t)fminf(fmaxf((e0 * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
    const uint32_t qy = (uint32_t)fminf(fmaxf((e1 * 0.5f + 0.5f) * 65535.0f, 0.0f), 65535.0f);
    const float    fx = (float)qx / 65535.0f * 2.0f - 1.0f, fy = (float)qy / 65535.0f * 2.0f - 1.0f;
    float          d0 = fx, d1 = fy, d2 = 1.0f - fabsf(fx) - fabsf(fy);
    if(d2 < 0.0f)
    {
      const float a0 = (1.0f - fabsf(d1)) * (d0 >= 0.0f ? 1.0f : -1.0f);
      const float a1 = (1.0f - fabsf(d0)) * (d1 >= 0.0f ? 1.0f : -1.0f);
      d0 = a0; d1 = a1;
    }
    const float dl = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    w0 = d0 / dl; w1 = d1 / dl; w2 = d2 / dl;
  }
  // .w = 1: two or more axes are degenerate and the normal is minus the RAY direction — the 3DGS mesh shader passes the
  // direction to the splat centre (what was computed above), the 3DGUT fragment shader its pixel's ray (k_gut.hip)
  return make_float4(w0, w1, w2, smallCount >= 2 ? 1.0f : 0.0f);
}


// NORMAL_METHOD_ISO_SURFACE (3DGUT fragment, threedgrt.h.slang:330-345 -> computeEllipsoidNormal :423-497): what the compositor keeps
// per record.  The normal of a regular particle depends on the pixel's ray: with h the point where the canonical ray enters the
// sphere of radius 3 (raySphereIntersection :502-540), normalModel = normalize(mul(normalize(h) / scale, rotMat)) and
// normalWorld = normalize(mul(normalModel, modelToWorldRS)) — one 3x3 N = W R S^-1 applied to h, normalised once.
//   out[0..2].xyz = rows of N;  out[0].w = 0
//   one degenerate axis (scale < max(0.02 maxScale, thinParticleThreshold)): out[0].xyz = the world normal, out[0].w = 1
//   two or three: out[0].w = 2 (minus the pixel's ray)
__device__ inline void splatIsoNormalRec(const FrameConst& F, const InstanceConst& I, uint32_t li, float4 out[3])
{
  const float  px = I.centers[3 * (size_t)li], py = I.centers[3 * (size_t)li + 1], pz = I.centers[3 * (size_t)li + 2];
  const float  s0 = expf(I.scales[3 * (size_t)li]), s1 = expf(I.scales[3 * (size_t)li + 1]),
              s2 = expf(I.scales[3 * (size_t)li + 2]);
  const float4 rq = *reinterpret_cast<const float4*>(I.rotations + 4 * (size_t)li);  // (w,x,y,z)
  const float  ql = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
  const float  w = rq.x / ql, x = rq.y / ql, y = rq.z / ql, z = rq.w / ql;
  const float  xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  const float  m[9] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy),
                       2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx),
                       2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy)};
  const float  th = fmaxf(0.02f * fmaxf(fmaxf(s0, s1), s2), F.thinParticleThreshold);
  const bool   t0 = s0 < th, t1 = s1 < th, t2 = s2 < th;
  const int    smallCount = (t0 ? 1 : 0) + (t1 ? 1 : 0) + (t2 ? 1 : 0);
  const float* M = I.model;  // glm column-major
  if(smallCount == 0)
  {
    const float is[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
    float       N[9];
#pragma unroll
    for(int j = 0; j < 3; ++j)
#pragma unroll
      for(int a = 0; a < 3; ++a)
        N[3 * j + a] = (M[j] * m[a] + M[4 + j] * m[3 + a] + M[8 + j] * m[6 + a]) * is[a];
    out[0] = make_float4(N[0], N[1], N[2], 0.0f);
    out[1] = make_float4(N[3], N[4], N[5], 0.0f);
    out[2] = make_float4(N[6], N[7], N[8], 0.0f);
  }
  else if(smallCount == 1)
  {
    const int   a  = t0 ? 0 : (t1 ? 1 : 2);
    float       n0 = m[a], n1 = m[3 + a], n2 = m[6 + a];
    const float l0 = I.camModel[0] - px, l1 = I.camModel[1] - py, l2 = I.camModel[2] - pz;
    if(n0 * l0 + n1 * l1 + n2 * l2 < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }
    float       w0 = M[0] * n0 + M[4] * n1 + M[8] * n2;
    float       w1 = M[1] * n0 + M[5] * n1 + M[9] * n2;
    float       w2 = M[2] * n0 + M[6] * n1 + M[10] * n2;
    const float wl = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    out[0] = make_float4(w0 / wl, w1 / wl, w2 / wl, 1.0f);
    out[1] = out[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  else
  {
    out[0] = make_float4(0.0f, 0.0f, 0.0f, 2.0f);
    out[1] = out[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

}  // namespace mgs
