// k_gut.hip — the 3DGUT raster pipeline (PIPELINE_MESH_3DGUT) for gfx950: SURVEY.md §8f rank 3.
//
// Replaces
//   shaders/threedgut_raster.mesh.slang:111-254   per-splat front end: colour + SH, alpha cull, unscented projection,
//                                                  quad extent (conic / eigen), quad placement
//   shaders/threedgut.h.slang:26-163              threedgutParticleProjection (7 sigma points, GUT_* of
//                                                  threedgut_definitions.h.slang), threedgutProjectedExtentConicOpacity
//   shaders/threedgut_camera_projections.h.slang:84-201  perfect pinhole / fisheye projection, global shutter
//   shaders/threedgut_raster.frag.slang:87-183    per-fragment: ray generation (cameras.h.slang:27-82), model-space ray,
//                                                  particleProcessHitGut (threedgrt.h.slang:57-135,238-278), blend source
// Shared with the 3DGS path and unchanged: depth keys + frustum cull (dist.comp.slang), the key sort, the per-bin lists.
// Structure: k_project_gut is k_project's phase 1 (key, cull, ordered compaction: same code, same bits) followed by the
// 3DGUT front end for the survivors; it writes one 96-byte GutRec per sorted splat.  k_composite_gut walks the bin
// lists nearest-first like k_composite (one workgroup per 16x16 tile, one pixel per thread, records staged through
// LDS in list order) and evaluates the particle response per fragment.  This pipeline is a "next" row: correct and
// reasonably fast, not tuned like the 3DGS compositor.
// Depth of field and stochastic splats are the XT variant of the compositor (random numbers: kernels_common.h).
// Kernel degrees other than 2 and the surface side outputs run in the XT variant too.  Not built (stated in DESIGN.md):
// rolling shutter (untested in the reference).
#include <cstdlib>
#include "kernels_common.h"
#include "sh_eval.h"
#include "surface_normal.h"
#include "sort_plan.h"
#include "slot_emit.h"
#include "partition_cull.h"

namespace mgs {

constexpr int kGutThreads = 256;
constexpr int kGutItems   = 8;
constexpr int kGutPart    = kGutThreads * kGutItems;  // == the project kernel's partition: same slots, same sort input

// This pipeline's arithmetic has a tolerance (>= 50 dB vs the oracle), no bit-exact part except the keys of phase 1: reciprocals
// and square roots are the 1-ulp hardware instructions, not hipcc's correctly rounded expansions (10 instructions each; the
// front end had ~30 divisions per splat, the compositor one per fragment).
__device__ __forceinline__ float gRcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float gSqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// threedgut_camera_projections.h.slang:32-44
__device__ __forceinline__ float gutStableNorm2(float x, float y)
{
  const float ax = fabsf(x), ay = fabsf(y);
  const float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
  if(mx <= 0.0f)
    return 0.0f;
  const float r = mn * gRcp(mx);
  return mx * gSqrt(1.0f + r * r);
}

// projectPointWithShutter (global shutter) + projectPoint for the perfect pinhole / fisheye models.  `cam` is the
// view-space point with z negated (RUB -> RUF: the flips of :188-196 amount to F (R p + t), F = diag(1,1,-1)).
__device__ __forceinline__ bool gutProjectCam(const FrameConst& F, float cx, float cy, float cz, float& ox, float& oy)
{
  const float resx = (float)F.width, resy = (float)F.height;
  bool        ok;
  if(F.cameraModel == 1)
  {  // projectPointFisheye, radial coefficients 0: :151-176
    const float rho       = fmaxf(gutStableNorm2(cx, cy), 1e-7f);
    const float thetaFull = atan2f(rho, cz);
    const float theta     = fminf(thetaFull, F.gutMaxAngle);
    const float delta     = theta * gRcp(rho);
    ox                    = F.gutFocal[0] * cx * delta + resx * 0.5f;
    oy                    = F.gutFocal[1] * cy * delta + resy * 0.5f;
    ok                    = theta < F.gutMaxAngle;
  }
  else
  {  // projectPointPinhole, distortion coefficients 0 (icD = 1, delta = 0): :85-137
    if(cz <= 0.0f)
    {
      ox = oy = 0.0f;
      return false;
    }
    const float rcz = gRcp(cz);
    ox = (cx * rcz) * F.gutFocal[0] + resx * 0.5f;
    oy = (cy * rcz) * F.gutFocal[1] + resy * 0.5f;
    ok = true;
  }
  const float mx = resx * 0.1f, my = resy * 0.1f;  // withinResolution, GUT_IN_IMAGE_MARGIN_FACTOR
  return ok && (ox > -mx) && (oy > -my) && (ox < resx + mx) && (oy < resy + my);
}

// the per-splat 3DGUT front end; returns false when the splat emits no quad
__device__ __forceinline__ bool projectSplatGut(const FrameConst& F, const InstanceConst& I, uint32_t li, GutRec& out, uint32_t& rectOut)
{
  // mesh.slang:116-122.  The colour (base + SH, :142-148) does not influence any decision of the front end: it is
  // evaluated by the compositor for the records it stages (deferred shading, as in the 3DGS path: 0.96 M staged
  // (tile, splat) pairs against 4.1 M sorted splats on the garden-sized frame; the SH records were half of this kernel's time)
  const float  px = I.centers[3 * (size_t)li], py = I.centers[3 * (size_t)li + 1], pz = I.centers[3 * (size_t)li + 2];
  const float  s0 = __expf(I.scales[3 * (size_t)li]), s1 = __expf(I.scales[3 * (size_t)li + 1]), s2 = __expf(I.scales[3 * (size_t)li + 2]);
  const float4 rq = *reinterpret_cast<const float4*>(I.rotations + 4 * (size_t)li);  // (w,x,y,z)
  const float  ql = rsqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
  const float  w = rq.x * ql, x = rq.y * ql, y = rq.z * ql, z = rq.w * ql;
  const float  xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  // quatToMat3 (quaternions.h.slang:39-58): row a = a-th principal axis
  const float R[3][3] = {{1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy)},
                         {2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx)},
                         {2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy)}};
  const float sc[3] = {s0, s1, s2};
  float alpha = I.alpha[li];  // the opacity as fetchColor returns it
  if(alpha < F.alphaCull)     // :150-155
    return false;
  // threedgutParticleProjection, threedgut.h.slang:26-110 (GUT_D 3, alpha 1, beta 2, kappa 0 -> lambda 0, delta sqrt 3)
  // the sigma points p +- delta_a through V*M: the map is affine, so the mean goes through the model-view matrix once and each
  // axis offset through its 3x3 (7 x 18 FMAs -> 9 + 3 x 9 + adds); only the camera model is evaluated per point
  const float* MV = I.modelView;
  const float  vcx = MV[0] * px + MV[4] * py + MV[8] * pz + MV[12];
  const float  vcy = MV[1] * px + MV[5] * py + MV[9] * pz + MV[13];
  const float  vcz = MV[2] * px + MV[6] * py + MV[10] * pz + MV[14];
  float spx[7], spy[7];
  int   nValid = gutProjectCam(F, vcx, vcy, -vcz, spx[0], spy[0]) ? 1 : 0;
  constexpr float kDelta = 1.73205080757f, kWI = 1.0f / 6.0f;
  float ccx = 0.0f, ccy = 0.0f;  // weight of the mean: lambda / (D + lambda) = 0
#pragma unroll
  for(int a = 0; a < 3; ++a)
  {
    const float ex = kDelta * sc[a] * R[a][0], ey = kDelta * sc[a] * R[a][1], ez = kDelta * sc[a] * R[a][2];
    const float dvx = MV[0] * ex + MV[4] * ey + MV[8] * ez;
    const float dvy = MV[1] * ex + MV[5] * ey + MV[9] * ez;
    const float dvz = MV[2] * ex + MV[6] * ey + MV[10] * ez;
    nValid += gutProjectCam(F, vcx + dvx, vcy + dvy, -(vcz + dvz), spx[a + 1], spy[a + 1]) ? 1 : 0;
    ccx += kWI * spx[a + 1];
    ccy += kWI * spy[a + 1];
    nValid += gutProjectCam(F, vcx - dvx, vcy - dvy, -(vcz - dvz), spx[a + 4], spy[a + 4]) ? 1 : 0;
    ccx += kWI * spx[a + 4];
    ccy += kWI * spy[a + 4];
  }
  if(nValid == 0)
    return false;
  float c0, c1, c2;
  {
    const float ex = spx[0] - ccx, ey = spy[0] - ccy;  // weight0 = 0 + (1 - alpha^2 + beta) = 2
    c0 = 2.0f * (ex * ex);
    c1 = 2.0f * (ex * ey);
    c2 = 2.0f * (ey * ey);
  }
#pragma unroll
  for(int a = 1; a < 7; ++a)
  {
    const float ex = spx[a] - ccx, ey = spy[a] - ccy;
    c0 += kWI * (ex * ex);
    c1 += kWI * (ex * ey);
    c2 += kWI * (ey * ey);
  }
  float h1x, h1y, h2x, h2y;
  if(F.extentMethod == 1)
  {  // threedgutProjectedExtentConicOpacity, threedgut.h.slang:113-160
    const float ddx = c0 + 0.3f, ddy = c1, ddz = c2 + 0.3f;
    const float det = ddx * ddz - ddy * ddy;
    if(det == 0.0f)
      return false;
    float wop = alpha;
    if(F.msAA)
      wop = alpha * gSqrt(fmaxf(0.000025f, (c0 * c2 - c1 * c1) * gRcp(det)));
    if(wop < 0.01f)
      return false;
    const float maxPower = __logf(wop * 100.0f);
    const float factor   = fminf(3.33f, gSqrt(2.0f * maxPower));
    const float mid      = 0.5f * (ddx + ddz);
    const float lambda   = mid + gSqrt(fmaxf(0.01f, mid * mid - det));
    const float radius   = factor * gSqrt(lambda);
    if(!(radius > 0.0f))
      return false;
    if(F.msAA)
      alpha = wop;
    h1x = fminf(factor * gSqrt(ddx), radius);
    h1y = 0.0f;
    h2x = 0.0f;
    h2y = fminf(factor * gSqrt(ddz), radius);
  }
  else
  {  // threedgsProjectedExtentBasis(cov, 3.33, splatScale, ...), threedgs.h.slang:60-121
    float a = c0, b = c1, d = c2, detOrig = 0.0f;
    if(F.msAA)
      detOrig = a * d - b * b;
    a += 0.3f;
    d += 0.3f;
    if(F.msAA)
      alpha *= gSqrt(fmaxf(detOrig * gRcp(a * d - b * b), 0.0f));
    const float D = a * d - b * b, half = 0.5f * (a + d);
    const float term2 = gSqrt(fmaxf(0.1f, half * half - D));
    float ev1 = half + term2, ev2 = half - term2;
    if(ev2 <= 0.0f)
      return false;
    if(F.debugFlags & 1)
      ev1 = ev2 = 0.2f;
    float       e1x = (fabsf(b) < 0.001f) ? 1.0f : b, e1y = ev1 - a;
    const float el  = rsqrtf(e1x * e1x + e1y * e1y);
    e1x *= el;
    e1y *= el;
    const float l1 = F.splatScale * fminf(3.33f * gSqrt(ev1), 2048.0f), l2 = F.splatScale * fminf(3.33f * gSqrt(ev2), 2048.0f);
    h1x = e1x * l1;
    h1y = e1y * l1;
    h2x = e1y * l2;
    h2y = -e1x * l2;
  }
  // depth of the quad from the pinhole projection matrix (":205-214": a coarse approximation for fisheye) and the
  // fixed-function clip of a quad emitted at z = ndc.z, w = 1
  const float* P  = F.proj;
  const float  tx = vcx, ty = vcy, tz = vcz;  // the mean in view space, computed above
  const float  tw = MV[3] * px + MV[7] * py + MV[11] * pz + MV[15];
  const float  cz = P[2] * tx + P[6] * ty + P[10] * tz + P[14] * tw;
  const float  cw = P[3] * tx + P[7] * ty + P[11] * tz + P[15] * tw;
  const float  ndcz = cz * gRcp(cw);
  if(!(ndcz >= 0.0f && ndcz <= 1.0f))
    return false;
  const float n1 = h1x * h1x + h1y * h1y, n2 = h2x * h2x + h2y * h2y;
  if(!(n1 > 0.0f && n2 > 0.0f))
    return false;
  // bounding box of the quad -> pixel centres covered -> bin rectangle (as the 3DGS path)
  const float bex = fabsf(h1x) + fabsf(h2x), bey = fabsf(h1y) + fabsf(h2y);
  const float fx0 = ceilf(ccx - bex - 0.5f), fx1 = floorf(ccx + bex - 0.5f);
  const float fy0 = ceilf(ccy - bey - 0.5f), fy1 = floorf(ccy + bey - 0.5f);
  const float ymin = (float)(F.stripRow0 * kTilePx), ymax = (float)(min(F.stripRow1 * kTilePx, F.height) - 1);
  if(!(fx1 >= fx0 && fy1 >= fy0 && fx1 >= 0.f && fx0 <= (float)(F.width - 1) && fy1 >= ymin && fy0 <= ymax))
    return false;
  const int x0 = (int)fmaxf(fx0, 0.f), x1 = (int)fminf(fx1, (float)(F.width - 1));
  const int y0 = (int)fmaxf(fy0, ymin), y1 = (int)fminf(fy1, ymax);
  const int sx = 4 + F.binShiftX, sy = 4 + F.binShiftY;
  rectOut = (uint32_t)(x0 >> sx) | ((uint32_t)(y0 >> sy) << 8) | ((uint32_t)(x1 >> sx) << 16) | ((uint32_t)(y1 >> sy) << 24);

  out.cx  = ccx;
  out.cy  = ccy;
  const float rn1 = gRcp(n1), rn2 = gRcp(n2);
  out.q1x = h1x * rn1;
  out.q1y = h1y * rn1;
  out.q2x = h2x * rn2;
  out.q2y = h2y * rn2;
  out.bex = bex + 0.01f;
  out.bey = bey + 0.01f;
  // canonical frame: A = S^-1 R^T, i.e. A[k][r] = R[k][r] / s_k with R's rows the axes;  B = A N, ro = A (M^-1 o - p)
  const float* Mi = I.modelInv;
  float        A[3][3];
  const float  rsc[3] = {gRcp(sc[0]), gRcp(sc[1]), gRcp(sc[2])};
#pragma unroll
  for(int k = 0; k < 3; ++k)
#pragma unroll
    for(int r = 0; r < 3; ++r)
      A[k][r] = R[k][r] * rsc[k];
#pragma unroll
  for(int k = 0; k < 3; ++k)
#pragma unroll
    for(int c = 0; c < 3; ++c)  // N(r,c) = Mi[c*4 + r]
      out.B[3 * k + c] = A[k][0] * Mi[c * 4 + 0] + A[k][1] * Mi[c * 4 + 1] + A[k][2] * Mi[c * 4 + 2];
  // camera origin in model space: M^-1 * (V^-1 * (0,0,0,1))
  const float ox = F.viewInv[12], oy = F.viewInv[13], oz = F.viewInv[14];
  const float mox = Mi[0] * ox + Mi[4] * oy + Mi[8] * oz + Mi[12];
  const float moy = Mi[1] * ox + Mi[5] * oy + Mi[9] * oz + Mi[13];
  const float moz = Mi[2] * ox + Mi[6] * oy + Mi[10] * oz + Mi[14];
  const float gx = mox - px, gy = moy - py, gz = moz - pz;
#pragma unroll
  for(int k = 0; k < 3; ++k)
    out.ro[k] = A[k][0] * gx + A[k][1] * gy + A[k][2] * gz;
  out.r = out.g = out.b = 0.0f;  // shaded by the compositor
  out.a = alpha;
  return true;
}

// Phase 1 (key + frustum cull + ordered compaction) is k_project's, statement for statement: the sorted (key, id)
// stream of a 3DGUT frame is bit-identical to the 3DGS frame's before the front-end rejections.
__global__ __launch_bounds__(kGutThreads) void k_project_gut(const FrameArgs* __restrict__ Ap, FrameCounters* __restrict__ ctr,
                                                             uint2* __restrict__ slotPairs, uint32_t* __restrict__ slotCount,
                                                             GutRec* __restrict__ rec, uint32_t* __restrict__ rect,
                                                             uint32_t* __restrict__ slotHist2,
                                                             uint32_t* __restrict__ top16Rec, uint32_t* __restrict__ top16Count,
                                                             OsPlan* __restrict__ osPlan, const uint32_t* __restrict__ order)
{
  const FrameArgs& A = *Ap;
  __shared__ uint32_t s_hist2[256];  // 2 x 256 sixteen-bit counters (slot_emit.h)
  __shared__ uint16_t s_li[kGutPart];
  __shared__ uint32_t s_key[kGutPart];
  __shared__ uint32_t s_cnt[32];
  __shared__ uint32_t s_base[33];
  const int      t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t part = order != nullptr ? order[blockIdx.x] : blockIdx.x;  // fullest slot of the previous frame first (k_project.hip)
  int            k    = 0;
  for(int i = 1; i < A.f.nInstances; ++i)
    if(part >= A.inst[i].blockBegin)
      k = i;
  const InstanceConst& I      = A.inst[k];
  const uint32_t       local0 = (part - I.blockBegin) * kGutPart;
  const PartitionBox pbox = partitionLoad(I, part - I.blockBegin);  // ahead of the centres (partition_cull.h)
  float px[kGutItems], py[kGutItems], pz[kGutItems];
#pragma unroll
  for(int it = 0; it < kGutItems; ++it)
  {
    const uint32_t li = min(local0 + it * kGutThreads + t, I.count - 1u);
    px[it] = I.centers[3 * (size_t)li];
    py[it] = I.centers[3 * (size_t)li + 1];
    pz[it] = I.centers[3 * (size_t)li + 2];
  }
  {  // the partition as a whole (partition_cull.h): no splat of it can survive the cull / reach the strip
    float partRadius;
    if(A.f.partitionCull && (partitionTest(A, I, pbox, partRadius) & 1u) != 0u)
    {
      emitEmptySlot<kGutThreads>(slotCount, slotHist2, top16Rec, part);
      return;
    }
  }
  // (s_hist2: the hand-over's small tables, slot_emit.h)
  uint32_t key[kGutItems];
  uint64_t bal[kGutItems];
  bool     vis[kGutItems];
#pragma unroll
  for(int it = 0; it < kGutItems; ++it)
  {
    const uint32_t li = local0 + it * kGutThreads + t;
    float          wp[4], vp[4], cp[4];
    mulMat4Exact(I.model, px[it], py[it], pz[it], 1.0f, wp);  // dist.comp.slang:58
    mulMat4Exact(A.f.view, wp[0], wp[1], wp[2], wp[3], vp);
    mulMat4Exact(A.f.proj, vp[0], vp[1], vp[2], vp[3], cp);   // :60
    const float nx = divExact(cp[0], cp[3]), ny = divExact(cp[1], cp[3]), nz = divExact(cp[2], cp[3]);  // :61
    bool        v  = li < I.count;
    if(A.f.cullMode == 1 && distStageCulled(A.f, nx, ny, nz, vp[0], vp[1], vp[2]))  // dist.comp.slang:64-91
      v = false;
    if(A.f.sizeCulling && v)
      v = !sizeCulled(I.maxScale[min(li, I.count - 1u)], A.f.splatScale, I.modelAxisMax, vp[2], A.f.maxFocal, A.f.sizeCullingMinPixels);
    vis[it] = v;
    key[it] = A.f.frontToBack ? encodeKey(nz) : encodeKey(-nz);
    bal[it] = __ballot(v);
    if(lane == 0)
      s_cnt[it * 4 + w] = (uint32_t)__popcll(bal[it]);
  }
  const uint32_t Mv = scanRoundWaveCounts(s_cnt, s_base);
#pragma unroll
  for(int it = 0; it < kGutItems; ++it)
    if(vis[it])
    {
      const uint32_t pos = s_base[it * 4 + w] + lanesBelow(bal[it]);
      s_li[pos]          = (uint16_t)(it * kGutThreads + t);
      s_key[pos]         = key[it];
    }
  __syncthreads();
  if(t == 0 && Mv)
    atomicAdd(&frameStatSlotFromOs(osPlan, part)[2], Mv);  // (sort_plan.h: 32 slots on 32 lines, not the counters' one line)
  // ---- 3DGUT front end over the survivors ----
  // The 96-byte records leave through LDS (as k_project's do): every lane builds one record, then the wave stores its 64 records
  // six lanes per record, so that a store instruction covers whole sectors wherever neighbouring ids both survive (a wave's
  // survivors are mostly consecutive ids: 6 KB contiguous).  Written lane-per-record, each of the six instructions put 16 bytes
  // into 64 different sectors.  Pitch 7 quads: conflict-free 16-byte LDS accesses.
  __shared__ float4   s_grec[4][64 * 7];
  __shared__ uint32_t s_ggid[4][64];
  // the bin rectangles' codes for the hand-over (kernels_common.h: rideEncode): in LDS here — this kernel runs at 169 VGPRs
  // and three workgroups per CU either way, and k_project's register chain cost it 60 us
  __shared__ uint16_t s_code[kGutPart];
  const uint32_t      rideShift = (uint32_t)A.f.rideShift;
  for(uint32_t j0 = 0; j0 < Mv; j0 += kGutThreads)
  {
    const uint32_t j = j0 + t;
    uint32_t       gidOk = 0xFFFFFFFFu;
    if(j < Mv)
    {
      const uint32_t li = local0 + s_li[j];
      GutRec         r;
      uint32_t       rc;
      if(projectSplatGut(A.f, I, li, r, rc))
      {
        gidOk       = I.globalOffset + li;
        float4* dst = &s_grec[w][lane * 7];
        dst[0]      = make_float4(r.cx, r.cy, r.q1x, r.q1y);
        dst[1]      = make_float4(r.q2x, r.q2y, r.bex, r.bey);
        dst[2]      = make_float4(r.B[0], r.B[1], r.B[2], r.B[3]);
        dst[3]      = make_float4(r.B[4], r.B[5], r.B[6], r.B[7]);
        dst[4]      = make_float4(r.B[8], r.ro[0], r.ro[1], r.ro[2]);
        dst[5]      = make_float4(r.r, r.g, r.b, r.a);
        uint32_t code = A.f.rideEscape;
        if(rideShift != 0u)
          s_code[j] = (uint16_t)(code = rideEncode(rc, A.f.binsX, A.f.binsY, A.f.rideShapes, A.f.rideEscape));  // own entry only
        if(rideShift == 0u || code == A.f.rideEscape)  // read back by id only where the code cannot say it (k_project.hip)
          rect[gidOk] = rc;
        s_li[j] |= 0x8000u;
      }
    }
    s_ggid[w][lane] = gidOk;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for(int i = 0; i < 6; ++i)
    {
      const uint32_t idx = (uint32_t)(i * 64 + lane), rr = idx / 6u, pt = idx - rr * 6u;
      const uint32_t g   = s_ggid[w][rr];
      if(g != 0xFFFFFFFFu)
        reinterpret_cast<float4*>(rec + g)[pt] = s_grec[w][rr * 7 + pt];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // second ordered compaction into the partition's slot + what the key sort needs up front (slot_emit.h)
  // (the record staging area is free now: 16 KB of it hold the grouped slot, 2 KB behind them the per-wave digit counts)
  EmitLds E;
  E.li    = s_li;
  E.key   = s_key;
  E.code  = rideShift != 0u ? s_code : nullptr;
  E.stage = reinterpret_cast<uint2*>(&s_grec[0][0]);
  E.whist = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(&s_grec[0][0]) + 16384);
  E.hist1 = s_hist2;
  E.start = reinterpret_cast<uint16_t*>(s_hist2 + 128);
  E.cnt   = s_cnt;
  static_assert(sizeof(s_grec) >= 16384 + 2048, "the hand-over's stage and counters live in the record staging area");
  emitSlot<kGutThreads, kGutItems>(Mv, false, E, slotPairs, slotCount, slotHist2, top16Rec, top16Count, osPlan, ctr, part,
                                   I.globalOffset + local0, rideShift | (A.f.rideSplit ? 0x100u : 0u));
}

// world-space ray direction of the pixel whose centre is (pcx, pcy) (threedgut_raster.frag.slang:101-111); false: outside the
// fisheye's field of view (the fragment is discarded)
__device__ __forceinline__ bool gutPixelRay(const FrameConst& F, float pcx, float pcy, float& dxw, float& dyw, float& dzw)
{
  const float* Vi = F.viewInv;
  float        cx, cy, cz;
  bool         rayOk = true;
  if(F.cameraModel == 1)
  {  // generateFisheyeRay(position.xy, viewport, fovRad, principal 0, viewInverse), cameras.h.slang:46-82
    const float u = (pcx / ((float)F.width - 1.0f)) * 2.0f - 1.0f, v = (pcy / ((float)F.height - 1.0f)) * 2.0f - 1.0f;
    const float r = sqrtf(u * u + v * v);
    rayOk         = !(r > 1.0f);
    float phiCos  = fabsf(r) > 1e-9f ? u / r : 0.0f;
    phiCos        = fminf(fmaxf(phiCos, -1.0f), 1.0f);
    float phi     = acosf(phiCos);
    phi           = v < 0.0f ? -phi : phi;
    const float theta = r * F.fovRad * 0.5f;
    cx = cosf(phi) * sinf(theta);
    cy = -sinf(phi) * sinf(theta);
    cz = -cosf(theta);
  }
  else
  {  // generatePinholeRay(position.xy, float2(0.5), ...): the 0.5 is added to SV_Position, as the reference writes it
    const float ux = ((pcx + 0.5f) / (float)F.width) * 2.0f - 1.0f, uy = ((pcy + 0.5f) / (float)F.height) * 2.0f - 1.0f;
    const float* Pi = F.projInv;
    cx = Pi[0] * ux + Pi[4] * uy + Pi[8] + Pi[12];
    cy = Pi[1] * ux + Pi[5] * uy + Pi[9] + Pi[13];
    cz = Pi[2] * ux + Pi[6] * uy + Pi[10] + Pi[14];
  }
  dxw = Vi[0] * cx + Vi[4] * cy + Vi[8] * cz;
  dyw = Vi[1] * cx + Vi[5] * cy + Vi[9] * cz;
  dzw = Vi[2] * cx + Vi[6] * cy + Vi[10] * cz;
  const float l = rsqrtf(dxw * dxw + dyw * dyw + dzw * dzw);
  dxw *= l; dyw *= l; dzw *= l;
  return rayOk;
}

// ---- compositor: one workgroup per 16x16 tile, one pixel per thread -------------------------------------------------------
constexpr int kGutBatch = 256;  // list entries scanned per round == staging capacity

// XT 1 (2: + the non-quadratic particle kernels, the surface side outputs and their LDS; 3: + NORMAL_METHOD_ISO_SURFACE): the variant with depth of field (frag.slang:104-109, cameras.h.slang:85-108), stochastic splats (:150-172) and/or
// a particle kernel other than the quadratic one, and/or the surface side outputs (picked depth, splat id, integrated normal)
template <int SHF, int XT>
__global__ __launch_bounds__(256) void k_composite_gut(const FrameArgs* __restrict__ Ap, const uint2* __restrict__ ranges,
                                                       const uint32_t* __restrict__ valX, const uint32_t* __restrict__ valY,
                                                       const SortPlan* __restrict__ plan, const GutRec* __restrict__ rec,
                                                       void* __restrict__ outImage, int halfOut, FrameCounters* __restrict__ ctr,
                                                       float* __restrict__ outDepth, uint32_t* __restrict__ outSplatId,
                                                       float4* __restrict__ outNormal)
{
  __shared__ float4   s_r[kGutBatch][6];
  __shared__ uint32_t s_gid[XT ? kGutBatch : 1];
  __shared__ float4   s_n[XT >= 2 ? kGutBatch : 1];  // surface outputs (XT 2): world normal of the record (.w = 1: minus the pixel's ray)
  __shared__ float    s_z[XT >= 2 ? kGutBatch : 1];  //                         fragCoord.z of the record's quad
  __shared__ float4   s_iso[XT == 3 ? kGutBatch : 1][3];  // NORMAL_METHOD_ISO_SURFACE: canonical -> world normal matrix (surface_normal.h)
  __shared__ uint32_t s_wc[4];
  __shared__ uint32_t s_live;
  const FrameConst& F = Ap->f;
  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  const int tilesInStrip = F.tilesX * (F.stripRow1 - F.stripRow0);
  if((int)blockIdx.x >= tilesInStrip)
    return;
  const int tx = (int)blockIdx.x % F.tilesX, ty = F.stripRow0 + (int)blockIdx.x / F.tilesX;
  const int px = tx * kTilePx + (t & 15), py = ty * kTilePx + (t >> 4);
  const bool inside = px < F.width && py < F.height;
  const float pcx = (float)px + 0.5f, pcy = (float)py + 0.5f;
  const float bcx = (float)(tx * kTilePx) + 8.0f, bcy = (float)(ty * kTilePx) + 8.0f;
  // ray of this pixel in world space (frag.slang:101-111)
  float dxw, dyw, dzw;
  const bool rayOk = gutPixelRay(F, pcx, pcy, dxw, dyw, dzw);
  // depth of field: the ray leaves a random point of the lens and still passes through the focal point of the pinhole ray
  float    lensX = 0.0f, lensY = 0.0f, lensZ = 0.0f;
  uint32_t seedPx = 0u;
  const bool stoch = XT && F.stochastic != 0 && !((F.debugFlags & 4) != 0);
  if constexpr(XT != 0)
  {
    seedPx = rngXxhash32((uint32_t)px, (uint32_t)py, (uint32_t)F.frameSampleId);  // int(position.x), int(position.y), frameSampleId
    if(F.dofMode != 0)
    {
      uint32_t    seed = seedPx;
      const float r1 = rngRand(seed) * 6.28318530717958647692f, r2 = rngRand(seed) * F.aperture;
      const float* Vi = F.viewInv;  // camRight = mul(float4(1,0,0,0), viewInverse), camUp = mul(float4(0,1,0,0), viewInverse)
      const float c = cosf(r1), sn = sinf(r1), sq = sqrtf(r2);
      lensX = (c * Vi[0] + sn * Vi[4]) * sq;
      lensY = (c * Vi[1] + sn * Vi[5]) * sq;
      lensZ = (c * Vi[2] + sn * Vi[6]) * sq;
      float fx = dxw * F.focusDist - lensX, fy = dyw * F.focusDist - lensY, fz = dzw * F.focusDist - lensZ;
      const float l = rsqrtf(fx * fx + fy * fy + fz * fz);
      dxw = fx * l; dyw = fy * l; dzw = fz * l;
    }
  }
  const bool  early   = F.alphaMode == 0;
  const bool  noGauss = (F.debugFlags & 4) != 0;
  const float tMin    = early ? 1.0e-4f : -1.0f;
  const uint32_t* vals = plan->finalSel ? valY : valX;
  const int      bin   = (ty >> F.binShiftY) * F.binsX + (tx >> F.binShiftX);
  const uint2    range = ranges[bin];
  float T = (inside && rayOk) ? 1.0f : 0.0f, cr = 0.f, cg = 0.f, cb = 0.f, asum = 0.f;
  const bool surf = XT >= 2 && F.surfaceOutputs != 0;
  float      nx = 0.f, ny = 0.f, nz = 0.f, pickZ = 0.f;
  uint32_t   pickId = 0xFFFFFFFFu;
  uint32_t hi = range.y;
  uint32_t statScanned = 0, statStaged = 0;
  while(hi > range.x)
  {
    // ---- stage: the next 256 nearest entries, culled against the tile, compacted in list order ----
    const uint32_t avail = hi - range.x;
    const bool     have  = (uint32_t)t < avail;
    uint32_t       g     = have ? vals[hi - 1u - (uint32_t)t] : 0u;
    float4         r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, -1.f, -1.f);
    if(have)
    {
      const float4* rp = reinterpret_cast<const float4*>(rec + g);
      r0 = rp[0];
      r1 = rp[1];
    }
    const bool     ok  = have && fabsf(r0.x - bcx) <= r1.z + 7.5f && fabsf(r0.y - bcy) <= r1.w + 7.5f;
    const uint64_t bal = __ballot(ok);
    if(lane == 0)
      s_wc[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0;
    if(w > 0) base += s_wc[0];
    if(w > 1) base += s_wc[1];
    if(w > 2) base += s_wc[2];
    const uint32_t fill = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    if(ok)
    {
      const uint32_t pos = base + lanesBelow(bal);
      const float4*  rp  = reinterpret_cast<const float4*>(rec + g);
      const float4   r2 = rp[2], r3 = rp[3], r4 = rp[4];
      float4         c5 = rp[5];
      // deferred shading (mesh.slang:142-148): base colour + SH in the splat's model coordinates, once per staged record
      int k = 0;
      for(int i = 1; i < F.nInstances; ++i)
        if(g >= Ap->inst[i].globalOffset)
          k = i;
      const InstanceConst& I  = Ap->inst[k];
      const uint32_t       li = g - I.globalOffset;
      const float4         col = reinterpret_cast<const float4*>(I.rgbaF32)[li];
      float dx = I.centers[3 * (size_t)li] - I.camModel[0], dy = I.centers[3 * (size_t)li + 1] - I.camModel[1],
            dz = I.centers[3 * (size_t)li + 2] - I.camModel[2];
      const float dl = rsqrtf(dx * dx + dy * dy + dz * dz);
      dx *= dl; dy *= dl; dz *= dl;
      const bool shOnly = (F.debugFlags & 2) != 0;
      c5.x = shOnly ? 0.5f : col.x;
      c5.y = shOnly ? 0.5f : col.y;
      c5.z = shOnly ? 0.5f : col.z;
      const int deg = (I.sh == nullptr) ? 0 : min(I.shDegree, F.shDegree);
      if(deg > 0)
        addShRadiance<SHF>(I.sh, li, deg, dx, dy, dz, c5.x, c5.y, c5.z);
      // acceptance (threedgrt.h.slang:259-267) with two compares less per fragment: density > alphaCull, alpha = min(clamp,
      // response * density) > 1/255 and response > kMin  <=>  response > max(kMin, 1 / (255 density)) — a per-record cutoff
      // that takes the slot of the box extents (needed above only)
      const float rcut = (c5.w > F.alphaCull) ? fmaxf(F.kernelMinResponse, 1.0f / (255.0f * fmaxf(c5.w, 1e-30f))) : 3.0e38f;
      s_r[pos][0] = r0;
      s_r[pos][1] = make_float4(r1.x, r1.y, rcut, 0.0f);
      s_r[pos][2] = r2;
      s_r[pos][3] = r3;
      s_r[pos][4] = r4;
      s_r[pos][5] = c5;
      if constexpr(XT != 0)
      {
        s_gid[pos] = g;
        if constexpr(XT >= 2)
        {  // frag.slang:127-131 -> particleProcessHitGutWithNormal (threedgrt.h.slang:281-345): the max-density-plane normal is a
          // per-splat quantity (ray ORIGIN only) except for particles with two degenerate axes; no octahedral round trip here
          // (the normal is computed in the fragment shader, not carried through an interstage variable).  fragCoord.z: the
          // quad sits at the pinhole depth of the centre (mesh.slang:221-226)
          if constexpr(XT == 3)
          {  // the fragment's own normal: where its ray enters the kernel ellipsoid (threedgrt.h.slang:330-335, 423-497)
            float4 isoRec[3];
            splatIsoNormalRec(F, I, li, isoRec);
            s_iso[pos][0] = isoRec[0];
            s_iso[pos][1] = isoRec[1];
            s_iso[pos][2] = isoRec[2];
            s_n[pos]      = make_float4(isoRec[0].x, isoRec[0].y, isoRec[0].z, isoRec[0].w == 2.0f ? 1.0f : 0.0f);
          }
          else
          {
            s_n[pos] = splatWorldNormal(F, I, li, false);
          }
          const float  cpx = I.centers[3 * (size_t)li], cpy = I.centers[3 * (size_t)li + 1], cpz = I.centers[3 * (size_t)li + 2];
          const float* MV = I.modelView;
          const float* P  = F.proj;
          const float  tx = MV[0] * cpx + MV[4] * cpy + MV[8] * cpz + MV[12];
          const float  ty = MV[1] * cpx + MV[5] * cpy + MV[9] * cpz + MV[13];
          const float  tz = MV[2] * cpx + MV[6] * cpy + MV[10] * cpz + MV[14];
          const float  tw = MV[3] * cpx + MV[7] * cpy + MV[11] * cpz + MV[15];
          const float  cz = P[2] * tx + P[6] * ty + P[10] * tz + P[14] * tw;
          const float  cw = P[3] * tx + P[7] * ty + P[11] * tz + P[15] * tw;
          s_z[pos]        = cz * gRcp(cw);
        }
      }
    }
    if(t == 0)
      s_live = 0u;
    __syncthreads();
    statScanned += min(avail, (uint32_t)kGutBatch);
    statStaged += fill;
    hi -= min(avail, (uint32_t)kGutBatch);
    // ---- blend front to back ----
    for(uint32_t j = 0; j < fill; ++j)
    {
      const float4 a0 = s_r[j][0], a1 = s_r[j][1];
      const float  ddx = pcx - a0.x, ddy = pcy - a0.y;
      const bool   inQuad = fabsf(ddx * a0.z + ddy * a0.w) <= 1.0f && fabsf(ddx * a1.x + ddy * a1.y) <= 1.0f;
      const float4 b0 = s_r[j][2], b1 = s_r[j][3], b2 = s_r[j][4], c4 = s_r[j][5];
      // canonical ray direction ~ B d, origin ro; dist^2 = |g x ro|^2 / |g|^2  (threedgrt.h.slang:57-81)
      const float gx = b0.x * dxw + b0.y * dyw + b0.z * dzw;
      const float gy = b0.w * dxw + b1.x * dyw + b1.y * dzw;
      const float gz = b1.z * dxw + b1.w * dyw + b2.x * dzw;
      float rox = b2.y, roy = b2.z, roz = b2.w;
      if constexpr(XT != 0)
      {  // rayOrigin += randomAperturePos: canonical origin + B * offset
        rox += b0.x * lensX + b0.y * lensY + b0.z * lensZ;
        roy += b0.w * lensX + b1.x * lensY + b1.y * lensZ;
        roz += b1.z * lensX + b1.w * lensY + b2.x * lensZ;
      }
      const float kx = gy * roz - gz * roy, ky = gz * rox - gx * roz, kz = gx * roy - gy * rox;
      const float dist2 = (kx * kx + ky * ky + kz * kz) * gRcp(gx * gx + gy * gy + gz * gz);
      float resp = __expf(-0.5f * dist2);                            // quadratic kernel, :127-131
      if constexpr(XT >= 2)
      {  // particleRayMaxKernelResponse<KERNEL_DEGREE>, threedgrt.h.slang:83-127 (its argument is the squared distance)
        switch(F.kernelDegree)
        {
          case 8: resp = __expf(-0.000685871056241f * (dist2 * dist2) * (dist2 * dist2)); break;
          case 5: resp = __expf(-0.0185185185185f * dist2 * dist2 * sqrtf(dist2)); break;
          case 4: resp = __expf(-0.0555555555556f * dist2 * dist2); break;
          case 3: resp = __expf(-0.166666666667f * dist2 * sqrtf(dist2)); break;
          case 1: resp = __expf(-1.5f * sqrtf(dist2)); break;
          case 0: resp = fmaxf(1.0f + -0.329630334487f * sqrtf(dist2), 0.0f); break;
          default: break;
        }
      }
      const float al    = fminf(F.alphaClamp, resp * c4.w);          // :263
      const bool  hit   = inQuad && resp > a1.z && T >= tMin;
      float       op    = hit ? (noGauss ? 1.0f : al) : 0.0f;
      if constexpr(XT != 0)
      {
        if(stoch)
        {  // frag.slang:153-158; primitive id as in the 3DGS compositor (k_raster.hip): 2 * (id mod 32) + triangle
          const uint32_t gid = s_gid[j];
          const float    qu = ddx * a0.z + ddy * a0.w, qv = ddx * a1.x + ddy * a1.y;
          uint32_t       h  = rngXxhash32(seedPx, gid, 2u * (gid & 31u) + (qu > qv ? 0u : 1u));
          op                = (hit && rngRand(h) < op) ? 1.0f : 0.0f;
        }
      }
      const float wgt   = op * T;
      cr += wgt * c4.x;
      cg += wgt * c4.y;
      cb += wgt * c4.z;
      asum += op;
      T -= wgt;
      if constexpr(XT >= 2)
      {
        if(surf)
        {  // frag.slang:195-228: normal attachment "under"-blended with (normal * opacity, opacity); picked depth = the
          // fragment after which the transmittance is below the threshold, and the splat that set it
          float4 n1  = s_n[j];
          bool   ray = n1.w != 0.0f;  // degenerate particle: -rayDirection (model -> world: minus the pixel's world ray)
          if constexpr(XT == 3)
          if(wgt != 0.0f && !ray && s_iso[j][0].w == 0.0f)
          {  // raySphereIntersection(canonical origin, canonical direction, 3, 0, inf), threedgrt.h.slang:502-540.  The same two
            // roots, written around the point of closest approach (t_mid = -(o.g)/(g.g), half chord = sqrt((9 - dist^2)/(g.g)))
            // instead of b^2 - 4ac: the canonical origin lies hundreds of radii from a small particle and the discriminant of
            // the textbook form cancels in fp32 (the oracle follows the shader; tests/test_oracle_cpu.py has the numbers)
            const float qa = gx * gx + gy * gy + gz * gz;
            const float tm = -(rox * gx + roy * gy + roz * gz) * gRcp(qa);
            const float dd = 9.0f - dist2;
            ray = true;  // no intersection in front of the origin: -rayDirection (:468-472)
            if(dd >= 0.0f)
            {
              const float half = sqrtf(dd * gRcp(qa));
              const float th   = (tm - half >= 0.0f) ? tm - half : tm + half;
              if(th >= 0.0f)
              {
                const float  hx = rox + th * gx, hy = roy + th * gy, hz = roz + th * gz;
                const float4 i0 = s_iso[j][0], i1 = s_iso[j][1], i2 = s_iso[j][2];
                const float  wx = i0.x * hx + i0.y * hy + i0.z * hz;
                const float  wy = i1.x * hx + i1.y * hy + i1.z * hz;
                const float  wz = i2.x * hx + i2.y * hy + i2.z * hz;
                const float  rl = rsqrtf(wx * wx + wy * wy + wz * wz);
                n1  = make_float4(wx * rl, wy * rl, wz * rl, 0.0f);
                ray = false;
              }
            }
          }
          nx += wgt * (ray ? -dxw : n1.x);
          ny += wgt * (ray ? -dyw : n1.y);
          nz += wgt * (ray ? -dzw : n1.z);
          if(op > 0.0f && pickZ == 0.0f && T < F.depthIsoThreshold)
          {
            pickZ  = s_z[j];
            pickId = s_gid[j];
          }
        }
      }
    }
    if(early)
    {
      if(T >= tMin)
        s_live = 1u;  // benign race: every writer stores 1
      __syncthreads();
      if(s_live == 0u)
        break;
    }
    else
      __syncthreads();
  }
  if(t == 0)
  {
    uint32_t* stat = frameStatSlot(plan, blockIdx.x >> 3);  // (sort_plan.h: one 128-byte line per slot)
    atomicAdd(&stat[1], statScanned);
    atomicAdd(&stat[0], statStaged);
  }
  if(!inside)
    return;
  const float alphaOut = F.alphaMode == 1 ? asum : 1.0f - ((inside && rayOk) ? T : 1.0f);
  const size_t pix = (size_t)py * F.width + px;
  if constexpr(XT >= 2)
  {
    if(surf)
    {
      outDepth[pix]   = pickZ;
      outSplatId[pix] = pickId;
      outNormal[pix]  = make_float4(nx, ny, nz, 1.0f - ((inside && rayOk) ? T : 1.0f));
    }
  }
  if(halfOut == 1)
  {
    const __half2 lo = __floats2half2_rn(cr, cg), hi2 = __floats2half2_rn(cb, alphaOut);
    uint2         o;
    o.x = *reinterpret_cast<const uint32_t*>(&lo);
    o.y = *reinterpret_cast<const uint32_t*>(&hi2);
    reinterpret_cast<uint2*>(outImage)[pix] = o;
  }
  else if(halfOut == 0)
    reinterpret_cast<float4*>(outImage)[pix] = make_float4(cr, cg, cb, alphaOut);
  else
  {
    auto q = [](float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f); };
    reinterpret_cast<uint32_t*>(outImage)[pix] = q(cr) | (q(cg) << 8) | (q(cb) << 16) | (q(alphaOut) << 24);
  }
}


// ---- packed compositor (plain mode: quadratic kernel, coverage alpha, no depth of field / stochastic / surface outputs) -------------
// The geometry of the 3DGS compositor (k_raster.hip): one workgroup per 32x16-pixel region, one wave per 16x8 quarter, TWO pixels
// per lane (x and x + 8) so that the per-fragment arithmetic runs on gfx950's packed-fp32 instructions with the particle's
// parameters broadcast — the one-pixel-per-lane kernel above spends ~45 instructions per (record, pixel), this one ~50 per
// (record, pixel PAIR) — and a region twice as large halves the list entries scanned per pixel.  Records are culled against the
// region, compacted in list order into an LDS batch together with a 4-bit mask of the quarters their quad's box touches, shaded
// there (deferred SH), and each wave walks only the records of its quarter.  Saturation is a per-fragment predicate (T >= 1e-4),
// so the frame does not depend on batch boundaries (strips == full frame).
typedef float gv2f __attribute__((ext_vector_type(2)));
constexpr int kGut2Cap = 256;  // LDS batch capacity == entries scanned per round

// XT 1: + depth of field (a lens offset per pixel) and stochastic splats (quadratic kernel only)
template <int SHF, int XT>
__global__ __launch_bounds__(256) void k_composite_gut2(const FrameArgs* __restrict__ Ap, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ valX, const uint32_t* __restrict__ valY,
                                                        const SortPlan* __restrict__ plan, const GutRec* __restrict__ rec,
                                                        void* __restrict__ outImage, int halfOut, FrameCounters* __restrict__ ctr)
{
  __shared__ float4   s_r[kGut2Cap][6];
  __shared__ uint8_t  s_m[kGut2Cap];
  __shared__ uint32_t s_gid[XT ? kGut2Cap : 1];
  __shared__ uint32_t s_wc[4];
  const FrameConst& F = Ap->f;
  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  // region order as in k_composite (k_raster.hip): all regions of a bin on one XCD (workgroup b runs on XCD b % 8; they read
  // the same list), consecutive bins on different XCDs, bins taken longest list first (ranked by the binning stage)
  const int colsX    = (F.tilesX + 1) >> 1;
  const int bw       = 1 << (F.binShiftX - 1), bh = 1 << F.binShiftY;  // bin size in regions
  const int binRow0  = F.stripRow0 >> F.binShiftY;
  const int binRows  = ((F.stripRow1 - 1) >> F.binShiftY) - binRow0 + 1;
  const int perBin   = bw * bh;
  const int seq      = (int)(blockIdx.x >> 3);
  const int ord      = (seq / perBin) * 8 + (int)(blockIdx.x & 7);  // bin ordinal
  const int inBin    = seq % perBin;
  const bool ordered = plan->ghist[2][0] != 0u;
  int cx2, ty;
  if(ordered)
  {
    if(ord >= F.binsX * F.binsY)
      return;
    const int b = (int)plan->ghist[1][ord];
    cx2         = (b % F.binsX) * bw + inBin % bw;
    ty          = (b / F.binsX) * bh + inBin / bw;
  }
  else
  {
    if(ord >= binRows * F.binsX)
      return;
    cx2 = (ord % F.binsX) * bw + inBin % bw;
    ty  = (binRow0 + ord / F.binsX) * bh + inBin / bw;
  }
  if(cx2 >= colsX || ty < F.stripRow0 || ty >= F.stripRow1)
    return;
  const int tx  = cx2 * 2;
  const int qx0 = tx * kTilePx + (w & 1) * 16, qy0 = ty * kTilePx + (w >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);  // second pixel: px + 8
  const bool in0 = px < F.width && py < F.height, in1 = px + 8 < F.width && py < F.height;
  const gv2f  pcx = {(float)px + 0.5f, (float)px + 8.5f};
  const float pcy = (float)py + 0.5f;
  const float bcx = (float)(tx * kTilePx) + 16.0f, bcy = (float)(ty * kTilePx) + 8.0f;  // region centre
  gv2f dxw, dyw, dzw;
  bool ok0, ok1;
  {
    float a, b, c;
    ok0 = gutPixelRay(F, pcx.x, pcy, a, b, c);
    dxw.x = a; dyw.x = b; dzw.x = c;
    ok1 = gutPixelRay(F, pcx.y, pcy, a, b, c);
    dxw.y = a; dyw.y = b; dzw.y = c;
  }
  // depth of field (frag.slang:104-109, cameras.h.slang:85-108): one lens sample per pixel and frame
  gv2f     lensX = {0.f, 0.f}, lensY = {0.f, 0.f}, lensZ = {0.f, 0.f};
  uint32_t seed0 = 0u, seed1 = 0u;
  if constexpr(XT != 0)
  {
    seed0 = rngXxhash32((uint32_t)px, (uint32_t)py, (uint32_t)F.frameSampleId);
    seed1 = rngXxhash32((uint32_t)px + 8u, (uint32_t)py, (uint32_t)F.frameSampleId);
    if(F.dofMode != 0)
    {
      const float* Vi = F.viewInv;
#pragma unroll
      for(int h = 0; h < 2; ++h)
      {
        uint32_t    sd = h ? seed1 : seed0;
        const float r1 = rngRand(sd) * 6.28318530717958647692f, r2 = rngRand(sd) * F.aperture;
        const float c = cosf(r1), sn = sinf(r1), sq = gSqrt(r2);
        const float lx = (c * Vi[0] + sn * Vi[4]) * sq, ly = (c * Vi[1] + sn * Vi[5]) * sq, lz = (c * Vi[2] + sn * Vi[6]) * sq;
        const float dx0 = h ? dxw.y : dxw.x, dy0 = h ? dyw.y : dyw.x, dz0 = h ? dzw.y : dzw.x;
        const float fx = dx0 * F.focusDist - lx, fy = dy0 * F.focusDist - ly, fz = dz0 * F.focusDist - lz;
        const float l  = rsqrtf(fx * fx + fy * fy + fz * fz);
        if(h) { lensX.y = lx; lensY.y = ly; lensZ.y = lz; dxw.y = fx * l; dyw.y = fy * l; dzw.y = fz * l; }
        else  { lensX.x = lx; lensY.x = ly; lensZ.x = lz; dxw.x = fx * l; dyw.x = fy * l; dzw.x = fz * l; }
      }
    }
  }
  const bool  stoch   = XT && F.stochastic != 0 && !((F.debugFlags & 4) != 0);
  const bool  noGauss = (F.debugFlags & 4) != 0;
  constexpr float tMin = 1.0e-4f;
  const uint32_t* vals = plan->finalSel ? valY : valX;
  const int      bin   = (ty >> F.binShiftY) * F.binsX + (tx >> F.binShiftX);
  const uint2    range = ranges[bin];
  gv2f T = {(in0 && ok0) ? 1.0f : 0.0f, (in1 && ok1) ? 1.0f : 0.0f}, cr = {0.f, 0.f}, cg = {0.f, 0.f}, cb = {0.f, 0.f};
  uint32_t hi = range.y;
  uint32_t statScanned = 0, statStaged = 0;
  while(hi > range.x)
  {
    // ---- stage: the next 256 nearest entries, culled against the region, compacted in list order ----
    const uint32_t avail = hi - range.x;
    const bool     have  = (uint32_t)t < avail;
    const uint32_t g     = have ? vals[hi - 1u - (uint32_t)t] : 0u;
    float4         r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, -1.f, -1.f);
    if(have)
    {
      const float4* rp = reinterpret_cast<const float4*>(rec + g);
      r0 = rp[0];
      r1 = rp[1];
    }
    // pixel centres of the region span bcx +- 15.5, bcy +- 7.5
    const bool     ok  = have && fabsf(r0.x - bcx) <= r1.z + 15.5f && fabsf(r0.y - bcy) <= r1.w + 7.5f;
    const uint64_t bal = __ballot(ok);
    if(lane == 0)
      s_wc[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = 0;
    if(w > 0) base += s_wc[0];
    if(w > 1) base += s_wc[1];
    if(w > 2) base += s_wc[2];
    const uint32_t fill = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    if(ok)
    {
      const uint32_t pos = base + lanesBelow(bal);
      const float4*  rp  = reinterpret_cast<const float4*>(rec + g);
      const float4   r2 = rp[2], r3 = rp[3], r4 = rp[4];
      float4         c5 = rp[5];
      // deferred shading (mesh.slang:142-148): base colour + SH in the splat's model coordinates, once per staged record
      int k = 0;
      for(int i = 1; i < F.nInstances; ++i)
        if(g >= Ap->inst[i].globalOffset)
          k = i;
      const InstanceConst& I  = Ap->inst[k];
      const uint32_t       li = g - I.globalOffset;
      const float4         col = reinterpret_cast<const float4*>(I.rgbaF32)[li];
      float dx = I.centers[3 * (size_t)li] - I.camModel[0], dy = I.centers[3 * (size_t)li + 1] - I.camModel[1],
            dz = I.centers[3 * (size_t)li + 2] - I.camModel[2];
      const float dl = rsqrtf(dx * dx + dy * dy + dz * dz);
      dx *= dl; dy *= dl; dz *= dl;
      const bool shOnly = (F.debugFlags & 2) != 0;
      c5.x = shOnly ? 0.5f : col.x;
      c5.y = shOnly ? 0.5f : col.y;
      c5.z = shOnly ? 0.5f : col.z;
      const int deg = (I.sh == nullptr) ? 0 : min(I.shDegree, F.shDegree);
      if(deg > 0)
        addShRadiance<SHF>(I.sh, li, deg, dx, dy, dz, c5.x, c5.y, c5.z);
      if(!(c5.w > F.alphaCull))
        c5.w = 0.0f;  // particleProcessHitGut rejects the whole particle (density <= alphaCullThreshold): no fragment can pass
      // acceptance of particleProcessHitGut (threedgrt.h.slang:259-267) as ONE compare per pixel: alpha = min(clamp, response *
      // density) > 1/255 and response > kernelMinResponse  <=>  response > max(kMin, 1 / (255 density))  <=>  (quadratic
      // kernel) dist^2 < -2 ln(that): a per-record cutoff, staged in the slot of the box extents (used above only)
      const float rcut  = fmaxf(F.kernelMinResponse, 1.0f / (255.0f * fmaxf(c5.w, 1e-30f)));
      const float d2cut = (c5.w > 0.0f && rcut < 1.0f) ? -2.0f * __logf(rcut) : -1.0f;
      s_r[pos][0] = r0;
      s_r[pos][1] = make_float4(r1.x, r1.y, d2cut, 0.0f);
      s_r[pos][2] = r2;
      s_r[pos][3] = r3;
      s_r[pos][4] = r4;
      s_r[pos][5] = c5;
      if constexpr(XT != 0)
        s_gid[pos] = g;
      // quarters (16 x 8 pixels; centres x in [bcx-15.5,bcx-0.5] / [bcx+0.5,bcx+15.5], y in [bcy-7.5,bcy-0.5] / [bcy+0.5,bcy+7.5])
      // the box of the quad touches
      const bool xl = r0.x - r1.z <= bcx - 0.5f, xr = r0.x + r1.z >= bcx + 0.5f;
      const bool yt = r0.y - r1.w <= bcy - 0.5f, yb = r0.y + r1.w >= bcy + 0.5f;
      s_m[pos] = (uint8_t)(((xl && yt) ? 1u : 0u) | ((xr && yt) ? 2u : 0u) | ((xl && yb) ? 4u : 0u) | ((xr && yb) ? 8u : 0u));
    }
    __syncthreads();
    statScanned += min(avail, (uint32_t)kGut2Cap);
    statStaged += fill;
    hi -= min(avail, (uint32_t)kGut2Cap);
    // ---- blend front to back: 64 records at a time, this wave's hit set from the quarter masks ----
    for(uint32_t j0 = 0; j0 < fill; j0 += 64)
    {
      const uint32_t jl   = j0 + (uint32_t)lane;
      const bool     mine = jl < fill && ((s_m[jl] >> w) & 1u);
      uint64_t       hits = __ballot(mine);
      while(hits != 0ull)
      {
        const uint32_t j = j0 + (uint32_t)__builtin_ctzll(hits);
        hits &= hits - 1ull;
        const float4 a0 = s_r[j][0], a1 = s_r[j][1], b0 = s_r[j][2], b1 = s_r[j][3], b2 = s_r[j][4], c4 = s_r[j][5];
        const gv2f   ddx = pcx - a0.x;
        const float  ddy = pcy - a0.y;
        const gv2f   qu = ddx * a0.z + ddy * a0.w, qv = ddx * a1.x + ddy * a1.y;
        // canonical ray direction ~ B d, origin ro; dist^2 = |g x ro|^2 / |g|^2  (threedgrt.h.slang:57-81)
        const gv2f gx = dxw * b0.x + (dyw * b0.y + dzw * b0.z);
        const gv2f gy = dxw * b0.w + (dyw * b1.x + dzw * b1.y);
        const gv2f gz = dxw * b1.z + (dyw * b1.w + dzw * b2.x);
        gv2f rox = {b2.y, b2.y}, roy = {b2.z, b2.z}, roz = {b2.w, b2.w};
        if constexpr(XT != 0)
        {  // rayOrigin += randomAperturePos: canonical origin + B * offset
          rox += lensX * b0.x + (lensY * b0.y + lensZ * b0.z);
          roy += lensX * b0.w + (lensY * b1.x + lensZ * b1.y);
          roz += lensX * b1.z + (lensY * b1.w + lensZ * b2.x);
        }
        const gv2f kx = gy * roz - gz * roy, ky = gz * rox - gx * roz, kz = gx * roy - gy * rox;
        const gv2f kk = kx * kx + (ky * ky + kz * kz), gg = gx * gx + (gy * gy + gz * gz);
        const gv2f dist2 = {kk.x * gRcp(gg.x), kk.y * gRcp(gg.y)};
        const gv2f resp  = {__expf(-0.5f * dist2.x), __expf(-0.5f * dist2.y)};  // quadratic kernel, :127-131
        const gv2f raw   = resp * c4.w;
        const gv2f al    = {fminf(F.alphaClamp, raw.x), fminf(F.alphaClamp, raw.y)};  // :263
        const bool h0 = fmaxf(fabsf(qu.x), fabsf(qv.x)) <= 1.0f && dist2.x < a1.z && T.x >= tMin;
        const bool h1 = fmaxf(fabsf(qu.y), fabsf(qv.y)) <= 1.0f && dist2.y < a1.z && T.y >= tMin;
        gv2f op  = {h0 ? (noGauss ? 1.0f : al.x) : 0.0f, h1 ? (noGauss ? 1.0f : al.y) : 0.0f};
        if constexpr(XT != 0)
        {
          if(stoch)
          {  // frag.slang:153-158; primitive id as in the other compositors: 2 * (id mod 32) + triangle (0: u > v)
            const uint32_t gid = s_gid[j], prim = 2u * (gid & 31u);
            uint32_t       q0 = rngXxhash32(seed0, gid, prim + (qu.x > qv.x ? 0u : 1u));
            uint32_t       q1 = rngXxhash32(seed1, gid, prim + (qu.y > qv.y ? 0u : 1u));
            op.x = (h0 && rngRand(q0) < op.x) ? 1.0f : 0.0f;
            op.y = (h1 && rngRand(q1) < op.y) ? 1.0f : 0.0f;
          }
        }
        const gv2f wgt = op * T;
        cr += wgt * c4.x;
        cg += wgt * c4.y;
        cb += wgt * c4.z;
        T -= wgt;
      }
    }
    // all four waves saturated: stop fetching (the predicate above already keeps saturated pixels unchanged)
    if(__syncthreads_and((T.x >= tMin || T.y >= tMin) ? 0 : 1))
      break;
  }
  if(t == 0)
  {
    uint32_t* stat = frameStatSlot(plan, blockIdx.x >> 3);  // (sort_plan.h: one 128-byte line per slot)
    atomicAdd(&stat[1], statScanned);
    atomicAdd(&stat[0], statStaged);
  }
#pragma unroll
  for(int h = 0; h < 2; ++h)
  {
    if(!(h ? in1 : in0))
      continue;
    const bool  rok = h ? ok1 : ok0;
    const float r = h ? cr.y : cr.x, g = h ? cg.y : cg.x, b = h ? cb.y : cb.x;
    const float alphaOut = 1.0f - (rok ? (h ? T.y : T.x) : 1.0f);
    const size_t pix = (size_t)py * F.width + (size_t)(px + 8 * h);
    if(halfOut == 1)
    {
      const __half2 lo = __floats2half2_rn(r, g), hi2 = __floats2half2_rn(b, alphaOut);
      uint2         o;
      o.x = *reinterpret_cast<const uint32_t*>(&lo);
      o.y = *reinterpret_cast<const uint32_t*>(&hi2);
      reinterpret_cast<uint2*>(outImage)[pix] = o;
    }
    else if(halfOut == 0)
      reinterpret_cast<float4*>(outImage)[pix] = make_float4(r, g, b, alphaOut);
    else
    {
      auto q = [](float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f); };
      reinterpret_cast<uint32_t*>(outImage)[pix] = q(r) | (q(g) << 8) | (q(b) << 16) | (q(alphaOut) << 24);
    }
  }
}

// ---------------------------------------------------------------------------------------------
void launchProjectGut(hipStream_t stream, const FrameArgs& args, const FrameArgs* dArgs, int shFormat, FrameCounters* ctr,
                      uint2* slotPairs, uint32_t* slotCount, GutRec* rec, uint32_t* rect,
                      uint32_t* slotHist2, uint32_t* top16Rec, uint32_t* top16Count, OsPlan* osPlan, const uint32_t* order)
{
  (void)shFormat;
  if(args.f.totalPartitions == 0)
    return;
  hipLaunchKernelGGL(k_project_gut, dim3(args.f.totalPartitions), dim3(kGutThreads), 0, stream, dArgs, ctr, slotPairs, slotCount, rec, rect,
                     slotHist2, top16Rec, top16Count, osPlan, order);
}

void launchCompositeGut(hipStream_t stream, const FrameArgs& A, const FrameArgs* dArgs, const uint2* ranges, const uint32_t* valX,
                        const uint32_t* valY, const SortPlan* planPairs, const GutRec* rec, void* image, int halfOut,
                        FrameCounters* ctr, int shFormat, float* outDepth, uint32_t* outSplatId, float4* outNormal)
{
  const int tiles = A.f.tilesX * (A.f.stripRow1 - A.f.stripRow0);
  if(tiles <= 0)
    return;
  const bool extras = A.f.dofMode != 0 || A.f.stochastic != 0 || A.f.kernelDegree != 2 || A.f.surfaceOutputs != 0;
  static const bool kPacked = [] { const char* e = std::getenv("MGS_GUT_PACKED"); return e ? std::atoi(e) != 0 : true; }();
  const bool packedOk = A.f.kernelDegree == 2 && A.f.surfaceOutputs == 0 && (A.f.alphaMode == 0 || A.f.stochastic != 0);
  if(kPacked && packedOk)
  {  // the quadratic-kernel modes without side outputs run on the packed two-pixels-per-lane compositor
    // all bins of the frame are enumerated (the bin order of the binning stage is over the whole frame; regions outside a
    // strip exit at once)
    const int nBins   = A.f.binsX * A.f.binsY;
    const int regions = ((nBins + 7) / 8) * (1 << (A.f.binShiftX - 1 + A.f.binShiftY)) * 8;
    const bool xt = A.f.dofMode != 0 || A.f.stochastic != 0;
#define MGS_LAUNCH2X(SHF, XTV)                                                                                                            \
  hipLaunchKernelGGL((k_composite_gut2<SHF, XTV>), dim3(regions), dim3(256), 0, stream, dArgs, ranges, valX, valY, planPairs, rec, image, \
                     halfOut, ctr)
#define MGS_LAUNCH2(SHF)   \
  do                       \
  {                        \
    if(xt)                 \
      MGS_LAUNCH2X(SHF, 1); \
    else                   \
      MGS_LAUNCH2X(SHF, 0); \
  } while(0)
    if(shFormat == 0)
      MGS_LAUNCH2(0);
    else if(shFormat == 1)
      MGS_LAUNCH2(1);
    else
      MGS_LAUNCH2(2);
#undef MGS_LAUNCH2X
#undef MGS_LAUNCH2
    return;
  }
#define MGS_LAUNCH(SHF, XT)                                                                                                          \
  hipLaunchKernelGGL((k_composite_gut<SHF, XT>), dim3(tiles), dim3(256), 0, stream, dArgs, ranges, valX, valY, planPairs, rec, image, \
                     halfOut, ctr, outDepth, outSplatId, outNormal)
#define MGS_LAUNCH_X(SHF)            \
  do                                 \
  {                                  \
    if(A.f.surfaceOutputs != 0 && A.f.normalMethod == 1) \
      MGS_LAUNCH(SHF, 3);            \
    else if(A.f.surfaceOutputs != 0 || A.f.kernelDegree != 2) \
      MGS_LAUNCH(SHF, 2);            \
    else if(extras)                  \
      MGS_LAUNCH(SHF, 1);            \
    else                             \
      MGS_LAUNCH(SHF, 0);            \
  } while(0)
  if(shFormat == 0)
    MGS_LAUNCH_X(0);
  else if(shFormat == 1)
    MGS_LAUNCH_X(1);
  else
    MGS_LAUNCH_X(2);
#undef MGS_LAUNCH_X
#undef MGS_LAUNCH
}

}  // namespace mgs
