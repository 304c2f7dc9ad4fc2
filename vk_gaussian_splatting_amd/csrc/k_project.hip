// k_project.hip — fused "dist + cull + project" kernel for gfx950.
//
// Replaces, in ONE pass over the splats in storage (id) order:
//   shaders/dist.comp.slang:40-171          depth key, frustum cull, survivor append
//   shaders/threedgs_raster.mesh.slang:111-291  per-splat raster front end (colour fetch, alpha cull,
//                                            covariance projection, extent basis) — except the SH sum
//                                            (:243), which the compositor evaluates for the splats it stages
//   shaders/threedgs.h.slang:26-121         threedgsCovarianceProjection / threedgsProjectedExtentBasis
// MI355X-first differences:
//   * the reference runs dist in id order and the mesh shader in SORTED order, so its 232 B/splat
//     attribute gather is uncoalesced; here the gather happens before the sort, in id order, and the
//     sort only moves 8-byte (key,id) pairs.  The 32-byte projected record (SplatRec) is indexed by global id.
//   * survivors are compacted per 2048-splat partition into the partition's own slot (ascending id, so
//     the order is deterministic); the radix sort's first pass gathers dense partitions from the slots
//     through the prefix of their counts — there is no global atomic append and no inter-workgroup
//     chain (slot_emit.h) — and gets the partition's two low-digit histograms, built here while the
//     keys are on chip.
//   * the partition as a whole is tested first, by the workgroup itself (partition_cull.h).
//   * splats that can never produce a fragment (alpha cull, lambda2<=0, clipped by z, footprint
//     outside the strip) are dropped BEFORE the sort.
//   * the 180-byte SH record (62 % of a splat's bytes) is not read here: shading is deferred.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_common.h"
#include "sort_plan.h"
#include "slot_emit.h"
#include "partition_cull.h"

namespace mgs {

#ifndef MGS_PRJ_THREADS
#define MGS_PRJ_THREADS 256
#endif
constexpr int kPrjThreads = MGS_PRJ_THREADS;
constexpr int kPrjWaves   = kPrjThreads / 64;
constexpr int kPrjItems   = 2048 / kPrjThreads;
constexpr int kPrjPart    = kPrjThreads * kPrjItems;  // 2048 splats per workgroup
static_assert(kPrjItems * kPrjWaves == 32 && kPrjPart == 2048, "the round x wave table of scanRoundWaveCounts has 32 entries");

struct Projected
{
  SplatRec rec;
  uint32_t rect;  // bin rectangle x0 | y0<<8 | x1<<16 | y1<<24 (inclusive, in bins)
};

// The per-splat raster front end.  Returns false when the splat cannot produce a fragment.
// Written BRANCH-FREE with every load issued first (fetchSplat: opacity, centre, covariance — 40 B per survivor):
// rocprof showed the waves of this kernel waiting on memory 70 % of their cycles when the fetches were staged behind
// the early-outs.  Colour and SH are not touched here: shading is deferred to the compositor.
struct SplatFetch
{
  float  alpha;
  float  px, py, pz;
  float4 cA;
  float2 cB;
};
__device__ __forceinline__ SplatFetch fetchSplat(const InstanceConst& I, uint32_t li)
{
  SplatFetch f;
  f.alpha = I.alpha[li];
  f.px  = I.centers[3 * (size_t)li + 0];
  f.py  = I.centers[3 * (size_t)li + 1];
  f.pz  = I.centers[3 * (size_t)li + 2];
  f.cA  = reinterpret_cast<const float4*>(I.cov6)[li];                        // planar: 16 B per lane
  f.cB  = reinterpret_cast<const float2*>(I.cov6 + 4 * (size_t)I.count)[li];  //          8 B per lane
  return f;
}

// The raster front end (phase 2) is floating-point work with a tolerance (centre / basis <= 1e-5 relative), not the
// bit-exact key: its reciprocals and square roots are the 1-ulp hardware instructions instead of hipcc's correctly rounded
// expansions (10 instructions each; five divisions and seven roots per splat were a fifth of this kernel's instructions).
__device__ __forceinline__ float fastRcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fastSqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ bool projectSplat(const FrameConst& F, const InstanceConst& I, int instIdx, const SplatFetch& in,
                                             Projected& out)
{
  float4       col = make_float4(0.f, 0.f, 0.f, in.alpha);  // the colour itself is fetched by the compositor
  const float  px = in.px, py = in.py, pz = in.pz;
  const float4 cA = in.cA;
  const float2 cB = in.cB;

  // ---- mesh.slang:164-190 ---------------------------------------------------------------------------------
  bool         ok = !(col.w < F.alphaCull);
  const float* MV = I.modelView;
  const float  tx = MV[0] * px + MV[4] * py + MV[8] * pz + MV[12];
  const float  ty = MV[1] * px + MV[5] * py + MV[9] * pz + MV[13];
  const float  tz = MV[2] * px + MV[6] * py + MV[10] * pz + MV[14];
  const float* P  = F.proj;
  float        tw, cx, cy, cz, cw;
  if(F.perspAffine && (I.modelIsIdentity | I.modelIsAffine))
  {  // the same products without the matrices' exact zeros (FrameConst::perspAffine): 7 instead of 20 multiply-adds
    tw = 1.0f;
    cx = P[0] * tx + P[8] * tz;
    cy = P[5] * ty + P[9] * tz;
    cz = P[10] * tz + P[14];
    cw = P[11] * tz;
  }
  else
  {
    tw = MV[3] * px + MV[7] * py + MV[11] * pz + MV[15];
    cx = P[0] * tx + P[4] * ty + P[8] * tz + P[12] * tw;
    cy = P[1] * tx + P[5] * ty + P[9] * tz + P[13] * tw;
    cz = P[2] * tx + P[6] * ty + P[10] * tz + P[14] * tw;
    cw = P[3] * tx + P[7] * ty + P[11] * tz + P[15] * tw;
  }
  if(F.cullMode == 2)
  {  // FRUSTUM_CULLING_AT_RASTER, mesh.slang:181-190
    const float c = (1.0f + F.frustumDilation) * cw;
    ok            = ok && !(fabsf(cx) > c || fabsf(cy) > c || cz < (0.0f - F.frustumDilation) * cw || cz > cw);
  }
  const float rw   = fastRcp(cw);
  const float ndcx = cx * rw, ndcy = cy * rw, ndcz = cz * rw;
  // fixed-function z clip of the emitted quad (all vertices at z = ndc.z, w = 1; no depth clamp)
  ok = ok && (ndcz >= 0.0f && ndcz <= 1.0f);

  // ---- covariance projection, threedgs.h.slang:26-56 ---------------------------------------------------------
  const float s00 = cA.x, s01 = cA.y, s02 = cA.z, s11 = cA.w, s12 = cB.x, s22 = cB.y;
  const float rz = fastRcp(tz), rz2 = rz * rz;
  const float j00 = F.focal[0] * rz, j02 = -(F.focal[0] * tx) * rz2;
  const float j11 = F.focal[1] * rz, j12 = -(F.focal[1] * ty) * rz2;
  // W(r,c) = MV(r,c);  T = J * W (rows 0 and 1 only)
  const float t00 = j00 * MV[0] + j02 * MV[2], t01 = j00 * MV[4] + j02 * MV[6], t02 = j00 * MV[8] + j02 * MV[10];
  const float t10 = j11 * MV[1] + j12 * MV[2], t11 = j11 * MV[5] + j12 * MV[6], t12 = j11 * MV[9] + j12 * MV[10];
  // cov2D = T * Sigma * T^T
  const float u0 = t00 * s00 + t01 * s01 + t02 * s02, u1 = t00 * s01 + t01 * s11 + t02 * s12,
              u2 = t00 * s02 + t01 * s12 + t02 * s22;
  const float v0 = t10 * s00 + t11 * s01 + t12 * s02, v1 = t10 * s01 + t11 * s11 + t12 * s12,
              v2 = t10 * s02 + t11 * s12 + t12 * s22;
  float a = u0 * t00 + u1 * t01 + u2 * t02;
  float b = u0 * t10 + u1 * t11 + u2 * t12;
  float d = v0 * t10 + v1 * t11 + v2 * t12;

  // ---- extent basis, threedgs.h.slang:60-121 -------------------------------------------------------------------
  float detOrig = 0.f;
  if(F.msAA)
    detOrig = a * d - b * b;
  a += 0.3f;
  d += 0.3f;
  if(F.msAA)
  {
    const float detBlur = a * d - b * b;
    col.w *= fastSqrt(fmaxf(detOrig * fastRcp(detBlur), 0.0f));
  }
  const float D     = a * d - b * b;
  const float half  = 0.5f * (a + d);
  const float term2 = fastSqrt(fmaxf(0.1f, half * half - D));
  float ev1 = half + term2, ev2 = half - term2;
  ok        = ok && !(ev2 <= 0.0f);
  if(F.debugFlags & 1)  // POINT_CLOUD_MODE, threedgs.h.slang:108-110
    ev1 = ev2 = 0.2f;
  float       e1x = (fabsf(b) < 0.001f) ? 1.0f : b, e1y = ev1 - a;
  const float el  = __builtin_amdgcn_rsqf(e1x * e1x + e1y * e1y);  // (1-ulp hardware instruction, like fastRcp / fastSqrt)
  e1x *= el;
  e1y *= el;
  const float kSqrt8 = 2.8284271247461903f;
  const float l1     = F.splatScale * fminf(kSqrt8 * fastSqrt(ev1), 2048.0f);
  const float l2     = F.splatScale * fminf(kSqrt8 * fastSqrt(fmaxf(ev2, 0.f)), 2048.0f);
  ok                 = ok && (l1 > 0.f && l2 > 0.f);
  const float b1x = e1x * l1, b1y = e1y * l1;   // basisVector1 (pixels)
  const float b2x = e1y * l2, b2y = -e1x * l2;  // basisVector2 = (e1.y, -e1.x) * l2

  // a fragment survives iff q = (d.p1)^2+(d.p2)^2 <= 4 (A<=8) and a*exp(-q) > 1/255 (frag.slang:242-262)
  const bool  noGauss = (F.debugFlags & 4) != 0;  // DISABLE_OPACITY_GAUSSIAN: alpha == 1 inside the ellipse
  const float a255    = col.w * 255.0f;
  ok                  = ok && (noGauss || a255 > 1.0f);
  const float qmax    = noGauss ? 4.0f : fminf(4.0f, __builtin_amdgcn_logf(fmaxf(a255, 1.0f)) * 0.6931471805599453f + 1e-3f);  // v_log_f32 (log2) * ln 2: the argument is >= 1, no denormal handling needed
  const float shrink = fastSqrt(qmax * 0.25f) * 1.0005f;
  const float ex = shrink * fastSqrt(b1x * b1x + b2x * b2x) + 0.01f;
  const float ey = shrink * fastSqrt(b1y * b1y + b2y * b2y) + 0.01f;

  const float pcx = (ndcx + 1.0f) * 0.5f * (float)F.width;
  const float pcy = (ndcy + 1.0f) * 0.5f * (float)F.height;
  // pixel centres (x+0.5) within [pc-e, pc+e]
  const float fx0 = ceilf(pcx - ex - 0.5f), fx1 = floorf(pcx + ex - 0.5f);
  const float fy0 = ceilf(pcy - ey - 0.5f), fy1 = floorf(pcy + ey - 0.5f);
  const float ymin = (float)(F.stripRow0 * kTilePx), ymax = (float)(min(F.stripRow1 * kTilePx, F.height) - 1);
  ok = ok && (fx1 >= fx0 && fy1 >= fy0 && fx1 >= 0.f && fx0 <= (float)(F.width - 1) && fy1 >= ymin && fy0 <= ymax);
  if(!ok)
    return false;  // nothing below touches memory except the caller's stores
  const int x0 = (int)fmaxf(fx0, 0.f), x1 = (int)fminf(fx1, (float)(F.width - 1));
  const int y0 = (int)fmaxf(fy0, ymin), y1 = (int)fminf(fy1, ymax);
  const int sx = 4 + F.binShiftX, sy = 4 + F.binShiftY;
  out.rect = (uint32_t)(x0 >> sx) | ((uint32_t)(y0 >> sy) << 8) | ((uint32_t)(x1 >> sx) << 16) | ((uint32_t)(y1 >> sy) << 24);

  // base colour, view direction and the SH sum (mesh.slang:205-207,240-243) are the compositor's business: it
  // shades the records it stages (kernels_common.h: viewDirection, k_raster.hip: shading phase)
  const float n1 = 2.0f * fastRcp(b1x * b1x + b1y * b1y), n2 = 2.0f * fastRcp(b2x * b2x + b2y * b2y);
  out.rec.cx  = pcx;
  out.rec.cy  = pcy;
  out.rec.p1x = b1x * n1;
  out.rec.p1y = b1y * n1;
  out.rec.p2x = b2x * n2;
  out.rec.p2y = b2y * n2;
  out.rec.a   = col.w;
  {  // cull-only extents, rounded up to fp16: conservative, so the frame does not depend on the rounding
    const __half2 h = __halves2half2(__float2half_ru(ex), __float2half_ru(ey));
    out.rec.exey    = *reinterpret_cast<const uint32_t*>(&h);
  }
  return true;
}

#ifdef MGS_PRJ_TRACE  // debug build (tools/prj_trace.py): per-workgroup wall-clock stamps (100 MHz) of the phases
__device__ uint64_t* g_prjTrace = nullptr;
#define MGS_PRJ_STAMP(i) if(threadIdx.x == 0) trc[i] = wall_clock64();
#else
#define MGS_PRJ_STAMP(i)
#endif
template <bool FULL>
__global__ __launch_bounds__(kPrjThreads, 6) void k_project(const FrameArgs* __restrict__ Ap, FrameCounters* __restrict__ ctr,
                                                         uint2* __restrict__ slotPairs, uint32_t* __restrict__ slotCount,
                                                         SplatRec* __restrict__ rec, uint32_t* __restrict__ rect,
                                                         uint32_t* __restrict__ slotHist2,
                                                         uint32_t* __restrict__ top16Rec, uint32_t* __restrict__ top16Count,
                                                         OsPlan* __restrict__ osPlan, const uint32_t* __restrict__ order)
{
  const FrameArgs& A = *Ap;  // frame constants live in device memory (same pointer every frame: graph-replayable)
#ifdef MGS_PRJ_TRACE
  __shared__ uint64_t trc[8];
  if(threadIdx.x < 8) trc[threadIdx.x] = 0;
  MGS_PRJ_STAMP(0)
#endif
  // One block of LDS, carved by hand, because the hand-over re-uses it (slot_emit.h: EmitLds):
  //   [0, 12 K)  s_rec   per wave: 64 records at a 48-byte pitch (phase 2)  | hand-over: the codes [0, 4 K), the per-wave digit
  //   [12, 16 K) s_li    bit 15: survived phase 2                           | counts [4 K, 6 K), then — [0, 16 K) — the slot itself
  //   [16, 24 K) s_key
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[12288 + 4096 + 8192];
  float4(*s_rec)[64 * 3] = reinterpret_cast<float4(*)[64 * 3]>(s_raw);
  uint16_t* s_li         = reinterpret_cast<uint16_t*>(s_raw + 12288);
  uint32_t* s_key        = reinterpret_cast<uint32_t*>(s_raw + 16384);
  __shared__ uint32_t s_small[256];  // hand-over: counts of key bits 8-15 [128 words], starts of the digit-0 groups [256 x 16 bit]
  __shared__ uint32_t s_cnt[32];
  __shared__ uint32_t s_base[33];
  EmitLds E;
  E.li    = s_li;
  E.key   = s_key;
  E.code  = reinterpret_cast<const uint16_t*>(s_raw);
  E.whist = reinterpret_cast<uint16_t*>(s_raw + 4096);
  E.hist1 = s_small;
  E.start = reinterpret_cast<uint16_t*>(s_small + 128);
  E.stage = reinterpret_cast<uint2*>(s_raw);
  E.cnt   = s_cnt;

  const int      t = threadIdx.x, lane = laneId(), w = t >> 6;
  // partitions are dispatched fullest slot of the PREVIOUS frame first (k_os_prepare leaves the order; scheduling only)
  const uint32_t part = order != nullptr ? order[blockIdx.x] : blockIdx.x;
  int            k    = 0;
  for(int i = 1; i < A.f.nInstances; ++i)  // instances own consecutive partition ranges
    if(part >= A.inst[i].blockBegin)
      k = i;
  const InstanceConst& I      = A.inst[k];
  const uint32_t       local0 = (part - I.blockBegin) * kPrjPart;

  const PartitionBox pbox = partitionLoad(I, part - I.blockBegin);  // ahead of the centres (partition_cull.h)
  // ---- phase 1: key + frustum cull for 8 splats per thread -------------------------------------------
  float px[kPrjItems], py[kPrjItems], pz[kPrjItems];
  // strips (multi-GPU): every splat's own footprint bound needs its largest axis (4 B per splat, requested beside the centres: a
  // load behind the dist stage would be one more dependent trip) — round 6, below
  const bool stripMode = FULL && A.f.partitionCull && (A.f.stripRow1 - A.f.stripRow0) < A.f.tilesY;
  [[maybe_unused]] float msc[kPrjItems];
#pragma unroll
  for(int it = 0; it < kPrjItems; ++it)
  {
    // clamped, not predicated: a predicated load becomes a branch + wait and serialises the 8 fetches
    const uint32_t li = min(local0 + it * kPrjThreads + t, I.count - 1u);
    px[it] = I.centers[3 * (size_t)li];
    py[it] = I.centers[3 * (size_t)li + 1];
    pz[it] = I.centers[3 * (size_t)li + 2];
    if constexpr(FULL)
    {
      msc[it] = 0.0f;
      if(stripMode)  // uniform
        msc[it] = I.maxScale[li];
    }
  }
  // the partition as a whole (partition_cull.h; behind the centres' loads, which are in flight either way): bit 0 no splat of
  // it can survive the cull / reach the strip, bit 1 every centre passes the frustum test, bit 2 all centres finite
  // (Strips: running the test BEFORE the centres are requested was measured in round 4 — a middle strip of eight skips only
  //  14 % of the partitions, 16 size classes make a partition's cell ~200 px tall on screen — 75 -> 78 us: the other 86 % pay a
  //  round trip for nothing.)
  float          partRadius = 3.0e38f;
  const uint32_t pflag      = A.f.partitionCull ? partitionTest(A, I, pbox, partRadius) : 0u;
  if(pflag & 1u)
  {
    emitEmptySlot<kPrjThreads>(slotCount, slotHist2, top16Rec, part);
    return;
  }
  uint32_t key[kPrjItems];
  uint64_t bal[kPrjItems];
  bool     vis[kPrjItems];
#ifdef MGS_PRJ_TRACE
  if(px[0] + py[1] + pz[2] == 12345.678f) trc[7] = 1;  // consume the loads: the stamp below follows their arrival
  MGS_PRJ_STAMP(1)
#endif
  const bool identityFast = (pflag & 4u) != 0u && I.modelIsIdentity != 0u;
  // affine model and view, perspective projection, finite and moderate coordinates: w stays exactly 1 through M and V and the
  // products with the matrices' zeros can be dropped without changing a bit (kernels_common.h: mulMat4ExactAffineW1 / mulPersp...)
  const bool w1Fast = (pflag & 12u) == 12u && A.f.perspAffine != 0 && (I.modelIsIdentity != 0u || I.modelIsAffine != 0u);
  // strips (multi-GPU): a splat whose centre lies further from this device's rows than the partition's footprint bound R
  // (partition_cull.h; valid for every splat of the partition) cannot reach them — dropped here, before the projection.
  // The exact footprint-vs-strip test of phase 2 would reject it anyway: the sorted set is unchanged.
  // Round 6: the bound is the SPLAT'S OWN — R_i = s (k_i f S sqrt(8) sigma_i / z_i + 3.2) with its own view depth z_i, its own
  // k_i = sqrt(2 + (x/z)^2 + (y/z)^2) (||J||_F = k f / z) and sigma_i = its largest axis (sqrt(lambda_max(Sigma3D)) exactly) — capped
  // by the partition's R (the same expression at the box's nearest corner with the partition's largest trace).  On a middle
  // strip of eight the partition's bound let 2.03 M candidates into the front end for 0.87 M that reach the rows
  // (profiles/r6_b_prj_trace_strip.log).
  const bool  stripPre = stripMode;
  const float stripR   = stripPre ? partRadius : 3.0e38f;
  const float stripK   = fmaxf(fabsf(A.f.focal[0]), fabsf(A.f.focal[1])) * I.modelScale * 2.8284271247461903f;
  const bool  gutFish  = A.f.cameraModel == 1 && A.f.pipeline == 1;  // (this kernel is the 3DGS one: never; kept for the rule's sake)
  const float stripY0  = (float)(A.f.stripRow0 * kTilePx), stripY1 = (float)min(A.f.stripRow1 * kTilePx, A.f.height);
  const bool  insideFast = (pflag & 2u) != 0u && A.f.cullMode == 1 && !stripPre;
#pragma unroll
  for(int it = 0; it < kPrjItems; ++it)
  {
    const uint32_t li = local0 + it * kPrjThreads + t;
    float          wp[4], vp[4], cp[4];
    if(identityFast)
    {  // M is bitwise the identity and the centres are finite: ((x*1 + y*0) + z*0) + 1*0 == x + 0.0f, bit for bit
#pragma clang fp contract(off)
      wp[0] = px[it] + 0.0f;
      wp[1] = py[it] + 0.0f;
      wp[2] = pz[it] + 0.0f;
      wp[3] = 1.0f;
    }
    else if(w1Fast)
    {
      mulMat4ExactAffineW1(I.model, px[it], py[it], pz[it], wp);  // dist.comp.slang:58
      wp[3] = 1.0f;
    }
    else
      mulMat4Exact(I.model, px[it], py[it], pz[it], 1.0f, wp);  // dist.comp.slang:58
    if(w1Fast)
    {
      mulMat4ExactAffineW1(A.f.view, wp[0], wp[1], wp[2], vp);   // :58
      vp[3] = 1.0f;
    }
    else
      mulMat4Exact(A.f.view, wp[0], wp[1], wp[2], wp[3], vp);   // :58
    bool  v = li < I.count;
    float nz;
    if(insideFast)
    {  // every centre of this partition passes :71-73 (all 8 corners of its box do, with margin, and the tests are linear
       // in the point): only clip z and w — the key — are needed
      if(w1Fast)
        mulPerspExactW1ZW(A.f.proj, vp[2], cp[2], cp[3]);
      else
        mulMat4ExactZW(A.f.proj, vp[0], vp[1], vp[2], vp[3], cp[2], cp[3]);
      nz = divExact(cp[2], cp[3]);
    }
    else
    {
      if(w1Fast)
      {
        mulPerspExactW1XY(A.f.proj, vp[0], vp[1], vp[2], cp[0], cp[1]);
        mulPerspExactW1ZW(A.f.proj, vp[2], cp[2], cp[3]);
      }
      else
        mulMat4Exact(A.f.proj, vp[0], vp[1], vp[2], vp[3], cp);   // :60
      const float nx = divExact(cp[0], cp[3]), ny = divExact(cp[1], cp[3]);  // :61
      nz             = divExact(cp[2], cp[3]);
      if(A.f.cullMode == 1 && distStageCulled(A.f, nx, ny, nz, vp[0], vp[1], vp[2]))  // :64-91, pinhole box or fisheye validity
        v = false;
      if(stripPre)
      {
        float R = stripR;
        if constexpr(FULL)
        {
          const float zv = -vp[2];  // view depth (the camera looks down -z)
          if(zv > 1.0e-4f && !gutFish)
          {
            const float iz = fastRcp(zv), qx = vp[0] * iz, qy = vp[1] * iz;
            const float kk = fastSqrt(2.0f + qx * qx + qy * qy);
            float       Ri = A.f.splatScale * (kk * stripK * msc[it] * iz + 3.2f);
            Ri             = fminf(Ri, 2897.0f * A.f.splatScale) * 1.01f + 2.0f;  // both bases are clamped at 2048 px
            R              = fminf(R, Ri);  // (a NaN Ri leaves the partition's bound)
          }
        }
        const float ypx = (ny + 1.0f) * 0.5f * (float)A.f.height;
        if(ypx + R < stripY0 || ypx - R > stripY1)
          v = false;
      }
    }
    if(A.f.sizeCulling && v)  // dist.comp.slang:93-134 (after the frustum test, like the shader)
      v = !sizeCulled(I.maxScale[min(li, I.count - 1u)], A.f.splatScale, I.modelAxisMax, vp[2], A.f.maxFocal,
                      A.f.sizeCullingMinPixels);
    vis[it] = v;
    key[it] = A.f.frontToBack ? encodeKey(nz) : encodeKey(-nz);  // :163-167
    bal[it] = __ballot(v);
    if(lane == 0)
      s_cnt[it * kPrjWaves + w] = (uint32_t)__popcll(bal[it]);
  }
  MGS_PRJ_STAMP(2)
  const uint32_t M = scanRoundWaveCounts(s_cnt, s_base);
#pragma unroll
  for(int it = 0; it < kPrjItems; ++it)
    if(vis[it])
    {
      const uint32_t pos = s_base[it * kPrjWaves + w] + lanesBelow(bal[it]);
      s_li[pos]          = (uint16_t)(it * kPrjThreads + t);
      s_key[pos]         = key[it];
    }
  __syncthreads();
  if(t == 0 && M)
    atomicAdd(&frameStatSlotFromOs(osPlan, part)[2], M);  // (sort_plan.h: 32 slots on 32 lines, not the counters' one line)

  MGS_PRJ_STAMP(3)
  if(M == 0u)
  {  // nothing of this partition survives the dist stage (a strip of the multi-GPU partition: 18 % of the workgroups that pass the
     // box test, profiles/r6_b_prj_trace_strip.log): the slot is empty — no front end, no hand-over (four barriers, 3 us)
    emitEmptySlot<kPrjThreads>(slotCount, slotHist2, top16Rec, part);
    return;
  }
  if constexpr(!FULL)
  {
    // sort-only hook: survivors of the dist stage, exactly dist.comp.slang's (key, id) stream
    E.code = nullptr;
    emitSlot<kPrjThreads, kPrjItems>(M, true, E, slotPairs, slotCount, slotHist2, top16Rec, top16Count, osPlan, ctr, part, I.globalOffset + local0);
    return;
  }
  else
  {
    // ---- phase 2: dense raster front end over the survivors (no barriers inside) ------------------------
    // The 32-byte records leave through LDS: every lane computes one record, then the wave stores them two
    // lanes per record, so a store instruction covers whole 64-byte sectors wherever neighbouring ids both survive
    // (32 records = 1 KB contiguous when the survivors are).  Written lane-per-record, each instruction put 16 bytes into 64 different sectors and
    // the kernel spent half its time on those partial-sector writes (0.20 ms -> 0.11 ms with the stores removed).
    // The fetches of the NEXT batch of 256 survivors are issued before this batch is computed (13 registers):
    // otherwise every batch starts with a full memory round trip and the kernel spent half its wave cycles parked.
    SplatFetch cur;
    uint32_t   liCur = 0;
    if(t < M)
    {
      liCur = local0 + s_li[t];
      cur   = fetchSplat(I, liCur);
    }
    static_assert(kPrjItems == 8, "RideCodes holds eight rounds");
    RideCodes      codes;
    bool           anySurvivor = false;
    const uint32_t rideShift = (uint32_t)A.f.rideShift;
    for(uint32_t j0 = 0; j0 < M; j0 += kPrjThreads)
    {
      uint32_t       code = 0u;
      const uint32_t j    = j0 + t;
      uint32_t       gidOk = 0xFFFFFFFFu;
      SplatFetch     nxt;
      uint32_t       liNxt = 0;
      const bool     haveNxt = j + kPrjThreads < M;
      liNxt = local0 + s_li[haveNxt ? j + kPrjThreads : 0u];
      liNxt = min(liNxt, I.count - 1u);
      nxt   = fetchSplat(I, liNxt);  // clamped, not predicated: straight-line loads
      if(j < M)
      {
        const uint32_t li = liCur;
        Projected      pr;
        if(projectSplat(A.f, I, k, cur, pr))
        {
          gidOk       = I.globalOffset + li;
          float4* dst = &s_rec[w][lane * 3];  // 48-byte pitch: conflict-free 16-byte accesses
          dst[0]      = make_float4(pr.rec.cx, pr.rec.cy, pr.rec.p1x, pr.rec.p1y);
          dst[1]      = make_float4(pr.rec.p2x, pr.rec.p2y, pr.rec.a, __uint_as_float(pr.rec.exey));
          if(rideShift != 0u)
            code = rideEncode(pr.rect, A.f.binsX, A.f.binsY, A.f.rideShapes, A.f.rideEscape);
          // the rectangle by id is read back only where the code cannot say it (escape), or when nothing rides: four scattered
          // bytes per survivor less
          if(rideShift == 0u || code == A.f.rideEscape)
            rect[gidOk] = pr.rect;
          s_li[j] |= 0x8000u;  // own entry only: no race
          anySurvivor = true;
        }
      }
      // the id rides in the pitch's third (padding) quad: a separate 1 KB array was what kept the workgroup above 26 KB of LDS
      // and the CU at 5 resident workgroups instead of 6
      reinterpret_cast<uint32_t*>(&s_rec[w][lane * 3 + 2])[0] = gidOk;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for(int i = 0; i < 2; ++i)
      {
        const int      rr = 32 * i + (lane >> 1), part = lane & 1;
        const uint32_t g  = reinterpret_cast<const uint32_t*>(&s_rec[w][rr * 3 + 2])[0];
        if(g != 0xFFFFFFFFu)
          reinterpret_cast<float4*>(rec + g)[part] = s_rec[w][rr * 3 + part];
      }
      __builtin_amdgcn_wave_barrier();
      cur   = nxt;
      liCur = liNxt;
      codes.push(code);
    }
    // the codes' slots are counted from the LAST of eight rounds: the rounds behind the survivors only shift
    for(uint32_t j0 = (M + kPrjThreads - 1u) / kPrjThreads * kPrjThreads; j0 < (uint32_t)kPrjPart; j0 += kPrjThreads)
      codes.push(0u);
    // ---- the survivors, grouped by the low byte of their keys, into the partition's slot (slot_emit.h) ----
    // (phase 2 is over behind this barrier: s_rec is free.  The barrier also tells whether ANY candidate survived the front end: in a
    //  strip 37 % of the workgroups hand nothing to the sort — every candidate's footprint misses the rows — and skip the hand-over)
    const bool anyOut = __syncthreads_or(anySurvivor ? 1 : 0) != 0;
    MGS_PRJ_STAMP(4)
    if(!anyOut)
    {
      emitEmptySlot<kPrjThreads>(slotCount, slotHist2, top16Rec, part);
#ifdef MGS_PRJ_TRACE
      MGS_PRJ_STAMP(5)
      if(threadIdx.x == 0 && g_prjTrace)
      {
        uint64_t* o = g_prjTrace + (size_t)blockIdx.x * 8;
        for(int i = 0; i < 6; ++i) o[i] = trc[i];
        o[6] = M;
        o[7] = 0;
      }
#endif
      return;
    }
    if(rideShift != 0u)
    {  // the codes leave their registers for the hand-over, which deals the candidates out anew (wave-contiguous)
      uint16_t* s_code = reinterpret_cast<uint16_t*>(s_raw);
#pragma unroll
      for(int r = 0; r < kPrjItems; ++r)
        s_code[r * kPrjThreads + t] = (uint16_t)codes.get(r);
    }
    const uint32_t outCount = emitSlot<kPrjThreads, kPrjItems>(M, false, E, slotPairs, slotCount, slotHist2, top16Rec, top16Count, osPlan, ctr, part,
                                                               I.globalOffset + local0, rideShift | (A.f.rideSplit ? 0x100u : 0u));
    (void)outCount;
#ifdef MGS_PRJ_TRACE
    MGS_PRJ_STAMP(5)
    if(threadIdx.x == 0 && g_prjTrace)
    {
      uint64_t* o = g_prjTrace + (size_t)blockIdx.x * 8;  // (dispatch order)
      for(int i = 0; i < 6; ++i) o[i] = trc[i];
      o[6] = M;
      o[7] = outCount;
    }
#endif
  }
}


// ---------------------------------------------------------------------------------------------
// host-callable launcher
void launchProject(hipStream_t stream, const FrameArgs& args, const FrameArgs* dArgs, bool full, FrameCounters* ctr, uint2* slotPairs,
                   uint32_t* slotCount, SplatRec* rec, uint32_t* rect, uint32_t* slotHist2,
                   uint32_t* top16Rec, uint32_t* top16Count, OsPlan* osPlan, const uint32_t* order)
{
  const dim3 grid(args.f.totalPartitions), block(kPrjThreads);
  if(args.f.totalPartitions == 0)
    return;
#ifdef MGS_PRJ_TRACE
  static uint64_t* traceBuf = nullptr;
  const char*      tracePath = std::getenv("MGS_PRJ_TRACE_FILE");
  const size_t     traceN = (size_t)args.f.totalPartitions * 8;
  if(tracePath && full)
  {
    if(!traceBuf)
    {
      (void)hipMalloc(&traceBuf, (size_t)1 << 24);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prjTrace), &traceBuf, sizeof(traceBuf));
    }
    (void)hipMemsetAsync(traceBuf, 0, traceN * 8, stream);
  }
#endif
#define MGS_LAUNCH(FULLV)                                                                                                \
  hipLaunchKernelGGL((k_project<FULLV>), grid, block, 0, stream, dArgs, ctr, slotPairs, slotCount, rec, rect, slotHist2, \
                     top16Rec, top16Count, osPlan, order)
  if(full)
    MGS_LAUNCH(true);
  else
    MGS_LAUNCH(false);
#undef MGS_LAUNCH
#ifdef MGS_PRJ_TRACE
  if(tracePath && full)
  {
    (void)hipStreamSynchronize(stream);
    std::vector<uint64_t> h(traceN);
    (void)hipMemcpy(h.data(), traceBuf, traceN * 8, hipMemcpyDeviceToHost);
    if(FILE* fp = std::fopen(tracePath, "wb"))
    {
      std::fwrite(h.data(), 8, traceN, fp);
      std::fclose(fp);
    }
  }
#endif
}

}  // namespace mgs
