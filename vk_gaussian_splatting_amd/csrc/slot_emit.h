// slot_emit.h — the hand-over from the project kernels (k_project.hip, k_gut.hip) to the frame's key sort (k_osort.hip).
// A project workgroup owns one 2048-splat partition.  After its raster front end it compacts the splats that can produce a
// fragment, in ascending id, and appends them as interleaved (key, id) pairs to ONE dense array: the sort's first pass reads
// full partitions (with per-partition slots it ran 1 424 workgroups that were 70 % full — two residency waves — against
// 1 020 full ones).  The position of a workgroup's pairs is the number of entries reserved by the workgroups before it, resolved
// inside the kernel by a two-level look-back of fan-in 64 (a wave reads a whole group of status words with one load, a
// second one covers 64 groups): workgroup = blockIdx.x, it waits only for lower-numbered ones, which the dispatcher has
// started (k_osort.hip's header has the argument); bounded spins, kErrSpinTimeout instead of a hang.
// It also leaves what the sort needs before its first pass, computed while the keys are still on chip:
//   * slotHist2[part][256]: the partition's histograms of key bits 0-7 and 8-15, two 16-bit counters per word (k_os_prepare
//     reduces them to digit totals);
//   * top16Rec[part][wave][32]: how often each value of key >> 16 occurs in the wave (the totals of the upper passes and the
//     pass-elision decision come from these; k_os_prepare folds them).
#pragma once
#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

constexpr uint32_t kPrjAgg     = 1u << 30, kPrjInc = 2u << 30, kPrjMask = (1u << 30) - 1u;
constexpr uint32_t kPrjSpinMax = 1u << 21;

__device__ __forceinline__ uint32_t prjLd(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void     prjSt(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- where a workgroup's pairs go: the entries reserved by the workgroups before it -----------------------------------
// status = [parts] workgroup words (flag | count), then the group words (flag | group total, later | inclusive prefix).
// prjReserve   as soon as the workgroup knows how many pairs it has: publishes the count.  The LAST member of a group of 64
//              also does the group's work there and then: sums its group (one wave-wide load, re-polled until all 63 are
//              in), publishes the group total, looks back over the groups (64 per load, down to the first inclusive
//              prefix), publishes the inclusive prefix.
// prjIssue /   ONE round of loads — the words of the members before me and the inclusive prefix of the group before mine —
// prjPosition  issued right behind the reservation and consumed behind the workgroup's other hand-overs.
// (Reserving by the dist-stage survivor count, known a whole raster front end earlier, and filling the entries the front end
//  rejects with holes was measured: k_project did not get faster — the waits are not its bottleneck — and the sort's first pass
//  got 1 059 instead of 1 020 partitions: one more than a residency wave, +8 us.)
// Called by every thread (uniform control flow); wave 0 works.  s_keep: 2 words that live from prjReserve to prjPosition.
__device__ __forceinline__ void prjReserve(uint32_t* __restrict__ status, uint32_t parts, uint32_t part, uint32_t count, uint32_t* s_keep,
                                           FrameCounters* __restrict__ ctr)
{
  if(threadIdx.x >= 64)
    return;
  const uint32_t lane = threadIdx.x, g = part / kPrjGroup, m = part % kPrjGroup;
  uint32_t*      gst  = status + parts;
  if(lane == 0)
    prjSt(&status[part], kPrjAgg | count);
  if(m != kPrjGroup - 1u)
    return;
  uint32_t spins = 0, intra = 0, base = 0;
  bool     bad = false;
  for(;;)
  {
    const uint32_t v = (lane < m) ? prjLd(&status[g * kPrjGroup + lane]) : kPrjAgg;
    if(__all((v >> 30) != 0u))
    {
      intra = waveSum(v & kPrjMask);
      break;
    }
    if(++spins > kPrjSpinMax)
    {
      bad = true;
      break;
    }
  }
  const uint32_t total = intra + count;
  if(lane == 0)
    prjSt(&gst[g], kPrjAgg | total);
  int q = (int)g - 1;
  while(q >= 0 && !bad)
  {
    const int      idx = q - (int)lane;
    const uint32_t v   = idx >= 0 ? prjLd(&gst[idx]) : kPrjInc;  // before group 0: inclusive 0
    const uint64_t inc = __ballot((v >> 30) == 2u), nr = __ballot((v >> 30) == 0u);
    const int      fi  = inc ? __builtin_ctzll(inc) : 64;       // nearest group with an inclusive prefix
    const int      fn  = nr ? __builtin_ctzll(nr) : 64;         // nearest group that has not published
    if(fn < fi)
    {
      if(++spins > kPrjSpinMax)
        bad = true;
      continue;
    }
    base += waveSum(((int)lane <= fi) ? (v & kPrjMask) : 0u);
    q = (fi < 64) ? -1 : q - 64;
  }
  if(lane == 0)
  {
    prjSt(&gst[g], kPrjInc | ((base + total) & kPrjMask));
    s_keep[0] = base + intra;  // this workgroup's own position
    if(bad)
      atomicOr(&ctr->errorFlags, kErrSpinTimeout);
  }
}

// lane l < m: the word of member l of my group; lane 63: the inclusive prefix of the group before mine (m <= 62 here)
__device__ __forceinline__ uint32_t prjIssue(const uint32_t* __restrict__ status, uint32_t parts, uint32_t part)
{
  const uint32_t lane = threadIdx.x, g = part / kPrjGroup, m = part % kPrjGroup;
  if(lane >= 64 || m == kPrjGroup - 1u)
    return 0u;
  if(lane < m)
    return prjLd(&status[g * kPrjGroup + lane]);
  if(lane == 63)
    return g > 0u ? prjLd(&status[parts + g - 1u]) : kPrjInc;
  return kPrjAgg;
}

__device__ __forceinline__ uint32_t prjPosition(const uint32_t* __restrict__ status, uint32_t parts, uint32_t part, uint32_t v,
                                                const uint32_t* s_keep, uint32_t* s_bcast, FrameCounters* __restrict__ ctr)
{
  if(threadIdx.x < 64)
  {
    const uint32_t lane = threadIdx.x, m = part % kPrjGroup;
    if(m == kPrjGroup - 1u)
    {
      if(lane == 0)
        s_bcast[0] = s_keep[0];
    }
    else
    {
      uint32_t spins = 0;
      bool     bad   = false;
      while(!__all(lane == 63 ? (v >> 30) == 2u : (v >> 30) != 0u))
      {
        if(++spins > kPrjSpinMax)
        {
          bad = true;
          break;
        }
        v = prjIssue(status, parts, part);
      }
      const uint32_t sum = waveSum(v & kPrjMask);
      if(lane == 0)
      {
        s_bcast[0] = sum;
        if(bad)
          atomicOr(&ctr->errorFlags, kErrSpinTimeout);
      }
    }
  }
  __syncthreads();
  return s_bcast[0];
}

// A partition that is skipped as a whole still takes part: count 0 in the look-back, zero histograms, empty records.
template <int THREADS>
__device__ __forceinline__ void emitEmptySlot(uint32_t* __restrict__ prjStatus, uint32_t parts, uint32_t* __restrict__ slotHist2,
                                              uint32_t* __restrict__ top16Rec, FrameCounters* __restrict__ ctr, uint32_t part)
{
  __shared__ uint32_t s_b[2];
  for(int i = threadIdx.x; i < 256; i += THREADS)
    slotHist2[(size_t)part * 256u + i] = 0u;
  if(threadIdx.x < THREADS / 64)
    top16Rec[((size_t)part * (THREADS / 64) + threadIdx.x) * 32u + 31u] = 0xFFFFFFFFu;
  prjReserve(prjStatus, parts, part, 0u, s_b, ctr);  // nothing to place, but a group's last member owes the group its prefix
}

// Returns the number of pairs written.  s_li[j] bit 15 marks a survivor of the front end (ignored when allSurvive); s_key[j]
// its depth key; s_hist2 must be zero.
template <int THREADS, int ITEMS>
__device__ __forceinline__ uint32_t emitSlot(uint32_t M, bool allSurvive, const uint16_t* s_li, const uint32_t* s_key, uint32_t* s_cnt /*32*/,
                                             uint32_t* s_base /*33*/, uint32_t* s_hist2 /*256, zero*/, uint32_t* s_keep /*2 words*/,
                                             uint2* __restrict__ densePairs,
                                             uint32_t* __restrict__ prjStatus, uint32_t parts, uint32_t* __restrict__ slotHist2,
                                             uint32_t* __restrict__ top16Rec, uint32_t* __restrict__ top16Count, OsPlan* __restrict__ osPlan,
                                             FrameCounters* __restrict__ ctr, uint32_t part, uint32_t idBase)
{
  constexpr int WAVES = THREADS / 64;
  static_assert(WAVES * ITEMS == 32, "the round x wave table of scanRoundWaveCounts has 32 entries");
  static_assert(WAVES == 4, "k_os_prepare folds four wave records per slot");
  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  bool      vis[ITEMS];
  uint64_t  bal[ITEMS];
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
  {
    const uint32_t j = r * THREADS + t;
    vis[r]           = (j < M) && (allSurvive || (s_li[j] & 0x8000u));
    bal[r]           = __ballot(vis[r]);
    if(lane == 0)
      s_cnt[r * WAVES + w] = (uint32_t)__popcll(bal[r]);
  }
  const uint32_t outCount = scanRoundWaveCounts(s_cnt, s_base);
  // the count goes out at once (a group's last member also resolves the group's prefix here); this workgroup's own position
  // is one round of loads, issued now and consumed behind its other hand-overs, right before the pairs are stored
  prjReserve(prjStatus, parts, part, outCount, s_keep, ctr);
  const uint32_t lbv = prjIssue(prjStatus, parts, part);
  uint32_t top[ITEMS], pos[ITEMS];
  uint32_t tmn = 0xFFFFu, tmx = 0u;
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
  {
    top[r] = 0xFFFFFFFFu;
    pos[r] = 0u;
    if(vis[r])
    {
      const uint32_t key = s_key[r * THREADS + t];
      pos[r]             = s_base[r * WAVES + w] + lanesBelow(bal[r]);
      // two 16-bit counters per word (a partition holds <= 2048 keys): the table is 1 KB, which keeps six workgroups per CU
      atomicAdd(&s_hist2[(key & 255u) >> 1], 1u << (16u * (key & 1u)));
      atomicAdd(&s_hist2[128u + (((key >> 8) & 255u) >> 1)], 1u << (16u * ((key >> 8) & 1u)));
      top[r] = key >> 16;
      tmn    = min(tmn, top[r]);
      tmx    = max(tmx, top[r]);
    }
  }
  sortTop16Post<WAVES>(tmn, tmx, s_cnt);
  if(t == 0 && outCount)
    atomicAdd(&ctr->sortedCount, outCount);
  __syncthreads();
  for(int i = t; i < 256; i += THREADS)
    slotHist2[(size_t)part * 256u + i] = s_hist2[i];  // packed as counted: bins 2 i and 2 i + 1 of digit i >> 7
  uint32_t lo, hi;
  sortTop16Range<WAVES>(outCount, s_cnt, lo, hi);
  // this wave's record of key >> 16: counts of lo .. lo + 24 in words 0-24, header in word 31
  uint32_t hdr = 0xFFFFFFFFu, myc = 0u;
  if(hi >= lo)
  {
    if(hi - lo <= 24u)
    {  // the usual case, 1-3 values: count with ballots (wave-uniform results), lane v - lo keeps the count of v
      hdr = lo | ((hi - lo) << 16);
      for(uint32_t v = lo; v <= hi; ++v)
      {
        uint32_t c = 0;
#pragma unroll
        for(int r = 0; r < ITEMS; ++r)
          c += (uint32_t)__popcll(__ballot(top[r] == v));
        if((uint32_t)lane == v - lo)
          myc = c;
      }
    }
    else if(top16Count != nullptr)
    {  // a cell around the camera: many values.  Counted in LDS (s_hist2 has been stored; lo / hi are the same for every
       // thread, so the barriers are uniform), one atomic per occurring value into the count table, and the occurring range
      // (top16Count == nullptr: this frame's key sort does not run — CPU sorting — and nobody would consume the counts)
      const bool inLds = hi - lo < 256u;
      if(inLds)
      {
        __syncthreads();
        for(int i = t; i < 256; i += THREADS)
          s_hist2[i] = 0u;
        __syncthreads();
      }
#pragma unroll
      for(int r = 0; r < ITEMS; ++r)
        if(vis[r])
          atomicAdd(inLds ? &s_hist2[top[r] - lo] : &top16Count[top[r]], 1u);
      if(inLds)
      {
        __syncthreads();
        for(uint32_t i = t; i <= hi - lo; i += THREADS)
          if(s_hist2[i])
            atomicAdd(&top16Count[lo + i], s_hist2[i]);
      }
      if(t == 0)
      {
        atomicMax(&osPlan->top16MinInv, 0x10000u - lo);
        atomicMax(&osPlan->top16MaxP1, hi + 1u);
      }
    }
  }
  if(lane < 32)
    top16Rec[((size_t)part * WAVES + w) * 32u + lane] = (lane == 31) ? hdr : myc;
  // where this workgroup's pairs go (s_base is free: the positions are in registers and a barrier lies behind their last read)
  const size_t dst0 = prjPosition(prjStatus, parts, part, lbv, s_keep, s_base, ctr);
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
    if(vis[r])
    {
      const uint32_t j          = r * THREADS + t;
      densePairs[dst0 + pos[r]] = make_uint2(s_key[j], idBase + (uint32_t)(s_li[j] & 0x7FFFu));
    }
  return outCount;
}

}  // namespace mgs
