// slot_emit.h — the hand-over from the project kernels (k_project.hip, k_gut.hip) to the frame's key sort (k_osort.hip).
// A project workgroup owns one 2048-splat partition.  After its raster front end it compacts the splats that can produce a
// fragment, in ascending id, and appends them as interleaved (key, id) pairs to ONE dense array: the sort's first pass reads
// full partitions (with per-partition slots it ran 1 424 workgroups that were 70 % full — two residency waves — against
// 1 020 full ones).  The position of a workgroup's pairs is the number of pairs of the workgroups before it, resolved
// inside the kernel by a two-level look-back of fan-in 64 (a wave reads a whole group of status words with one load, a
// second one covers 64 groups): workgroup = blockIdx.x, it waits only for lower-numbered ones, which the dispatcher has
// started (k_osort.hip's header has the argument); bounded spins, kErrSpinTimeout instead of a hang.
// It also leaves what the sort needs before its first pass, computed while the keys are still on chip:
//   * slotHist2[part][2][256]: the partition's histograms of key bits 0-7 and 8-15 (k_os_prepare reduces them to digit totals);
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

// Number of pairs of the workgroups before `part`.  Called by every thread of the workgroup (uniform); wave 0 works, the result
// comes back through s_bcast[0] behind a barrier.  status = [parts] workgroup words, then the group words.
__device__ __forceinline__ uint32_t prjExclusiveBase(uint32_t* __restrict__ status, uint32_t parts, uint32_t part, uint32_t count,
                                                     uint32_t* s_bcast, FrameCounters* __restrict__ ctr)
{
  if(threadIdx.x < 64)
  {
    const uint32_t lane = threadIdx.x, g = part / kPrjGroup, m = part % kPrjGroup;
    uint32_t*      gst  = status + parts;
    if(lane == 0)
      prjSt(&status[part], kPrjAgg | count);
    uint32_t spins = 0, intra = 0, base = 0;
    bool     bad = false;
    for(;;)
    {  // level 1: the members of my group before me
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
    if(m == kPrjGroup - 1u)
    {  // the group's last member: total -> aggregate; look back over the groups, 64 per load, down to the first inclusive prefix
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
        prjSt(&gst[g], kPrjInc | ((base + total) & kPrjMask));
    }
    else if(g > 0u)
    {
      uint32_t v;
      while(((v = prjLd(&gst[g - 1u])) >> 30) != 2u)
        if(++spins > kPrjSpinMax)
        {
          bad = true;
          break;
        }
      base = v & kPrjMask;
    }
    if(lane == 0)
    {
      s_bcast[0] = base + intra;
      if(bad)
        atomicOr(&ctr->errorFlags, kErrSpinTimeout);
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
  __shared__ uint32_t s_b[1];
  for(int i = threadIdx.x; i < 512; i += THREADS)
    slotHist2[(size_t)part * 512u + i] = 0u;
  if(threadIdx.x < THREADS / 64)
    top16Rec[((size_t)part * (THREADS / 64) + threadIdx.x) * 32u + 31u] = 0xFFFFFFFFu;
  (void)prjExclusiveBase(prjStatus, parts, part, 0u, s_b, ctr);
}

// Returns the number of pairs written.  s_li[j] bit 15 marks a survivor of the front end (ignored when allSurvive); s_key[j]
// its depth key; s_hist2 must be zero.
template <int THREADS, int ITEMS>
__device__ __forceinline__ uint32_t emitSlot(uint32_t M, bool allSurvive, const uint16_t* s_li, const uint32_t* s_key, uint32_t* s_cnt /*32*/,
                                             uint32_t* s_base /*33*/, uint32_t* s_hist2 /*512*/, uint2* __restrict__ densePairs,
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
  // where this workgroup's pairs go (s_cnt is free again: its sums live in s_base)
  const size_t dst0 = prjExclusiveBase(prjStatus, parts, part, outCount, s_cnt, ctr);
  uint32_t     top[ITEMS];
  uint32_t     tmn = 0xFFFFu, tmx = 0u;
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
  {
    top[r] = 0xFFFFFFFFu;
    if(vis[r])
    {
      const uint32_t j   = r * THREADS + t;
      const uint32_t pos = s_base[r * WAVES + w] + lanesBelow(bal[r]);
      const uint32_t key = s_key[j];
      densePairs[dst0 + pos] = make_uint2(key, idBase + (uint32_t)(s_li[j] & 0x7FFFu));
      atomicAdd(&s_hist2[key & 255u], 1u);
      atomicAdd(&s_hist2[256u + ((key >> 8) & 255u)], 1u);
      top[r] = key >> 16;
      tmn    = min(tmn, top[r]);
      tmx    = max(tmx, top[r]);
    }
  }
  __syncthreads();  // s_cnt: everybody has read the base before the per-wave min / max land in it
  sortTop16Post<WAVES>(tmn, tmx, s_cnt);
  if(t == 0 && outCount)
    atomicAdd(&ctr->sortedCount, outCount);
  __syncthreads();
  for(int i = t; i < 512; i += THREADS)
    slotHist2[(size_t)part * 512u + i] = s_hist2[i];
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
      const bool inLds = hi - lo < 512u;
      if(inLds)
      {
        __syncthreads();
        for(int i = t; i < 512; i += THREADS)
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
  return outCount;
}

}  // namespace mgs
