// slot_emit.h — the hand-over from the project kernels (k_project.hip, k_gut.hip) to the frame's key sort (k_osort.hip).
// A project workgroup owns one 2048-splat partition.  After its raster front end it compacts the splats that can produce a
// fragment, in ascending id, into ITS OWN SLOT of the pair array — slot p = entries [2048 p, 2048 p + count_p) — and leaves
// count_p.  Nothing here waits for another workgroup: the sort's first pass reads DENSE partitions of 4096 pairs by gathering
// them from the slots through the exclusive prefix of the counts, which k_os_prepare computes on the side (k_osort.hip).
// (Round 3 first appended to one dense array from inside this kernel, ordered by an in-kernel look-back.  A workgroup's
//  position needs the exact counts of ALL workgroups before it, so everybody waited for the slowest early partition with
//  its CU slot held: 19 us of a 63 us workgroup lifetime on a full frame, and of a 45 us one on a multi-GPU strip, where
//  that wait was most of the kernel — tools/prj_trace.py.)
// It also leaves what the sort needs before its first pass, computed while the keys are still on chip:
//   * slotHist2[part][256]: the partition's histograms of key bits 0-7 and 8-15, two 16-bit counters per word (k_os_prepare
//     reduces them to digit totals);
//   * top16Rec[part][wave][32]: how often each value of key >> 16 occurs in the wave (the totals of the upper passes and the
//     pass-elision decision come from these; k_os_prepare folds them).
#pragma once
#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

// A partition that is skipped as a whole still takes part: count 0, zero histograms, empty records.
template <int THREADS>
__device__ __forceinline__ void emitEmptySlot(uint32_t* __restrict__ slotCount, uint32_t* __restrict__ slotHist2,
                                              uint32_t* __restrict__ top16Rec, uint32_t part)
{
  for(int i = threadIdx.x; i < 256; i += THREADS)
    slotHist2[(size_t)part * 256u + i] = 0u;
  if(threadIdx.x < THREADS / 64)
    top16Rec[((size_t)part * (THREADS / 64) + threadIdx.x) * 32u + 31u] = 0xFFFFFFFFu;
  if(threadIdx.x == 0)
    slotCount[part] = 0u;
}

// Returns the number of pairs written.  s_li[j] bit 15 marks a survivor of the front end (ignored when allSurvive); s_key[j]
// its depth key; s_hist2 must be zero.
template <int THREADS, int ITEMS>
__device__ __forceinline__ uint32_t emitSlot(uint32_t M, bool allSurvive, const uint16_t* s_li, const uint32_t* s_key, uint32_t* s_cnt /*32*/,
                                             uint32_t* s_base /*33*/, uint32_t* s_hist2 /*256, zero*/,
                                             uint2* __restrict__ slotPairs,
                                             uint32_t* __restrict__ slotCount, uint32_t* __restrict__ slotHist2,
                                             uint32_t* __restrict__ top16Rec, uint32_t* __restrict__ top16Count, OsPlan* __restrict__ osPlan,
                                             FrameCounters* __restrict__ ctr, uint32_t part, uint32_t idBase,
                                             uint32_t rideShift = 0u, const RideCodes* codes = nullptr,
                                             const uint16_t* s_code = nullptr /* [2048] in LDS, instead of `codes` */)
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
  if(t == 0)
  {  // k_os_prepare turns the counts into the slots' positions in the dense order and their sum into ctr->sortedCount; a frame
    // whose key sort does not run (CPU sorting) counts here
    slotCount[part] = outCount;
    if(top16Count == nullptr && outCount)
      atomicAdd(&ctr->sortedCount, outCount);
  }
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
  const size_t dst0 = (size_t)part * (size_t)(THREADS * ITEMS);  // this partition's slot
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
    if(vis[r])
    {
      const uint32_t j         = r * THREADS + t;
      uint32_t id = idBase + (uint32_t)(s_li[j] & 0x7FFFu);
      if(rideShift != 0u)  // the bin rectangle's code rides through the sort above the id (kernels_common.h: rideEncode)
        id |= (s_code != nullptr ? (uint32_t)s_code[j] : codes->get(r)) << rideShift;
      slotPairs[dst0 + pos[r]] = make_uint2(s_key[j], id);
    }
  return outCount;
}

}  // namespace mgs
