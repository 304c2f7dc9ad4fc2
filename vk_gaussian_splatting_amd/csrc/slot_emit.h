// slot_emit.h — the hand-over from the project kernels (k_project.hip, k_gut.hip) to the frame's key sort (k_osort.hip).
// A project workgroup owns one 2048-splat partition.  After its raster front end it hands the splats that can produce a
// fragment to the sort in ITS OWN SLOT of the pair array — slot p = entries [2048 p, 2048 p + count_p) — and leaves count_p.
// Nothing here waits for another workgroup.
//
// Round 4: the slot is written GROUPED BY THE LOW BYTE OF THE KEY (a stable multi-split of <= 2048 pairs in LDS, ascending
// id inside a group), and the slot's digit-0 group starts go out with it.  That IS the first pass of the LSD sort, done
// while the keys are on chip: where pair (slot s, digit d, i-th of its group) stands after a stable pass on bits 0-7 is
//     D[d] + sum over slots s' < s of count[s'][d] + i        (D = exclusive scan of the digit totals),
// a function of the groups' counts alone — so that pass is never run: the sort's first kernel (bits 8-15) reads its dense
// partitions straight from the slots, in digit-0 order, through k_os_prepare's tables of those sums (k_osort.hip, "virtual
// pass 0").  Round 3 wrote the slot in id order and ran pass 0 as a kernel of its own (38 us of a 118 us sort); the ranking
// it did there (8 ballots per key) is done here instead, and its load / look-back / re-order / scatter are gone.
//
// What a slot leaves besides its pairs (slotHist[part][384], 16-bit counters, two per word):
//   words   0-127  counts of key bits 0-7  (digit-0 groups)       -> chunk sums, digit totals
//   words 128-255  counts of key bits 8-15                          -> digit totals of the sort's first real pass
//   words 256-383  exclusive starts of the digit-0 groups inside the slot
//   top16Rec[part][wave][32]: how often each value of key >> 16 occurs in the wave (the totals of the upper passes and the
//   pass-elision decision come from these; k_os_prepare folds them).
#pragma once
#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

// Among the lanes of `act`, those that hold the same 8-bit digit as this lane: `lower` = how many of them are below it,
// `cnt` = how many there are (stable multi-split inside a wave).  Per bit: x = 0 / ~0 from the bit (one bfe), the ballot
// of the bit, and mask &= ~(ballot ^ x) on both halves as one v_bitop3 each (truth table 0x90: a & !(b ^ c)).
__device__ __forceinline__ void waveMatch8(uint32_t d, uint64_t act, uint32_t& lower, uint32_t& cnt)
{
  uint32_t mlo = (uint32_t)act, mhi = (uint32_t)(act >> 32);
#pragma unroll
  for(int b = 0; b < 8; ++b)
  {
    uint32_t x = (uint32_t)((int32_t)(d << (31 - b)) >> 31);
    asm volatile("" : "+v"(x));  // the ballot compares x itself: left alone the compiler re-derives it from d (a shift per bit)
    const uint64_t bal = __ballot(x != 0u);
    mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bal, x, 0x90);
    mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bal >> 32), x, 0x90);
  }
  lower = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
  cnt   = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
}

// A partition that is skipped as a whole still takes part: count 0, zero histograms, empty records.
template <int THREADS>
__device__ __forceinline__ void emitEmptySlot(uint32_t* __restrict__ slotCount, uint32_t* __restrict__ slotHist,
                                              uint32_t* __restrict__ top16Rec, uint32_t part)
{
  for(int i = threadIdx.x; i < (int)kSlotHistWords; i += THREADS)
    slotHist[(size_t)part * kSlotHistWords + i] = 0u;
  if(threadIdx.x < THREADS / 64)
    top16Rec[((size_t)part * (THREADS / 64) + threadIdx.x) * 32u + 31u] = 0xFFFFFFFFu;
  if(threadIdx.x == 0)
    slotCount[part] = 0u;
}

// LDS the hand-over works in (the caller owns the memory; what may alias what is part of the contract):
struct EmitLds
{
  const uint16_t* li;     // [2048] in: candidate j's local index; bit 15 marks a survivor of the front end (ignored when allSurvive)
  const uint32_t* key;    // [2048] in: its depth key
  const uint16_t* code;   // [2048] in: its bin-rectangle code (null: nothing rides)
  uint16_t*       whist;  // [4][256] per-wave digit counts -> offsets; must not alias li / key / code; may alias stage
  uint32_t*       hist1;  // [128]   counts of key bits 8-15, two per word; aliases nothing
  uint16_t*       start;  // [256]   digit-0 group starts; aliases nothing
  uint2*          stage;  // [2048]  the slot, grouped; may alias li / key / code / whist (all dead by then)
  uint32_t*       cnt;    // [32]    small scratch
};

// Returns the number of pairs written.  Every thread of the workgroup calls it (barriers inside); the inputs are complete and
// visible (a barrier lies between their last write and the call).
template <int THREADS, int ITEMS>
__device__ __forceinline__ uint32_t emitSlot(uint32_t M, bool allSurvive, const EmitLds& S, uint2* __restrict__ slotPairs,
                                             uint32_t* __restrict__ slotCount, uint32_t* __restrict__ slotHist,
                                             uint32_t* __restrict__ top16Rec, uint32_t* __restrict__ top16Count, OsPlan* __restrict__ osPlan,
                                             FrameCounters* __restrict__ ctr, uint32_t part, uint32_t idBase, uint32_t rideShift = 0u)
{
  constexpr int WAVES = THREADS / 64;
  static_assert(WAVES == 4 && THREADS == 256, "k_os_prepare folds four wave records per slot; thread t scans digit t");
  static_assert(THREADS * ITEMS == (int)kOsSlot, "a slot holds one partition");
  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  for(int i = t; i < WAVES * 256 / 2; i += THREADS)
    reinterpret_cast<uint32_t*>(S.whist)[i] = 0u;
  if(t < 128)
    S.hist1[t] = 0u;
  __syncthreads();
  // ---- items in registers; wave w owns candidates [512 w, 512 w + 512), lane-interleaved: (wave, round, lane) is id order ----
  uint32_t key[ITEMS], idc[ITEMS], rd[ITEMS];  // rd: digit << 16 | rank among the wave's earlier keys of that digit, later the position
  uint32_t vmask = 0u;
  uint32_t tmn = 0xFFFFu, tmx = 0u;
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
  {
    const uint32_t j   = (uint32_t)w * (ITEMS * 64u) + (uint32_t)r * 64u + (uint32_t)lane;
    const uint32_t liw = S.li[j];
    const bool     vis = (j < M) && (allSurvive || (liw & 0x8000u));
    key[r]             = S.key[j];
    idc[r]             = idBase + (liw & 0x7FFFu);
    rd[r]              = 0u;
    if(rideShift != 0u)
    {  // the bin rectangle's code rides through the sort above the id (kernels_common.h: rideEncode) ...
      const uint32_t code = S.code[j];
      if(rideShift & 0x100u)
      {  // ... or, when the id leaves too few bits (bit 8 of rideShift: FrameConst::rideSplit), split: the low byte takes the
         // place of the KEY's low byte when the pair is staged — the slot is grouped by that byte, the group says it, and no
         // later pass looks at it again — and waits in the top byte of rd[r] until then
        idc[r] |= (code >> 8) << (rideShift & 31u);
        rd[r] = code << 24;
      }
      else
        idc[r] |= code << rideShift;
    }
    const uint64_t act = __ballot(vis);
    if(act == 0ull)  // wave-uniform
      continue;
    vmask |= vis ? (1u << r) : 0u;
    const uint32_t d = key[r] & 255u;
    uint32_t       lower, cnt;
    waveMatch8(d, act, lower, cnt);
    uint32_t pre = 0u;
    if(vis)
      pre = S.whist[w * 256 + d];
    rd[r] |= (d << 16) | (pre + lower);
    __builtin_amdgcn_wave_barrier();  // every lane of the group has read `pre` before the leader bumps it
    if(vis && lower == 0u)
      S.whist[w * 256 + d] = (uint16_t)(pre + cnt);
    __builtin_amdgcn_wave_barrier();
    if(vis)
    {
      atomicAdd(&S.hist1[((key[r] >> 8) & 255u) >> 1], 1u << (16u * ((key[r] >> 8) & 1u)));
      tmn = min(tmn, key[r] >> 16);
      tmx = max(tmx, key[r] >> 16);
    }
  }
  sortTop16Post<WAVES>(tmn, tmx, S.cnt);
  __syncthreads();
  // ---- thread t == digit t: the waves' counts -> offsets, the groups' starts ----
  uint32_t tot = 0;
#pragma unroll
  for(int q = 0; q < WAVES; ++q)
  {
    const uint32_t c       = S.whist[q * 256 + t];
    S.whist[q * 256 + t]   = (uint16_t)tot;
    tot += c;
  }
  const uint32_t inc = waveInclusiveScan(tot);
  if(lane == 63)
    S.cnt[8 + w] = inc;
  __syncthreads();
  uint32_t base = 0, outCount = 0;
#pragma unroll
  for(int q = 0; q < WAVES; ++q)
  {
    const uint32_t v = S.cnt[8 + q];
    base += (q < w) ? v : 0u;
    outCount += v;
  }
  const uint32_t st = base + inc - tot;
  S.start[t]        = (uint16_t)st;
  {  // the slot's row: two 16-bit values per word, digits 2 i and 2 i + 1
    const uint32_t totN = (uint32_t)__shfl_down((int)tot, 1, 64), stN = (uint32_t)__shfl_down((int)st, 1, 64);
    uint32_t*      row  = slotHist + (size_t)part * kSlotHistWords;
    if((t & 1) == 0)
    {
      row[t >> 1]        = tot | (totN << 16);
      row[256 + (t >> 1)] = st | (stN << 16);
    }
    else
      row[128 + (t >> 1)] = S.hist1[t >> 1];
  }
  if(t == 0)
  {  // k_os_prepare sums the counts into ctr->sortedCount; a frame whose key sort does not run (CPU sorting) counts here
    slotCount[part] = outCount;
    if(top16Count == nullptr && outCount)
      atomicAdd(&ctr->sortedCount, outCount);
  }
  __syncthreads();
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
  {
    const uint32_t d = (rd[r] >> 16) & 255u;
    rd[r]            = (rd[r] & 0xFF000000u) | ((uint32_t)S.start[d] + (uint32_t)S.whist[w * 256 + d] + (rd[r] & 0xFFFFu));
  }
  // ---- this wave's record of key >> 16: counts of lo .. lo + 24 in words 0-24, header in word 31 ----
  uint32_t lo, hi;
  sortTop16Range<WAVES>(outCount, S.cnt, lo, hi);
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
          c += (uint32_t)__popcll(__ballot(((vmask >> r) & 1u) != 0u && (key[r] >> 16) == v));
        if((uint32_t)lane == v - lo)
          myc = c;
      }
    }
    else if(top16Count != nullptr)
    {  // a cell around the camera: many values.  Counted in LDS (the per-wave table is dead: every position has been computed;
       // lo / hi are the same for every thread, so the barriers are uniform), one atomic per occurring value into the count
       // table, and the occurring range
      // (top16Count == nullptr: this frame's key sort does not run — CPU sorting — and nobody would consume the counts)
      uint32_t*  tab   = reinterpret_cast<uint32_t*>(S.whist);  // 256 words of its 512
      const bool inLds = hi - lo < 256u;
      if(inLds)
      {
        __syncthreads();
        for(int i = t; i < 256; i += THREADS)
          tab[i] = 0u;
        __syncthreads();
      }
#pragma unroll
      for(int r = 0; r < ITEMS; ++r)
        if((vmask >> r) & 1u)
          atomicAdd(inLds ? &tab[(key[r] >> 16) - lo] : &top16Count[key[r] >> 16], 1u);
      if(inLds)
      {
        __syncthreads();
        for(uint32_t i = t; i <= hi - lo; i += THREADS)
          if(tab[i])
            atomicAdd(&top16Count[lo + i], tab[i]);
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
  // ---- the slot, grouped by digit 0: through LDS, so that the stores are contiguous ----
  __syncthreads();  // every position is in registers, the tables are dead: the stage may overwrite them (and the inputs)
#pragma unroll
  for(int r = 0; r < ITEMS; ++r)
    if((vmask >> r) & 1u)
      S.stage[rd[r] & 0xFFFFu] = make_uint2((rideShift & 0x100u) ? ((key[r] & 0xFFFFFF00u) | (rd[r] >> 24)) : key[r], idc[r]);
  __syncthreads();
  uint2* dst = slotPairs + (size_t)part * (size_t)kOsSlot;
  for(uint32_t i = t; i < outCount; i += THREADS)
    dst[i] = S.stage[i];
  return outCount;
}

}  // namespace mgs
