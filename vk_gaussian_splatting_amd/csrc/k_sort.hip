// k_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, device-side count.
//
// Behavioural spec: vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/include/vk_radix_sort.h:73-78,
// 3rdparty/vrdx/src/vk_radix_sort.cc:262-416): stable, ascending, 8-bit digits, element count read on
// the device.  The implementation is new and wave64-native:
//   * reduce-then-scan per pass (partition histograms -> per-digit row scan -> ranked scatter).  A
//     decoupled-look-back (onesweep) chain was rejected for MI355X: a cross-CU hand-off costs
//     ~1-3 us (MI355X_MICROARCH.md "handoff-1to1"), and with every partition co-resident on 256 CUs
//     the chain serialises at one hop per partition (measured: tools/micro/lookback_rate.hip, 1.3 ms for the look-back of
//     1 020 partitions x 256 digits alone = 1.27 us per hop).  Reduce-then-scan has no inter-workgroup
//     dependency inside a launch, so nothing can spin or hang.
//   * ranking inside a partition uses 64-lane ballots (8 per key) to find the lanes holding the same
//     digit, one LDS counter row per wave, then an LDS re-order so the global scatter is coalesced.
//   * pass 0 can read "slotted" input (the project kernel's per-partition survivor lists), which
//     fuses the stream compaction into the sort; the producer also hands over the pass-0 digit
//     histogram of every slot, so the in-frame sort starts directly with a scan.
//   * no global atomics anywhere: the scan kernel leaves each digit row's total in the plan and the
//     scatter workgroups turn the 256 totals into digit bases themselves (a 256-wide scan is noise
//     next to ranking 8192 keys).
//   * a pass whose digit is identical for every key (the top byte of a depth key, typically) is
//     detected by its scan kernel and its scatter exits; the ping-pong selection is derived from the
//     skip flags on the device.
#include <cstdlib>

#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

constexpr int kSlotPart = 2048;  // slotted pass-0 partitions == the project kernel's partitions

// Uniform partition p covers [p*part, p*part+count).  A slotted partition covers part/2048 consecutive
// slots of the project kernel (each a 2048-entry region holding slotCount[slot] compacted survivors).
__device__ __forceinline__ uint32_t partitionCount(const uint32_t* slotCount, uint32_t n, uint32_t p, uint32_t part)
{
  const uint64_t base = (uint64_t)p * part;
  return (n > base) ? (uint32_t)min((uint64_t)part, (uint64_t)n - base) : 0u;
}

// LDS histogram add with run aggregation: lanes holding the same digit as their left neighbour are
// folded into the run's first lane, so long runs of equal digits (the high byte of a tile id, the top
// bytes of depth keys) cost one LDS atomic instead of a 64-way serialised one.
__device__ __forceinline__ void histAddRuns(uint32_t* hist, uint32_t digit, bool valid)
{
  const int      lane = laneId();
  const uint32_t d    = valid ? digit : 0xFFFFFFFFu;
  const uint32_t prev = __shfl_up(d, 1, 64);
  const bool     lead = (lane == 0) || (prev != d);
  const uint64_t mask = __ballot(lead);
  if(lead && valid)
  {
    const uint64_t above = (lane == 63) ? 0ull : (mask >> (lane + 1));
    const uint32_t len   = above ? (uint32_t)__builtin_ctzll(above) + 1u : (uint32_t)(64 - lane);
    atomicAdd(&hist[digit], len);
  }
}

// digit of pass 2 when the top 16 bits are remapped: the rank of key >> 16 among the occurring values, read from a
// per-workgroup LDS table indexed by (key >> 16) - base (the occurring values span < 4096, checked by the scan kernel;
// padding keys clamp to the last entry, which holds the largest rank)
constexpr uint32_t kRemapSpan = 4096;
__device__ __forceinline__ void remapBuildTable(uint8_t* s_tab, const SortPlan* plan, int t, int threads)
{
  const uint32_t count = plan->remapCount, base = plan->remapBase;
  for(uint32_t i = t; i < count; i += threads)
    s_tab[(uint32_t)plan->remapVals[i] - base] = (uint8_t)i;
  if(t == 0)
    s_tab[kRemapSpan - 1] = (uint8_t)(count - 1u);
}
__device__ __forceinline__ uint32_t remapDigit(const uint8_t* s_tab, uint32_t base, uint32_t key)
{
  return s_tab[min((key >> 16) - base, kRemapSpan - 1u)];
}

// where pass `pass` (> 0) reads from: 0 = X, 1 = Y.  Pass 0 writes X; every executed pass flips.
__device__ __forceinline__ uint32_t planSrcSel(const SortPlan* __restrict__ plan, int pass)
{
  uint32_t cur = 0;
  for(int q = 1; q < pass; ++q)
    cur ^= plan->skip[q] ? 0u : 1u;
  return cur;
}

// (a) per-partition digit histogram of one pass (pass 0 of a slotted sort gets it from the producer instead)
__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t* __restrict__ keysX, const uint32_t* __restrict__ keysY,
                                                   const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ nPtr,
                                                   const SortPlan* __restrict__ plan, uint32_t* __restrict__ partHist,
                                                   uint32_t pStride, int pass, int beginBit, uint32_t part)
{
  __shared__ uint32_t s_h[256];
  __shared__ uint8_t s_rv[kRemapSpan];
  const int       t     = threadIdx.x;
  const uint32_t  n     = *nPtr;
  const bool      remap = plan->remapOn != 0u;
  if(remap && pass == 3)
    return;  // pass 2 sorted on the rank of the top 16 bits: nothing left to do
  const uint32_t rcount = plan->remapBase;
  if(remap && pass == 2)
    remapBuildTable(s_rv, plan, t, 256);
  const uint32_t* keys  = (pass == 0) ? keys0 : (planSrcSel(plan, pass) ? keysY : keysX);
  const uint32_t  parts = (uint32_t)(((uint64_t)n + part - 1) / part);
  const int       shift = beginBit + 8 * pass;
  for(uint32_t p = blockIdx.x; p < parts; p += gridDim.x)
  {
    s_h[t] = 0;
    __syncthreads();
    const uint32_t  count = partitionCount(nullptr, n, p, part);
    const uint32_t* src   = keys + (size_t)p * part;
    // 8 loads in flight per thread, then the LDS work (the loop was one dependent round trip per key)
    for(uint32_t i0 = 0; i0 < count; i0 += 2048u)
    {
      uint32_t kk[8];
#pragma unroll
      for(int u = 0; u < 8; ++u)
      {
        const uint32_t i = min(i0 + (uint32_t)u * 256u + (uint32_t)t, count - 1u);  // clamped, not predicated
        kk[u]            = src[i];
      }
#pragma unroll
      for(int u = 0; u < 8; ++u)
        histAddRuns(s_h, (remap && pass == 2) ? remapDigit(s_rv, rcount, kk[u]) : ((kk[u] >> shift) & 255u),
                    i0 + (uint32_t)u * 256u + (uint32_t)t < count);
    }
    __syncthreads();
    partHist[(size_t)t * pStride + p] = s_h[t];
    __syncthreads();
  }
}

// (b) one workgroup per digit: exclusive scan of that digit's row of partition counts, in place.  The row
// total goes to plan->ghist[pass][d]; a row that holds every key marks the pass as skippable.
__global__ __launch_bounds__(256) void k_sort_scan(const uint32_t* __restrict__ nPtr, uint32_t partsSlotted,
                                                   SortPlan* __restrict__ plan, uint32_t* __restrict__ partHist,
                                                   uint32_t pStride, int pass, uint32_t part, int allowRemap)
{
  __shared__ uint32_t s_tmp[4];
  const int      t = threadIdx.x, d = blockIdx.x;
  const uint32_t n     = *nPtr;
  if(pass == 3 && plan->remapOn != 0u)
    return;  // skip[3] was set together with remapOn
  // slotted pass 0: the producer wrote one histogram per 2048-key slot; a partition takes spp = part/2048 slots.
  // In place is safe: slot index >= partition index, and the block scan's barriers sit between reads and writes.
  const uint32_t spp   = partsSlotted ? part / kSlotPart : 1u;
  const uint32_t parts = partsSlotted ? (partsSlotted + spp - 1) / spp : (uint32_t)(((uint64_t)n + part - 1) / part);
  uint32_t       carry = 0;
  uint32_t*      row   = partHist + (size_t)d * pStride;
  for(uint32_t base = 0; base < parts; base += 2048)
  {
    const uint32_t p0 = base + t * 8;
    uint32_t       v[8], sum = 0;
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
      v[i] = 0u;
      if(p0 + i < parts)
      {
        if(spp == 1u)
          v[i] = row[p0 + i];
        else
          for(uint32_t q = 0; q < spp; ++q)
            v[i] += ((p0 + i) * spp + q < partsSlotted) ? row[(p0 + i) * spp + q] : 0u;
      }
      sum += v[i];
    }
    uint32_t chunk;
    uint32_t run = carry + blockExclusiveScan256(sum, s_tmp, &chunk);
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
      if(p0 + i < parts)
        row[p0 + i] = run;
      run += v[i];
    }
    carry += chunk;
  }
  if(t == 0)
  {
    plan->ghist[pass][d] = carry;
    if(pass > 0 && n > 0 && carry == n)
      plan->skip[pass] = 1u;  // every key has this digit: the pass would be the identity permutation
  }
  // pass elision: after the producer's marks are complete (this kernel runs behind it), workgroup 0 of pass 1 turns the
  // presence bitmap of key >> 16 into the ascending value table; <= 256 values -> pass 2 takes their rank as its digit
  if(allowRemap && pass == 1 && d == 0)
  {
    uint32_t c[8], sum = 0;
#pragma unroll
    for(int q = 0; q < 8; ++q)
    {
      c[q] = (uint32_t)__popc(plan->topBitmap[t * 8 + q]);
      sum += c[q];
    }
    uint32_t total;
    uint32_t run = blockExclusiveScan256(sum, s_tmp, &total);
    // smallest / largest occurring value (wave + block reductions over the per-thread words)
    uint32_t vmin = 0xFFFFFFFFu, vmax = 0u;
#pragma unroll
    for(int q = 0; q < 8; ++q)
    {
      const uint32_t bits = plan->topBitmap[t * 8 + q];
      if(bits)
      {
        vmin = min(vmin, (uint32_t)((t * 8 + q) * 32 + __builtin_ctz(bits)));
        vmax = max(vmax, (uint32_t)((t * 8 + q) * 32 + 31 - __builtin_clz(bits)));
      }
    }
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
    {
      vmin = min(vmin, (uint32_t)__shfl_xor(vmin, o, 64));
      vmax = max(vmax, (uint32_t)__shfl_xor(vmax, o, 64));
    }
    __shared__ uint32_t s_mm[8];
    if((t & 63) == 0)
    {
      s_mm[t >> 6]       = vmin;
      s_mm[4 + (t >> 6)] = vmax;
    }
    __syncthreads();
    vmin = min(min(s_mm[0], s_mm[1]), min(s_mm[2], s_mm[3]));
    vmax = max(max(s_mm[4], s_mm[5]), max(s_mm[6], s_mm[7]));
    const bool on = total >= 1u && total <= 256u && (vmax - vmin) < kRemapSpan - 1u;
    if(on)
    {
#pragma unroll
      for(int q = 0; q < 8; ++q)
      {
        uint32_t bits = plan->topBitmap[t * 8 + q];
        while(bits)
        {
          const uint32_t b = (uint32_t)__builtin_ctz(bits);
          bits &= bits - 1u;
          plan->remapVals[run++] = (uint16_t)((t * 8 + q) * 32 + b);
        }
      }
    }
    if(t == 0 && on)
    {
      plan->remapCount = total;
      plan->remapBase  = vmin;
      plan->remapOn    = 1u;
      plan->skip[3]    = 1u;
    }
  }
}

// exclusive scan over the first 256 threads of a THREADS-wide block (one value per digit)
template <int THREADS>
__device__ __forceinline__ uint32_t digitExclusiveScan(uint32_t v, uint32_t* s_tmp /*THREADS/64*/)
{
  const int      lane = laneId(), w = threadIdx.x >> 6;
  const uint32_t inc  = waveInclusiveScan(v);
  if(lane == 63)
    s_tmp[w] = inc;
  __syncthreads();
  uint32_t base = 0;
  if(w > 0) base += s_tmp[0];
  if(w > 1) base += s_tmp[1];
  if(w > 2) base += s_tmp[2];
  __syncthreads();
  return base + inc - v;  // meaningful for threads < 256 only
}

// (c) ranked scatter of one partition of THREADS*KPT keys.  Workgroup 0 of the last pass also publishes the
// outcome (which buffer holds the result, how many passes ran); on a single-pass sort the digit histogram IS the
// sorted layout, so the caller's per-digit ranges are written there too.
template <bool FIRST, int THREADS, int KPT, bool GATHER = false>
__global__ __launch_bounds__(THREADS) void k_sort_scatter(const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ vals0,
                                                          uint32_t* __restrict__ keysX, uint32_t* __restrict__ valsX,
                                                          uint32_t* __restrict__ keysY, uint32_t* __restrict__ valsY,
                                                          const uint32_t* __restrict__ slotCount, const uint32_t* __restrict__ nPtr,
                                                          uint32_t partsSlotted, SortPlan* __restrict__ plan,
                                                          const uint32_t* __restrict__ partHist, uint32_t pStride, int pass,
                                                          int beginBit, int nPasses, uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ gatherSrc, uint32_t* __restrict__ gatherDst)
{
  // GATHER (last pass of the in-frame key sort only): the keys are dead after this pass, so the workgroup writes
  // gatherDst[sortedPos] = gatherSrc[value] in their place (the splat's bin rect, which the binning stage needs in
  // sorted order).  The random 4-byte gathers are issued as soon as the values are loaded and land while the keys are
  // being ranked, instead of being the exposed dependent trip of their own kernel.
  constexpr int PART  = THREADS * KPT;
  constexpr int SPP   = PART / kSlotPart;  // slots per slotted partition
  constexpr int WAVES = THREADS / 64;
  __shared__ uint32_t s_whist[WAVES][256];
  __shared__ uint32_t s_k[PART];
  __shared__ uint32_t s_v[PART];
  __shared__ uint32_t s_loff[256];
  __shared__ uint32_t s_gbase[256];
  __shared__ uint32_t s_tmp[WAVES];
  __shared__ uint8_t  s_dig[GATHER ? PART : 1];
  __shared__ uint8_t  s_rv[FIRST ? 1 : kRemapSpan];

  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  const bool     skipped = !FIRST && plan->skip[pass] != 0u;
  const bool     slotted = FIRST && (slotCount != nullptr);
  const uint32_t n       = *nPtr;
  if(blockIdx.x == 0 && t == 0 && pass == nPasses - 1)
  {
    uint32_t run = 1;
    for(int q = 1; q < nPasses; ++q)
      run += plan->skip[q] ? 0u : 1u;
    plan->finalSel  = planSrcSel(plan, nPasses);
    plan->passesRun = run;
    plan->n         = n;
    if(GATHER)
      plan->reserved[0] = skipped ? 0u : 1u;  // gatherDst is valid: consumers need not gather themselves
  }
  if(skipped)
    return;
  const uint32_t parts   = slotted ? (partsSlotted + SPP - 1) / SPP : (uint32_t)(((uint64_t)n + PART - 1) / PART);
  const uint32_t p       = blockIdx.x;
  if(p >= parts)
    return;
  // slotted: prefix of the SPP slot counts -> compact index inside the partition maps to (slot, offset)
  uint32_t pre[SPP + 1];
  pre[0] = 0;
  if(slotted)
  {
#pragma unroll
    for(int q = 0; q < SPP; ++q)
    {
      const uint32_t slot = p * SPP + q;
      pre[q + 1]          = pre[q] + (slot < partsSlotted ? slotCount[slot] : 0u);
    }
  }
  const uint32_t count = slotted ? pre[SPP] : partitionCount(nullptr, n, p, PART);
  const uint32_t *kin, *vin;
  uint32_t *      kout, *vout;
  if(FIRST)
  {
    kin  = keys0;
    vin  = vals0;
    kout = keysX;
    vout = valsX;
  }
  else if(planSrcSel(plan, pass))
  {
    kin  = keysY;
    vin  = valsY;
    kout = keysX;
    vout = valsX;
  }
  else
  {
    kin  = keysX;
    vin  = valsX;
    kout = keysY;
    vout = valsY;
  }
  const int shift = beginBit + 8 * pass;
  // pass 2 of a remapped sort: digit = rank of key >> 16 among the occurring values (see SortPlan)
  const bool     remap  = !FIRST && pass == 2 && plan->remapOn != 0u;
  const uint32_t rcount = plan->remapBase;
  if constexpr(!FIRST)
    if(remap)
      remapBuildTable(s_rv, plan, t, THREADS);

  for(int i = t; i < WAVES * 256; i += THREADS)
    (&s_whist[0][0])[i] = 0;

  // wave-striped load: wave w owns keys [w*64*KPT, (w+1)*64*KPT) of the partition, lane-interleaved, so
  // (round i, lane) order == memory order inside the wave, and waves are in memory order too.
  const uint32_t wofs = w * (64 * KPT);
  uint32_t       key[KPT], val[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t idx = wofs + i * 64 + lane;
    const bool     in  = idx < count;
    size_t         src = (size_t)p * PART + idx;
    if(slotted)
    {
      uint32_t q = 0;
#pragma unroll
      for(int z = 1; z < SPP; ++z)
        q += (idx >= pre[z]) ? 1u : 0u;
      uint32_t pq = pre[0];
#pragma unroll
      for(int z = 1; z < SPP; ++z)
        pq = (q >= (uint32_t)z) ? pre[z] : pq;
      src = ((size_t)p * SPP + q) * kSlotPart + (idx - pq);
    }
    key[i] = in ? kin[src] : 0xFFFFFFFFu;
    val[i] = in ? vin[src] : 0u;
  }
  uint32_t gat[GATHER ? KPT : 1];
  if constexpr(GATHER)
  {
#pragma unroll
    for(int i = 0; i < KPT; ++i)
      gat[i] = (wofs + i * 64 + lane < count) ? gatherSrc[val[i]] : 0u;
  }
  __syncthreads();

  // per-wave multi-split: rank of each key among the keys of its wave with the same digit
  uint32_t rank[KPT];
  uint8_t  dig[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    // padding keys (idx >= count) carry digit 255 and sit behind every real key of the partition
    const uint32_t d = remap ? remapDigit(s_rv, rcount, key[i]) : ((key[i] >> shift) & 255u);
    dig[i]           = (uint8_t)d;
    uint64_t       m = ~0ull;
#pragma unroll
    for(int b = 0; b < 8; ++b)
    {
      const bool     bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t lower = lanesBelow(m);
    const uint32_t cnt   = (uint32_t)__popcll(m);
    const uint32_t pre   = s_whist[w][d];
    rank[i]              = pre + lower;
    __builtin_amdgcn_wave_barrier();  // every lane of the group has read `pre` before the leader bumps it
    if(lower == 0)
      s_whist[w][d] = pre + cnt;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // thread t (< 256) == digit t: wave offsets, partition-local exclusive scan over digits, global base
  uint32_t tot = 0;
  if(t < 256)
  {
    uint32_t acc = 0;
#pragma unroll
    for(int q = 0; q < WAVES; ++q)
    {
      const uint32_t c = s_whist[q][t];
      s_whist[q][t]    = acc;
      acc += c;
    }
    tot = acc;
  }
  const uint32_t loff  = digitExclusiveScan<THREADS>(tot, s_tmp);
  const uint32_t gtot  = (t < 256) ? plan->ghist[pass][t] : 0u;
  const uint32_t below = digitExclusiveScan<THREADS>(gtot, s_tmp);  // keys with a smaller digit
  if(t < 256)
  {
    s_loff[t]  = loff;
    s_gbase[t] = below + partHist[(size_t)t * pStride + p] - loff;  // wraps are fine: only base+idx is used
    if(ranges != nullptr && nPasses == 1 && p == 0)
      ranges[t] = make_uint2(below, below + gtot);
  }
  __syncthreads();

  // re-order through LDS so that keys with equal digits are contiguous
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t d   = dig[i];
    const uint32_t pos = s_loff[d] + s_whist[w][d] + rank[i];
    s_v[pos]           = val[i];
    if constexpr(GATHER)
    {
      s_k[pos]   = gat[i];
      s_dig[pos] = (uint8_t)d;
    }
    else
      s_k[pos] = key[i];
  }
  __syncthreads();

  // coalesced scatter: consecutive threads write consecutive addresses inside each digit run
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t idx = i * THREADS + t;
    if(idx < count)
    {
      const uint32_t k   = s_k[idx];
      const uint32_t v   = s_v[idx];
      const uint32_t d   = GATHER ? (uint32_t)s_dig[idx] : (remap ? remapDigit(s_rv, rcount, k) : ((k >> shift) & 255u));
      const uint32_t dst = s_gbase[d] + idx;
      if constexpr(GATHER)
        gatherDst[dst] = k;
      else
        kout[dst] = k;
      vout[dst] = v;
    }
  }
}

// zero the plan (ghist etc.) — folded into the frame-init kernel when sorting inside a frame
__global__ void k_sort_plan_clear(SortPlan* plan)
{
  uint32_t* w = reinterpret_cast<uint32_t*>(plan);
  for(uint32_t i = threadIdx.x; i < sizeof(SortPlan) / 4; i += blockDim.x)
    w[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// Host-side driver.  src0 -> X on pass 0, then X <-> Y.  The result is in (plan->finalSel ? Y : X).
void launchSortClearPlan(hipStream_t stream, SortPlan* plan)
{
  hipLaunchKernelGGL(k_sort_plan_clear, dim3(1), dim3(256), 0, stream, plan);
}

void launchRadixSort(hipStream_t stream, const SortLaunch& s)
{
  const int nPasses = (s.endBit - s.beginBit + 7) / 8;
  if(nPasses <= 0 || s.maxElems == 0)
    return;
  const bool     slotted = s.slotCount != nullptr;  // pass 0: 2048-key slots + the producer's slot histograms
  // big sorts use 4096-key partitions (digit runs of ~16 keys = 64-byte scatter segments), small ones keep 2048 so
  // that 256 CUs still see enough workgroups.  8192 (512 threads, 72 KB of LDS) is 2 % faster on an idle GPU but its
  // workgroups wait for a whole free half CU when frames overlap: 77 us instead of 21.5 us per scatter with three
  // frames in flight (profiles/r1_h), 2450 vs 2540 frames/s.
  static const uint32_t kPartOverride = [] { const char* e = std::getenv("MGS_SORT_PART"); return e ? (uint32_t)std::atoi(e) : 0u; }();
  const uint32_t part    = kPartOverride ? kPartOverride : ((s.maxElems >= (2u << 20)) ? 4096u : 2048u);
  static const uint32_t kSlotPartOverride = [] { const char* e = std::getenv("MGS_SLOT_PART"); return e ? (uint32_t)std::atoi(e) : 0u; }();
  auto partOf  = [&](int pass) { return (pass == 0 && slotted && kSlotPartOverride) ? kSlotPartOverride : part; };
  auto partsOf = [&](int pass) {
    const uint32_t pp = partOf(pass);
    return (pass == 0 && slotted) ? (s.partsSlotted + pp / 2048u - 1) / (pp / 2048u) : (uint32_t)(((uint64_t)s.maxElems + pp - 1) / pp);
  };
  for(int pass = 0; pass < nPasses; ++pass)
  {
    const uint32_t parts = partsOf(pass), pp = partOf(pass);
    if(parts == 0)
      continue;
    if(!(pass == 0 && slotted))
      hipLaunchKernelGGL(k_sort_hist, dim3(parts), dim3(256), 0, stream, s.keysX, s.keysY, s.keys0, s.nPtr, s.plan, s.partHist,
                         s.pStride, pass, s.beginBit, pp);
    hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, stream, s.nPtr, (pass == 0 && slotted) ? s.partsSlotted : 0u, s.plan,
                       s.partHist, s.pStride, pass, pp, (s.allowRemap && nPasses == 4 && s.beginBit == 0) ? 1 : 0);
#define MGS_SCATTER(FIRSTV, TH, KP, SLOTS)                                                                                  \
  hipLaunchKernelGGL((k_sort_scatter<FIRSTV, TH, KP>), dim3(parts), dim3(TH), 0, stream, s.keys0, s.vals0, s.keysX, s.valsX, \
                     s.keysY, s.valsY, SLOTS, s.nPtr, s.partsSlotted, s.plan, s.partHist, s.pStride, pass, s.beginBit, nPasses, \
                     s.ranges, s.gatherSrc, s.gatherDst)
#define MGS_SCATTER_G(TH, KP)                                                                                               \
  hipLaunchKernelGGL((k_sort_scatter<false, TH, KP, true>), dim3(parts), dim3(TH), 0, stream, s.keys0, s.vals0, s.keysX,     \
                     s.valsX, s.keysY, s.valsY, (const uint32_t*)nullptr, s.nPtr, s.partsSlotted, s.plan, s.partHist,        \
                     s.pStride, pass, s.beginBit, nPasses, s.ranges, s.gatherSrc, s.gatherDst)
    if(pass == 0)
    {
      if(pp == 2048u)
        MGS_SCATTER(true, 256, 8, s.slotCount);
      else if(pp == 4096u)
        MGS_SCATTER(true, 256, 16, s.slotCount);
      else
        MGS_SCATTER(true, 512, 16, s.slotCount);
    }
    else if(s.gatherSrc != nullptr && pass == nPasses - 1 && pp != 4096u)
    {
      if(pp == 2048u)
        MGS_SCATTER_G(256, 8);
      else
        MGS_SCATTER_G(512, 16);
    }
    else
    {
      if(pp == 2048u)
        MGS_SCATTER(false, 256, 8, (const uint32_t*)nullptr);
      else if(pp == 4096u)
        MGS_SCATTER(false, 256, 16, (const uint32_t*)nullptr);
      else
        MGS_SCATTER(false, 512, 16, (const uint32_t*)nullptr);
    }
#undef MGS_SCATTER_G
#undef MGS_SCATTER
  }
}

}  // namespace mgs
