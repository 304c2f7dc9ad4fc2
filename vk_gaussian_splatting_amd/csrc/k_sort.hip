// k_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, device-side count.
//
// Behavioural spec: vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/include/vk_radix_sort.h:73-78,
// 3rdparty/vrdx/src/vk_radix_sort.cc:262-416): stable, ascending, 8-bit digits, element count read on
// the device.  The implementation is new and wave64-native:
//   * reduce-then-scan per pass (partition histograms -> per-digit row scan -> ranked scatter).  A
//     decoupled-look-back (onesweep) chain was rejected for MI355X: a cross-CU hand-off costs
//     ~1-3 us (MI355X_MICROARCH.md "handoff-1to1"), and with every partition co-resident on 256 CUs
//     the chain serialises at one hop per partition.  Reduce-then-scan has no inter-workgroup
//     dependency inside a launch, so nothing can spin or hang.
//   * ranking inside a partition uses 64-lane ballots (8 per key) to find the lanes holding the same
//     digit, one LDS counter row per wave, then an LDS re-order so the global scatter is coalesced.
//   * pass 0 can read "slotted" input (the project kernel's per-partition survivor lists), which
//     fuses the stream compaction into the sort.
//   * a pass whose digit is identical for every key (the top byte of a depth key, typically) is
//     skipped on the device; the ping-pong selection lives in a device-side plan.
#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

constexpr int kSortThreads = 256;
constexpr int kSortKpt     = 8;
constexpr int kSortPart    = kSortThreads * kSortKpt;  // 2048 keys per partition (== project partition)
constexpr int kSortWaves   = kSortThreads / 64;

__device__ __forceinline__ void partitionExtent(bool slotted, const uint32_t* slotCount, uint32_t n, uint32_t p,
                                                uint32_t& count)
{
  if(slotted)
    count = slotCount[p];
  else
  {
    const uint32_t base = p * kSortPart;
    count               = (n > base) ? min((uint32_t)kSortPart, n - base) : 0u;
  }
}

// (a) per-partition digit histogram.  FIRST additionally accumulates the global histogram of every
// pass (order independent), flushed once per workgroup.
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint32_t* __restrict__ keysX, const uint32_t* __restrict__ keysY,
                                                            const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ slotCount,
                                                            const uint32_t* __restrict__ nPtr, uint32_t partsSlotted,
                                                            SortPlan* __restrict__ plan, uint32_t* __restrict__ partHist,
                                                            uint32_t pStride, int pass, int beginBit, int nPasses)
{
  __shared__ uint32_t s_h[256];
  __shared__ uint32_t s_g[4][256];
  const int  t       = threadIdx.x;
  const bool slotted = FIRST && (slotCount != nullptr);
  const uint32_t n   = *nPtr;
  if(!FIRST && plan->skip[pass])
    return;
  const uint32_t* keys  = FIRST ? keys0 : (plan->srcSel[pass] ? keysY : keysX);
  const uint32_t  parts = slotted ? partsSlotted : (n + kSortPart - 1) / kSortPart;
  const int       shift = beginBit + 8 * pass;
  if(FIRST)
  {
#pragma unroll
    for(int q = 0; q < 4; ++q)
      s_g[q][t] = 0;
  }
  for(uint32_t p = blockIdx.x; p < parts; p += gridDim.x)
  {
    s_h[t] = 0;
    __syncthreads();
    uint32_t count;
    partitionExtent(slotted, slotCount, n, p, count);
    const uint32_t* src = keys + (size_t)p * kSortPart;
    for(uint32_t i = t; i < count; i += kSortThreads)
    {
      const uint32_t key = src[i];
      atomicAdd(&s_h[(key >> shift) & 255u], 1u);
      if(FIRST)
      {
        atomicAdd(&s_g[0][(key >> shift) & 255u], 1u);
        for(int q = 1; q < nPasses; ++q)
          atomicAdd(&s_g[q][(key >> (shift + 8 * q)) & 255u], 1u);
      }
    }
    __syncthreads();
    partHist[(size_t)t * pStride + p] = s_h[t];
    __syncthreads();
  }
  if(FIRST)
  {
    __syncthreads();
    for(int q = 0; q < nPasses; ++q)
      if(s_g[q][t])
        atomicAdd(&plan->ghist[q][t], s_g[q][t]);
  }
}

// decides which passes run and where each pass reads from (pass 0 always runs: src0 -> X)
__global__ __launch_bounds__(256) void k_sort_plan(SortPlan* __restrict__ plan, const uint32_t* __restrict__ nPtr, int nPasses)
{
  __shared__ uint32_t s_skip[4];
  const int      t = threadIdx.x;
  const uint32_t n = *nPtr;
  if(t < 4)
    s_skip[t] = 0;
  __syncthreads();
  for(int q = 1; q < nPasses; ++q)
    if(n > 0 && plan->ghist[q][t] == n)
      s_skip[q] = 1;  // every key has the same digit: the pass would be the identity permutation
  __syncthreads();
  if(t == 0)
  {
    uint32_t cur = 0;  // after pass 0 the data is in X (sel 0)
    uint32_t run = 1;
    plan->skip[0]   = 0;
    plan->srcSel[0] = 0;
    for(int q = 1; q < nPasses; ++q)
    {
      plan->skip[q]   = s_skip[q];
      plan->srcSel[q] = cur;
      if(!s_skip[q])
      {
        cur ^= 1u;
        ++run;
      }
    }
    plan->finalSel  = cur;
    plan->passesRun = run;
    plan->n         = n;
  }
}

// (b) one workgroup per digit: exclusive scan of that digit's row of partition counts, offset by the
// number of keys with a smaller digit.  In place: partHist[d][p] becomes the global destination of
// the first key of partition p with digit d.
__global__ __launch_bounds__(256) void k_sort_scan(const uint32_t* __restrict__ nPtr, const uint32_t* __restrict__ slotCount,
                                                   uint32_t partsSlotted, SortPlan* __restrict__ plan,
                                                   uint32_t* __restrict__ partHist, uint32_t pStride, int pass)
{
  __shared__ uint32_t s_tmp[4];
  const int      t = threadIdx.x, d = blockIdx.x;
  if(plan->skip[pass])
    return;
  const uint32_t n     = *nPtr;
  const bool     slotted = (pass == 0) && (slotCount != nullptr);
  const uint32_t parts = slotted ? partsSlotted : (n + kSortPart - 1) / kSortPart;
  uint32_t       total;
  const uint32_t below = (t < d) ? plan->ghist[pass][t] : 0u;
  (void)blockExclusiveScan256(below, s_tmp, &total);
  uint32_t  carry = total;  // keys with a smaller digit
  uint32_t* row   = partHist + (size_t)d * pStride;
  for(uint32_t base = 0; base < parts; base += 256)
  {
    const uint32_t p = base + t;
    const uint32_t v = (p < parts) ? row[p] : 0u;
    uint32_t       chunk;
    const uint32_t ex = blockExclusiveScan256(v, s_tmp, &chunk);
    if(p < parts)
      row[p] = carry + ex;
    carry += chunk;
  }
}

// (c) ranked scatter of one partition.
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ vals0,
                                                               uint32_t* __restrict__ keysX, uint32_t* __restrict__ valsX,
                                                               uint32_t* __restrict__ keysY, uint32_t* __restrict__ valsY,
                                                               const uint32_t* __restrict__ slotCount, const uint32_t* __restrict__ nPtr,
                                                               uint32_t partsSlotted, const SortPlan* __restrict__ plan,
                                                               const uint32_t* __restrict__ partHist, uint32_t pStride, int pass,
                                                               int beginBit)
{
  __shared__ uint32_t s_whist[kSortWaves][256];
  __shared__ uint32_t s_k[kSortPart];
  __shared__ uint32_t s_v[kSortPart];
  __shared__ uint32_t s_loff[256];
  __shared__ uint32_t s_gbase[256];
  __shared__ uint32_t s_tmp[4];

  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  if(!FIRST && plan->skip[pass])
    return;
  const bool     slotted = FIRST && (slotCount != nullptr);
  const uint32_t n       = *nPtr;
  const uint32_t parts   = slotted ? partsSlotted : (n + kSortPart - 1) / kSortPart;
  const uint32_t p       = blockIdx.x;
  if(p >= parts)
    return;
  uint32_t count;
  partitionExtent(slotted, slotCount, n, p, count);
  const uint32_t *kin, *vin;
  uint32_t *      kout, *vout;
  if(FIRST)
  {
    kin  = keys0;
    vin  = vals0;
    kout = keysX;
    vout = valsX;
  }
  else if(plan->srcSel[pass])
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

#pragma unroll
  for(int i = 0; i < kSortWaves; ++i)
    s_whist[i][t] = 0;

  // wave-striped load: wave w owns keys [w*512, w*512+512) of the partition, lane-interleaved, so
  // (round i, lane) order == memory order inside the wave, and waves are in memory order too.
  const size_t   base = (size_t)p * kSortPart + (size_t)w * (64 * kSortKpt);
  const uint32_t wofs = w * (64 * kSortKpt);
  uint32_t       key[kSortKpt], val[kSortKpt];
#pragma unroll
  for(int i = 0; i < kSortKpt; ++i)
  {
    const uint32_t idx = wofs + i * 64 + lane;
    const bool     in  = idx < count;
    key[i]             = in ? kin[base + i * 64 + lane] : 0xFFFFFFFFu;
    val[i]             = in ? vin[base + i * 64 + lane] : 0u;
  }
  __syncthreads();

  // per-wave multi-split: rank of each key among the keys of its wave with the same digit
  uint32_t rank[kSortKpt];
#pragma unroll
  for(int i = 0; i < kSortKpt; ++i)
  {
    // padding keys (idx >= count) use digit 255 of 0xFFFFFFFF shifted — they sort behind every real key
    const uint32_t d = (key[i] >> shift) & 255u;
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
    // make sure every lane of the group has read `pre` before the leader bumps it
    __builtin_amdgcn_wave_barrier();
    if(lower == 0)
      s_whist[w][d] = pre + cnt;
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();

  // thread t == digit t: wave offsets, partition-local exclusive scan over digits, global base
  {
    const uint32_t c0 = s_whist[0][t], c1 = s_whist[1][t], c2 = s_whist[2][t], c3 = s_whist[3][t];
    s_whist[0][t] = 0;
    s_whist[1][t] = c0;
    s_whist[2][t] = c0 + c1;
    s_whist[3][t] = c0 + c1 + c2;
    const uint32_t tot = c0 + c1 + c2 + c3;
    uint32_t       dummy;
    const uint32_t loff = blockExclusiveScan256(tot, s_tmp, &dummy);
    s_loff[t]           = loff;
    s_gbase[t]          = partHist[(size_t)t * pStride + p] - loff;  // wraps are fine: only base+idx is used
  }
  __syncthreads();

  // re-order through LDS so that keys with equal digits are contiguous
#pragma unroll
  for(int i = 0; i < kSortKpt; ++i)
  {
    const uint32_t d   = (key[i] >> shift) & 255u;
    const uint32_t pos = s_loff[d] + s_whist[w][d] + rank[i];
    s_k[pos]           = key[i];
    s_v[pos]           = val[i];
  }
  __syncthreads();

  // coalesced scatter: consecutive threads write consecutive addresses inside each digit run
#pragma unroll
  for(int i = 0; i < kSortKpt; ++i)
  {
    const uint32_t idx = i * kSortThreads + t;
    if(idx < count)
    {
      const uint32_t k   = s_k[idx];
      const uint32_t d   = (k >> shift) & 255u;
      const uint32_t dst = s_gbase[d] + idx;
      kout[dst]          = k;
      vout[dst]          = s_v[idx];
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
  if(nPasses <= 0 || s.maxParts == 0)
    return;
  const uint32_t fatGrid = s.maxParts < 1024u ? s.maxParts : 1024u;
  hipLaunchKernelGGL((k_sort_hist<true>), dim3(fatGrid), dim3(kSortThreads), 0, stream, s.keysX, s.keysY, s.keys0,
                     s.slotCount, s.nPtr, s.partsSlotted, s.plan, s.partHist, s.pStride, 0, s.beginBit, nPasses);
  hipLaunchKernelGGL(k_sort_plan, dim3(1), dim3(256), 0, stream, s.plan, s.nPtr, nPasses);
  for(int pass = 0; pass < nPasses; ++pass)
  {
    if(pass > 0)
      hipLaunchKernelGGL((k_sort_hist<false>), dim3(s.maxParts), dim3(kSortThreads), 0, stream, s.keysX, s.keysY, s.keys0,
                         (const uint32_t*)nullptr, s.nPtr, 0u, s.plan, s.partHist, s.pStride, pass, s.beginBit, nPasses);
    hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, stream, s.nPtr, pass == 0 ? s.slotCount : nullptr,
                       s.partsSlotted, s.plan, s.partHist, s.pStride, pass);
    if(pass == 0)
      hipLaunchKernelGGL((k_sort_scatter<true>), dim3(s.maxParts), dim3(kSortThreads), 0, stream, s.keys0, s.vals0, s.keysX,
                         s.valsX, s.keysY, s.valsY, s.slotCount, s.nPtr, s.partsSlotted, s.plan, s.partHist, s.pStride,
                         pass, s.beginBit);
    else
      hipLaunchKernelGGL((k_sort_scatter<false>), dim3(s.maxParts), dim3(kSortThreads), 0, stream, s.keys0, s.vals0, s.keysX,
                         s.valsX, s.keysY, s.valsY, (const uint32_t*)nullptr, s.nPtr, 0u, s.plan, s.partHist, s.pStride,
                         pass, s.beginBit);
  }
}

}  // namespace mgs
