// k_sort.hip — generic stable LSD radix sort of (u32 key, u32 value) pairs for gfx950: any bit range, device-side count.
// Users: the record-path pair sort of the binning fallback (> 256 bins) and the stand-alone sort API with a partial bit
// range.  The frame's depth-key sort is k_osort.hip (single-kernel passes).
//
// Behavioural spec: vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/include/vk_radix_sort.h:73-78,
// 3rdparty/vrdx/src/vk_radix_sort.cc:262-416): stable, ascending, 8-bit digits, element count read on
// the device.  The implementation is wave64-native:
//   * reduce-then-scan per pass (partition histograms -> per-digit row scan -> ranked scatter): no inter-workgroup
//     dependency inside a launch.
//   * ranking inside a partition uses 64-lane ballots (8 per key) to find the lanes holding the same
//     digit, one LDS counter row per wave, then an LDS re-order so the global scatter is coalesced.
//   * no global atomics anywhere: the scan kernel leaves each digit row's total in the plan and the
//     scatter workgroups turn the 256 totals into digit bases themselves.
//   * a pass whose digit is identical for every key (the high byte of a bin id) is detected by its scan kernel and
//     its scatter exits; the ping-pong selection is derived from the skip flags on the device.
#include <cstdlib>

#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

// partition p covers [p*part, p*part+count)
__device__ __forceinline__ uint32_t partitionCount(uint32_t n, uint32_t p, uint32_t part)
{
  const uint64_t base = (uint64_t)p * part;
  return (n > base) ? (uint32_t)min((uint64_t)part, (uint64_t)n - base) : 0u;
}

// LDS histogram add with run aggregation: lanes holding the same digit as their left neighbour are
// folded into the run's first lane, so long runs of equal digits (the high byte of a bin id) cost one
// LDS atomic instead of a 64-way serialised one.
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

// where pass `pass` (> 0) reads from: 0 = X, 1 = Y.  Pass 0 writes X; every executed pass flips.
__device__ __forceinline__ uint32_t planSrcSel(const SortPlan* __restrict__ plan, int pass)
{
  uint32_t cur = 0;
  for(int q = 1; q < pass; ++q)
    cur ^= plan->skip[q] ? 0u : 1u;
  return cur;
}

// (a) per-partition digit histogram of one pass
__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t* __restrict__ keysX, const uint32_t* __restrict__ keysY,
                                                   const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ nPtr,
                                                   const SortPlan* __restrict__ plan, uint32_t* __restrict__ partHist,
                                                   uint32_t pStride, int pass, int beginBit, uint32_t part)
{
  __shared__ uint32_t s_h[256];
  const int       t     = threadIdx.x;
  const uint32_t  n     = *nPtr;
  const uint32_t* keys  = (pass == 0) ? keys0 : (planSrcSel(plan, pass) ? keysY : keysX);
  const uint32_t  parts = (uint32_t)(((uint64_t)n + part - 1) / part);
  const int       shift = beginBit + 8 * pass;
  for(uint32_t p = blockIdx.x; p < parts; p += gridDim.x)
  {
    s_h[t] = 0;
    __syncthreads();
    const uint32_t  count = partitionCount(n, p, part);
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
        histAddRuns(s_h, (kk[u] >> shift) & 255u, i0 + (uint32_t)u * 256u + (uint32_t)t < count);
    }
    __syncthreads();
    partHist[(size_t)t * pStride + p] = s_h[t];
    __syncthreads();
  }
}

// (b) one workgroup per digit: exclusive scan of that digit's row of partition counts, in place.  The row
// total goes to plan->ghist[pass][d]; a row that holds every key marks the pass as skippable.
__global__ __launch_bounds__(256) void k_sort_scan(const uint32_t* __restrict__ nPtr, SortPlan* __restrict__ plan,
                                                   uint32_t* __restrict__ partHist, uint32_t pStride, int pass, uint32_t part)
{
  __shared__ uint32_t s_tmp[4];
  const int      t = threadIdx.x, d = blockIdx.x;
  const uint32_t n     = *nPtr;
  const uint32_t parts = (uint32_t)(((uint64_t)n + part - 1) / part);
  uint32_t       carry = 0;
  uint32_t*      row   = partHist + (size_t)d * pStride;
  for(uint32_t base = 0; base < parts; base += 2048)
  {
    const uint32_t p0 = base + t * 8;
    uint32_t       v[8], sum = 0;
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
      v[i] = (p0 + i < parts) ? row[p0 + i] : 0u;
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
template <bool FIRST, int THREADS, int KPT>
__global__ __launch_bounds__(THREADS) void k_sort_scatter(const uint32_t* __restrict__ keys0, const uint32_t* __restrict__ vals0,
                                                          uint32_t* __restrict__ keysX, uint32_t* __restrict__ valsX,
                                                          uint32_t* __restrict__ keysY, uint32_t* __restrict__ valsY,
                                                          const uint32_t* __restrict__ nPtr, SortPlan* __restrict__ plan,
                                                          const uint32_t* __restrict__ partHist, uint32_t pStride, int pass,
                                                          int beginBit, int nPasses, uint2* __restrict__ ranges)
{
  constexpr int PART  = THREADS * KPT;
  constexpr int WAVES = THREADS / 64;
  __shared__ uint32_t s_whist[WAVES][256];
  __shared__ uint32_t s_k[PART];
  __shared__ uint32_t s_v[PART];
  __shared__ uint32_t s_loff[256];
  __shared__ uint32_t s_gbase[256];
  __shared__ uint32_t s_tmp[WAVES];

  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  const bool     skipped = !FIRST && plan->skip[pass] != 0u;
  const uint32_t n       = *nPtr;
  if(blockIdx.x == 0 && t == 0 && pass == nPasses - 1)
  {
    uint32_t run = 1;
    for(int q = 1; q < nPasses; ++q)
      run += plan->skip[q] ? 0u : 1u;
    plan->finalSel  = planSrcSel(plan, nPasses);
    plan->passesRun = run;
    plan->n         = n;
  }
  if(skipped)
    return;
  const uint32_t parts = (uint32_t)(((uint64_t)n + PART - 1) / PART);
  const uint32_t p     = blockIdx.x;
  if(p >= parts)
    return;
  const uint32_t count = partitionCount(n, p, PART);
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
    const size_t   src = (size_t)p * PART + idx;
    key[i] = in ? kin[src] : 0xFFFFFFFFu;
    val[i] = in ? vin[src] : 0u;
  }
  __syncthreads();

  // per-wave multi-split: rank of each key among the keys of its wave with the same digit
  uint32_t rank[KPT];
  uint8_t  dig[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    // padding keys (idx >= count) carry digit 255 and sit behind every real key of the partition
    const uint32_t d = (key[i] >> shift) & 255u;
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
    s_k[pos]           = key[i];
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
      const uint32_t dst = s_gbase[(k >> shift) & 255u] + idx;
      kout[dst]          = k;
      vout[dst]          = v;
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
  // big sorts use 4096-key partitions (digit runs of ~16 keys = 64-byte scatter segments), small ones keep 2048 so
  // that 256 CUs still see enough workgroups
  static const uint32_t kPartOverride = [] { const char* e = std::getenv("MGS_SORT_PART"); return e ? (uint32_t)std::atoi(e) : 0u; }();
  const uint32_t part  = kPartOverride ? kPartOverride : ((s.maxElems >= (2u << 20)) ? 4096u : 2048u);
  const uint32_t parts = (uint32_t)(((uint64_t)s.maxElems + part - 1) / part);
  for(int pass = 0; pass < nPasses; ++pass)
  {
    hipLaunchKernelGGL(k_sort_hist, dim3(parts), dim3(256), 0, stream, s.keysX, s.keysY, s.keys0, s.nPtr, s.plan, s.partHist, s.pStride,
                       pass, s.beginBit, part);
    hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, stream, s.nPtr, s.plan, s.partHist, s.pStride, pass, part);
#define MGS_SCATTER(FIRSTV, TH, KP)                                                                                              \
  hipLaunchKernelGGL((k_sort_scatter<FIRSTV, TH, KP>), dim3(parts), dim3(TH), 0, stream, s.keys0, s.vals0, s.keysX, s.valsX, s.keysY, \
                     s.valsY, s.nPtr, s.plan, s.partHist, s.pStride, pass, s.beginBit, nPasses, s.ranges)
    if(pass == 0)
    {
      if(part == 2048u)
        MGS_SCATTER(true, 256, 8);
      else if(part == 4096u)
        MGS_SCATTER(true, 256, 16);
      else
        MGS_SCATTER(true, 512, 16);
    }
    else
    {
      if(part == 2048u)
        MGS_SCATTER(false, 256, 8);
      else if(part == 4096u)
        MGS_SCATTER(false, 256, 16);
      else
        MGS_SCATTER(false, 512, 16);
    }
#undef MGS_SCATTER
  }
}

}  // namespace mgs
