// k_ssort.hip — two-round-trip stable sort of (u32 key, u32 value) pairs for gfx950: a SAMPLE SORT whose second half
// runs entirely in the 160 KB LDS of a CU.
//
// Behavioural spec (unchanged): vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/include/vk_radix_sort.h:73-78,
// 3rdparty/vrdx/src/vk_radix_sort.cc:262-416): stable, ascending, full 32-bit keys, element count read on the device.
//
// Why not four LSD passes (k_sort.hip, kept for partial bit ranges): depth keys are a terrible radix input — on the
// benchmark frame 70 % of 4.2 M keys fall into ~370 K consecutive values of one exponent, so the top byte is (almost)
// constant, yet a handful of near-plane splats keeps every pass alive: 4 global round trips + 7 latency-floor
// histogram/scan launches = 0.15 ms.  What MI355X has that the reference's target did not is 160 KB of LDS per CU:
//   1. k_ss_sample + k_ss_splitters: <= 16 K keys sampled proportionally from the input, sorted by ONE workgroup in LDS;
//      every (S/nb)-th sample becomes a splitter.  nb = ceil(n / 3584) buckets, chosen on the device from the
//      device-side count (a strip's 40 K keys get 12 buckets, a 12 K-key sort gets one: a single-workgroup LDS sort).
//      Splitters are quantiles of the actual keys, so the buckets are balanced whatever the key distribution is.
//   2. k_ss_classify: one workgroup per 2048-key partition (== one slot of the project kernel).  bucket(key) =
//      #splitters <= key (binary search in LDS), a stable wave64 ballot multi-split orders the partition by bucket,
//      and the partition is written back IN PLACE (fully coalesced).  Per non-empty (partition, bucket) pair one
//      8-byte descriptor (source index, partition, count) is appended to the bucket's list with one 64-bit atomic that
//      also accumulates the bucket total.  No histogram matrix, no scan kernel, no scatter to bucket regions.
//   3. k_ss_finish: one 1024-thread workgroup per bucket.  The descriptors arrive in atomic (= arbitrary) order; a
//      bitmap over partition indices turns each into its canonical rank (ascending partition == ascending input
//      position), so the slices are laid into LDS in INPUT ORDER and a stable LSD sort of the bits in which the
//      bucket's keys actually differ (typically 8-10: one or two LDS passes) yields exactly what a stable global sort
//      would — bit-identical to std::stable_sort, independent of the order the atomics retired in.  The bucket's
//      output offset is the sum of the totals of the buckets before it (computed by the workgroup itself).
//   A bucket that does not fit in LDS (> 12288 keys: only possible when few distinct key values dominate, because
//   equal keys cannot be split) is sorted by the same workgroup streaming through global memory — slow, never wrong,
//   never hangs; an all-equal bucket is just copied.
// Global traffic: 8 B/key read + 8 written by classify, 8 + 8 by finish (+ descriptors, ~0.4 B/key) = 32 B/key
// against 68 B/key for the four-pass algorithm the roofline is priced on; 4 launches instead of 11.
#include <cstdlib>

#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

constexpr int      kSsPart       = 2048;   // classify partition == slot of the project kernel
constexpr int      kSsMaxSamples = 16384;  // 64 KB of LDS in the splitter kernel (~7 samples per bucket at 2400 buckets)
constexpr int      kSsFinThreads = 512;    // two finisher workgroups per CU (78 KB of LDS each): one hides the other's
constexpr int      kSsFinKpt     = 12;     // global-memory phases
constexpr int      kSsCap        = kSsFinThreads * kSsFinKpt;  // 6144 keys per LDS bucket
constexpr uint32_t kSsTarget     = 1792;   // mean keys per bucket: max/mean of sampled quantiles stays < 3.4

// plan->reserved[]: [1] sample cursor, [2] number of buckets in use, [3] buckets that took the streaming path
enum { kSsSampleCursor = 1, kSsBucketCount = 2, kSsBigBuckets = 3 };

__device__ __forceinline__ uint32_t ssPartitionCount(const uint32_t* slotCount, uint32_t n, uint32_t p)
{
  if(slotCount)
    return slotCount[p];
  const uint64_t base = (uint64_t)p * kSsPart;
  return (n > base) ? (uint32_t)min((uint64_t)kSsPart, (uint64_t)n - base) : 0u;
}

// exclusive scan of one value per thread over a THREADS-wide block (THREADS/64 <= 16 waves)
template <int THREADS>
__device__ __forceinline__ uint32_t blockExclusiveScan(uint32_t v, uint32_t* s_tmp /*THREADS/64*/, uint32_t* total)
{
  constexpr int  WAVES = THREADS / 64;
  const int      lane = laneId(), w = threadIdx.x >> 6;
  const uint32_t inc  = waveInclusiveScan(v);
  if(lane == 63)
    s_tmp[w] = inc;
  __syncthreads();
  uint32_t base = 0, sum = 0;
#pragma unroll
  for(int q = 0; q < WAVES; ++q)
  {
    const uint32_t c = s_tmp[q];
    base += (q < w) ? c : 0u;
    sum += c;
  }
  *total = sum;
  __syncthreads();
  return base + inc - v;
}

template <int THREADS>
__device__ __forceinline__ uint32_t blockMax(uint32_t v, uint32_t* s_tmp /*THREADS/64*/)
{
  constexpr int WAVES = THREADS / 64;
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
    v = max(v, (uint32_t)__shfl_xor(v, o, 64));
  if(laneId() == 0)
    s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t m = 0;
#pragma unroll
  for(int q = 0; q < WAVES; ++q)
    m = max(m, s_tmp[q]);
  __syncthreads();
  return m;
}

// Stable LSD sort, in LDS, of the low `hiBits` bits of THREADS*KPT (key[, value]) pairs; elements beyond the real count
// must hold key 0xFFFFFFFF (they stay at the end).  Wave-striped ownership: (round i, lane) order == memory order
// inside a wave, waves in memory order, so ballot ranks are ranks in input order.
template <int THREADS, int KPT, bool VALS>
__device__ __forceinline__ void ldsSortBits(uint32_t* s_key, uint32_t* s_val, uint32_t* s_whist /*[THREADS/64][256]*/,
                                            uint32_t* s_dbase /*256*/, uint32_t* s_tmp /*THREADS/64*/, int hiBits)
{
  constexpr int WAVES = THREADS / 64;
  const int     t = threadIdx.x, lane = laneId(), w = t >> 6;
  for(int shift = 0; shift < hiBits; shift += 8)
  {
    for(int i = t; i < WAVES * 256; i += THREADS)
      s_whist[i] = 0;
    uint32_t key[KPT], val[VALS ? KPT : 1], rank[KPT];
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
      const int idx = w * (64 * KPT) + i * 64 + lane;
      key[i]        = s_key[idx];
      if constexpr(VALS)
        val[i] = s_val[idx];
    }
    __syncthreads();
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
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
      const uint32_t pre   = s_whist[w * 256 + d];
      rank[i]              = pre + lower;
      __builtin_amdgcn_wave_barrier();  // every lane of the group has read `pre` before the leader bumps it
      if(lower == 0)
        s_whist[w * 256 + d] = pre + cnt;
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    uint32_t tot = 0;
    if(t < 256)
    {
      uint32_t acc = 0;
#pragma unroll
      for(int q = 0; q < WAVES; ++q)
      {
        const uint32_t c     = s_whist[q * 256 + t];
        s_whist[q * 256 + t] = acc;
        acc += c;
      }
      tot = acc;
    }
    uint32_t       dummy;
    const uint32_t below = blockExclusiveScan<THREADS>(tot, s_tmp, &dummy);  // meaningful for t < 256 (others add 0 behind them)
    if(t < 256)
      s_dbase[t] = below;
    __syncthreads();
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
      const uint32_t d   = (key[i] >> shift) & 255u;
      const uint32_t pos = s_dbase[d] + s_whist[w * 256 + d] + rank[i];
      s_key[pos]         = key[i];
      if constexpr(VALS)
        s_val[pos] = val[i];
    }
    __syncthreads();
  }
}

// ---- 1a. proportional sample of the input keys ------------------------------------------------------------------
// One thread per partition.  A partition with c keys contributes about c/R of them (R = smallest power of two with
// n/R <= 16 K; probabilistic rounding by a hash of the partition index, so thinly populated partitions are not
// systematically under-sampled), evenly spaced.  Also zeroes the bucket counters of this sort.
__global__ __launch_bounds__(256) void k_ss_sample(const uint32_t* keys0, const uint32_t* __restrict__ slotCount,
                                                   const uint32_t* __restrict__ nPtr, uint32_t parts, SortPlan* plan,
                                                   uint32_t* __restrict__ samples, unsigned long long* __restrict__ bucketCount,
                                                   uint32_t maxBuckets)
{
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  for(uint32_t i = gid; i < maxBuckets; i += gridDim.x * blockDim.x)
    bucketCount[i] = 0ull;
  const uint32_t n = *nPtr;
  uint32_t       R = 1;  // expected sample count n/R + parts/2 (rounding) must fit; beyond R = 2048 the tail is dropped
  while(R < 2048u && (n + R - 1) / R + parts / 2 > (uint32_t)kSsMaxSamples)
    R <<= 1;
  uint32_t cnt = 0, m = 0;
  if(gid < parts)
  {
    cnt = ssPartitionCount(slotCount, n, gid);
    m   = (cnt + (((gid * 0x9E3779B1u) >> 9) & (R - 1u))) / R;
    m   = min(m, cnt);
  }
  const uint32_t inc  = waveInclusiveScan(m);
  const uint32_t wtot = __shfl(inc, 63, 64);
  uint32_t       base = 0;
  if(laneId() == 0 && wtot)
    base = atomicAdd(&plan->reserved[kSsSampleCursor], wtot);
  base = __shfl(base, 0, 64) + inc - m;
  const uint32_t* src = keys0 + (size_t)gid * kSsPart;
  for(uint32_t j = 0; j < m; ++j)
  {
    const uint32_t idx = (uint32_t)(((uint64_t)(2 * j + 1) * cnt) / (2 * m));
    if(base + j < (uint32_t)kSsMaxSamples)
      samples[base + j] = src[idx];
  }
}

// ---- 1b. splitters: one workgroup sorts the sample in LDS and picks quantiles --------------------------------------
__global__ __launch_bounds__(1024) void k_ss_splitters(SortPlan* plan, const uint32_t* __restrict__ samples,
                                                       uint32_t* __restrict__ splitters, const uint32_t* __restrict__ nPtr,
                                                       uint32_t maxBuckets)
{
  constexpr int THREADS = 1024, KPT = kSsMaxSamples / THREADS;  // 16
  __shared__ uint32_t s_key[kSsMaxSamples];
  __shared__ uint32_t s_whist[16 * 256];
  __shared__ uint32_t s_dbase[256];
  __shared__ uint32_t s_tmp[16];
  const int      t = threadIdx.x;
  const uint32_t S = min(plan->reserved[kSsSampleCursor], (uint32_t)kSsMaxSamples);
  const uint32_t n = *nPtr;
  uint32_t       lo = 0xFFFFFFFFu, hi = 0u;
  for(int i = t; i < kSsMaxSamples; i += THREADS)
  {
    const uint32_t k = (uint32_t)i < S ? samples[i] : 0xFFFFFFFFu;
    s_key[i]         = k;
    if((uint32_t)i < S)
    {
      lo = min(lo, k);
      hi = max(hi, k);
    }
  }
  // only the bits in which the samples differ need sorting: everything above the highest set bit of min ^ max is common
  const uint32_t gmax = blockMax<THREADS>(hi, s_tmp);
  const uint32_t gmin = ~blockMax<THREADS>(~lo, s_tmp);
  const uint32_t x      = S ? (gmin ^ gmax) : 0u;
  const int      hiBits = x ? 32 - __builtin_clz(x) : 0;
  __syncthreads();
  // the padding keys 0xFFFFFFFF must stay behind the real ones: they do if every sorted digit of a pad is 255, which
  // holds for any bit range
  ldsSortBits<THREADS, KPT, false>(s_key, nullptr, s_whist, s_dbase, s_tmp, hiBits);
  uint32_t nb = (uint32_t)(((uint64_t)n + kSsTarget - 1) / kSsTarget);
  nb          = max(1u, min(nb, maxBuckets));
  nb          = min(nb, max(1u, S));
  for(uint32_t j = t; j + 1 < nb; j += THREADS)
    splitters[j] = s_key[(uint32_t)(((uint64_t)(j + 1) * S) / nb)];
  if(t == 0)
    plan->reserved[kSsBucketCount] = nb;
}

// ---- 2. classify: partition-local stable multi-split by bucket, in place, + descriptors ---------------------------
// Written for throughput, not latency: the 8 binary searches of a thread advance in lock step (8 independent LDS reads
// per step), the four wave counters of a bucket sit in one 8-byte word (one LDS access per bucket in the scan), and the
// non-empty buckets are compacted so that every descriptor atomic is issued by a different thread.
template <int NBMAX>
__global__ __launch_bounds__(256) void k_ss_classify(uint32_t* keys0, uint32_t* vals0, const uint32_t* __restrict__ slotCount,
                                                     const uint32_t* __restrict__ nPtr, const SortPlan* __restrict__ plan,
                                                     const uint32_t* __restrict__ splitters,
                                                     unsigned long long* __restrict__ bucketCount, uint2* __restrict__ desc,
                                                     uint32_t descStride)
{
  constexpr int THREADS = 256, KPT = kSsPart / THREADS, WAVES = 4, DPT = NBMAX / THREADS;
  __shared__ uint16_t s_wh[NBMAX][WAVES];  // [bucket][wave]: 8 bytes per bucket
  __shared__ uint16_t s_loff[NBMAX];
  __shared__ uint32_t s_split[NBMAX];
  __shared__ uint32_t s_k[kSsPart];
  __shared__ uint32_t s_v[kSsPart];
  __shared__ uint32_t s_ne[kSsPart];       // compacted non-empty buckets: bucket << 12 | (count - 1)
  __shared__ uint32_t s_tmp[WAVES];
  const int      t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t p     = blockIdx.x;
  const uint32_t n     = *nPtr;
  const uint32_t count = ssPartitionCount(slotCount, n, p);
  if(count == 0)
    return;
  const uint32_t nb    = plan->reserved[kSsBucketCount];
  const size_t   gbase = (size_t)p * kSsPart;
  const uint32_t wofs  = w * (64 * KPT);
  uint32_t       key[KPT], val[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t idx = min(wofs + i * 64 + lane, count - 1u);  // clamped, not predicated: 16 loads in flight
    key[i]             = keys0[gbase + idx];
    val[i]             = vals0[gbase + idx];
  }
  for(uint32_t i = t; i + 1 < nb; i += THREADS)
    s_split[i] = splitters[i];
  for(int i = t; i < NBMAX * WAVES / 2; i += THREADS)
    reinterpret_cast<uint32_t*>(&s_wh[0][0])[i] = 0u;
#pragma unroll
  for(int i = 0; i < KPT; ++i)
    if(wofs + i * 64 + lane >= count)
    {
      key[i] = 0xFFFFFFFFu;
      val[i] = 0u;
    }
  __syncthreads();
  // bucket = number of splitters <= key (upper bound): equal keys always share a bucket
  int steps = 0;
  while((1u << steps) < nb)
    ++steps;
  uint32_t lo[KPT], hi[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    lo[i] = 0;
    hi[i] = nb - 1;  // answer in [lo, hi]
  }
  for(int s = 0; s < steps; ++s)
  {
    uint32_t sv[KPT];
#pragma unroll
    for(int i = 0; i < KPT; ++i)
      sv[i] = s_split[min((lo[i] + hi[i]) >> 1, nb - 2u + (nb < 2u ? 1u : 0u))];
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
      const uint32_t mid  = (lo[i] + hi[i]) >> 1;
      const bool     open = lo[i] < hi[i];
      const bool     ge   = open && (sv[i] <= key[i]);
      const bool     lt   = open && !ge;
      lo[i]               = ge ? mid + 1 : lo[i];
      hi[i]               = lt ? mid : hi[i];
    }
  }
  uint32_t rank[KPT];
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    if(wofs + i * 64 + lane >= count)
      lo[i] = nb - 1;  // padding rides at the end of the last bucket
    const uint32_t d = lo[i];
    uint64_t       m = ~0ull;
    for(int b = 0; b < steps; ++b)
    {
      const bool     bit = (d >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const uint32_t lower = lanesBelow(m);
    const uint32_t cnt   = (uint32_t)__popcll(m);
    const uint32_t pre   = s_wh[d][w];
    rank[i]              = pre + lower;
    __builtin_amdgcn_wave_barrier();
    if(lower == 0)
      s_wh[d][w] = (uint16_t)(pre + cnt);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // thread t owns buckets [t*DPT, (t+1)*DPT): one 8-byte LDS word per bucket holds the four wave counts
  uint32_t tot[DPT], sum = 0, ne = 0;
  {
    uint2 c[DPT];
#pragma unroll
    for(int j = 0; j < DPT; ++j)
      c[j] = *reinterpret_cast<const uint2*>(&s_wh[t * DPT + j][0]);
#pragma unroll
    for(int j = 0; j < DPT; ++j)
    {
      const uint32_t c0 = c[j].x & 0xFFFFu, c1 = c[j].x >> 16, c2 = c[j].y & 0xFFFFu, c3 = c[j].y >> 16;
      tot[j]            = c0 + c1 + c2 + c3;
      *reinterpret_cast<uint2*>(&s_wh[t * DPT + j][0]) = make_uint2(0u | (c0 << 16), (c0 + c1) | ((c0 + c1 + c2) << 16));
      sum += tot[j];
    }
  }
  const uint32_t pads = (uint32_t)kSsPart - count;
#pragma unroll
  for(int j = 0; j < DPT; ++j)
  {
    if((uint32_t)(t * DPT + j) == nb - 1)
      tot[j] |= 0x80000000u;  // remember: this one carries the padding (real count = tot - pads)
    const uint32_t real = (tot[j] & 0x7FFFFFFFu) - ((tot[j] >> 31) ? pads : 0u);
    ne += (real > 0 && (uint32_t)(t * DPT + j) < nb) ? 1u : 0u;
  }
  uint32_t dummy, neTotal;
  uint32_t run   = blockExclusiveScan<THREADS>(sum, s_tmp, &dummy);
  uint32_t neRun = blockExclusiveScan<THREADS>(ne, s_tmp, &neTotal);
#pragma unroll
  for(int j = 0; j < DPT; ++j)
  {
    const uint32_t d    = t * DPT + j;
    const uint32_t full = tot[j] & 0x7FFFFFFFu;
    const uint32_t real = full - ((tot[j] >> 31) ? pads : 0u);
    s_loff[d]           = (uint16_t)run;
    if(real > 0 && d < nb)
      s_ne[neRun++] = (d << 12) | (real - 1u);
    run += full;
  }
  __syncthreads();
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t d   = lo[i];
    const uint32_t pos = (uint32_t)s_loff[d] + (uint32_t)s_wh[d][w] + rank[i];
    s_k[pos]           = key[i];
    s_v[pos]           = val[i];
  }
  // one descriptor per non-empty bucket, one thread each: the atomics of a partition are all in flight together
  for(uint32_t e = t; e < neTotal; e += THREADS)
  {
    const uint32_t           ent = s_ne[e];
    const uint32_t           d = ent >> 12, c = (ent & 4095u) + 1u;
    const unsigned long long old  = atomicAdd(&bucketCount[d], (1ull << 32) | (unsigned long long)c);
    const uint32_t           slot = (uint32_t)(old >> 32);
    desc[(size_t)d * descStride + slot] = make_uint2((uint32_t)(gbase + s_loff[d]), (p << 11) | (c - 1u));
  }
  __syncthreads();
#pragma unroll
  for(int i = 0; i < KPT; ++i)
  {
    const uint32_t idx = i * THREADS + t;
    if(idx < count)
    {
      keys0[gbase + idx] = s_k[idx];
      vals0[gbase + idx] = s_v[idx];
    }
  }
}

// relaxed agent-scope accesses (L2-served): what the streaming path uses for data one wave writes and another wave of the
// same workgroup reads a sweep later
__device__ __forceinline__ uint32_t ldg(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void     stg(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- 3. finish: one workgroup per bucket ---------------------------------------------------------------------------
__global__ __launch_bounds__(kSsFinThreads) void k_ss_finish(const uint32_t* __restrict__ keysIn, const uint32_t* __restrict__ valsIn,
                                                             uint32_t* keysX, uint32_t* valsX, uint32_t* keysY, uint32_t* valsY,
                                                             const uint32_t* __restrict__ nPtr, SortPlan* plan,
                                                             const unsigned long long* __restrict__ bucketCount,
                                                             const uint2* __restrict__ desc, uint32_t descStride,
                                                             const uint32_t* __restrict__ gatherSrc, uint32_t* __restrict__ gatherDst,
                                                             int dbgStop)
{
  constexpr int THREADS = kSsFinThreads, KPT = kSsFinKpt, CAP = kSsCap, WAVES = THREADS / 64;
  __shared__ uint32_t s_key[CAP];
  __shared__ uint32_t s_id[CAP];
  __shared__ uint32_t s_whist[WAVES * 256 > 4096 ? WAVES * 256 : 4096];  // also: the partition bitmap (<= 4096 words = 131072 partitions)
  __shared__ uint16_t s_bpre[4096];          // popcount prefix per bitmap word
  __shared__ uint16_t s_cnt[CAP];            // slice sizes by canonical rank -> LDS offsets
  __shared__ uint32_t s_dbase[256];
  __shared__ uint32_t s_tmp[WAVES];
  __shared__ uint32_t s_run[256];
  const int      t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t b = blockIdx.x;
  const uint32_t n = *nPtr;
  if(b == 0 && t == 0)
  {
    plan->finalSel  = 0;  // the result is always in X
    plan->passesRun = 2;
    plan->n         = n;
    if(gatherDst != nullptr)
      plan->reserved[0] = 1u;  // gatherDst is valid: consumers need not gather themselves
  }
  const uint32_t nb = plan->reserved[kSsBucketCount];
  if(b >= nb)
    return;
  const unsigned long long packed = bucketCount[b];
  const uint32_t           D = (uint32_t)(packed >> 32), T = (uint32_t)packed;
  if(T == 0)
    return;
  if(t == 0)
    atomicAdd(&plan->pad[0], D);  // statistics: slices of this sort
  uint32_t base;
  {
    uint32_t part = 0;
    for(uint32_t j = t; j < b; j += THREADS)
      part += (uint32_t)bucketCount[j];
    part = waveSum(part);
    if(lane == 0)
      s_tmp[w] = part;
    __syncthreads();
    base = 0;
#pragma unroll
    for(int q = 0; q < WAVES; ++q)
      base += s_tmp[q];
    __syncthreads();
  }
  if(dbgStop == 1) return;
  const uint2* dlist = desc + (size_t)b * descStride;
  // canonical rank of a slice = number of this bucket's slices from lower partitions: bitmap + popcount prefix
  uint32_t*      s_bm  = s_whist;
  const uint32_t words = (descStride + 31u) >> 5;  // <= 4096 (host-checked)
  for(uint32_t i = t; i < words; i += THREADS)
    s_bm[i] = 0u;
  __syncthreads();
  for(uint32_t j = t; j < D; j += THREADS)
  {
    const uint32_t part = dlist[j].y >> 11;
    atomicOr(&s_bm[part >> 5], 1u << (part & 31u));
  }
  __syncthreads();
  {
    constexpr int WPT = 4096 / THREADS;  // bitmap words per thread
    uint32_t      c[WPT], sum = 0;
#pragma unroll
    for(int q = 0; q < WPT; ++q)
    {
      const uint32_t wi = t * WPT + q;
      c[q]              = wi < words ? (uint32_t)__popc(s_bm[wi]) : 0u;
      sum += c[q];
    }
    uint32_t dummy;
    uint32_t run = blockExclusiveScan<THREADS>(sum, s_tmp, &dummy);
#pragma unroll
    for(int q = 0; q < WPT; ++q)
    {
      const uint32_t wi = t * WPT + q;
      if(wi < words)
        s_bpre[wi] = (uint16_t)min(run, 65535u);  // > 65535 slices only on the streaming path, which re-derives ranks in 32 bits
      run += c[q];
    }
  }
  __syncthreads();
  auto rankOf = [&](uint32_t part) { return (uint32_t)s_bpre[part >> 5] + (uint32_t)__popc(s_bm[part >> 5] & ((1u << (part & 31u)) - 1u)); };
  if(dbgStop == 2) return;

  if(T <= (uint32_t)CAP)
  {
    // ---- LDS path ---------------------------------------------------------------------------------------------
    for(uint32_t j = t; j < D; j += THREADS)
    {
      const uint2 d       = dlist[j];
      s_cnt[rankOf(d.y >> 11)] = (uint16_t)((d.y & 2047u) + 1u);
    }
    __syncthreads();
    {  // exclusive scan of s_cnt[0..D) in place (D <= T <= CAP; sums <= CAP fit 16 bits)
      uint32_t v[KPT], sum = 0;
#pragma unroll
      for(int q = 0; q < KPT; ++q)
      {
        const uint32_t i = t * KPT + q;
        v[q]             = i < D ? (uint32_t)s_cnt[i] : 0u;
        sum += v[q];
      }
      uint32_t dummy;
      uint32_t run = blockExclusiveScan<THREADS>(sum, s_tmp, &dummy);
#pragma unroll
      for(int q = 0; q < KPT; ++q)
      {
        const uint32_t i = t * KPT + q;
        if(i < D)
          s_cnt[i] = (uint16_t)run;
        run += v[q];
      }
    }
    // slice heads: s_key[o] = o + 1 marks a head, s_id[o] = its source index
    for(int i = t; i < CAP; i += THREADS)
      s_key[i] = 0u;
    __syncthreads();
    for(uint32_t j = t; j < D; j += THREADS)
    {
      const uint2    d = dlist[j];
      const uint32_t o = s_cnt[rankOf(d.y >> 11)];
      s_key[o]         = o + 1u;
      s_id[o]          = d.x;
    }
    __syncthreads();
    if(dbgStop == 3) return;
    {  // inclusive max-scan of the head marks: every element learns the head of its slice
      uint32_t v[KPT], mx = 0;
#pragma unroll
      for(int q = 0; q < KPT; ++q)
      {
        mx   = max(mx, s_key[t * KPT + q]);
        v[q] = mx;
      }
      // exclusive max over the threads before this one
      uint32_t pre = mx;
#pragma unroll
      for(int o = 1; o < 64; o <<= 1)
      {
        const uint32_t u = __shfl_up(pre, o, 64);
        if(lane >= o)
          pre = max(pre, u);
      }
      if(lane == 63)
        s_tmp[w] = pre;
      uint32_t excl = __shfl_up(pre, 1, 64);
      if(lane == 0)
        excl = 0u;
      __syncthreads();
      uint32_t wpre = 0;
#pragma unroll
      for(int q = 0; q < WAVES; ++q)
        wpre = max(wpre, q < w ? s_tmp[q] : 0u);
      excl = max(excl, wpre);
#pragma unroll
      for(int q = 0; q < KPT; ++q)
        s_key[t * KPT + q] = max(v[q], excl);
    }
    __syncthreads();
    if(dbgStop == 4) return;
    uint32_t k[KPT], v[KPT];
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
      const uint32_t e = i * THREADS + t;
      k[i]             = 0xFFFFFFFFu;
      v[i]             = 0u;
      if(e < T)
      {
        const uint32_t h   = s_key[e] - 1u;
        const uint32_t src = s_id[h] + (e - h);
        k[i]               = keysIn[src];
        v[i]               = valsIn[src];
      }
    }
#pragma unroll
    for(int i = 0; i < KPT; ++i)
      if((uint32_t)(i * THREADS + t) < T)
      {
        lo = min(lo, k[i]);
        hi = max(hi, k[i]);
      }
    __syncthreads();  // every head has been read
#pragma unroll
    for(int i = 0; i < KPT; ++i)
    {
      s_key[i * THREADS + t] = k[i];
      s_id[i * THREADS + t]  = v[i];
    }
    if(dbgStop == 5) return;
    const uint32_t gmax = blockMax<THREADS>(hi, s_tmp);
    const uint32_t gmin = ~blockMax<THREADS>(~lo, s_tmp);
    const uint32_t x    = gmin ^ gmax;
    ldsSortBits<THREADS, KPT, true>(s_key, s_id, s_whist, s_dbase, s_tmp, x ? 32 - __builtin_clz(x) : 0);
    if(dbgStop == 6) return;
    for(uint32_t i = t; i < T; i += THREADS)
    {
      const uint32_t id = s_id[i];
      if(gatherDst != nullptr)
        gatherDst[base + i] = gatherSrc[id];
      else
        keysX[base + i] = s_key[i];
      valsX[base + i] = id;
    }
    if(gatherDst != nullptr)  // the keys are still wanted by the sort-download hook: second, independent store stream
      for(uint32_t i = t; i < T; i += THREADS)
        keysX[base + i] = s_key[i];
    return;
  }

  // ---- streaming path: the bucket does not fit in LDS ------------------------------------------------------------
  if(t == 0)
    atomicAdd(&plan->reserved[kSsBigBuckets], 1u);
  // slice sizes by canonical rank -> keysY[base + r] (T >= D), scanned into valsY[base + r]; ranks in 32 bits
  {
    // 32-bit popcount prefix: recompute per word group (s_bpre saturates at 65535)
    for(uint32_t j = t; j < D; j += THREADS)
    {
      const uint2    d    = dlist[j];
      const uint32_t part = d.y >> 11;
      uint32_t       r    = (uint32_t)__popc(s_bm[part >> 5] & ((1u << (part & 31u)) - 1u));
      if(D < 65535u)
        r += s_bpre[part >> 5];
      else
        for(uint32_t q = 0; q < (part >> 5); ++q)
          r += (uint32_t)__popc(s_bm[q]);
      stg(&keysY[base + r], (d.y & 2047u) + 1u);
    }
    __syncthreads();
    uint32_t carry = 0;
    for(uint32_t c0 = 0; c0 < D; c0 += THREADS)
    {
      const uint32_t i = c0 + t;
      const uint32_t v = i < D ? ldg(&keysY[base + i]) : 0u;
      uint32_t       total;
      const uint32_t ex = blockExclusiveScan<THREADS>(v, s_tmp, &total);
      if(i < D)
        stg(&valsY[base + i], carry + ex);
      carry += total;
    }
    __syncthreads();
    for(uint32_t j = w; j < D; j += WAVES)  // one slice per wave
    {
      const uint2    d    = dlist[j];
      const uint32_t part = d.y >> 11, cnt = (d.y & 2047u) + 1u;
      uint32_t       r    = (uint32_t)__popc(s_bm[part >> 5] & ((1u << (part & 31u)) - 1u));
      if(D < 65535u)
        r += s_bpre[part >> 5];
      else
        for(uint32_t q = 0; q < (part >> 5); ++q)
          r += (uint32_t)__popc(s_bm[q]);
      const uint32_t o = ldg(&valsY[base + r]);
      for(uint32_t l = lane; l < cnt; l += 64)
      {
        stg(&keysX[base + o + l], keysIn[d.x + l]);
        stg(&valsX[base + o + l], valsIn[d.x + l]);
      }
    }
    __syncthreads();
  }
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  for(uint32_t i = t; i < T; i += THREADS)
  {
    const uint32_t k = ldg(&keysX[base + i]);
    lo               = min(lo, k);
    hi               = max(hi, k);
  }
  const uint32_t gmax = blockMax<THREADS>(hi, s_tmp);
  const uint32_t gmin = ~blockMax<THREADS>(~lo, s_tmp);
  const uint32_t x    = gmin ^ gmax;
  const int      hiBits = x ? 32 - __builtin_clz(x) : 0;
  uint32_t *curK = keysX, *curV = valsX, *othK = keysY, *othV = valsY;
  for(int shift = 0; shift < hiBits; shift += 8)
  {
    if(t < 256)
      s_run[t] = 0u;
    __syncthreads();
    for(uint32_t i = t; i < T; i += THREADS)
      atomicAdd(&s_run[(ldg(&curK[base + i]) >> shift) & 255u], 1u);
    __syncthreads();
    {
      uint32_t       dummy;
      const uint32_t v  = t < 256 ? s_run[t] : 0u;
      const uint32_t ex = blockExclusiveScan<THREADS>(v, s_tmp, &dummy);
      if(t < 256)
        s_run[t] = ex;  // running base of every digit
    }
    __syncthreads();
    for(uint32_t c0 = 0; c0 < T; c0 += THREADS)
    {
      const uint32_t i     = c0 + t;
      const bool     valid = i < T;
      const uint32_t key   = valid ? ldg(&curK[base + i]) : 0u;
      const uint32_t val   = valid ? ldg(&curV[base + i]) : 0u;
      const uint32_t d     = (key >> shift) & 255u;
      for(int q = t; q < WAVES * 256; q += THREADS)
        s_whist[q] = 0u;
      __syncthreads();
      uint64_t m = __ballot(valid);
      m          = valid ? m : ~m;
#pragma unroll
      for(int bb = 0; bb < 8; ++bb)
      {
        const bool     bit = (d >> bb) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      const uint32_t lower = lanesBelow(m);
      if(valid && lower == 0)
        s_whist[w * 256 + d] = (uint32_t)__popcll(m);
      __syncthreads();
      if(t < 256)
      {
        uint32_t acc = s_run[t];
#pragma unroll
        for(int q = 0; q < WAVES; ++q)
        {
          const uint32_t c     = s_whist[q * 256 + t];
          s_whist[q * 256 + t] = acc;
          acc += c;
        }
        s_run[t] = acc;
      }
      __syncthreads();
      if(valid)
      {
        const uint32_t dst = s_whist[w * 256 + d] + lower;
        stg(&othK[base + dst], key);
        stg(&othV[base + dst], val);
      }
      __syncthreads();
    }
    uint32_t* tk = curK; curK = othK; othK = tk;
    uint32_t* tv = curV; curV = othV; othV = tv;
  }
  if(curK != keysX)
    for(uint32_t i = t; i < T; i += THREADS)
    {
      stg(&keysX[base + i], ldg(&curK[base + i]));
      stg(&valsX[base + i], ldg(&curV[base + i]));
    }
  if(gatherDst != nullptr)
  {
    __syncthreads();
    for(uint32_t i = t; i < T; i += THREADS)
      gatherDst[base + i] = gatherSrc[ldg(&valsX[base + i])];
  }
}

// ---------------------------------------------------------------------------------------------
uint32_t sampleSortBuckets(uint32_t maxElems)
{
  uint64_t want = ((uint64_t)maxElems + kSsTarget - 1) / kSsTarget;
  uint32_t nb   = 64;
  while(nb < want && nb < 4096u)
    nb <<= 1;
  return nb;
}

bool sampleSortSupported(const SortLaunch& s)
{
  const uint32_t parts = s.slotCount ? s.partsSlotted : (uint32_t)(((uint64_t)s.maxElems + kSsPart - 1) / kSsPart);
  // more than 4096 buckets' worth of keys would overfill the LDS buckets (the classify kernel ranks 12 bucket bits)
  return s.beginBit == 0 && s.endBit == 32 && parts <= 131072u && s.ss.desc != nullptr
         && ((uint64_t)s.maxElems + kSsTarget - 1) / kSsTarget <= 4096u;
}

// src0 (slotted or flat) is permuted IN PLACE inside its 2048-key partitions; the sorted result lands in X.
void launchSampleSort(hipStream_t stream, const SortLaunch& s)
{
  if(s.maxElems == 0)
    return;
  const uint32_t parts = s.slotCount ? s.partsSlotted : (uint32_t)(((uint64_t)s.maxElems + kSsPart - 1) / kSsPart);
  if(parts == 0)
    return;
  const uint32_t NB = s.ss.maxBuckets;
  static const int kDbgStop = [] { const char* e = std::getenv("MGS_SS_STOP"); return e ? std::atoi(e) : 0; }();  // phase bisect (timing only)
  uint32_t*      k0 = const_cast<uint32_t*>(s.keys0);
  uint32_t*      v0 = const_cast<uint32_t*>(s.vals0);
  hipLaunchKernelGGL(k_ss_sample, dim3((parts + 255) / 256), dim3(256), 0, stream, s.keys0, s.slotCount, s.nPtr, parts, s.plan,
                     s.ss.samples, s.ss.bucketCount, NB);
  hipLaunchKernelGGL(k_ss_splitters, dim3(1), dim3(1024), 0, stream, s.plan, s.ss.samples, s.ss.splitters, s.nPtr, NB);
  if(NB <= 2048u)
    hipLaunchKernelGGL((k_ss_classify<2048>), dim3(parts), dim3(256), 0, stream, k0, v0, s.slotCount, s.nPtr, s.plan, s.ss.splitters,
                       s.ss.bucketCount, s.ss.desc, parts);
  else
    hipLaunchKernelGGL((k_ss_classify<4096>), dim3(parts), dim3(256), 0, stream, k0, v0, s.slotCount, s.nPtr, s.plan, s.ss.splitters,
                       s.ss.bucketCount, s.ss.desc, parts);
  hipLaunchKernelGGL(k_ss_finish, dim3(NB), dim3(kSsFinThreads), 0, stream, s.keys0, s.vals0, s.keysX, s.valsX, s.keysY, s.valsY,
                     s.nPtr, s.plan, s.ss.bucketCount, s.ss.desc, parts, s.gatherSrc, s.gatherDst, kDbgStop);
}

}  // namespace mgs
