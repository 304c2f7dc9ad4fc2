// sort_plan.h — device-resident plan of one radix sort and the host-side launch descriptor.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mgs {

struct SortPlan
{
  uint32_t ghist[4][256];  // digit totals of every pass, written by that pass's scan kernel
  uint32_t skip[4];        // pass is the identity permutation (single occupied digit) -> its scatter exits
  uint32_t reserved[4];    // [0]: the last pass left gatherDst filled (fused gather); [1..3]: sample sort (cursor, buckets, streamed buckets)
  uint32_t finalSel;       // result lives in X (0) or Y (1); written by the last pass
  uint32_t passesRun;
  uint32_t n;
  uint32_t pad[5];
  // Pass elision for depth keys (k_sort.hip): the producer marks which values of key >> 16 occur (as small ranges per
  // partition); when at most 256 do, pass 2 sorts on the RANK of key >> 16 among them — an order-preserving 8-bit
  // digit that covers the top 16 bits at once — and pass 3 is skipped.
  uint32_t remapOn;          // decided by the pass-1 scan kernel
  uint32_t remapCount;
  uint32_t remapBase;        // smallest occurring value: the kernels index a 4096-entry LDS table with (key >> 16) - remapBase
  uint32_t remapPad;
  uint32_t topBitmap[2048];  // presence of key >> 16
  uint16_t remapVals[256];   // the occurring values, ascending
};

// mark one value of key >> 16 as present.  Thousands of partitions mark the same handful of values: look first (a stale
// miss only costs a redundant atomic), so that the atomics on those few words do not serialise the grid.
__device__ __forceinline__ void sortMarkTop16(SortPlan* plan, uint32_t v)
{
  if(((__hip_atomic_load(&plan->topBitmap[v >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (v & 31u)) & 1u) == 0u)
    atomicOr(&plan->topBitmap[v >> 5], 1u << (v & 31u));
}

// Pass elision of the key sort: the producer's workgroup marks which values of key >> 16 it hands to the sort.  A partition is
// a compact cell of space, so its keys span one to three values: thread 0 marks the range; a partition that spans many (a cell
// around the camera) has every thread mark its own keys.  Split in two so that it needs no barrier of its own and the marking's
// memory latency overlaps the kernel's last stores: post the per-wave min / max before the kernel's final barrier, mark after it.
template <int WAVES>
__device__ __forceinline__ void sortTop16Post(uint32_t mn, uint32_t mx, uint32_t* s_red /* 2 * WAVES free words */)
{
#pragma unroll
  for(int o = 32; o > 0; o >>= 1)
  {
    mn = min(mn, (uint32_t)__shfl_xor(mn, o, 64));
    mx = max(mx, (uint32_t)__shfl_xor(mx, o, 64));
  }
  if((threadIdx.x & 63u) == 0u)
  {
    s_red[threadIdx.x >> 6]           = mn;
    s_red[WAVES + (threadIdx.x >> 6)] = mx;
  }
}
// returns true when the caller's threads must mark their own keys
template <int WAVES>
__device__ __forceinline__ bool sortTop16Mark(SortPlan* plan, uint32_t count, const uint32_t* s_red)
{
  if(plan == nullptr || count == 0u)
    return false;
  uint32_t lo = s_red[0], hi = s_red[WAVES];
#pragma unroll
  for(int i = 1; i < WAVES; ++i)
  {
    lo = min(lo, s_red[i]);
    hi = max(hi, s_red[WAVES + i]);
  }
  if(hi - lo > 24u)
    return true;
  if(threadIdx.x == 0)
    for(uint32_t v = lo; v <= hi; ++v)
      sortMarkTop16(plan, v);
  return false;
}

// scratch of the sample sort (k_ssort.hip); all device pointers, owned by the caller
struct SampleSortBuffers
{
  uint32_t*           samples     = nullptr;  // [16384]
  uint32_t*           splitters   = nullptr;  // [maxBuckets]
  unsigned long long* bucketCount = nullptr;  // [maxBuckets]: slices << 32 | keys, per bucket
  uint2*              desc        = nullptr;  // [maxBuckets][parts]: (source index, partition << 11 | count - 1)
  uint32_t            maxBuckets  = 0;        // sampleSortBuckets(maxElems)
};

struct SortLaunch
{
  const uint32_t* keys0;  // pass-0 source
  const uint32_t* vals0;
  uint32_t*       keysX;
  uint32_t*       valsX;
  uint32_t*       keysY;
  uint32_t*       valsY;
  const uint32_t* slotCount;     // non-null: pass 0 reads slotted partitions (stride 2048) with these counts, and the
                                 // producer has already written their pass-0 digit histograms into partHist
  uint32_t        partsSlotted;  // number of slotted partitions
  const uint32_t* nPtr;          // device-side element count (uniform partitions)
  SortPlan*       plan;          // must be zeroed before the launch (launchSortClearPlan / frame init)
  uint32_t*       partHist;      // [256][pStride], pStride >= max(partsSlotted, ceil(maxElems/2048))
  uint32_t        pStride;
  uint32_t        maxElems;      // host-side upper bound of the element count (sizes the grids)
  int             beginBit, endBit;
  uint2*          ranges    = nullptr;  // optional, single-pass sorts: ranges[digit] = [begin,end) in the sorted output
  const uint32_t* gatherSrc = nullptr;  // optional (multi-pass sorts): the LAST pass writes gatherDst[pos] = gatherSrc[value]
  uint32_t*       gatherDst = nullptr;  //   instead of the keys, and sets plan->reserved[0] when it ran (not skipped)
  SampleSortBuffers ss;                 // non-null desc: full 32-bit sorts take the sample sort (k_ssort.hip)
  bool            allowRemap = false;   // the producer marked plan->topBitmap (full 32-bit key sorts of a frame only)
};

void launchSortClearPlan(hipStream_t stream, SortPlan* plan);
void launchRadixSort(hipStream_t stream, const SortLaunch& s);
// sample sort: 4 launches, 32 B/key; src0 is permuted in place inside its 2048-key partitions, result in X
uint32_t sampleSortBuckets(uint32_t maxElems);
bool     sampleSortSupported(const SortLaunch& s);
void     launchSampleSort(hipStream_t stream, const SortLaunch& s);

}  // namespace mgs
