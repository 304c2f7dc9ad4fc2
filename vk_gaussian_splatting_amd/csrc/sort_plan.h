// sort_plan.h — device-resident plans of the radix sorts and the host-side launch descriptors.
//   k_sort.hip   generic reduce-then-scan LSD sort (any bit range; the record-path pair sort, the stand-alone sort API)
//   k_osort.hip  the frame's depth-key sort: single-kernel passes with an in-kernel two-level look-back; its pass 0 is virtual
//                (done by the project kernels' hand-over, slot_emit.h)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.h"

namespace mgs {

struct SortPlan
{
  uint32_t ghist[4][256];  // digit totals of every pass, written by that pass's scan kernel
  uint32_t skip[4];        // pass is the identity permutation (single occupied digit) -> its scatter exits
  uint32_t reserved[4];
  uint32_t finalSel;       // result lives in X (0) or Y (1); written by the last pass
  uint32_t passesRun;
  uint32_t n;
  uint32_t pad[5];
};

struct SortLaunch
{
  const uint32_t* keys0;  // pass-0 source
  const uint32_t* vals0;
  uint32_t*       keysX;
  uint32_t*       valsX;
  uint32_t*       keysY;
  uint32_t*       valsY;
  const uint32_t* nPtr;      // device-side element count
  SortPlan*       plan;      // must be zeroed before the launch (launchSortClearPlan / frame init)
  uint32_t*       partHist;  // [256][pStride], pStride >= ceil(maxElems/2048)
  uint32_t        pStride;
  uint32_t        maxElems;  // host-side upper bound of the element count (sizes the grids)
  int             beginBit, endBit;
  uint2*          ranges = nullptr;  // optional, single-pass sorts: ranges[digit] = [begin,end) in the sorted output
};

void launchSortClearPlan(hipStream_t stream, SortPlan* plan);
void launchRadixSort(hipStream_t stream, const SortLaunch& s);

// ---- the frame's key sort (k_osort.hip) -------------------------------------------------------------------------------
#ifndef MGS_OS_PART
#define MGS_OS_PART 4096
#endif
constexpr uint32_t kOsPart    = MGS_OS_PART;  // pairs per partition (256 threads x 16)
constexpr uint32_t kOsSlot    = 2048;  // pairs a project workgroup's slot can hold (== its partition of splats): 11-bit starts
constexpr uint32_t kSlotHistWords = 384;  // what a slot leaves per partition (slot_emit.h): counts of key bits 0-7 and 8-15,
                                          // starts of the digit-0 groups — 16-bit values, two per word
constexpr uint32_t kOsChunk   = 32;    // slots per chunk of the virtual pass 0 (k_os_prepare: one reduce workgroup per chunk)
constexpr uint32_t kOsGroup   = 32;    // partitions per look-back group
constexpr uint64_t kOsMaxPairs = 1ull << 30;  // exclusive: look-back words = 2 flag bits + a 30-bit prefix (callers guard)
constexpr uint32_t kRemapSpan = 4096;  // pass 2 of a depth-key sort indexes a 4096-entry LDS table with (key >> 16) - remapBase

struct OsPlan
{
  uint32_t total[4][256];  // digit totals of every pass, complete before pass 0 starts (pass 2: per rank when remapOn)
  uint32_t reserved4[4];   // (round 3 tried a start-order ticket per pass here: measured and dropped, k_osort.hip header)
  // Pass elision for depth keys: when at most 256 values of key >> 16 occur (within a span < 4096), pass 2 sorts on the RANK
  // of key >> 16 among them — an order-preserving 8-bit digit that covers the top 16 bits at once — and pass 3 does not run.
  uint32_t remapOn, remapCount, remapBase;
  uint32_t n;               // element count (copied from the device-side counter by k_os_prepare)
  uint32_t remapPadRank;    // pass 2 when remapOn: the rank of keys outside the table (padding lanes): the largest
  uint32_t arrivedUnused;   // (rounds 3-4: arrival counter of k_os_prepare's reduce workgroups; the fold moved to k_os_pass<3>)
  uint32_t top16MinInv;     // 0x10000 - (smallest occurring value of key >> 16); 0 = none   } both kept as maxima: the plan
  uint32_t top16MaxP1;      // largest occurring value + 1; 0 = none                          } starts zeroed
  uint32_t pad[4];
  uint16_t remapVals[256];  // the occurring values, ascending
};

// host descriptor of the frame key sort / the stand-alone full-width sort
struct OsLaunch
{
  // input, exactly one of: the project kernels' slots of (key, id) pairs with their per-partition histograms and records ...
  const uint2*    pairs0     = nullptr;  // slot p = pairs [kOsSlot p, kOsSlot p + slotCount[p]), grouped by key bits 0-7 (slot_emit.h);
                                         // may alias pairB (the first pass reads it, the second one is the first to write B)
  uint32_t        prjParts   = 0;        // project partitions == slots (2048 splats each)
  const uint32_t* slotCount  = nullptr;  // [prjParts] pairs in every slot
  uint32_t*       chunkSum   = nullptr;  // [osSortChunks(prjParts)][256] scratch (k_os_prepare): pairs per digit-0 value in every chunk of 32 slots
  uint32_t*       runTab     = nullptr;  // [256][32 osSortChunks(prjParts)] scratch (k_os_prepare): per (digit-0 value, slot) the pairs of
                                         // that value in the chunk's earlier slots | the group's start inside its slot << 16
  uint32_t        rideShift  = 0;        // frame only: the ids carry the bin rectangles' codes above bit rideShift (rideEncode) ...
  uint32_t        rideSplit  = 0;        // ... (1: split between the key's low byte and the id's spare bits, FrameConst::rideSplit) ...
  uint32_t        rideInfo   = 0;        // ... shapes | code bits << 8, handed to the binning stage in planOut->reserved[0] ...
  uint16_t*       outCode16  = nullptr;  // ... and the final pass writes clean ids and, here, the codes in sorted order
  uint32_t*       prjOrderOut = nullptr; // [prjParts] the next frame's dispatch order of the project kernel: fullest slot first (k_os_prepare)
  uint32_t*       nOut       = nullptr;  // the frame's count of sorted pairs (== *nPtr afterwards), written by k_os_prepare
  const uint32_t* slotHist   = nullptr;  // [partition][kSlotHistWords] (slot_emit.h)
  const uint32_t* top16Rec   = nullptr;  // [partition][4 waves][32]: counts of key >> 16 per producer wave (slot_emit.h)
  uint32_t*       top16Count = nullptr;  // [65536] occurrences of key >> 16 (filled, consumed and cleared by k_os_prepare;
                                         // partitions that span > 24 values add their keys themselves)
  // ... or a uniform array of keys and values
  const uint32_t* keys0 = nullptr;
  const uint32_t* vals0 = nullptr;
  const uint32_t* nPtr     = nullptr;  // device-side element count
  uint32_t        maxElems = 0;        // host-side upper bound (sizes the grids)
  uint2*          pairA    = nullptr;  // ping-pong, maxElems pairs each
  uint2*          pairB    = nullptr;
  uint32_t*       outVals  = nullptr;  // the sorted values
  uint32_t*       outKeys  = nullptr;  // the sorted keys (null: not needed)
  OsPlan*         plan     = nullptr;  // zeroed before the launch (frame init / launchOsSortClearPlan)
  SortPlan*       planOut  = nullptr;  // n / finalSel = 0 / passesRun for the consumers of the sorted ids
  uint32_t*       status   = nullptr;  // 3 x osSortStatusWords(osSortMaxParts(...)) words, zero on first use
  FrameCounters*  ctr      = nullptr;  // errorFlags |= kErrSpinTimeout if a look-back wait ever gives up
  bool            allowRemap = true;
  uint32_t        partMin    = 0;      // smallest partition size the passes may choose on the device (k_osort.hip: osPartOf); 0 = MGS_OS_PART_MIN or, by default, kOsPart = fixed
  uint32_t        resSlots   = 0;      // workgroups of a pass the device holds at once; 0 = 4 per CU of the current device (126 VGPRs, 36-40 KB of LDS)
};

// the per-frame sort state of one context, contiguous so that the frame's first kernel zeroes it in one sweep
struct FramePlans
{
  SortPlan keys;   // what the consumers of the sorted ids read (n, finalSel, passesRun)
  SortPlan pairs;  // the record path's pair sort / the direct binning's per-bin totals
  OsPlan   os;     // the key sort's own plan
};

static_assert(offsetof(FramePlans, pairs) == sizeof(SortPlan) && offsetof(FramePlans, os) == 2 * sizeof(SortPlan),
              "frameStatSlot: the keys plan sits right before the pairs plan, the key sort's plan right behind it");
// The compositors' frame statistics (staged records, scanned list entries; mgs_frame_stats): every region adds its two counts
// with fire-and-forget atomics.  Rounds 1-4 kept 8 + 8 words for them in the frame counters — ONE 128-byte line; at 4K that is
// 33 K atomics on one line, and the line's atomic unit (~90 per microsecond) was what the compositor's tail waited for (round 5
// ablation: composite 361 -> 331 us at 4K, 118 -> 114 at 1080p without them).  Now 32 slots, one per 128-byte line of the KEYS
// plan's histogram rows, which a frame's key sort does not use and the frame's upload zeroes: word 0 staged, word 1 scanned.
constexpr uint32_t kFrameStatSlots = 32;
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t* frameStatSlot(const SortPlan* planPairs /* &FramePlans::pairs */, uint32_t slot)
{
  return const_cast<uint32_t*>(&(planPairs - 1)->ghist[0][0]) + 32u * (slot & (kFrameStatSlots - 1u));
}
// the same slots from the project kernels, which hold &FramePlans::os: word 2 = survivors of the dist-stage cull (one add per
// partition: 22.8 K of them at configs[4] were 23 us of the frame counters' line)
__device__ __forceinline__ uint32_t* frameStatSlotFromOs(const OsPlan* osPlan /* &FramePlans::os */, uint32_t slot)
{
  return frameStatSlot(reinterpret_cast<const SortPlan*>(osPlan) - 1, slot);
}
#endif

uint32_t osSortMaxParts(uint32_t maxElems);
inline uint32_t osSortChunks(uint32_t prjParts) { return (prjParts + kOsChunk - 1u) / kOsChunk; }
size_t   osSortStatusWords(uint32_t maxParts);
void     launchOsSortClearPlan(hipStream_t stream, OsPlan* plan);
void     launchOsSort(hipStream_t stream, const OsLaunch& L);

// ---- what the project kernels hand to the key sort (k_project.hip, k_gut.hip; device side in slot_emit.h) -----------------

// The producer counts which values of key >> 16 it hands to the sort.  A partition is a compact cell of space, so its keys
// span one to three values: every wave leaves a 32-word record (counts of the values lo .. lo + 24, header lo | span << 16
// in word 31) and k_os_prepare folds the records; a partition that spans more than 24 values (a cell around the camera)
// adds its keys to the count table one by one.  Split in two so that it needs no barrier of its own: post the per-wave
// min / max before the kernel's last barrier, count after it.
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
template <int WAVES>
__device__ __forceinline__ void sortTop16Range(uint32_t count, const uint32_t* s_red, uint32_t& lo, uint32_t& hi)
{
  lo = s_red[0];
  hi = s_red[WAVES];
#pragma unroll
  for(int i = 1; i < WAVES; ++i)
  {
    lo = min(lo, s_red[i]);
    hi = max(hi, s_red[WAVES + i]);
  }
  if(count == 0u)
  {
    lo = 1u;
    hi = 0u;
  }
}

}  // namespace mgs
