// k_osort.hip — the frame's depth-key sort: stable LSD radix sort of (u32 key, u32 id) pairs in single-kernel passes.
//
// Behavioural spec: vrdxCmdSortKeyValueIndirect (3rdparty/vrdx/src/vk_radix_sort.cc:262-416: stable, ascending, 8-bit
// digits, element count read on the device).  vrdx runs upsweep / spine / downsweep per pass (12 dispatches); the generic
// sort of this build (k_sort.hip) runs histogram / scan / scatter per pass.  This file is what the in-frame key sort uses
// instead: every pass is ONE kernel — each partition publishes its digit counts, resolves its digit prefixes from the
// partitions before it while it ranks its keys, and scatters ("onesweep" structure; the look-back is re-designed for
// MI355X, below) — and what the passes need up front comes for free from the producer:
//   * PASS 0 IS VIRTUAL (round 4).  k_project writes its slot grouped by the key's low byte (a stable multi-split of <= 2048
//     pairs in LDS, slot_emit.h) together with the groups' counts and starts; where a pair stands after a stable pass on
//     bits 0-7 is then a function of those counts alone, so that pass is never run: k_os_prepare turns the counts into two
//     small tables, and the sort's first kernel (bits 8-15) gathers its dense partitions straight from the slots in digit-0
//     order (k_os_pass<3>).  One pass of ranking, look-back, re-order, scatter and 16 B/key of traffic less than round 3;
//   * k_project also leaves the slot's histogram of bits 8-15 and, per wave, a record of how often each value of key >> 16
//     occurs (a partition is a compact cell of space: 1-3 values);
//   * k_os_prepare reduces the histograms to totals, folds the records into a 64 K-entry count table (one atomic per
//     occurring value per 32 slots) and turns the table into the totals of the upper passes: when at
//     most 256 values of key >> 16 occur (depth keys span one or two binades) pass 2 sorts on the RANK of key >> 16 among
//     them — an order-preserving 8-bit digit that covers the top 16 bits — and pass 3 does not run; otherwise plain digits.
//   => launches per frame sort: prepare + 2 passes (+ 1 that exits at once); round 3: prepare + 3 (+ 1); reduce-then-scan: 11.
//
// Look-back, re-designed: the classic chain resolves partition p from p-1, one dependent cross-CU load per hop; with all
// ~1 000 partitions of a frame's sort resident at once that serialises (measured 1.27 us per hop, tools/micro/
// lookback_rate.hip).  Here the prefix is resolved in TWO LEVELS of fan-in 32, every level one batch of independent loads:
//   1. member m of a group sums the published counts of the m members before it (<= 31 loads in flight per thread);
//   2. the group's last member publishes the group total, resolves the group's base with a windowed look-back over the
//      GROUP totals (16 per batch, stopping at the first inclusive prefix) and publishes the inclusive group prefix;
//   3. every member reads ONE word: the inclusive prefix of the previous group.
// Three round trips whatever the partition count, overlapped with the ranking.  Status words carry flag and value in one
// 32-bit word written by one relaxed agent-scope store (sc1: the data is the flag, no fence to order), polled with relaxed
// agent-scope loads.  The level-1 loads are issued right after a partition has published its own counts and consumed behind
// its ranking, the group level behind its LDS re-order: the round trips overlap the pass's own work.
// Partition = blockIdx.x.  A workgroup waits only for lower-numbered ones, and the dispatcher starts the workgroups of a 1-D
// grid in index order on every XCD, so whatever is waited for is running or done.  A ticket per workgroup (partitions in
// observed start order) would not have to lean on that; it was measured and dropped: one word takes ~88 returning atomics
// per microsecond (MI355X_MICROARCH.md "dequeue"), so the ~1 000 workgroups of a pass that start together wait up to 10 us
// for their number.  Every spin is bounded: if the order were ever violated, the wait gives up and raises kErrSpinTimeout
// (the frame is wrong, the GPU does not hang).
//
// Pairs travel interleaved (uint2): one 8-byte access per element, digit runs of 16 elements are 128 contiguous bytes; the
// last pass writes the ids alone (the keys are dead; the sort-only hook asks for them explicitly).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_common.h"
#include "sort_plan.h"

namespace mgs {

namespace {
constexpr int      kThreads = 256;
constexpr int      kKpt     = kOsPart / kThreads;  // 16
constexpr int      kWaves   = kThreads / 64;
constexpr uint32_t kAgg = 1u << 30, kInc = 2u << 30, kValMask = (1u << 30) - 1u;
constexpr int      kGroupWindow = 16;
constexpr uint32_t kSpinMax     = 1u << 21;  // polls before a wait gives up (seconds; a healthy wait is a few polls)
static_assert(kOsGroup == 32, "the member mask is one 32-bit word");

__device__ __forceinline__ uint32_t ldAgent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void     stAgent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive scan of one value per digit over the 256 threads of the block
__device__ __forceinline__ uint32_t scan256(uint32_t v, uint32_t* s_tmp /*4*/)
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
  return base + inc - v;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// (a) uniform input only (the stand-alone sort API): digit totals of all four passes in one read of the keys
__global__ __launch_bounds__(256) void k_os_hist(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ nPtr, OsPlan* __restrict__ plan)
{
  __shared__ uint32_t s_h[4][256];
  const int      t = threadIdx.x;
  const uint32_t n = *nPtr;
  for(int q = 0; q < 4; ++q)
    s_h[q][t] = 0u;
  __syncthreads();
  for(uint64_t i0 = (uint64_t)blockIdx.x * 2048u; i0 < n; i0 += (uint64_t)gridDim.x * 2048u)
  {
    uint32_t kk[8];
#pragma unroll
    for(int u = 0; u < 8; ++u)
    {
      const uint64_t i = i0 + (uint64_t)u * 256u + (uint64_t)t;
      kk[u]            = keys[i < n ? i : (uint64_t)n - 1u];  // clamped, not predicated
    }
#pragma unroll
    for(int u = 0; u < 8; ++u)
      if(i0 + (uint64_t)u * 256u + (uint64_t)t < n)
      {
#pragma unroll
        for(int q = 0; q < 4; ++q)
          atomicAdd(&s_h[q][(kk[u] >> (8 * q)) & 255u], 1u);
      }
  }
  __syncthreads();
#pragma unroll
  for(int q = 0; q < 4; ++q)
    if(s_h[q][t])
      atomicAdd(&plan->total[q][t], s_h[q][t]);
}

#ifdef MGS_OS_TRACE  // debug build (tools/os_trace.py): per-workgroup wall-clock stamps (100 MHz) of the phases of every pass
#define MGS_OS_STAMP(i) if(threadIdx.x == 0) trc[i] = wall_clock64();
#define MGS_OS_GSTAMP(i) if(threadIdx.x == 0 && gtr[i] == 0) gtr[i] = wall_clock64();  // the virtual pass 0's table (first time only)
__device__ uint64_t* g_osPrepTrace = nullptr;  // [reduce workgroup][8]
#else
#define MGS_OS_STAMP(i)
#define MGS_OS_GSTAMP(i)
#endif
// ---------------------------------------------------------------------------------------------------------------------
// (b) prepare: the digit totals of all passes and the tables of the virtual pass 0, from what the producer left.
// Grid: reduce workgroups of 1024 threads, one per CHUNK of 32 slots.  A reduce workgroup
//   * turns its slots' digit-0 group counts (slot_emit.h) into the chunk's part of the virtual pass 0: per digit-0 value d the
//     chunk's total (chunkSum[chunk][d]) and, per slot, how many pairs of value d the chunk's earlier slots hold together with
//     where the slot's group of d starts inside the slot (runTab[d][slot], 16 + 16 bits) — and adds the totals to
//     plan->total[0];  the pair that a stable pass on bits 0-7 would put at position x of its output is then found from
//     D[d] (scan of the totals), the prefix of chunkSum[.][d] and one runTab row segment, all of which the sort's first
//     kernel reads for just the part of the order it owns (k_os_pass<3>).  Nothing here waits for another workgroup;
//   * sums its slots' histograms of key bits 8-15 into plan->total[1] (<= 256 atomics, one per non-empty bin);
//   * folds its slots' key >> 16 records (slot_emit.h: one 32-word record per producer wave: counts of the values lo..lo+24)
//     in an LDS table and adds each occurring value ONCE to the 64 K-entry count table.  The producers do not touch that
//     table themselves: thousands of partitions hold the same handful of values, and that many atomics on a few addresses
//     serialise (measured: 0.13 -> 0.55 ms for k_project).  Partitions that span more than 24 values (a cell around the
//     camera) are the exception: their keys are spread over many addresses and were added one by one.
// Turning the count table into what the upper passes sort on (at most 256 values within a span < 4096 -> pass 2 sorts on their
// rank; otherwise plain digits for passes 2 and 3) is NOT done here any more: the table is complete when this kernel ends and
// its fold is needed by the second sort kernel only, so workgroups beyond the partitions of the FIRST sort kernel do it (foldTop16
// below; rounds 3-4a had the reduce workgroup that arrived last do it here, 4 us at the end of this kernel's critical path).
// One more workgroup (the last of the grid) sums the slots' counts — the frame's number of sorted pairs — and leaves the project
// kernel's dispatch order for the NEXT frame, fullest slot first (round 5).
__global__ __launch_bounds__(1024) void k_os_prepare(const uint32_t* __restrict__ slotHist, const uint32_t* __restrict__ top16Rec, uint32_t prjParts,
                                                     uint32_t* __restrict__ top16Count, OsPlan* __restrict__ plan, const uint32_t* __restrict__ nPtr,
                                                     int allowRemap, uint32_t reduceWgs, const uint32_t* __restrict__ slotCount,
                                                     uint32_t* __restrict__ chunkSum, uint32_t* __restrict__ runTab,
                                                     uint32_t* __restrict__ nOut, uint32_t* __restrict__ orderOut)
{
  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
#ifdef MGS_OS_TRACE
  __shared__ uint64_t trc[8];
  if(t < 8) trc[t] = 0;
  MGS_OS_STAMP(0)
#endif
  if(slotHist == nullptr)
  {  // uniform input (stand-alone sort): k_os_hist has the totals
    if(blockIdx.x == 0 && t == 0)
      plan->n = *nPtr;
    return;
  }
  __shared__ uint32_t s_tab[2048];
  if(blockIdx.x == reduceWgs)
  {
    __shared__ uint32_t s_scan[16];
    uint32_t sum = 0;
    for(uint32_t q = t; q < prjParts; q += 1024u)
      sum += slotCount[q];
    const uint32_t inc = waveInclusiveScan(sum);
    if(lane == 63)
      s_scan[w] = inc;
    __syncthreads();
    if(t == 0)
    {
      uint32_t total = 0;
      for(int q = 0; q < 16; ++q)
        total += s_scan[q];
      plan->n = total;
      *nOut   = total;
    }
    if(orderOut != nullptr)
    {  // The NEXT frame's dispatch order of the project kernel's partitions: fullest slot first (round 5).  That kernel runs
       // 1.85 residency waves of workgroups that take 19 .. 80 us each in storage order, and a third of its span was a draining
       // tail (profiles/r4_z_prj_trace.log); a partition's cost follows its survivor count, which hardly changes from one frame of
       // a sequence to the next.  Scheduling only: slots are per partition, the frame does not depend on the order.  A counting
       // sort on count / 64 (33 classes), the order inside a class is whatever the atomics give.
      __shared__ uint32_t s_cls[40];
      if(t < 40)
        s_cls[t] = 0u;
      __syncthreads();
      for(uint32_t q = t; q < prjParts; q += 1024u)
        atomicAdd(&s_cls[32u - min(slotCount[q] >> 6, 32u)], 1u);
      __syncthreads();
      if(t == 0)
      {
        uint32_t run = 0;
        for(int c = 0; c < 33; ++c)
        {
          const uint32_t v = s_cls[c];
          s_cls[c]         = run;
          run += v;
        }
      }
      __syncthreads();
      for(uint32_t q = t; q < prjParts; q += 1024u)
        orderOut[atomicAdd(&s_cls[32u - min(slotCount[q] >> 6, 32u)], 1u)] = q;
    }
    return;
  }
  if(blockIdx.x >= reduceWgs)
    return;
  __shared__ uint32_t s_part[512];
  __shared__ uint32_t s_lo, s_hi;
  const uint32_t slot0 = blockIdx.x * 32u;
  // the records' loads go out with the histograms' (one round trip instead of three): thread t owns words j0 .. j0 + 3 of
  // record t / 8, and reads that record's header itself
  const uint32_t recR = (uint32_t)t >> 3, recJ0 = ((uint32_t)t & 7u) * 4u, recSlot = slot0 + recR / 4u;
  uint32_t       recHdr = 0xFFFFFFFFu;
  uint4          recC   = make_uint4(0u, 0u, 0u, 0u);
  if(recSlot < prjParts)
  {
    const uint32_t* rp = &top16Rec[((size_t)recSlot * 4u + (recR & 3u)) * 32u];
    recHdr             = rp[31];
    recC               = *reinterpret_cast<const uint4*>(rp + recJ0);
  }
  {  // the slots' rows: thread = (packed column c: digits 2 c and 2 c + 1, group j of four consecutive slots)
    __shared__ uint32_t s_grp[8][256];
    const uint32_t c = (uint32_t)t & 127u, j = (uint32_t)t >> 7;
    uint32_t       w0[4], w1[4], w2[4];
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
      const uint32_t  sl  = slot0 + 4u * j + (uint32_t)i;
      const uint32_t* row = slotHist + (size_t)min(sl, prjParts - 1u) * kSlotHistWords;
      const bool      ok  = sl < prjParts;
      w0[i]               = row[c];
      w1[i]               = row[128u + c];
      w2[i]               = row[256u + c];
      if(!ok)
        w0[i] = w1[i] = w2[i] = 0u;
    }
    s_tab[t]         = 0u;
    s_tab[t + 1024u] = 0u;
    if(t < 512)
      s_part[t] = 0u;
    if(t == 0)
    {
      s_lo = 0xFFFFu;
      s_hi = 0u;
    }
    uint32_t pLo[4], pHi[4], sLo = 0, sHi = 0, lo1 = 0, hi1 = 0;
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
      pLo[i] = sLo;
      pHi[i] = sHi;
      sLo += w0[i] & 0xFFFFu;
      sHi += w0[i] >> 16;
      lo1 += w1[i] & 0xFFFFu;
      hi1 += w1[i] >> 16;
    }
    s_grp[j][2u * c]      = sLo;
    s_grp[j][2u * c + 1u] = sHi;
    __syncthreads();
    if(lo1)
      atomicAdd(&s_part[256u + 2u * c], lo1);
    if(hi1)
      atomicAdd(&s_part[256u + 2u * c + 1u], hi1);
    uint32_t bLo = 0, bHi = 0;
#pragma unroll
    for(uint32_t q = 0; q < 7u; ++q)
      if(q < j)
      {
        bLo += s_grp[q][2u * c];
        bHi += s_grp[q][2u * c + 1u];
      }
    if(j == 7u)
    {  // the chunk's totals of digits 2 c and 2 c + 1
      const uint32_t tLo = bLo + sLo, tHi = bHi + sHi;
      *reinterpret_cast<uint2*>(&chunkSum[(size_t)blockIdx.x * 256u + 2u * c]) = make_uint2(tLo, tHi);
      if(tLo)
        atomicAdd(&plan->total[0][2u * c], tLo);
      if(tHi)
        atomicAdd(&plan->total[0][2u * c + 1u], tHi);
    }
    // (pairs of the value in the chunk's earlier slots: <= 31 x 2048, 16 bits) | (start of the value's group in its slot: <= 2048) << 16
    const size_t spad = (size_t)reduceWgs * kOsChunk, col = (size_t)slot0 + 4u * j;
    *reinterpret_cast<uint4*>(&runTab[(size_t)(2u * c) * spad + col]) =
        make_uint4((bLo + pLo[0]) | (w2[0] << 16), (bLo + pLo[1]) | (w2[1] << 16), (bLo + pLo[2]) | (w2[2] << 16), (bLo + pLo[3]) | (w2[3] << 16));
    *reinterpret_cast<uint4*>(&runTab[(size_t)(2u * c + 1u) * spad + col]) =
        make_uint4((bHi + pHi[0]) | (w2[0] & 0xFFFF0000u), (bHi + pHi[1]) | (w2[1] & 0xFFFF0000u), (bHi + pHi[2]) | (w2[2] & 0xFFFF0000u),
                   (bHi + pHi[3]) | (w2[3] & 0xFFFF0000u));
    __syncthreads();
    if(t >= 256 && t < 512 && s_part[t])
      atomicAdd(&plan->total[1][t & 255u], s_part[t]);
  }
  MGS_OS_STAMP(1)
  // key >> 16 records of the 32 slots x 4 producer waves: header word 31 = lo | span << 16 (0xFFFFFFFF: nothing to fold)
  if((t & 7) == 0 && recHdr != 0xFFFFFFFFu)
  {
    atomicMin(&s_lo, recHdr & 0xFFFFu);
    atomicMax(&s_hi, (recHdr & 0xFFFFu) + (recHdr >> 16));
  }
  __syncthreads();
  if(recHdr != 0xFFFFFFFFu && recJ0 <= (recHdr >> 16))
  {
    const uint32_t cv[4] = {recC.x, recC.y, recC.z, recC.w};
#pragma unroll
    for(int j = 0; j < 4; ++j)
      if(recJ0 + j <= (recHdr >> 16) && cv[j])
      {
        const uint32_t v = (recHdr & 0xFFFFu) + recJ0 + j, idx = v - s_lo;
        if(idx < 2048u)
          atomicAdd(&s_tab[idx], cv[j]);
        else
          atomicAdd(&top16Count[v], cv[j]);
      }
  }
  __syncthreads();
  for(uint32_t b = t; b < 2048u; b += 1024u)
    if(s_tab[b])
      atomicAdd(&top16Count[s_lo + b], s_tab[b]);
  if(t == 0 && s_hi >= s_lo)  // (s_lo = 0xFFFF, s_hi = 0 is the empty state: 0xFFFF itself is a legal value of key >> 16)
  {  // occurring range over all workgroups (the plan is zeroed per sort: both as maxima)
    atomicMax(&plan->top16MinInv, 0x10000u - s_lo);
    atomicMax(&plan->top16MaxP1, s_hi + 1u);
  }
  // (Rounds 3 and the first half of round 4 went on here: the workgroup that arrived last — drain, arrival counter, barrier —
  //  folded the count table into the rank table, 4 us at the end of this kernel's critical path.  The fold needs the table
  //  complete and is needed by the SECOND sort kernel only: it now runs in workgroups beyond the partitions of the first one,
  //  beside that kernel's own work — foldTop16 below.)
#ifdef MGS_OS_TRACE
  MGS_OS_STAMP(2)
  if(t == 0 && g_osPrepTrace)
    for(int i = 0; i < 8; ++i) g_osPrepTrace[(size_t)blockIdx.x * 8 + i] = trc[i];
#endif
  (void)allowRemap;  // (the rank digit's switch travels with the first sort kernel now: OsPassArgs::allowRemap)
}

// The count table of key >> 16 -> what the upper passes sort on.  At most 256 occurring values within a span < 4096: the second
// kernel sorts on their RANK (plan->remapVals, totals per rank) and is final; otherwise plain digits for passes 2 and 3 (totals of
// bits 16-23 and 24-31).  Clears what it read: the table is clean for the next sort of this context.  One workgroup; the table is
// complete (the kernel that filled it has ended).
// kOsFoldWgs workgroups take part: the narrow case (the rule) is workgroup 0's alone; a wide range — up to 64 K values — is split
// between all of them (one 256-thread workgroup walking it alone held the first sort kernel up for 50 us).
constexpr uint32_t kOsFoldWgs = 8;
template <int THREADS>
__device__ __forceinline__ void foldTop16(OsPlan* __restrict__ plan, uint32_t* __restrict__ top16Count, int allowRemap, uint32_t wg,
                                          uint32_t* s_part /*512*/, uint32_t* s_sum /*THREADS / 64*/)
{
  constexpr uint32_t kPer = kRemapSpan / THREADS;  // consecutive values per thread (ordered compaction)
  const int          t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t     minInv = plan->top16MinInv, maxP1 = plan->top16MaxP1;
  if(minInv == 0u || maxP1 == 0u)
  {  // no real keys at all
    if(t == 0 && wg == 0u)
      plan->remapOn = plan->remapCount = plan->remapBase = 0u;
    return;
  }
  const uint32_t vlo = 0x10000u - minInv, span = maxP1 - vlo;  // values vlo .. vlo + span - 1
  for(int i = t; i < 512; i += THREADS)
    s_part[i] = 0u;
  if(span <= kRemapSpan - 1u)
  {
    if(wg != 0u)
      return;
    uint32_t c[kPer], nz = 0;
#pragma unroll
    for(uint32_t i = 0; i < kPer; ++i)
    {
      const uint32_t k = (uint32_t)t * kPer + i;
      c[i]             = k < span ? top16Count[vlo + k] : 0u;
      nz += c[i] ? 1u : 0u;
    }
#pragma unroll
    for(uint32_t i = 0; i < kPer; ++i)
      if(c[i])
        top16Count[vlo + (uint32_t)t * kPer + i] = 0u;  // consumed
    const uint32_t inc = waveInclusiveScan(nz);
    if(lane == 63)
      s_sum[w] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for(int q = 0; q < THREADS / 64; ++q)
    {
      if(q < w) base += s_sum[q];
      total += s_sum[q];
    }
    const bool on = allowRemap != 0 && total >= 1u && total <= 256u;
    uint32_t   run = base + inc - nz;
#pragma unroll
    for(uint32_t i = 0; i < kPer; ++i)
      if(c[i])
      {
        const uint32_t v = vlo + (uint32_t)t * kPer + i;
        if(on)
        {
          plan->remapVals[run] = (uint16_t)v;
          plan->total[2][run]  = c[i];  // pass 2 sorts on the rank
          ++run;
        }
        else
        {
          atomicAdd(&s_part[v & 255u], c[i]);
          atomicAdd(&s_part[256u + (v >> 8)], c[i]);
        }
      }
    __syncthreads();
    if(!on)
      for(int i = t; i < 256; i += THREADS)
      {
        plan->total[2][i] = s_part[i];
        plan->total[3][i] = s_part[256 + i];
      }
    if(t == 0)
    {
      plan->remapOn      = on ? 1u : 0u;
      plan->remapCount   = on ? total : 0u;
      plan->remapBase    = on ? vlo : 0u;
      plan->remapPadRank = on ? total - 1u : 0u;  // keys outside the table (padding) take the largest rank
    }
    return;
  }
  // wide range (camera inside the cloud): plain digits of bits 16-23 and 24-31; this workgroup's share of the range, eight loads
  // in flight per thread, the totals added to the plan's (zero when the frame starts)
  __syncthreads();
  const uint32_t share = (span + kOsFoldWgs - 1u) / kOsFoldWgs, k0 = wg * share, k1 = min(span, k0 + share);
  for(uint32_t kb = k0; kb < k1; kb += 8u * THREADS)
  {
    uint32_t c[8];
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
      const uint32_t k = kb + (uint32_t)i * THREADS + (uint32_t)t;
      c[i]             = k < k1 ? top16Count[vlo + k] : 0u;
    }
#pragma unroll
    for(int i = 0; i < 8; ++i)
      if(c[i])
      {
        const uint32_t v = vlo + kb + (uint32_t)i * THREADS + (uint32_t)t;
        top16Count[v]    = 0u;
        atomicAdd(&s_part[v & 255u], c[i]);
        atomicAdd(&s_part[256u + (v >> 8)], c[i]);
      }
  }
  __syncthreads();
  for(int i = t; i < 256; i += THREADS)
  {
    if(s_part[i])
      atomicAdd(&plan->total[2][i], s_part[i]);
    if(s_part[256 + i])
      atomicAdd(&plan->total[3][i], s_part[256 + i]);
  }
  if(t == 0 && wg == 0u)
    plan->remapOn = plan->remapCount = plan->remapBase = 0u;
}

// ---------------------------------------------------------------------------------------------------------------------
// (c) one pass.  IN: 0 dense pairs (the output of the pass before); 2 split key / value arrays (pass 0 of the stand-alone sort);
// 3 the project kernels' slots, read in the order a stable pass on key bits 0-7 WOULD have left them in ("virtual pass 0": the
// first kernel of a frame's sort sorts on bits 8-15).  Dense partition p = positions [4096 p, 4096 p + count) of that order.
// Position x holds a pair of digit-0 value d = the one with D[d] <= x < D[d + 1] (D = exclusive scan of plan->total[0]); within
// d the slots follow each other, each with its group of d (slot_emit.h), so with c = the chunk of 32 slots and s the slot that
// x - D[d] falls into: x = D[d] + (sum of chunkSum[c'][d], c' < c) + runTab[d][s].low16 + i, and the pair is entry
// runTab[d][s].high16 + i of slot s.  The workgroup expands exactly the runs that overlap its partition into a 4096-entry
// source table in LDS (one prefix scan over the chunks per digit value it touches — usually one or two —, then one coalesced
// read of the runTab segment) and gathers: loads of <= 8-byte runs of ~6 pairs instead of round 3's contiguous reads, but
// everything else of a pass 0 — ranking, look-back, re-order, scatter, 67 MB of traffic — is not done at all.
struct OsPassArgs
{
#ifdef MGS_OS_TRACE
  uint64_t* trace;  // [partition][8]
#endif
  const uint2*    srcPairs;
  const uint32_t* chunkSum;  // IN 3: [chunks][256]
  const uint32_t* runTab;    // IN 3: [256][32 chunks]
  uint32_t        chunks;
  uint32_t        srcLimit;  // IN 3: the last valid pair index (a corrupted table must not turn into a wild read)
  uint32_t*       top16Count;  // IN 3: the grid's LAST workgroup folds the count table of key >> 16 for the kernel behind this one
  int             allowRemap;
  // the bin rectangles' codes ride above the ids (kernels_common.h: rideEncode); the final pass of a frame separates them:
  // clean ids for everybody, the codes in sorted order for the binning stage
  uint32_t        rideShift;   // bits of the id proper; 0 = nothing rides
  uint32_t        rideSplit;   // 1: the code's low 8 bits lie in the key's low byte, the rest above the id (slot_emit.h)
  uint32_t        rideInfo;    // what planOut->reserved[0] tells k_dbin_count: shapes | code bits << 8
  uint16_t*       dstCode16;
  const uint32_t* srcKeys;
  const uint32_t* srcVals;
  uint2*          dstPairs;
  uint32_t*       dstKeys;  // final pass: may be null (the frame does not need the keys again)
  uint32_t*       dstVals;
  OsPlan*         plan;
  SortPlan*       planOut;  // what the consumers of the sorted ids read: n, finalSel (always 0 here), passesRun
  uint32_t*       status;   // [maxParts][256] this pass: one 1 KB row of digit counts per partition
  uint32_t*       gstatus;  // [ceil(maxParts / 32)][256]
  uint32_t*       zStatus;  // look-back words no pass is using: cleared here for a later pass (launchOsSort has the rota)
  uint32_t        zWords;
  const uint32_t* nPtr;
  FrameCounters*  ctr;
  int             pass;
  int             digitMode;  // 0 plain byte `pass`; 1 pass 2: rank of key >> 16 when the plan says remap, else plain; 2 pass 3: exits when remapped
  int             finalMode;  // 0 writes pairs; 1 writes the result; 2 writes the result iff the plan says remap (pass 2)
  // Round 6: the partition size chosen ON THE DEVICE from the element count (osPartOf below): ~384 partitions between partMin and kOsPart
  uint32_t        partMin;    // smallest partition size allowed (kOsPart: the fixed size of rounds 3-5; A/B)
  uint32_t        resSlots;   // workgroups of this kernel the chip holds at once
  uint32_t        flatLookback;  // 1: sorts of at most 32 groups resolve the groups before a partition from COUNTED SUMS instead of the chain of group prefixes (below; MGS_OS_FLAT)
};

// A sort of few keys used to run on few workgroups: a strip's 0.32 M keys were 78 partitions of 4096 on 256 CUs, a train-sized
// frame's 0.74 M were 181 (64 us and 54 us, a third of a strip's frame).  The rounds of a partition adapt to its element
// count anyway (the ragged last one), so the SAME kernel takes smaller partitions: every workgroup derives the size from n
// (device-side count, identical for all) and the grid the host launched — no host-side guess, nothing in the graph key.  The size:
// ~384 partitions, in steps of 512 pairs, between partMin (1 536) and 4 096 — so every sort above 1.5 M keys keeps 4 096 — and only
// while all partitions are resident at once.  Round 6's first measurement of this (with the CHAIN as level 2 of the look-back) found
// no gain: what small partitions saved in the table's expansion and the ranking they paid in the look-back (four times the
// members and groups to wait for).  With the flat level 2 nothing waits, and the smaller partitions pay: same box x3 alternating
// (profiles/r6_x2_sort_part_adaptive_ab.log), sort of a top strip of eight 72.2 -> 51.1 us (its frame +14 %), a middle strip 62.0
// -> 58.2, a 400 K-splat scene 52.6 -> 42.2, train-sized 52.0 -> 49.1, sparse 54.7 -> 52.0; 1.3 M keys and more: unchanged.
// (Fixed 2 048 instead: within 2 us of this rule on every workload, 5 us worse on the top strip; 2 048 for a 2 M-key sort — 980
// partitions — is 6 us WORSE than 4 096: profiles/r6_x_sort_part_sweep.log, r6_x3_sort_part_rules.log.)  MGS_OS_PART_MIN=4096: the
// fixed size of rounds 3-5 (A/B; the variants test runs it).
__device__ __forceinline__ uint32_t osPartOf(uint32_t n, uint32_t grid, uint32_t partMin, uint32_t resSlots)
{
  if(partMin >= kOsPart)
    return kOsPart;
  // ~384 partitions, in steps of 512 pairs, between partMin and kOsPart (round 6, with the flat level 2: below)
  const uint32_t cap  = min(grid, resSlots);
  const uint32_t want = ((n / 384u + 511u) / 512u) * 512u;
  const uint32_t part = min(max(want, partMin), kOsPart);
  return (n + part - 1u) / part <= cap ? part : kOsPart;
}

#ifndef MGS_OS_WAVES
#define MGS_OS_WAVES 4
#endif
template <int IN, bool REMAP>
__global__ __launch_bounds__(kThreads, MGS_OS_WAVES) void k_os_pass(const OsPassArgs a)
{
  __shared__ uint2    s_pair[kOsPart];       // 32 KB: the partition's pairs ordered by digit
  __shared__ uint16_t s_whist[kWaves][256];  //  2 KB: per wave digit counts -> offsets
  __shared__ uint16_t s_loff[256];           //  partition-local exclusive digit offsets
  __shared__ uint32_t s_cnt[256];            //  digit counts of the partition, later the global base of every digit
  __shared__ uint8_t  s_rv[REMAP ? kRemapSpan : 4];  // 4 KB: rank table of key >> 16 (remapped pass)
  __shared__ uint32_t s_tmp[8];

  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  OsPlan*   plan = a.plan;
  const bool remapped = plan->remapOn != 0u;
  if(a.digitMode == 2 && remapped)
    return;  // pass 2 sorted on the rank of the top 16 bits and wrote the result
#ifdef MGS_OS_TRACE
  __shared__ uint64_t trc[8];
  __shared__ uint64_t gtr[8];
  if(threadIdx.x < 8) gtr[threadIdx.x] = 0;
  MGS_OS_STAMP(0)
#endif
  const uint32_t p = blockIdx.x;  // partitions in dispatch order (header: why no ticket)
  const uint32_t n = *a.nPtr;
  [[maybe_unused]] uint4 tot0q = make_uint4(0u, 0u, 0u, 0u);
  if constexpr(IN == 3)  // the virtual pass 0 starts from the digit-0 totals (lane l: values 4 l .. 4 l + 3): their round trip passes behind the set-up
    tot0q = *reinterpret_cast<const uint4*>(&plan->total[0][4 * lane]);
  // clear look-back words for a later pass (stream order: nobody reads them any more)
  for(uint32_t i = blockIdx.x * kThreads + t; i < a.zWords; i += gridDim.x * kThreads)
    a.zStatus[i] = 0u;
  if constexpr(IN == 3)
    if(blockIdx.x >= gridDim.x - kOsFoldWgs)
    {  // workgroups beyond the partitions: what the SECOND kernel sorts on (foldTop16), beside this kernel's own work
      foldTop16<kThreads>(plan, a.top16Count, a.allowRemap, blockIdx.x - (gridDim.x - kOsFoldWgs), reinterpret_cast<uint32_t*>(s_pair), s_tmp);
      return;
    }
  const uint32_t part  = osPartOf(n, gridDim.x - (IN == 3 ? kOsFoldWgs : 0u), a.partMin, a.resSlots);  // wave-uniform, the same in every workgroup
  const uint32_t parts = (uint32_t)(((uint64_t)n + part - 1u) / part);
  if(p >= parts)
    return;
  const bool finalOut = a.finalMode == 1 || (a.finalMode == 2 && remapped);
  if(p == 0 && t == 0 && a.planOut != nullptr && (finalOut || a.digitMode == 2))
  {
    a.planOut->n         = plan->n;
    a.planOut->finalSel  = 0u;
    a.planOut->passesRun = (uint32_t)a.pass + 1u;
    if(finalOut)
      a.planOut->reserved[0] = a.rideShift != 0u ? a.rideInfo : 0u;  // the binning stage finds the rectangles in sorted order
  }

  // ---- load: wave w owns a contiguous quarter of the partition, lane-interleaved, so (wave, round, lane) is memory order.
  // The rounds adapt to the element count (the last partition is ragged).
  const uint32_t count = (uint32_t)min((uint64_t)part, (uint64_t)n - (uint64_t)p * part);
  const uint32_t rounds = (count + kThreads - 1u) / kThreads;  // per wave: `rounds` rounds of 64 keys
  const uint32_t wofs   = w * 64u * rounds;
  uint32_t       key[kKpt], val[kKpt];
  MGS_OS_STAMP(1)
  [[maybe_unused]] uint32_t srcAt[IN == 3 ? kKpt : 1];
  auto loadPairs = [&]() {
    // clamped, not predicated: a predicated load becomes a branch + wait and serialises the fetches
#pragma unroll
    for(int i = 0; i < kKpt; ++i)
    {
      key[i] = 0xFFFFFFFFu;
      val[i] = 0u;
      if((uint32_t)i < rounds)  // wave-uniform
      {
        const uint32_t idx = min(wofs + (uint32_t)i * 64u + lane, count - 1u);
        if(IN == 2)
        {
          key[i] = a.srcKeys[(size_t)p * part + idx];
          val[i] = a.srcVals[(size_t)p * part + idx];
        }
        else
        {
          const uint2 kv = IN == 3 ? a.srcPairs[srcAt[IN == 3 ? i : 0]] : a.srcPairs[(size_t)p * part + idx];
          key[i] = kv.x;
          val[i] = kv.y;
        }
      }
    }
#pragma unroll
    for(int i = 0; i < kKpt; ++i)
      if(wofs + (uint32_t)i * 64u + lane >= count)
      {
        key[i] = 0xFFFFFFFFu;
        val[i] = 0u;
      }
  };
  // (round 6) a pass over contiguous input requests its pairs BEFORE it sets up its LDS (the zeroed wave histograms, the rank table
  // and their two barriers: 1.9 us of the workgroup's life that the loads' round trip used to follow); the first pass of a frame
  // cannot — its source table is built in LDS first
  if constexpr(IN != 3)
    loadPairs();
  for(int i = t; i < kWaves * 256; i += kThreads)
    (&s_whist[0][0])[i] = 0;
  const bool useRemap = REMAP && remapped;
  if constexpr(REMAP)
    if(useRemap)
    {  // every entry holds the largest rank first: a value outside the table (padding keys) sorts behind every real key
      const uint32_t count = plan->remapCount, base = plan->remapBase;
      const uint32_t fill  = plan->remapPadRank * 0x01010101u;
      for(int i = t; i < (int)kRemapSpan / 4; i += kThreads)
        reinterpret_cast<uint32_t*>(s_rv)[i] = fill;
      __syncthreads();
      for(uint32_t i = t; i < count; i += kThreads)
        s_rv[(uint32_t)plan->remapVals[i] - base] = (uint8_t)i;
    }
  __syncthreads();
  // IN 3: the source table of the virtual pass 0 (header of OsPassArgs).  s_pair is not in use before the re-order: its first
  // half holds the table.  Everything that locates the runs — the digit bases D, the prefix over the chunks — is computed by
  // EVERY WAVE FOR ITSELF (identical results, a few dozen loads each): the construction has no workgroup barrier except the
  // one before the table is read (with block-wide scans it had nine, 8 us per digit value; profiles/r4_c_os_trace.log).
  if constexpr(IN == 3)
  {
    uint32_t* s_src = reinterpret_cast<uint32_t*>(s_pair);       // [4096] pair index (slot * 2048 + entry) of every position of the partition
    uint32_t* s_cpw = s_src + kOsPart + (uint32_t)w * 264u;      // [257] this wave's prefix of chunkSum[.][d] over a tile of chunks
    constexpr uint32_t kCpTile = 256;                            // chunks per tile: four per lane
    const uint32_t a0 = p * part, e0 = a0 + count;
    // D: lane l holds digit-0 values 4 l .. 4 l + 3
    const uint32_t tk[4] = {tot0q.x, tot0q.y, tot0q.z, tot0q.w};
    uint32_t       d     = 256u;
    {
      const uint32_t s4 = tk[0] + tk[1] + tk[2] + tk[3];
      uint32_t       dq = waveInclusiveScan(s4) - s4;
#pragma unroll
      for(int k = 0; k < 4; ++k)
      {
        s_cnt[4 * lane + k] = dq;  // (s_cnt is free until the scatter; the four waves write the same values)
        const uint64_t bk   = __ballot(tk[k] != 0u && dq <= a0 && a0 < dq + tk[k]);
        if(bk != 0ull)
          d = 4u * (uint32_t)__builtin_ctzll(bk) + (uint32_t)k;  // the value the partition starts in
        dq += tk[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
    MGS_OS_GSTAMP(0)
    auto loadCs = [&](uint32_t dd, uint32_t c0, uint32_t cs[4]) {
#pragma unroll
      for(int k = 0; k < 4; ++k)
      {
        const uint32_t c = c0 + 4u * (uint32_t)lane + (uint32_t)k;
        cs[k]            = (dd < 256u && c < a.chunks) ? a.chunkSum[(size_t)c * 256u + dd] : 0u;
      }
    };
    const uint32_t spad = a.chunks * kOsChunk;
    uint32_t       csN[4];
    loadCs(d, 0u, csN);
    while(d < 256u)
    {  // uniform (every wave computes the same): the digit-0 values whose range [Dd, De) overlaps [a0, e0)
      const uint32_t Dd = s_cnt[d], De = d < 255u ? s_cnt[d + 1u] : n;
      if(Dd >= e0)
        break;
      uint32_t cs[4] = {csN[0], csN[1], csN[2], csN[3]};
      loadCs(De < e0 ? d + 1u : 256u, 0u, csN);  // the next value's first tile travels behind this one's expansion
      if(De > a0 && De != Dd)
      {
        uint32_t carry = Dd;  // position of the first pair of value d in the tile's first chunk
        for(uint32_t c0 = 0;;)
        {
          const uint32_t sum = cs[0] + cs[1] + cs[2] + cs[3];
          const uint32_t inc = waveInclusiveScan(sum);
          MGS_OS_GSTAMP(1)  // chunk sums arrived
          const uint32_t tileTotal = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
          uint32_t       run = inc - sum, cA = kCpTile, cB = 0u;  // chunks [cA, cB) of the tile hold pairs inside [a0, e0)
#pragma unroll
          for(int k = 0; k < 4; ++k)
          {
            s_cpw[4 * lane + k] = run;
            const uint32_t lo   = carry + run, hi = lo + cs[k];
            const uint64_t bk   = __ballot(cs[k] != 0u && lo < e0 && hi > a0);
            if(bk != 0ull)
            {
              cA = min(cA, 4u * (uint32_t)__builtin_ctzll(bk) + (uint32_t)k);
              cB = max(cB, 4u * (63u - (uint32_t)__builtin_clzll(bk)) + (uint32_t)k + 1u);
            }
            run += cs[k];
          }
          if(lane == 63)
            s_cpw[kCpTile] = run;
          __builtin_amdgcn_wave_barrier();
          MGS_OS_GSTAMP(2)  // chunk range known
          if(cB != 0u)
          {
            const uint32_t  entries = (cB - cA) * kOsChunk;
            const uint32_t* row     = a.runTab + (size_t)d * spad + (size_t)(c0 + cA) * kOsChunk;
            // lane == run: 32 consecutive lanes hold a chunk's 32 slots; the workgroup's 256 threads share the runs.  Four loads in
            // flight per thread (the usual partition overlaps ~700 runs: one batch).
            auto expand = [&](uint32_t i, uint32_t ic, uint32_t v) {
              const uint32_t cl  = cA + ic / kOsChunk;  // chunk within the tile
              const uint32_t inC = v & 0xFFFFu;
              const uint32_t nxt = (uint32_t)__shfl_down((int)inC, 1, 64);
              const uint32_t len = ((ic & (kOsChunk - 1u)) == kOsChunk - 1u ? s_cpw[cl + 1u] - s_cpw[cl] : nxt) - inC;
              const uint32_t R   = carry + s_cpw[cl] + inC;  // position of the run's first pair
              uint32_t       lo  = max(R, a0), hi = min(R + len, e0);
              if(i >= entries || hi <= lo)
                lo = hi = 0u;
              const uint32_t src = ((c0 + cl) * kOsChunk + (ic & (kOsChunk - 1u))) * kOsSlot + (v >> 16);  // the run's first pair
              // short runs (the rule: ~6 pairs) are written by their lane; long ones by the wave together
              const bool     big = hi - lo > 16u;
              if(!big)
                for(uint32_t x = lo; x < hi; ++x)
                  s_src[x - a0] = src + (x - R);
              uint64_t bm = __ballot(big);
              while(bm != 0ull)
              {
                const int      l   = (int)__builtin_ctzll(bm);
                bm &= bm - 1ull;
                const uint32_t loL = (uint32_t)__builtin_amdgcn_readlane((int)lo, l), hiL = (uint32_t)__builtin_amdgcn_readlane((int)hi, l);
                const uint32_t dl  = (uint32_t)__builtin_amdgcn_readlane((int)(src - R), l);
                for(uint32_t x = loL + (uint32_t)lane; x < hiL; x += 64u)
                  s_src[x - a0] = dl + x;
              }
            };
            for(uint32_t i0 = 0; i0 < entries; i0 += 4u * kThreads)
            {
              uint32_t vv[4];
#pragma unroll
              for(int k = 0; k < 4; ++k)
                vv[k] = row[min(i0 + (uint32_t)k * kThreads + (uint32_t)t, entries - 1u)];
#ifdef MGS_OS_TRACE
              if(vv[0] + vv[1] + vv[2] + vv[3] == 0x12345678u) gtr[7] = 1;  // consume the loads: the stamp follows their arrival
              MGS_OS_GSTAMP(3)
#endif
#pragma unroll
              for(int k = 0; k < 4; ++k)
              {
                const uint32_t i = i0 + (uint32_t)k * kThreads + (uint32_t)t;
                if(i0 + (uint32_t)k * kThreads + (uint32_t)(t & ~63) < entries)  // wave-uniform
                  expand(i, min(i, entries - 1u), vv[k]);
              }
            }
          }
          carry += tileTotal;
          c0 += kCpTile;
          if(c0 >= a.chunks || carry >= e0)
            break;
          __builtin_amdgcn_wave_barrier();  // the wave's prefix table is rewritten
          loadCs(d, c0, cs);
        }
        MGS_OS_GSTAMP(4)  // first digit value expanded
      }
      ++d;
    }
    __syncthreads();
    MGS_OS_STAMP(1)  // (trace build: the table's construction counts as "table + zeroing", the gather itself as "loads")
#pragma unroll
    for(int i = 0; i < kKpt; ++i)
      srcAt[i] = min(s_src[min(wofs + (uint32_t)i * 64u + lane, count - 1u)], a.srcLimit);
  }
  if constexpr(IN == 3)
    loadPairs();
  const int      shift = 8 * a.pass;
  const uint32_t rbase = plan->remapBase;
  uint32_t       rd[kKpt];  // digit << 16 | rank inside the wave (one register per key)
#pragma unroll
  for(int i = 0; i < kKpt; ++i)
  {
    uint32_t d;
    if(REMAP && useRemap)
      d = s_rv[min((key[i] >> 16) - rbase, kRemapSpan - 1u)];
    else
      d = (key[i] >> shift) & 255u;
    rd[i] = d << 16;
  }
  const uint32_t g = p / kOsGroup, m = p % kOsGroup;
  // bits that tell the digits of this pass apart: 8, or — sorting on the rank of key >> 16 — as many as the largest rank has
  [[maybe_unused]] const int rankBits = (REMAP && useRemap) ? 32 - __builtin_clz(plan->remapPadRank | 1u) : 8;
  MGS_OS_STAMP(2)
  // Round 6 — flat level 2 (sorts of at most 32 groups = 1 024 partitions: a frame's).  The chain of rounds 3-5 — a group's last
  // member folds its rows, publishes the group total, looks back over the group totals, publishes the inclusive prefix; every
  // other member polls that word — is three dependent round trips that START when the last member has ranked its keys, the moment
  // everybody else starts waiting: the partitions of a pass run in step (15-24 polls per workgroup, 5.2-7.3 us in "re-order +
  // level 2": profiles/r6_l_os_trace_flat0.log).  Instead every partition ADDS its counts to its group's row — one fire-and-forget
  // atomic per digit; the word carries the sum (< 2^20: 32 x 4096) and, above it, how many partitions have added — and reads the
  // rows of ALL groups before its own (<= 31 words per digit thread, requested behind the level-1 fold, consumed behind the LDS
  // re-order); a row counts once 32 partitions have arrived.  No last member, no dependent trips: polls per workgroup 23 -> 1 and
  // 15 -> 0, "re-order + level 2" 7.3 -> 2.5 and 5.2 -> 2.4 us (profiles/r6_l_os_trace_flat1.log).  (Publishing the counts BEFORE the
  // ranking as well — an LDS histogram, one atomic per key — was measured on top: no polls left to remove, and the histogram costs
  // the ranking phase +1 us / +2.3 us even on four copies per wave: the rank pass has 5-8 digit values — dropped.)
  const uint32_t groupsAll = (parts + kOsGroup - 1u) / kOsGroup;
  const bool     flat      = a.flatLookback != 0u && groupsAll <= 32u;

  // ---- (the look-back is software-pipelined with the rest of the pass: level 1 is issued right after the partition's own
  // counts are published, behind the ranking, and consumed behind the scans; level 2 travels behind the LDS re-order) ----

  // ---- per-wave multi-split: rank of each key among the keys of its wave with the same digit (stable) ----
#pragma unroll
  for(int i = 0; i < kKpt; ++i)
  {
    if((uint32_t)i < rounds)  // wave-uniform: skipped rounds hold nothing
    {
      // padding lanes (idx >= count) carry key 0xFFFFFFFF: the largest digit, behind every real key of the partition
      // lanes holding the same digit: per bit, x = 0 / ~0 from the bit (one bfe), the ballot of the bit, and
      // mask &= ~(ballot ^ x) on both halves — 6 vector instructions per bit
      const uint32_t d = rd[i] >> 16;
      uint32_t       mlo = ~0u, mhi = ~0u;
#pragma unroll
      for(int b = 0; b < 8; ++b)
      {
        if(REMAP && b >= rankBits)  // wave-uniform: the ranks of a remapped pass need ceil(log2(values)) bits — 3 for the usual 5-8
          break;
        uint32_t x = (uint32_t)((int32_t)(d << (31 - b)) >> 31);
        asm volatile("" : "+v"(x));  // the ballot compares x itself: left alone the compiler re-derives it from d (a shift per bit)
        const uint64_t bal = __ballot(x != 0u);
        // m & ~(bal ^ x) in one v_bitop3 per half (truth table 0x90: a & !(b ^ c))
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t)bal, x, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t)(bal >> 32), x, 0x90);
      }
      const uint32_t lower = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
      const uint32_t cnt   = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
      const uint32_t pre   = s_whist[w][d];
      rd[i] |= pre + lower;
      __builtin_amdgcn_wave_barrier();  // every lane of the group has read `pre` before the leader bumps it
      if(lower == 0)
        s_whist[w][d] = (uint16_t)(pre + cnt);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  MGS_OS_STAMP(3)
  // thread t == digit t: the partition's count of digit t (the ranking counted the padding lanes too: they sit in the
  // largest digit) -> published; per wave offsets
  uint32_t tot = 0;
#pragma unroll
  for(int q = 0; q < kWaves; ++q)
  {
    const uint32_t c = s_whist[q][t];
    s_whist[q][t]    = (uint16_t)tot;
    tot += c;
  }
  const uint32_t padDigit = (REMAP && useRemap) ? plan->remapPadRank : 255u;
  const uint32_t myCount  = tot - (((uint32_t)t == padDigit) ? rounds * kThreads - count : 0u);
  stAgent(&a.status[(size_t)p * 256u + t], kAgg | myCount);
  if(flat)
    __hip_atomic_fetch_add(&a.gstatus[(size_t)g * 256u + t], myCount | (1u << 20), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Level 1 by rows: the counts a member published are one 1 KB row (256 digits).  Wave w folds the rows w, w + 4, ... of the
  // members before me for four digits per lane (lane l: digits 4 l .. 4 l + 3);
  // the four waves' partial sums meet in LDS.  Issued now, consumed behind the scans.
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane(w);  // the compiler does not know that the wave index is uniform
  // four dword sc1 loads per lane and row (a wave instruction covers 256 contiguous bytes).  One 16-byte
  // buffer_load_dwordx4 ... sc1 per lane was tried and does NOT work: re-polls kept returning the zeros of the first read
  // (every wait ran into its bound), while dword / dwordx2 sc1 loads of the same words see the update.
  auto loadRow = [&](uint32_t row) -> v4u {
    const uint32_t* q = a.status + ((size_t)g * kOsGroup + row) * 256u + (uint32_t)lane * 4u;
    return v4u{ldAgent(q), ldAgent(q + 1), ldAgent(q + 2), ldAgent(q + 3)};
  };
  v4u rv[8];
#pragma unroll
  for(int k = 0; k < 8; ++k)
  {
    const uint32_t row = wu + 4u * k;
    rv[k]              = v4u{kAgg, kAgg, kAgg, kAgg};
    if(row < m)
      rv[k] = loadRow(row);
  }
  uint32_t spins = 0, intra = 0, base = 0;
  bool     bad   = false;
  // level 1, consume: fold the rows (a member that had not published all four words when they were read is re-polled);
  // the four waves' partial sums meet in the first 4 KB of s_pair, which is not in use before the re-order
  auto foldRows = [&]() {
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for(int k = 0; k < 8; ++k)
    {
      const uint32_t row = wu + 4u * k;
      if(row < m)
      {
        while(((rv[k].x >> 30) == 0u || (rv[k].y >> 30) == 0u || (rv[k].z >> 30) == 0u || (rv[k].w >> 30) == 0u) && !bad)
        {
          rv[k] = loadRow(row);
          if(++spins > kSpinMax)
            bad = true;
        }
        acc[0] += rv[k].x & kValMask;
        acc[1] += rv[k].y & kValMask;
        acc[2] += rv[k].z & kValMask;
        acc[3] += rv[k].w & kValMask;
      }
    }
    reinterpret_cast<uint4*>(s_pair)[wu * 64 + lane] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    const uint32_t* s_l1 = reinterpret_cast<const uint32_t*>(s_pair);
    intra                = s_l1[t] + s_l1[256 + t] + s_l1[512 + t] + s_l1[768 + t];
  };
  // level 2.  The group's LAST member does the group's work at once, before its own scans: it folds its rows (the other
  // members published at the same moment), publishes the group total, looks back over the groups (16 per batch, down to the
  // first inclusive prefix) and publishes the inclusive prefix — so that the 31 other members, which fold behind their scans
  // and need that one word behind their LDS re-order, find it there.  It finishes ~3 us behind them instead of all of them
  // waiting ~5 us for it.
  const bool lastMember = !flat && m == kOsGroup - 1u;
  if(lastMember)
  {
    foldRows();
    const uint32_t gtotal = intra + myCount;
    stAgent(&a.gstatus[(size_t)g * 256u + t], kAgg | gtotal);
    int gq = (int)g - 1;
    while(gq >= 0 && !bad)
    {
      uint32_t gw[kGroupWindow];
#pragma unroll
      for(int j = 0; j < kGroupWindow; ++j)
        gw[j] = (gq - j >= 0) ? ldAgent(&a.gstatus[(size_t)(gq - j) * 256u + t]) : kInc;
      bool done = false;
#pragma unroll
      for(int j = 0; j < kGroupWindow; ++j)
      {
        if(done)
          continue;
        if((gw[j] >> 30) == 0u)
        {  // not published yet: resume the window at this group
          done = true;
          gq -= j;
          if(++spins > kSpinMax)
            bad = true;
          continue;
        }
        base += gw[j] & kValMask;
        if((gw[j] >> 30) == 2u)
        {
          done = true;
          gq   = -1;
        }
        else if(j == kGroupWindow - 1)
        {
          done = true;
          gq -= kGroupWindow;
        }
      }
    }
    stAgent(&a.gstatus[(size_t)g * 256u + t], kInc | ((base + gtotal) & kValMask));
  }
  // local exclusive scan over the digits (padding included)
  const uint32_t loff = scan256(tot, s_tmp);
  s_loff[t]           = (uint16_t)loff;
  // digit bases of the whole array: exclusive scan of the totals (known before the pass started)
  const uint32_t below = scan256(plan->total[a.pass][t], s_tmp);
  if(!lastMember)
    foldRows();
  // flat level 2: the counted sums of the groups before mine, thread t == digit t, all requested now (the registers of the
  // level-1 rows are free), consumed behind the re-order
  uint32_t gl[31];
  if(flat)
  {
#pragma unroll
    for(uint32_t k = 0; k < 31u; ++k)
    {
      gl[k] = 32u << 20;
      if(k < g)
        gl[k] = ldAgent(&a.gstatus[(size_t)k * 256u + t]);
    }
  }
  __syncthreads();  // everybody has read the partial sums: s_pair may be overwritten
  MGS_OS_STAMP(4)
  // ---- re-order through LDS so that equal digits are contiguous ----
#pragma unroll
  for(int i = 0; i < kKpt; ++i)
    if((uint32_t)i < rounds)
    {
      const uint32_t d   = rd[i] >> 16;
      const uint32_t pos = (uint32_t)s_loff[d] + (uint32_t)s_whist[w][d] + (rd[i] & 0xFFFFu);
      s_pair[pos]        = make_uint2(key[i], val[i]);
    }
  if(flat)
  {
#pragma unroll
    for(uint32_t k = 0; k < 31u; ++k)
      if(k < g)
      {
        while((gl[k] >> 20) != kOsGroup && !bad)
        {
          gl[k] = ldAgent(&a.gstatus[(size_t)k * 256u + t]);
          if(++spins > kSpinMax)
            bad = true;
        }
        base += gl[k] & 0xFFFFFu;
      }
  }
  else if(!lastMember && g > 0u)
  {  // the inclusive prefix of the previous group is one word
    uint32_t v;
    while(((v = ldAgent(&a.gstatus[(size_t)(g - 1u) * 256u + t])) >> 30) != 2u)
      if(++spins > kSpinMax)
      {
        bad = true;
        break;
      }
    base = v & kValMask;
  }
  if(bad)
    atomicOr(&a.ctr->errorFlags, kErrSpinTimeout);
  s_cnt[t] = below + (base + intra) - loff;  // wraps are fine: only base + idx is used
  __syncthreads();
  MGS_OS_STAMP(5)

  // ---- coalesced scatter: consecutive threads write consecutive addresses inside each digit run ----
#pragma unroll
  for(int i = 0; i < kKpt; ++i)
    if((uint32_t)i < rounds)
    {
      const uint32_t idx = (uint32_t)i * kThreads + t;
      if(idx < count)
      {
        const uint2 kv = s_pair[idx];
        uint32_t    d;
        if(REMAP && useRemap)
          d = s_rv[min((kv.x >> 16) - rbase, kRemapSpan - 1u)];
        else
          d = (kv.x >> shift) & 255u;
        const uint32_t dst = s_cnt[d] + idx;
        if(finalOut && a.rideShift != 0u)
        {
          a.dstVals[dst]   = kv.y & ((1u << a.rideShift) - 1u);
          a.dstCode16[dst] = a.rideSplit ? (uint16_t)((kv.x & 255u) | ((kv.y >> a.rideShift) << 8)) : (uint16_t)(kv.y >> a.rideShift);
        }
        else if(finalOut)
        {
          a.dstVals[dst] = kv.y;
          if(a.dstKeys != nullptr)
            a.dstKeys[dst] = kv.x;
        }
        else
          a.dstPairs[dst] = kv;
      }
    }
#ifdef MGS_OS_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MGS_OS_STAMP(6)
  if(t == 0 && a.trace)
  {
    uint64_t* o = a.trace + (size_t)p * 8;
    for(int i = 0; i < 7; ++i) o[i] = trc[i];
    o[7] = ((uint64_t)count << 32) | spins;
    if(IN == 3)
    {  // the table's sub-stamps go where the (virtual) pass 0 would have put its own
      uint64_t* g = a.trace - (size_t)(gridDim.x - kOsFoldWgs) * 8 + (size_t)p * 8;  // (this kernel's grid has workgroups beyond the partitions)
      g[0] = trc[0];
      for(int i = 0; i < 5; ++i) g[1 + i] = gtr[i];
      g[6] = trc[1];
      g[7] = 0;
    }
  }
#endif
}

// small state clear for the stand-alone sort (inside a frame the frame-init kernel zeroes the plan)
__global__ void k_os_plan_clear(OsPlan* plan)
{
  uint32_t* w = reinterpret_cast<uint32_t*>(plan);
  for(uint32_t i = threadIdx.x; i < sizeof(OsPlan) / 4; i += blockDim.x)
    w[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Grid of a pass (and rows of its look-back words): partitions of kOsPart pairs for the largest count, and — round 6 — at least
// as many workgroups as 1024-pair partitions of that count, up to kOsSmallGrid, so that a sort of few keys can spread over the chip
// (osPartOf).  Workgroups beyond the partitions exit after the set-up.
constexpr uint32_t kOsSmallGrid = 1024;
// MGS_OS_PART_MIN = the smallest partition size the passes may choose on the device (osPartOf above; a multiple of 256 in
// [1024, 4096]; default 1 536; 4096 = the fixed size of rounds 3-5).
static uint32_t osPartMinEnv()
{
  static const uint32_t v = [] {
    const char* e = std::getenv("MGS_OS_PART_MIN");
    const int   x = e ? std::atoi(e) : 0;
    return (x >= 1024 && x <= (int)kOsPart && x % 256 == 0) ? (uint32_t)x : 1536u;
  }();
  return v;
}
uint32_t osSortMaxParts(uint32_t maxElems)
{
  const uint32_t big = (uint32_t)(((uint64_t)maxElems + kOsPart - 1u) / kOsPart);
  if(osPartMinEnv() >= kOsPart)
    return big;
  const uint32_t small = (uint32_t)std::min<uint64_t>(((uint64_t)maxElems + 1023u) / 1024u, kOsSmallGrid);
  return std::max(big, small);
}
size_t osSortStatusWords(uint32_t maxParts)
{  // per buffer: partition rows [partition][digit] (rounded up to whole groups) followed by group rows [group][digit]
  const size_t groups = (maxParts + kOsGroup - 1u) / kOsGroup + 1u;
  return groups * 256u * kOsGroup + groups * 256u;
}

void launchOsSortClearPlan(hipStream_t stream, OsPlan* plan)
{
  hipLaunchKernelGGL(k_os_plan_clear, dim3(1), dim3(256), 0, stream, plan);
}

void launchOsSort(hipStream_t stream, const OsLaunch& L)
{
  if(L.maxElems == 0 || (uint64_t)L.maxElems >= kOsMaxPairs)
    return;  // (the callers reject / re-route 2^30 pairs and more: a prefix would wrap inside its status word)
  const bool     frame    = L.pairs0 != nullptr;  // the project kernels' slots of pairs + their histograms / records
  const uint32_t maxParts = osSortMaxParts(L.maxElems);
  // the device-side choice of the partition size (osPartOf): off unless MGS_OS_PART_MIN asks for it (osPartMinEnv above)
  static uint32_t       devSlots[64] = {};
  int                   dev = 0;
  (void)hipGetDevice(&dev);
  if(dev >= 0 && dev < 64 && devSlots[dev] == 0u)
  {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    devSlots[dev] = (uint32_t)std::max(cus, 1) * (uint32_t)MGS_OS_WAVES;
  }
  const uint32_t partMin  = L.partMin ? L.partMin : osPartMinEnv();
  const uint32_t resSlots = L.resSlots ? L.resSlots : ((dev >= 0 && dev < 64) ? devSlots[dev] : 1024u);
  const uint32_t sWords   = (uint32_t)osSortStatusWords(maxParts);
  // Three sets of look-back words: pass 0 uses set 0, pass 1 set 1, pass 2 set 2, pass 3 set 1 again.  Every set is zero when its
  // pass starts: pass 0 clears sets 1 and 2 (whatever the previous sort left there, whether its pass 3 ran or not), pass 1
  // clears set 0 for the next sort, pass 2 clears set 1 for pass 3.
  uint32_t*      st[3]    = {L.status, L.status + sWords, L.status + 2 * (size_t)sWords};
  auto gOf = [&](uint32_t* s) { return s + (size_t)((maxParts + kOsGroup - 1u) / kOsGroup + 1u) * 256u * kOsGroup; };
#ifdef MGS_OS_TRACE
  static uint64_t* traceBuf = nullptr;
  const char*      tracePath = std::getenv("MGS_OS_TRACE_FILE");
  const size_t     traceN = (size_t)maxParts * 8;
  static uint64_t* prepBuf = nullptr;
  if(tracePath && frame)
  {
    if(!traceBuf)
    {
      (void)hipMalloc(&traceBuf, (size_t)4 << 24);
      (void)hipMalloc(&prepBuf, 4096 * 64);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_osPrepTrace), &prepBuf, sizeof(prepBuf));
    }
    (void)hipMemsetAsync(traceBuf, 0, 4 * traceN * 8, stream);
    (void)hipMemsetAsync(prepBuf, 0, 4096 * 64, stream);
  }
#endif
  if(!frame)
    hipLaunchKernelGGL(k_os_hist, dim3(std::min<uint32_t>((L.maxElems + 2047u) / 2048u, 1024u)), dim3(256), 0, stream, L.keys0, L.nPtr, L.plan);
  const uint32_t reduceWgs = frame ? osSortChunks(L.prjParts) : 0u;
  hipLaunchKernelGGL(k_os_prepare, dim3(reduceWgs + 1u), dim3(1024), 0, stream, frame ? L.slotHist : nullptr, L.top16Rec, L.prjParts,
                     L.top16Count, L.plan, L.nPtr, (frame && L.allowRemap) ? 1 : 0, reduceWgs, L.slotCount, L.chunkSum, L.runTab, L.nOut, L.prjOrderOut);
  // A frame's sort starts at pass 1: its pass 0 is virtual (slot_emit.h, OsPassArgs).
  for(int pass = frame ? 1 : 0; pass < 4; ++pass)
  {
    OsPassArgs a{};
#ifdef MGS_OS_TRACE
    a.trace = (tracePath && frame) ? traceBuf + (size_t)pass * traceN : nullptr;
#endif
    a.plan    = L.plan;
    a.planOut = L.planOut;
    // look-back words.  Stand-alone sort: pass 0 uses set 0, 1 -> 1, 2 -> 2, 3 -> 1 again; pass 0 clears sets 1 and 2 (whatever the
    // previous sort left there, whether its pass 3 ran or not), pass 1 clears set 0 for the next sort, pass 2 clears set 1 for
    // pass 3.  Frame: pass 1 uses set 0 and clears sets 1 and 2, pass 2 uses set 1 and clears set 0, pass 3 uses set 2.
    const int set = frame ? pass - 1 : (pass == 3 ? 1 : pass);
    a.status  = st[set];
    a.gstatus = gOf(st[set]);
    if(frame)
    {
      a.zStatus = pass == 1 ? st[1] : st[0];
      a.zWords  = pass == 1 ? 2u * sWords : (pass == 2 ? sWords : 0u);
    }
    else
    {
      a.zStatus = pass == 1 ? st[0] : st[1];
      a.zWords  = pass == 0 ? 2u * sWords : (pass == 3 ? 0u : sWords);
    }
    a.nPtr    = L.nPtr;
    a.ctr     = L.ctr;
    a.pass    = pass;
    a.partMin  = partMin;
    a.resSlots = resSlots;
    static const uint32_t kFlat = [] { const char* e = std::getenv("MGS_OS_FLAT"); return e ? (uint32_t)std::atoi(e) : 1u; }();
    a.flatLookback = kFlat;
    a.dstKeys = L.outKeys;
    a.dstVals = L.outVals;
    // stand-alone: pass 0 -> A, 1 -> B, 2 -> A, 3 -> the result.  Frame: pass 1 reads the project kernels' slots, which live in B,
    // -> A, 2 -> B (or the result), 3 -> the result.
    if(frame)
    {
      a.srcPairs = pass == 1 ? L.pairs0 : (pass == 2 ? L.pairA : L.pairB);
      a.dstPairs = pass == 1 ? L.pairA : L.pairB;
    }
    else
    {
      a.srcPairs = (pass & 1) ? L.pairA : L.pairB;
      a.dstPairs = (pass & 1) ? L.pairB : L.pairA;
    }
    a.chunkSum  = L.chunkSum;
    a.runTab    = L.runTab;
    a.chunks    = reduceWgs;
    a.srcLimit  = L.prjParts * kOsSlot - 1u;
    a.top16Count = L.top16Count;
    a.allowRemap = (frame && L.allowRemap) ? 1 : 0;
    a.rideShift = frame ? L.rideShift : 0u;
    a.rideInfo  = L.rideInfo;
    a.rideSplit = frame ? L.rideSplit : 0u;
    a.dstCode16 = L.outCode16;
    a.srcKeys   = L.keys0;
    a.srcVals   = L.vals0;
    a.digitMode = (pass == 2 && frame) ? 1 : (pass == 3 ? 2 : 0);
    a.finalMode = (pass == 3) ? 1 : ((pass == 2 && frame) ? 2 : 0);
    const uint32_t grid = maxParts;
    if(grid == 0)
      continue;
    if(pass == 0)
      hipLaunchKernelGGL((k_os_pass<2, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if(pass == 1 && frame)
      hipLaunchKernelGGL((k_os_pass<3, false>), dim3(grid + kOsFoldWgs), dim3(kThreads), 0, stream, a);  // + the workgroups that fold the count table
    else if(pass == 2 && frame)
      hipLaunchKernelGGL((k_os_pass<0, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    else
      hipLaunchKernelGGL((k_os_pass<0, false>), dim3(grid), dim3(kThreads), 0, stream, a);
  }
#ifdef MGS_OS_TRACE
  if(tracePath && frame)
  {
    (void)hipStreamSynchronize(stream);
    std::vector<uint64_t> h(4 * traceN);
    (void)hipMemcpy(h.data(), traceBuf, 4 * traceN * 8, hipMemcpyDeviceToHost);
    if(FILE* fp = std::fopen(tracePath, "wb"))
    {
      const uint64_t hdr[2] = {maxParts, 0};
      std::fwrite(hdr, 8, 2, fp);
      std::fwrite(h.data(), 8, 4 * traceN, fp);
      std::vector<uint64_t> hp(4096 * 8);
      (void)hipMemcpy(hp.data(), prepBuf, 4096 * 64, hipMemcpyDeviceToHost);
      std::fwrite(hp.data(), 8, hp.size(), fp);
      std::fclose(fp);
    }
  }
#endif
}

}  // namespace mgs
