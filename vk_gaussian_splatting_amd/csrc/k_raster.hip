// k_raster.hip — screen-tile binning and per-pixel compositing for gfx950.
//
// The reference has no counterpart for binning: it emits one quad per splat and lets the
// fixed-function rasterizer + ROP blend in primitive (= sorted) order
// (shaders/threedgs_raster.frag.slang:223-309, src/gaussian_splatting.cpp:2066-2087).
// Contract kept here: every pixel centre inside the splat's ellipse
// (A <= 8) with alpha > 1/255 is blended, in the global depth order produced by the sort.
// The reference blends back-to-front ("over"); we walk the same order from the other end and
// accumulate front-to-back with transmittance, which is the same sum in exact arithmetic:
//   C = sum_i c_i a_i prod_{j nearer than i} (1 - a_j).
#include <cstring>
#include <cstdio>
#include <vector>

#include "kernels_common.h"
#include "sh_eval.h"
#include "surface_normal.h"
#include "sort_plan.h"

namespace mgs {

constexpr int kBinThreads = 256;
constexpr int kBinItems   = 8;
constexpr int kBinPart    = kBinThreads * kBinItems;  // 2048 sorted splats per workgroup

__device__ __forceinline__ uint32_t rectTiles(uint32_t r)
{
  const uint32_t x0 = r & 255u, y0 = (r >> 8) & 255u, x1 = (r >> 16) & 255u, y1 = r >> 24;
  return (x1 - x0 + 1u) * (y1 - y0 + 1u);
}

// ---- frame init of the record path: empty tile ranges (counters and sort plans arrive zeroed with the frame's upload) ----
__global__ void k_frame_init(uint2* ranges, uint32_t nTiles)
{
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  for(uint32_t i = gid; i < nTiles; i += gsz)
    ranges[i] = make_uint2(0u, 0u);
}

// ---- binning, record path (frames with more than 256 bins; the default is the direct multi-split below) ----
// Contract: per bin, the splats whose footprint box overlaps it, in global depth order.
// Records (bin id, global id) are emitted in sorted-splat order; a STABLE sort by bin id then keeps
// every bin's list depth-ordered.  The expansion is partitioned by OUTPUT range, not by splat: the
// nearest splats are the largest on screen and sit together at the end of the sorted list, so a
// splat-partitioned expansion leaves a ~1 ms tail on a handful of workgroups (profiles/r1_a).
//   k_bin_count   : tile count of every sorted splat (+ its rect, re-laid out in sorted order) -> block sums
//   k_bin_scan    : exclusive scan of the block sums, total D
//   k_bin_offsets : per-splat exclusive offsets; marks, for every 2048-record output chunk, the sorted
//                   splat that contains the chunk's first record
//   k_bin_expand  : one workgroup per output chunk, balanced and fully coalesced writes
constexpr int kChunk = 2048;  // output records per expand workgroup

__global__ __launch_bounds__(kBinThreads) void k_bin_count(const uint32_t* __restrict__ idsX, const uint32_t* __restrict__ idsY,
                                                           const SortPlan* __restrict__ plan, const uint32_t* __restrict__ rect,
                                                           uint32_t* __restrict__ sortedRect, uint32_t* __restrict__ blockCount,
                                                           int gather)
{
  __shared__ uint32_t s_tmp[4];
  const uint32_t      n     = plan->n;
  const uint32_t      parts = (n + kBinPart - 1) / kBinPart;
  if(blockIdx.x >= parts)
    return;
  const uint32_t* ids = plan->finalSel ? idsY : idsX;
  uint32_t        sum = 0;
#pragma unroll
  for(int i = 0; i < kBinItems; ++i)
  {
    const uint32_t e = blockIdx.x * kBinPart + i * kBinThreads + threadIdx.x;
    if(e < n)
    {
      uint32_t r;
      if(gather)  // CPU-sort mode: nothing has produced sortedRect yet
      {
        r             = rect[ids[e]];
        sortedRect[e] = r;
      }
      else  // GPU sort: the last radix pass wrote the rects in sorted order (fused gather)
        r = sortedRect[e];
      sum += rectTiles(r);
    }
  }
  sum = waveSum(sum);
  if(laneId() == 0)
    s_tmp[threadIdx.x >> 6] = sum;
  __syncthreads();
  if(threadIdx.x == 0)
    blockCount[blockIdx.x] = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

__global__ __launch_bounds__(256) void k_bin_scan(const SortPlan* __restrict__ plan, uint32_t* __restrict__ blockCount,
                                                  FrameCounters* __restrict__ ctr, uint32_t capacity)
{
  __shared__ uint32_t s_tmp[4];
  const uint32_t      n     = plan->n;
  const uint32_t      parts = (n + kBinPart - 1) / kBinPart;
  uint64_t            carry = 0;
  for(uint32_t base = 0; base < parts; base += 256)
  {
    const uint32_t p = base + threadIdx.x;
    const uint32_t v = (p < parts) ? blockCount[p] : 0u;
    uint32_t       chunk;
    const uint32_t ex = blockExclusiveScan256(v, s_tmp, &chunk);
    if(p < parts)
      blockCount[p] = (uint32_t)min(carry + ex, (uint64_t)0xFFFFFFFFull);
    carry += chunk;
  }
  if(threadIdx.x == 0)
  {
    if(carry > capacity)
    {
      atomicOr(&ctr->errorFlags, kErrPairOverflow);
      carry = capacity;
    }
    ctr->pairCount = (uint32_t)carry;
  }
}

__global__ __launch_bounds__(kBinThreads) void k_bin_offsets(const SortPlan* __restrict__ plan, const uint32_t* __restrict__ sortedRect,
                                                             const uint32_t* __restrict__ blockOffset,
                                                             uint32_t* __restrict__ splatOffset, uint32_t* __restrict__ chunkStart,
                                                             uint32_t maxChunks)
{
  __shared__ uint32_t s_tmp[4];
  const uint32_t      n     = plan->n;
  const uint32_t      parts = (n + kBinPart - 1) / kBinPart;
  if(blockIdx.x >= parts)
    return;
  // thread t owns the 8 consecutive sorted entries [base + 8t, base + 8t + 8)
  const uint32_t e0 = blockIdx.x * kBinPart + threadIdx.x * kBinItems;
  uint32_t       cnt[kBinItems], sum = 0;
#pragma unroll
  for(int i = 0; i < kBinItems; ++i)
  {
    cnt[i] = (e0 + i < n) ? rectTiles(sortedRect[e0 + i]) : 0u;
    sum += cnt[i];
  }
  uint32_t total;
  uint32_t run = blockOffset[blockIdx.x] + blockExclusiveScan256(sum, s_tmp, &total);
#pragma unroll
  for(int i = 0; i < kBinItems; ++i)
  {
    if(e0 + i < n)
    {
      splatOffset[e0 + i] = run;
      if(cnt[i])
      {  // every multiple of kChunk inside [run, run+cnt) starts an output chunk inside this splat
        const uint32_t first = (run + kChunk - 1) / kChunk, last = (run + cnt[i] - 1) / kChunk;
        for(uint32_t m = first; m <= last && m < maxChunks; ++m)
          chunkStart[m] = e0 + i;
      }
    }
    run += cnt[i];
  }
}

__global__ __launch_bounds__(kBinThreads) void k_bin_expand(const uint32_t* __restrict__ idsX, const uint32_t* __restrict__ idsY,
                                                            const SortPlan* __restrict__ plan, const FrameCounters* __restrict__ ctr,
                                                            const uint32_t* __restrict__ sortedRect,
                                                            const uint32_t* __restrict__ splatOffset,
                                                            const uint32_t* __restrict__ chunkStart, uint32_t* __restrict__ pairKey,
                                                            uint32_t* __restrict__ pairVal, int binsX)
{
  constexpr int       kWin = 2048;
  __shared__ uint32_t s_off[kWin + 1];
  __shared__ uint32_t s_rect[kWin];
  __shared__ uint32_t s_gid[kWin];
  const uint32_t      D = ctr->pairCount;  // already clamped to the capacity
  const uint32_t      n = plan->n;
  const uint32_t      o0 = blockIdx.x * (uint32_t)kChunk;
  if(o0 >= D)
    return;
  const uint32_t  o1  = min(o0 + (uint32_t)kChunk, D);
  const uint32_t* ids = plan->finalSel ? idsY : idsX;
  const int       t   = threadIdx.x;
  const uint32_t  s0  = chunkStart[blockIdx.x];
  // the splat holding the next chunk's first record may also hold the tail of this chunk
  const uint32_t s1 = (o1 < D) ? chunkStart[blockIdx.x + 1] : (n - 1);
  for(uint32_t w0 = s0; w0 <= s1; w0 += kWin)
  {
    const uint32_t wn = min((uint32_t)kWin, s1 + 1 - w0);
    __syncthreads();
    for(uint32_t i = t; i < wn; i += kBinThreads)
    {
      s_off[i]  = splatOffset[w0 + i];
      s_rect[i] = sortedRect[w0 + i];
      s_gid[i]  = ids[w0 + i];
    }
    if(t == 0)
      s_off[wn] = (w0 + wn < n) ? splatOffset[w0 + wn] : 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lo = max(o0, s_off[0]), hi = min(o1, s_off[wn]);
    for(uint32_t o = lo + t; o < hi; o += kBinThreads)
    {
      // last window entry whose offset <= o
      uint32_t a = 0, b = wn;
      while(b - a > 1)
      {
        const uint32_t mid = (a + b) >> 1;
        if(s_off[mid] <= o)
          a = mid;
        else
          b = mid;
      }
      const uint32_t r  = s_rect[a];
      const uint32_t x0 = r & 255u, y0 = (r >> 8) & 255u, x1 = (r >> 16) & 255u;
      const uint32_t wd = x1 - x0 + 1u;
      const uint32_t k  = o - s_off[a];
      pairKey[o]        = (y0 + k / wd) * (uint32_t)binsX + (x0 + k % wd);
      pairVal[o]        = s_gid[a];
    }
  }
}

// ---- direct binning (<= 256 bins, <= 32 bin columns and <= 32 bin rows) ----------------------------
// With coarse bins the (bin, splat) records never need to exist as sortable pairs: splitting the
// depth-sorted splat list into per-bin lists is ONE stable multi-split, done reduce-then-scan like a
// radix pass whose "digit" is a set (every bin of the splat's rect):
//   k_dbin_count : per chunk of 1024 sorted splats, how many of them touch each bin -> binHist[bin][chunk]
//                  (also re-lays the rects out in sorted order: the only random gather of the stage)
//   k_dbin_scan  : one workgroup per bin: exclusive scan of its row, row total
//   k_dbin_emit  : per chunk, append the ids to every bin list at binBase + rowOffset, in sorted order
// Both per-chunk kernels work on bit matrices: per round of 64 splats, column mask c[bx] = lanes whose rect
// spans column bx (one ballot), row mask r[by] likewise; the lanes hitting bin (bx,by) are c[bx] & r[by].
// Counting is a popcount per bin (lane = bin), emission walks the set bits (lane = bin, appending to its own
// list), so the cost per splat does not depend on how many bins it covers — a splat covering the whole
// screen is one more bit in every mask (per-lane loops over the rect were tried: every round of 64
// depth-neighbours contains some large splat, and the wave pays its trip count).  Gone with the records:
// their 8-byte round trips, the output-partitioned expansion and the whole pair sort.
#ifndef MGS_DB_ROUNDS
#define MGS_DB_ROUNDS 4
#endif
#ifndef MGS_DB_STAGE
#define MGS_DB_STAGE 3072
#endif
constexpr int kDbRounds = MGS_DB_ROUNDS;     // rounds of 64 splats per wave
constexpr int kDbChunk  = 256 * kDbRounds;   // sorted splats per workgroup
constexpr int kDbStage  = MGS_DB_STAGE;      // list entries staged in LDS per chunk so that the appends are coalesced
constexpr int kDbMaxDim = 32;
#ifndef MGS_DB_CNT_MUL
#define MGS_DB_CNT_MUL 1
#endif
constexpr int kDbCntMul = MGS_DB_CNT_MUL;  // chunks counted per workgroup of k_dbin_count
constexpr int kDbMaxSum = 40;  // binsX + binsY of a frame the direct binning takes (each <= 32, product <= 256: 32 + 8)

// v_writelane_b32: a wave-uniform value into ONE lane's register
__device__ __forceinline__ void writeLane(uint32_t& dst, uint32_t value, uint32_t lane)
{
  // one SGPR per VALU op on gfx9: the lane select goes through m0.  Round 6: m0 is an INPUT ("{m0}") that the compiler sets up and
  // tracks itself — rounds 3-5 wrote it inside the asm and listed it as a clobber, which hipcc flags as possibly undefined; this
  // clang has no __builtin_amdgcn_writelane.
  asm volatile("v_writelane_b32 %0, %1, m0" : "+v"(dst) : "s"(value), "{m0}"(lane));
}

// column / row hit masks of one round of 64 rects.  Lane b < binsX ends up holding the mask of column b, lane binsX + b the
// mask of row b (the layout of maskBuf): each ballot is kept by ONE lane through a select — no exec juggling, no branch, one
// LDS write at the end instead of one per ballot.  (A third fewer instructions than lane-0 stores per ballot; the kernel's
// time did not move — 38 us is what 4.2 M random 4-byte gathers of `rect` cost, not what its instructions cost.)
__device__ __forceinline__ uint64_t rectMasks(uint32_t r, bool valid, int binsX, int binsY, uint64_t* s_col, uint64_t* s_row)
{
  const int lane = laneId();
  if(!valid)
    r = 1u;  // x0 = 1 > x1 = 0: no bins
  const uint32_t x0 = r & 255u, y0 = (r >> 8) & 255u, dx = ((r >> 16) & 255u) - x0, dy = (r >> 24) - y0;
  const bool     ok = (int)dx >= 0 && (int)dy >= 0;
  const uint32_t ux = ok ? dx : 0u, nx0 = ok ? x0 : 0xFFFFu;  // rejected: b - nx0 wraps far above ux
  // (v_writelane: the ballot is a scalar pair, lane b's registers take it directly — no compare, no select)
  uint32_t mlo = 0u, mhi = 0u;
  for(int b = 0; b < binsX; ++b)
  {
    const uint64_t m = __ballot((uint32_t)b - nx0 <= ux);
    writeLane(mlo, (uint32_t)m, (uint32_t)b);
    writeLane(mhi, (uint32_t)(m >> 32), (uint32_t)b);
  }
  for(int b = 0; b < binsY; ++b)
  {
    const uint64_t m = __ballot((uint32_t)b - y0 <= dy);
    writeLane(mlo, (uint32_t)m, (uint32_t)(binsX + b));
    writeLane(mhi, (uint32_t)(m >> 32), (uint32_t)(binsX + b));
  }
  const uint64_t mine = ((uint64_t)mhi << 32) | mlo;
  if(lane < binsX)
    s_col[lane] = mine;
  else if(lane < binsX + binsY)
    s_row[lane - binsX] = mine;
  return mine;
}

// the (up to 4) bins lane `lane` is responsible for: b = lane + 64 j
struct LaneBins
{
  int  bx[4], by[4];
  bool on[4];
};
__device__ __forceinline__ LaneBins laneBins(int binsX, int nb)
{
  LaneBins       L;
  const uint32_t inv = (65536u + (uint32_t)binsX - 1u) / (uint32_t)binsX;  // wave-uniform; (b*inv)>>16 == b/binsX for b < 256, binsX <= 32
#pragma unroll
  for(int j = 0; j < 4; ++j)
  {
    const int b = laneId() + 64 * j;
    L.on[j]     = b < nb;
    const int q = (int)(((uint32_t)b * inv) >> 16);
    L.by[j]     = L.on[j] ? q : 0;
    L.bx[j]     = L.on[j] ? b - q * binsX : 0;
  }
  return L;
}

// The same masks by a bit-matrix transpose (round 5, third session): a lane's rectangle IS a row of the (splat x bin-column / bin-row)
// bit matrix — columns x0..x1 in bits [0, binsX), rows y0..y1 in bits [binsX, binsX + binsY) — and the masks are its columns.
// Five butterfly stages (partner's word by DPP for lane distances 1, 2, 8, by ds_swizzle for 4 and 16; v_alignbit + v_bfi)
// transpose the 32 x 32 blocks of both wave halves at once, one ds_bpermute brings the upper half's word to lane b: ~25 vector
// instructions per round of 64 splats, whatever binsX + binsY is, where the ballots cost (binsX + binsY) x 7 (compare, ballot
// into an SGPR pair, two v_writelane through m0) — 119 at 1080p, 224 at 4K — in one dependent chain.  Needs binsX + binsY <= 32.
struct TransposeConst
{
  uint32_t keep[5];  // the bits a lane keeps at stage k (distance 1 << k): those whose index has bit k like the lane's own
  uint32_t rot[5];   // v_alignbit shift that brings the partner's other bits under the complement of keep
};
__device__ __forceinline__ TransposeConst transposeConst()
{
  TransposeConst C;
  const uint32_t lowMask[5] = {0x55555555u, 0x33333333u, 0x0F0F0F0Fu, 0x00FF00FFu, 0x0000FFFFu};
  const uint32_t lane = (uint32_t)laneId();
#pragma unroll
  for(int k = 0; k < 5; ++k)
  {
    const bool up = (lane >> k) & 1u;
    C.keep[k]     = up ? ~lowMask[k] : lowMask[k];
    C.rot[k]      = up ? (1u << k) : 32u - (1u << k);  // rotate right by d (upper lane) / left by d (lower lane)
  }
  return C;
}
template <int N>
__device__ __forceinline__ void transposeWords(uint32_t (&x)[N], const TransposeConst& C)
{
#define MGS_TR_STAGE(K, PARTNER)                                                                        \
  _Pragma("unroll") for(int i = 0; i < N; ++i)                                                          \
  {                                                                                                     \
    const uint32_t p = (uint32_t)(PARTNER);                                                             \
    const uint32_t r = __builtin_amdgcn_alignbit(p, p, C.rot[K]);                                       \
    x[i]             = (C.keep[K] & x[i]) | (~C.keep[K] & r);                                           \
  }
  MGS_TR_STAGE(0, __builtin_amdgcn_update_dpp(0, (int)x[i], 0xB1, 0xF, 0xF, true))   // quad_perm [1,0,3,2]: lane ^ 1
  MGS_TR_STAGE(1, __builtin_amdgcn_update_dpp(0, (int)x[i], 0x4E, 0xF, 0xF, true))   // quad_perm [2,3,0,1]: lane ^ 2
  MGS_TR_STAGE(2, __builtin_amdgcn_ds_swizzle((int)x[i], (4 << 10) | 0x1F))          // BITMASK_PERM xor 4
  MGS_TR_STAGE(3, __builtin_amdgcn_update_dpp(0, (int)x[i], 0x128, 0xF, 0xF, true))  // row_ror:8 = lane ^ 8 within a row of 16
  MGS_TR_STAGE(4, __builtin_amdgcn_ds_swizzle((int)x[i], (16 << 10) | 0x1F))         // BITMASK_PERM xor 16
#undef MGS_TR_STAGE
}
// a lane's row of the bit matrix: bins columns x0..x1 | bin rows y0..y1 << binsX (empty for an invalid / inverted rectangle)
__device__ __forceinline__ uint32_t rectWord(uint32_t r, bool valid, int binsX, uint32_t colAll, uint32_t rowAll)
{
  const uint32_t x0 = r & 255u, y0 = (r >> 8) & 255u, dx = ((r >> 16) & 255u) - x0, dy = (r >> 24) - y0;
  const bool     ok = valid && (int)dx >= 0 && (int)dy >= 0;
  const uint32_t cb = (((2u << (dx & 31u)) - 1u) << (x0 & 31u)) & colAll;
  const uint32_t rb = (((2u << (dy & 31u)) - 1u) << (y0 & 31u)) & rowAll;
  return ok ? (cb | (rb << binsX)) : 0u;
}

__global__ __launch_bounds__(256) void k_dbin_count(const uint32_t* __restrict__ idsX, const uint32_t* __restrict__ idsY,
                                                    const SortPlan* __restrict__ plan, const uint32_t* __restrict__ rect,
                                                    const uint16_t* __restrict__ sortedCode16, uint64_t* __restrict__ maskBuf,
                                                    uint32_t* __restrict__ binHist, uint32_t pStride, int binsX, int binsY, int transpose)
{
  __shared__ uint64_t s_col[4][kDbMaxDim], s_row[4][kDbMaxDim];
  __shared__ uint64_t s_msk[4][kDbRounds][32];  // transpose path: the rounds' masks, columns then rows (binsX + binsY <= 32)
  __shared__ uint32_t s_cnt[4][256];
  const uint32_t n      = plan->n;
  const uint32_t chunks = (n + kDbChunk - 1) / kDbChunk;
  // a workgroup counts kDbCntMul consecutive chunks (k_dbin_emit's unit stays one chunk): the kernel is a chain of dependent round
  // trips per workgroup (plan -> codes + ids -> the escapes' rectangles -> masks), so with every chunk's loads in flight at once
  // the grid passes through the chip in one residency wave instead of two
  const uint32_t chunk0 = blockIdx.x * (uint32_t)kDbCntMul;
  if(chunk0 >= chunks)
    return;
  const int       t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t* ids = plan->finalSel ? idsY : idsX;
  uint32_t        r[kDbCntMul][kDbRounds];
  const uint32_t  ride = plan->reserved[0];
  const uint32_t  eW   = (uint32_t)w * (kDbRounds * 64) + (uint32_t)lane;
  if(ride != 0u)
  {  // the rectangles rode through the key sort as codes above the ids and lie in sorted order (kernels_common.h: rideEncode);
    // only the escapes — splats over more than 2 x 2 bins — are looked up by id.  The ids are requested beside the codes
    // (coalesced, 4 B per splat): an escape's rectangle is two dependent trips away, not three
    const uint32_t escape = (1u << (ride >> 8)) - 1u;
    RideDecoder    dec    = rideDecoder(binsX, binsY);
    asm volatile("" : "+s"(dec.i0), "+s"(dec.i1));  // made once: left alone the compiler repeats the uniform divisions per decode
    uint32_t       v[kDbCntMul][kDbRounds], id[kDbCntMul][kDbRounds];
#pragma unroll
    for(int c = 0; c < kDbCntMul; ++c)
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
      {
        const uint32_t e = min((chunk0 + c) * (uint32_t)kDbChunk + eW + i * 64u, n - 1u);
        v[c][i]          = sortedCode16[e];
        id[c][i]         = ids[e];
      }
#pragma unroll
    for(int c = 0; c < kDbCntMul; ++c)
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
        r[c][i] = (v[c][i] == escape) ? rect[id[c][i]] : rideDecode(v[c][i], dec);
    // how many rectangles did NOT fit a code (MgsFrameOut::escape_count: the only rect[id] stores / gathers of the frame): one
    // fire-and-forget atomic per wave on the frame's statistics lines (sort_plan.h: frameStatSlot, word 3)
    uint32_t esc = 0u;
#pragma unroll
    for(int c = 0; c < kDbCntMul; ++c)
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
        esc += (uint32_t)__popcll(__ballot(v[c][i] == escape && (chunk0 + c) * (uint32_t)kDbChunk + eW + i * 64u < n));
    if(lane == 0 && esc != 0u)
      atomicAdd(const_cast<uint32_t*>(&plan->ghist[0][0]) + 32u * ((blockIdx.x * 4u + (uint32_t)w) & (kFrameStatSlots - 1u)) + 3u, esc);
  }
  else
  {
    uint32_t id[kDbCntMul][kDbRounds];
#pragma unroll
    for(int c = 0; c < kDbCntMul; ++c)
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
        id[c][i] = ids[min((chunk0 + c) * (uint32_t)kDbChunk + eW + i * 64u, n - 1u)];  // clamped, not predicated: all loads in flight
#pragma unroll
    for(int c = 0; c < kDbCntMul; ++c)
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
        r[c][i] = rect[id[c][i]];  // one random gather per splat: 4.2 M of them run at ~120 G/s (L2-miss sectors), 35 us, wherever
                                   // they are issued (moving them into the sort's final pass was measured twice: +36 us there for -16 us here)
  }
  const int            nb = binsX * binsY, S = binsX + binsY;
  const LaneBins       L  = laneBins(binsX, nb);
  const bool           viaTranspose = S <= 32 && transpose != 0;
  const TransposeConst C  = transposeConst();
#pragma unroll
  for(int c = 0; c < kDbCntMul; ++c)
  {
    const uint32_t chunk = chunk0 + (uint32_t)c;
    if(chunk >= chunks)
      break;
    const uint32_t e0     = chunk * (uint32_t)kDbChunk + eW;
    uint32_t       cnt[4] = {0u, 0u, 0u, 0u};
    // the masks of every round are kept for k_dbin_emit (binsX + binsY words of 8 B per round instead of re-reading 64 rects and
    // redoing the masks): maskBuf[((chunk*4 + wave)*rounds + round)*S + {column masks, row masks}]
    uint64_t* mOut = maskBuf + ((size_t)chunk * 4 + w) * kDbRounds * S;
    if(viaTranspose)
    {  // masks by transpose (above): the four rounds' butterflies are independent and interleave
      const uint32_t colAll = (1u << binsX) - 1u, rowAll = (1u << binsY) - 1u;  // (binsX, binsY <= 31 here)
      uint32_t       x[kDbRounds];
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
        x[i] = rectWord(r[c][i], e0 + i * 64u < n, binsX, colAll, rowAll);
      transposeWords(x, C);
      const int up = ((lane + 32) & 63) << 2;
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
      {
        const uint32_t hi   = (uint32_t)__builtin_amdgcn_ds_bpermute(up, (int)x[i]);  // lane b: the word of lane 32 + b
        const uint64_t mine = ((uint64_t)hi << 32) | x[i];
        if(lane < S)
        {
          s_msk[w][i][lane] = mine;
          mOut[i * S + lane] = mine;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for(int i = 0; i < kDbRounds; ++i)
#pragma unroll
        for(int j = 0; j < 4; ++j)
          if(j * 64 < nb)
          {
            const uint64_t m = L.on[j] ? (s_msk[w][i][L.bx[j]] & s_msk[w][i][binsX + L.by[j]]) : 0ull;
            cnt[j] += (uint32_t)__popcll(m);
          }
    }
    else
#pragma unroll
    for(int i = 0; i < kDbRounds; ++i)
    {
      const uint64_t mine = rectMasks(r[c][i], e0 + i * 64u < n, binsX, binsY, s_col[w], s_row[w]);
      __builtin_amdgcn_wave_barrier();
      if(lane < S)
        mOut[i * S + lane] = mine;
#pragma unroll
      for(int j = 0; j < 4; ++j)
        if(j * 64 < nb)
        {
          const uint64_t m = L.on[j] ? (s_col[w][L.bx[j]] & s_row[w][L.by[j]]) : 0ull;
          cnt[j] += (uint32_t)__popcll(m);
        }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for(int j = 0; j < 4; ++j)
      s_cnt[w][lane + 64 * j] = cnt[j];
    __syncthreads();
    if(t < nb)
      binHist[(size_t)t * pStride + chunk] = s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
    if(c + 1 < kDbCntMul)
      __syncthreads();  // s_cnt (and a wave's s_msk rows) are written again for the next chunk
  }
}

__global__ __launch_bounds__(256) void k_dbin_scan(const SortPlan* __restrict__ plan, uint32_t* __restrict__ binHist,
                                                   uint32_t pStride, uint32_t* __restrict__ binTotal)
{
  __shared__ uint32_t s_tmp[4];
  const int      t      = threadIdx.x;
  const uint32_t n      = plan->n;
  const uint32_t chunks = (n + kDbChunk - 1) / kDbChunk;
  uint32_t*      row    = binHist + (size_t)blockIdx.x * pStride;
  uint32_t       carry  = 0;
  // 16 values per thread: a garden-sized frame's ~4 080 chunks are ONE trip of loads, one block scan, one trip of stores (8 per
  // thread were two dependent rounds of that: the kernel is its latency)
  constexpr int kPer = 16;
  for(uint32_t base = 0; base < chunks; base += 256 * kPer)
  {
    const uint32_t p0 = base + t * kPer;
    uint32_t       v[kPer], sum = 0;
#pragma unroll
    for(int i = 0; i < kPer; ++i)
    {
      v[i] = (p0 + i < chunks) ? row[p0 + i] : 0u;
      sum += v[i];
    }
    uint32_t chunk;
    uint32_t run = carry + blockExclusiveScan256(sum, s_tmp, &chunk);
#pragma unroll
    for(int i = 0; i < kPer; ++i)
    {
      if(p0 + i < chunks)
        row[p0 + i] = run;
      run += v[i];
    }
    carry += chunk;
  }
  if(t == 0)
    binTotal[blockIdx.x] = carry;
}

#ifdef MGS_DB_TRACE  // debug build (tools/db_trace.py): per-workgroup wall-clock stamps (100 MHz) of k_dbin_emit's phases
__device__ uint64_t* g_dbTrace = nullptr;
#define MGS_DB_STAMP(i) if(threadIdx.x == 0) dbt[i] = wall_clock64();
#else
#define MGS_DB_STAMP(i)
#endif
__global__ __launch_bounds__(256) void k_dbin_emit(const uint32_t* __restrict__ idsX, const uint32_t* __restrict__ idsY,
                                                   const SortPlan* __restrict__ plan, const uint64_t* __restrict__ maskBuf,
                                                   const uint32_t* __restrict__ binHist, uint32_t pStride,
                                                   const uint32_t* __restrict__ binTotal, uint32_t* __restrict__ binList,
                                                   uint2* __restrict__ ranges, FrameCounters* __restrict__ ctr,
                                                   uint32_t capacity, int binsX, int binsY, uint32_t* __restrict__ binOrder,
                                                   const uint16_t* __restrict__ sortedCode16, uint32_t* __restrict__ binCost)
{
  // LDS diet (round 5): this kernel's residency is set by its LDS — 32 KB were 5 workgroups per CU, and 3 KB more cost a fifth of
  // them and 9 us (seen by accident) — so: column and row masks share one row of binsX + binsY <= 40 words per (wave, round)
  // (8 -> 5 KB), a staged entry is a 16-bit position + an 8-bit bin in two arrays (12 -> 9 KB), the bin ranking borrows the stage.
  __shared__ uint64_t s_msk[4][kDbRounds][kDbMaxSum];  // masks of every round: columns [0, binsX), rows [binsX, binsX + binsY)
  __shared__ uint32_t s_cnt[4][256];  // per-wave counts, then per-wave write cursors
  __shared__ uint32_t s_ids[kDbChunk];
  __shared__ __attribute__((aligned(16))) uint16_t s_spos[kDbStage];  // staged entry: position inside the chunk ...
  __shared__ uint8_t  s_sbin[kDbStage];                              // ... and its bin
  __shared__ uint32_t s_gdst[256], s_loc[256];
  __shared__ uint32_t s_tmp[4], s_tmp64[2];
  const uint32_t n      = plan->n;
  const uint32_t chunks = (n + kDbChunk - 1) / kDbChunk;
  if(chunks == 0u && blockIdx.x == 0u && (int)threadIdx.x < binsX * binsY)
    ranges[threadIdx.x] = make_uint2(0u, 0u);  // nothing sorted: every list is empty (nobody else writes the ranges)
  if(blockIdx.x >= chunks)
    return;
#ifdef MGS_DB_TRACE
  __shared__ uint64_t dbt[8];
  MGS_DB_STAMP(0)
#endif
  // nearest splats (the end of the list) are the largest: start their chunks first
  const uint32_t  chunk = chunks - 1u - blockIdx.x;
  const int       t = threadIdx.x, lane = laneId(), w = t >> 6;
  const uint32_t* ids = plan->finalSel ? idsY : idsX;
  const uint32_t  wbase = (uint32_t)w * (kDbRounds * 64);
  const uint32_t  e0    = chunk * (uint32_t)kDbChunk + wbase + (uint32_t)lane;
  const int       nb    = binsX * binsY;
  const LaneBins  L     = laneBins(binsX, nb);
  uint32_t        cnt[4] = {0u, 0u, 0u, 0u};
  // where every bin's list starts = exclusive prefix of the bins' totals: the same for every chunk, so one wave does it on the
  // side (four bins per lane, wave scans, no barrier of its own) instead of three block-wide scans per workgroup
  uint32_t bt[4] = {0u, 0u, 0u, 0u};
  if(w == 3)
  {
#pragma unroll
    for(int i = 0; i < 4; ++i)
      bt[i] = (4 * lane + i < nb) ? binTotal[4 * lane + i] : 0u;
  }
  // the splats' own rectangles, where they rode through the key sort as codes (k_dbin_count has the story): lane == splat for the
  // walk below
  const uint32_t ride = plan->reserved[0];
  uint32_t       code[kDbRounds];
#pragma unroll
  for(int i = 0; i < kDbRounds; ++i)
    code[i] = ride != 0u ? (uint32_t)sortedCode16[min(e0 + i * 64u, n - 1u)] : 0u;
  {
    const int       S   = binsX + binsY;
    const uint64_t* mIn = maskBuf + ((size_t)chunk * 4 + w) * kDbRounds * S;
    uint64_t        mk[kDbRounds];
#pragma unroll
    for(int i = 0; i < kDbRounds; ++i)
    {
      s_ids[wbase + i * 64 + lane] = ids[min(e0 + i * 64u, n - 1u)];
      mk[i]                        = (lane < S) ? mIn[i * S + lane] : 0ull;
    }
#pragma unroll
    for(int i = 0; i < kDbRounds; ++i)
      if(lane < S)
      {
        s_msk[w][i][lane] = mk[i];  // (the masks arrive in this order: maskBuf holds columns, then rows)
      }
  }
  __builtin_amdgcn_wave_barrier();
  // pop[j]: the populations of bin lane + 64 j's mask in the wave's rounds, 8 bits each (<= 64) — the cursors advance by them below
  uint32_t pop[4] = {0u, 0u, 0u, 0u};
  static_assert(kDbRounds <= 4, "pop[] packs one byte per round");
#pragma unroll
  for(int i = 0; i < kDbRounds; ++i)
#pragma unroll
    for(int j = 0; j < 4; ++j)
      if(j * 64 < nb)
      {
        const uint64_t m = L.on[j] ? (s_msk[w][i][L.bx[j]] & s_msk[w][i][binsX + L.by[j]]) : 0ull;
        const uint32_t c = (uint32_t)__popcll(m);
        cnt[j] += c;
        pop[j] |= c << (8 * i);
      }
#pragma unroll
  for(int j = 0; j < 4; ++j)
    s_cnt[w][lane + 64 * j] = cnt[j];
  __syncthreads();
  MGS_DB_STAMP(1)

  // thread t == bin t: where this chunk's run starts in the bin's list, and in the LDS stage
  const uint32_t c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t], c3 = s_cnt[3][t];
  const uint32_t tot   = (t < nb) ? c0 + c1 + c2 + c3 : 0u;
  uint32_t       P;
  if(w == 3)
  {
    const uint32_t sum4 = bt[0] + bt[1] + bt[2] + bt[3];
    uint32_t       run  = waveInclusiveScan(sum4) - sum4;
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
      s_gdst[4 * lane + i] = run;  // bin base (consumed below, behind the block scan's barriers)
      run += bt[i];
    }
    // the 64-bit total without 64-bit shuffles: the halves summed separately
    const uint32_t dLo = waveSum((bt[0] & 0xFFFFu) + (bt[1] & 0xFFFFu) + (bt[2] & 0xFFFFu) + (bt[3] & 0xFFFFu));
    const uint32_t dHi = waveSum((bt[0] >> 16) + (bt[1] >> 16) + (bt[2] >> 16) + (bt[3] >> 16));
    if(lane == 0)
    {
      s_tmp64[0] = dLo;
      s_tmp64[1] = dHi;
    }
  }
  const uint32_t local   = blockExclusiveScan256(tot, s_tmp, &P);  // (two barriers: the bases above are visible behind them)
  const uint32_t binBase = s_gdst[t];
  const uint64_t D64     = ((uint64_t)s_tmp64[1] << 16) + s_tmp64[0];
  const bool     wrapped = D64 > 0xFFFFFFFFull;  // bin bases are meaningless: emit nothing, report the overflow
  const bool     staged  = P <= (uint32_t)kDbStage;
  const uint32_t gdst    = binBase + ((t < nb) ? binHist[(size_t)t * pStride + chunk] : 0u);  // (requested here, not at the kernel's head: 256 strided loads beside the ids and masks cost 1.5 us — measured)
  __syncthreads();  // everybody has read its base: s_gdst is overwritten
  s_gdst[t] = gdst;
  s_loc[t]  = local;
  {
    const uint32_t start = staged ? local : gdst;
    s_cnt[0][t] = start;
    s_cnt[1][t] = start + c0;
    s_cnt[2][t] = start + c0 + c1;
    s_cnt[3][t] = start + c0 + c1 + c2;
  }
  if(chunk == 0)
  {
    // longest list first: the compositor hands its workgroups out in this bin order, so the regions with the most
    // to blend start early instead of forming the kernel's tail (binOrder[rank] = bin; ties by bin index)
    // The order: by how long the bin's slowest region took in the PREVIOUS frame of this context (k_composite leaves it in
    // binCost; consumed and cleared here) — the regions that never saturate are the long ones, and they are the same from one
    // frame of a sequence to the next; scheduling only, the frame does not depend on it.  Without a history (first frame, the
    // frame before was not composited by k_composite): longest list first.
    uint32_t*      s_tot = reinterpret_cast<uint32_t*>(s_spos);  // [256] (the stage is filled only after this block)
    const uint32_t btot = (t < nb) ? binTotal[t] : 0u;
    uint32_t       cost = 0u;
    if(binCost != nullptr && t < nb)
    {
      cost       = binCost[t];
      binCost[t] = 0u;
    }
    const bool     history = __syncthreads_or(cost != 0u) != 0;
    const uint32_t sortKey = history ? cost : btot;
    s_tot[t] = sortKey;
    __syncthreads();
    if(t < nb)
    {
      uint32_t rank = 0;
      for(int u = 0; u < nb; ++u)
        rank += (s_tot[u] > sortKey || (s_tot[u] == sortKey && u < t)) ? 1u : 0u;
      binOrder[rank] = (uint32_t)t;
    }
    if(t == 0)
      binOrder[256] = 1u;  // valid
    if(t < nb)
      ranges[t] = wrapped ? make_uint2(0u, 0u)
                          : make_uint2(min(binBase, capacity), (uint32_t)min((uint64_t)binBase + btot, (uint64_t)capacity));
    if(t == 0)
    {
      ctr->pairCount = (uint32_t)min(D64, (uint64_t)capacity);
      // (round 6) ... and beside the compositor's statistics (sort_plan.h: frameStatSlot, slot 0 word 4), so that ONE small copy
      // tells the host how much of their lists the regions scan: the adaptive bin size's input (mgs_api.hip: BinPolicy)
      const_cast<uint32_t*>(&plan->ghist[0][0])[4] = (uint32_t)min(D64, (uint64_t)capacity);
      if(D64 > capacity)
        atomicOr(&ctr->errorFlags, kErrPairOverflow);
    }
  }
  __syncthreads();
  MGS_DB_STAMP(2)
  if(wrapped)
    return;

  if(staged && ride != 0u)
  {
    // The rectangles are known per splat: most splats cover 1-4 bins (the coded shapes), so LANE == SPLAT places its few entries
    // directly — position = the bin's cursor + the splats before it in this round's mask of the bin —, and only the splats with
    // larger rectangles (escape code) are found by LANE == BIN walking the bits of its mask.  (Walking every bit that way, below,
    // keeps 2-3 % of the lanes busy: the trip count of a round is the population of its densest bin.)  Cursors live in LDS,
    // advanced by the bin lanes once per round, between two wave barriers.  (Cursor-free — every entry summing the bin's
    // population over the wave's earlier rounds itself — was measured: no faster, the bin lanes run anyway because nearly every
    // wave holds an escape.)
    const uint32_t escape = (1u << (ride >> 8)) - 1u;
    // (round 5, third session: the kernel is bound by VALU issue — 14 M instructions, 26 of its 37 us — so the placement was put on
    //  a diet: a coded splat's two column and two row masks are read once instead of per bin, its rank in a bin's mask is
    //  v_mbcnt_lo / hi instead of and + popcount on both halves, the bin lanes keep their cursors in registers and advance them by
    //  the populations the counting above already found, and a round without escapes does not touch its bin masks again)
    uint32_t run[4];
#pragma unroll
    for(int j = 0; j < 4; ++j)
      run[j] = s_cnt[w][lane + 64 * j];
    RideDecoder dec = rideDecoder(binsX, binsY);
    asm volatile("" : "+s"(dec.i0), "+s"(dec.i1));  // (made once, see k_dbin_count)
    const uint32_t xLast = (uint32_t)binsX - 1u, yLast = (uint32_t)(binsX + binsY) - 1u;
#pragma unroll
    for(int i = 0; i < kDbRounds; ++i)
    {
      const bool     valid = e0 + (uint32_t)i * 64u < n;
      const bool     coded = valid && code[i] != escape;
      const uint32_t idx   = wbase + (uint32_t)i * 64u + (uint32_t)lane;
      if(coded)
      {
        const uint32_t r  = rideDecode(code[i], dec);
        const uint32_t x0 = r & 255u, y0 = (r >> 8) & 255u, dx = ((r >> 16) & 255u) - x0, dy = (r >> 24) - y0;
        const uint64_t cm[2] = {s_msk[w][i][x0], s_msk[w][i][min(x0 + 1u, xLast)]};
        const uint64_t rm[2] = {s_msk[w][i][(uint32_t)binsX + y0], s_msk[w][i][min((uint32_t)binsX + y0 + 1u, yLast)]};
        const uint32_t b0    = y0 * (uint32_t)binsX + x0;
#pragma unroll
        for(uint32_t ky = 0; ky < 2u; ++ky)
#pragma unroll
          for(uint32_t kx = 0; kx < 2u; ++kx)
            if(kx <= dx && ky <= dy)
            {
              const uint32_t b  = b0 + ky * (uint32_t)binsX + kx;
              const uint64_t m  = cm[kx] & rm[ky];
              const uint32_t at = s_cnt[w][b] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
              s_spos[at]        = (uint16_t)idx;
              s_sbin[at]        = (uint8_t)b;
            }
      }
      const uint64_t escM = __ballot(valid && !coded);
      __builtin_amdgcn_wave_barrier();
      // (The escapes one after the other in a SCALAR loop — bin lane b takes escape e iff bit e of its mask is set, one-hot and
      //  below-mask as scalar pairs, no per-lane ctz / 64-bit shifts — was measured: bin 64.6 -> 67.4 us, 4K 91.8 -> 94.5: a round
      //  holds more escapes than its fullest bin takes trips.)
#pragma unroll
      for(int j = 0; j < 4; ++j)
        if(j * 64 < nb)
        {
          if(escM != 0ull)  // wave-uniform
          {
            const uint64_t m    = L.on[j] ? (s_msk[w][i][L.bx[j]] & s_msk[w][i][binsX + L.by[j]]) : 0ull;
            uint64_t       me   = m & escM;
            const uint8_t  btag = (uint8_t)(lane + 64 * j);
            while(__ballot(me != 0ull) != 0ull)
              if(me != 0ull)
              {
                const uint32_t bit = (uint32_t)__builtin_ctzll(me);
                const uint32_t at  = run[j] + (uint32_t)__popcll(m & ((1ull << bit) - 1ull));
                s_spos[at]         = (uint16_t)(wbase + (uint32_t)i * 64u + bit);
                s_sbin[at]         = btag;
                me &= me - 1ull;
              }
          }
          run[j] += (pop[j] >> (8 * i)) & 255u;
          if(L.on[j] && i + 1 < kDbRounds)
            s_cnt[w][lane + 64 * j] = run[j];
        }
      __builtin_amdgcn_wave_barrier();
    }
  }
  else
  // lane == bin: walk the set bits of its mask (low word, then high word), appending to its own list
#pragma unroll
  for(int j = 0; j < 4; ++j)
    if(j * 64 < nb)
    {
      uint32_t       run  = s_cnt[w][lane + 64 * j];
      const uint8_t  btag = (uint8_t)(lane + 64 * j);
      for(int i = 0; i < kDbRounds; ++i)
      {
        const uint64_t m = L.on[j] ? (s_msk[w][i][L.bx[j]] & s_msk[w][i][binsX + L.by[j]]) : 0ull;
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
          uint32_t       mh  = h ? (uint32_t)(m >> 32) : (uint32_t)m;
          const uint32_t pos0 = wbase + (uint32_t)i * 64u + (uint32_t)h * 32u;
          if(staged)
          {  // no memory read in the loop: the entry is (bin, position), the id is looked up at copy-out
            while(__ballot(mh != 0u) != 0ull)
              if(mh != 0u)
              {
                s_spos[run]   = (uint16_t)(pos0 + (uint32_t)__builtin_ctz(mh));
                s_sbin[run++] = btag;
                mh &= mh - 1u;
              }
          }
          else
          {
            while(__ballot(mh != 0u) != 0ull)
              if(mh != 0u)
              {
                const uint32_t id = s_ids[pos0 + (uint32_t)__builtin_ctz(mh)];
                mh &= mh - 1u;
                if(run < capacity)
                  binList[run] = id;
                ++run;
              }
          }
        }
      }
    }
  if(!staged)
    return;
  __syncthreads();
  MGS_DB_STAMP(3)
  for(uint32_t i = t; i < P; i += 256)
  {
    const uint32_t b   = s_sbin[i];
    const uint32_t dst = s_gdst[b] + (i - s_loc[b]);
    if(dst < capacity)
      binList[dst] = s_ids[s_spos[i]];
  }
#ifdef MGS_DB_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MGS_DB_STAMP(4)
  if(t == 0 && g_dbTrace)
  {
    uint64_t* o = g_dbTrace + (size_t)blockIdx.x * 8;
    for(int i = 0; i < 5; ++i) o[i] = dbt[i];
    o[5] = P;
  }
#endif
}

// ---- tile ranges over the tile-sorted pair list ------------------------------------------------
// 4 keys per thread (one 16-byte load) + the two neighbours.
__global__ void k_tile_ranges(const uint32_t* __restrict__ keyX, const uint32_t* __restrict__ keyY,
                              const SortPlan* __restrict__ plan, uint2* __restrict__ ranges)
{
  const uint32_t  n    = plan->n;
  const uint32_t* keys = plan->finalSel ? keyY : keyX;
  const uint32_t  n4   = (n + 3u) >> 2;
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x)
  {
    const uint32_t j = i << 2;
    uint32_t       k[6];
    if(j + 4 <= n)
    {
      const uint4 v = *reinterpret_cast<const uint4*>(keys + j);
      k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w;
    }
    else
    {
#pragma unroll
      for(int q = 0; q < 4; ++q)
        k[1 + q] = (j + q < n) ? keys[j + q] : 0xFFFFFFFFu;
    }
    k[0] = (j > 0) ? keys[j - 1] : 0xFFFFFFFFu;
    k[5] = (j + 4 < n) ? keys[j + 4] : 0xFFFFFFFFu;
#pragma unroll
    for(int q = 0; q < 4; ++q)
    {
      if(j + q < n)
      {
        if(k[q] != k[q + 1] || (j + q) == 0)
          ranges[k[q + 1]].x = j + q;
        if(k[q + 2] != k[q + 1] || (j + q) == n - 1)
          ranges[k[q + 1]].y = j + q + 1;
      }
    }
  }
}

// ---- compositing ------------------------------------------------------------------------------
// One workgroup per 32x16-pixel region (two 16x16 tiles side by side), one wave per 16x8 quarter, TWO
// pixels per lane (x and x+8): the per-fragment arithmetic then runs on gfx950's packed-fp32 pipe
// (v_pk_fma_f32 & co, two pixels per instruction) with the splat's parameters broadcast through op_sel.
// Lists exist per BIN (a block of tiles, e.g. 256x128 px), not per tile: duplicating every splat into every
// 16x16 tile it touches costs ~13 records per splat and made the (tile,id) sort the most expensive stage
// (profiles/r1_a/b).  Instead each workgroup walks its bin's depth-ordered list nearest-first and does
// the fine culling on chip:
//   stage A  every thread fetches 2 list entries (id -> the 32-byte record: centre, p1, p2, opacity, fp16 extents),
//            tests the footprint against the workgroup's region, and the survivors are compacted
//            IN ORDER (ballot + mbcnt) into an LDS batch, their records as they are;
//   finish   one thread per STAGED record (dense lanes, at the head of the shading loop) turns it into the blend walk's operands:
//            axes scaled by sqrt(log2 e) (the exponential is then a bare v_exp_f32), centre terms around the region centre,
//            log2 of the opacity, the fragment rule's cutoff term, the mask of the quarters its ellipse can reach;
//   shading  colour + SH of the staged records (deferred from the projection: a sixth of the sorted splats are ever staged);
//   stage B  each wave walks the batch, skips records that miss its quarter, and blends front-to-back:
//            q=(d.p1)^2+(d.p2)^2 (== A/2, frag.slang:236), alpha = 2^(log2 a - q), discard q>4 and alpha<=1/255 (one cutoff per
//            record, applied as a packed fma with the clamp modifier).
// A wave retires when all 128 pixels have T < 1e-4; the workgroup stops fetching when all 4 have.  In MGS_ALPHA_SUM mode, where the
// reference's additive alpha must see every fragment, a saturated wave goes on SUMMING alpha (a walk of its own: two LDS quads,
// seven packed fmas, two v_exp per record), and the batches staged after all four are saturated are staged and walked as a
// quadratic in the pixel offset.
#ifndef MGS_CMP_ENTRIES
#define MGS_CMP_ENTRIES 2
#endif
constexpr int kCmpEntries = MGS_CMP_ENTRIES;                  // list entries per thread per stage-A round (sweep with shading in the kernel: 1/2/3/4/6 -> 0.137/0.134/0.140/0.142/0.157 ms; 4K 0.374/0.379/-/0.42/-)
// (a stage-A round scans 256 * kCmpEntries list entries)
#ifndef MGS_CMP_CAP
#define MGS_CMP_CAP 288
#endif
#ifndef MGS_CMP_GO
#define MGS_CMP_GO 32
#endif
constexpr int kCmpCap     = MGS_CMP_CAP;                // LDS batch capacity (records): 18 KB -> 8 workgroups per CU (sweep: 384/512/768 -> 0.220/0.230/0.275 ms)
constexpr int kCmpGo      = MGS_CMP_GO;                // blend as soon as this many records are staged (<= kCmpCap-256)
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE bit 0: additive alpha (no early-out), bit 1: DISABLE_OPACITY_GAUSSIAN, bit 2: surface side outputs,
// bit 3: stochastic splats (frag.slang:265-290: a fragment is accepted with probability alpha and written opaque; the
// depth test keeps the nearest accepted one == the first accepted one of the nearest-first list); SHF: SH storage format
constexpr bool surf_lds(int mode) { return (mode & 4) != 0; }
constexpr bool sum_walk(int mode) { return (mode & 1) != 0 && (mode & 4) == 0; }  // MGS_ALPHA_SUM without surface outputs
#ifndef MGS_SUM_ENTRIES
#define MGS_SUM_ENTRIES MGS_CMP_ENTRIES
#endif
#ifndef MGS_SUM_CAP
#define MGS_SUM_CAP 448  // (sweep, garden-sized frame: 288/32 2.89 ms, 352/96 2.82, 416/160 2.80, 448/192 2.79 — 25.6 KB, still 6 workgroups
#endif                   //  per CU —, 544/288 2.77, 672/416 and beyond lose more to the residency than the longer batches gain)
#ifndef MGS_SUM_GO
#define MGS_SUM_GO 192
#endif
// MGS_CMP_FOLD: the blend walk's fragment alpha as 2^(log2 a - q) with the fragment rule as one packed fma + clamp (17 instead of
// 23 vector instructions per record and wave)
#ifndef MGS_CMP_FOLD
#define MGS_CMP_FOLD 1
#endif
#ifdef MGS_CMP_EXACT_PRED
#undef MGS_CMP_FOLD
#define MGS_CMP_FOLD 0
#endif
#ifndef MGS_SUM_POLY  // 0: A/B — batches of an all-saturated region are walked in the (s, u) form as well
#define MGS_SUM_POLY 1
#endif
#ifndef MGS_SUM_UNROLL
#define MGS_SUM_UNROLL 2
#endif
#ifndef MGS_CMP_WAVES
#define MGS_CMP_WAVES 6
#endif
template <int MODE, int SHF>
__global__ __launch_bounds__(256, MGS_CMP_WAVES) void k_composite(const CompositeArgs F, const uint2* __restrict__ ranges,
                                                   const uint32_t* __restrict__ valX, const uint32_t* __restrict__ valY,
                                                   const SortPlan* __restrict__ plan, const SplatRec* __restrict__ rec,
                                                   void* __restrict__ outImage, int halfOut, FrameCounters* __restrict__ ctr,
                                                   float* __restrict__ outDepth, uint32_t* __restrict__ outSplatId,
                                                   const FrameArgs* __restrict__ Ap, float4* __restrict__ outNormal)
{
  // batch geometry: the additive-alpha mode (every fragment of a region's whole list is composited) has its own
  constexpr int cEnt = sum_walk(MODE) ? MGS_SUM_ENTRIES : kCmpEntries, cRound = 256 * cEnt;
  constexpr int cCap = sum_walk(MODE) ? MGS_SUM_CAP : kCmpCap, cGo = sum_walk(MODE) ? MGS_SUM_GO : kCmpGo;
  static_assert(cGo <= cCap - 256, "sub-group 0 of a stage-A round must always fit");
  uint32_t statStaged = 0, statScanned = 0;
  const uint64_t costT0 = wall_clock64();  // this region's duration feeds the next frame's bin order (F.binCost)
#ifdef MGS_CMP_TRACE  // debug build (tools/cmp_trace.py): per-workgroup wall-clock stamps, 100 MHz
  const uint64_t traceT0 = wall_clock64();
  uint64_t       traceA = 0, traceS = 0, traceB = 0, traceLast;  // time spent in stage A / shading / blending
  uint32_t       traceIters = 0, traceRounds = 0, traceNeed = 0, traceNeeded = 0;
  __shared__ uint32_t s_traceNeed;
  traceLast = traceT0;
#define MGS_TRACE_PHASE(acc) { const uint64_t now_ = wall_clock64(); acc += now_ - traceLast; traceLast = now_; }
#else
#define MGS_TRACE_PHASE(acc)
#endif
  __shared__ float4   s_a[cCap];  // k1, k2 (centre terms of d.p1, d.p2), fragment cutoff, -
  __shared__ float4   s_b[cCap];  // p1, p2 (scaled by sqrt(log2 e))
  __shared__ float4   s_c[cCap];  // r, g, b, a
  __shared__ uint32_t s_g[cCap];  // global id: what the deferred shading needs
  __shared__ float    s_z[surf_lds(MODE) ? cCap : 1];  // fragCoord.z of the record (surface outputs only)
  __shared__ float4   s_n[surf_lds(MODE) ? cCap : 1];  // world normal of the record (surface outputs only)
  __shared__ uint32_t s_wc[2][cEnt][4];
  __shared__ uint8_t  s_m[cCap];  // which of the 4 quarters (waves) the record's footprint touches
  // MGS_ALPHA_SUM without surface outputs: s_a = (k1, k2, log2 opacity, cutoff term) feeds the saturated waves' two-quad walk
  // (below); the unsaturated walk's fragment cutoff lives here instead
  __shared__ float    s_t[(sum_walk(MODE) && MGS_CMP_FOLD == 0) ? cCap : 1];

  const int t = threadIdx.x, lane = laneId(), w = t >> 6;
  // regions are 32-px columns x 16-px rows ("tile pairs"); binShiftX >= 1, so a pair never straddles two bins.
  // XCD-aware mapping: workgroup b lands on XCD b%8 (observed dispatch rule).  All regions of a bin — which
  // read the same list — go to one XCD (shared L2), and consecutive bins go to different XCDs so that the dense
  // part of the image is spread over all eight (contiguous screen bands per XCD left most of them idle).
  const int colsX    = (F.tilesX + 1) >> 1;
  const int bw       = 1 << (F.binShiftX - 1), bh = 1 << F.binShiftY;  // bin size in regions
  const int binRow0  = F.stripRow0 >> F.binShiftY;
  const int binRows  = ((F.stripRow1 - 1) >> F.binShiftY) - binRow0 + 1;
  const int perBin   = bw * bh;
  const int seq      = (int)(blockIdx.x >> 3);
  int       ord      = (seq / perBin) * 8 + (int)(blockIdx.x & 7);  // bin ordinal
  const int inBin    = seq % perBin;
  const bool ordered = plan->ghist[2][0] != 0u;  // the binning stage ranked the bins by list length (all bins of the frame)
  int cx2, ty;
  if(ordered)
  {
    if(ord >= F.binsX * F.binsY)
      return;
    const int b = (int)plan->ghist[1][ord];
    cx2         = (b % F.binsX) * bw + inBin % bw;
    ty          = (b / F.binsX) * bh + inBin / bw;
  }
  else
  {
    if(ord >= binRows * F.binsX)
      return;
    cx2 = (ord % F.binsX) * bw + inBin % bw;
    ty  = (binRow0 + ord / F.binsX) * bh + inBin / bw;
  }
  if(cx2 >= colsX || ty < F.stripRow0 || ty >= F.stripRow1)
    return;
  const int      tx  = cx2 * 2;  // left 16-px tile of the region
  const int      qx0 = tx * kTilePx + (w & 1) * 16, qy0 = ty * kTilePx + (w >> 1) * 8;
  const int      px = qx0 + (lane & 7), py = qy0 + (lane >> 3);  // second pixel: px + 8
  const v2f      pcx = {(float)px + 0.5f, (float)px + 8.5f};
  const float    pcy = (float)py + 0.5f;
  const float    bcx = (float)(tx * kTilePx) + 16.0f, bcy = (float)(ty * kTilePx) + 8.0f;  // region centre
  const v2f      lx  = {pcx.x - bcx, pcx.y - bcx};  // pixel centre relative to the region centre (exact: half-integers)
  const float    ly  = pcy - bcy;
  const bool     in0 = px < F.width && py < F.height, in1 = px + 8 < F.width && py < F.height;
  constexpr bool early   = (MODE & 1) == 0;
  constexpr bool noGauss = (MODE & 2) != 0;
  constexpr bool surf    = (MODE & 4) != 0;  // FTB side outputs: picked depth + the splat that set it (frag.slang:320-349)
  constexpr bool stoch   = (MODE & 8) != 0;
  constexpr bool sumWalk = sum_walk(MODE);
  constexpr bool fold    = MGS_CMP_FOLD != 0;  // s_a = (k1, k2, log2 a, cutoff term), s_b = (p1x, p2x, p1y, p2y)  // saturated waves only sum alpha (s_a layout: see s_t)
  constexpr float kSumBig = 1073741824.0f;   // 2^30: (log2 alpha - cutoff) * 2^30, clamped to [0, 1], is the fragment's 0 / 1 weight
  // frag.slang:271: seed = xxhash32(uint3(fragCoord.xy, frameSampleId)); the sample id changes every frame: read through
  // the per-frame constants, not the by-value arguments a captured graph freezes
  uint32_t seedPx0 = 0u, seedPx1 = 0u;
  if constexpr(stoch)
  {
    const uint32_t sid = (uint32_t)Ap->f.frameSampleId;
    seedPx0            = rngXxhash32((uint32_t)px, (uint32_t)py, sid);
    seedPx1            = rngXxhash32((uint32_t)px + 8u, (uint32_t)py, sid);
  }
  v2f            pickZ   = {0.0f, 0.0f};
  uint32_t       pickId0 = 0xFFFFFFFFu, pickId1 = 0xFFFFFFFFu;
  v2f            nx = {0.f, 0.f}, ny = {0.f, 0.f}, nz = {0.f, 0.f};  // integrated normal (surface outputs)
  // a saturated pixel (T < 1e-4) takes no further fragments: the result must not depend on WHEN its wave
  // notices (batch boundaries differ between a strip and the full frame, the frames must not)
  const float    tMin    = early ? 1.0e-4f : -1.0f;
  const uint32_t* vals   = plan->finalSel ? valY : valX;
  const int      bin     = (ty >> F.binShiftY) * F.binsX + (tx >> F.binShiftX);
  const uint2    range   = ranges[bin];
  constexpr float kSqrtLog2e = 1.2011224087864498f;   // sqrt(log2 e): exp(-q) == exp2(-(q * log2 e))
  constexpr float kQMax      = 4.0f * 1.4426950408889634f;

  // pixels outside the image start saturated (they are never stored), so "any T >= tMin" is the wave's liveness
  v2f  T = {in0 ? 1.0f : 0.0f, in1 ? 1.0f : 0.0f}, cr = {0.f, 0.f}, cg = {0.f, 0.f}, cb = {0.f, 0.f}, asum = {0.f, 0.f};
  bool waveDone = (__ballot(in0 || in1) == 0ull);
  // MGS_ALPHA_SUM (no early-out: the additive alpha sees every fragment): once every pixel of the wave is saturated (T < 1e-4)
  // the fragments that follow cannot change the colour any more (each weighs < 1e-4, as in the default mode, where the wave
  // retires at this point) — they are only SUMMED: no colour, no transmittance, and once all four waves are there the batch is
  // not shaded either (the SH sum is most of a staged record's cost).  The switch is checked per record, like the retirement.
  bool waveSat = waveDone, allSat = false;  // (a wave without pixels inside the image has nothing to saturate)

  uint32_t hi   = range.y;
  uint32_t fill = 0;  // records currently in the LDS batch
  int      rnd  = 0;
  uint32_t gNext[cEnt];
  bool     prefValid = false;
#pragma unroll
  for(int k = 0; k < cEnt; ++k)
    gNext[k] = 0u;
  for(;;)
  {
    // ---- stage A: scan list entries (nearest first) until enough records are staged or the list ends ----
    // A round looks at up to 1024 entries as 4 sub-groups of 256 (k-th sub-group = entries k*256+t).  The
    // sub-groups are accepted in order while they fit into the batch; since fill < cGo <= cap-256 on
    // entry, sub-group 0 always fits, so every round makes progress and the batch can never overflow.
    // The ids of the NEXT round are fetched before this round's barrier (one dependent trip instead of two).
    while(hi > range.x && fill < (uint32_t)cGo)
    {
#ifdef MGS_CMP_TRACE
      ++traceRounds;
#endif
      const uint32_t avail = hi - range.x;
      uint32_t       g[cEnt];
      float4         a[cEnt], pb[cEnt];  // (cx, cy, ex, ey), (p1, p2)
      float          al[cEnt];                  // opacity
      bool           ok[cEnt];
      uint64_t       bal[cEnt];
#pragma unroll
      for(int k = 0; k < cEnt; ++k)
      {
        const uint32_t e = k * 256u + (uint32_t)t;  // e-th nearest remaining entry
        ok[k]            = e < avail;
        g[k]             = prefValid ? gNext[k] : (ok[k] ? vals[hi - 1u - e] : 0u);
      }
#pragma unroll
      for(int k = 0; k < cEnt; ++k)
      {  // the whole 32-byte record (half a sector): centre + p1 | p2 + opacity + fp16 extents
        const float4* r  = reinterpret_cast<const float4*>(rec + g[k]);
        const float4  r0 = ok[k] ? r[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4  r1 = ok[k] ? r[1] : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xBC00BC00u));  // extents -1: fails the test
        const uint32_t eb = __float_as_uint(r1.w);
        const float2   e  = __half22float2(*reinterpret_cast<const __half2*>(&eb));
        a[k]  = make_float4(r0.x, r0.y, e.x, e.y);
        pb[k] = make_float4(r0.z, r0.w, r1.x, r1.y);
        al[k] = r1.z;
      }
      // speculative prefetch of the next round's ids (valid if this round is consumed completely)
      {
        const uint32_t hiN = hi - min(avail, (uint32_t)cRound);
        const uint32_t avN = hiN - range.x;
#pragma unroll
        for(int k = 0; k < cEnt; ++k)
        {
          const uint32_t e = k * 256u + (uint32_t)t;
          gNext[k]         = (e < avN) ? vals[hiN - 1u - e] : 0u;
        }
      }
#pragma unroll
      for(int k = 0; k < cEnt; ++k)
      {
        // pixel centres of the region span bcx +- 15.5, bcy +- 7.5
        ok[k]  = ok[k] && fabsf(a[k].x - bcx) <= a[k].z + 15.5f && fabsf(a[k].y - bcy) <= a[k].w + 7.5f;
        bal[k] = __ballot(ok[k]);
        if(lane == 0)
          s_wc[rnd & 1][k][w] = (uint32_t)__popcll(bal[k]);
      }
      __syncthreads();
      const uint32_t(*cnt)[4] = s_wc[rnd & 1];
      ++rnd;
      uint32_t base = fill;
      int      used = 0;  // sub-groups consumed
#pragma unroll
      for(int k = 0; k < cEnt; ++k)
      {
        const uint32_t m = cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
        if(used == k && base + m <= (uint32_t)cCap)
        {
          uint32_t wb = 0;
          if(w > 0) wb += cnt[k][0];
          if(w > 1) wb += cnt[k][1];
          if(w > 2) wb += cnt[k][2];
          if(ok[k])
          {
            // the survivor's record goes to its place in the batch as it is (centre + extents | p1, p2 | opacity | id); what the
            // blend walk reads is computed from it by finishRecord below, once per STAGED record and on dense lanes — here,
            // under the survivors' predicate, the arithmetic ran for every round of 64 entries (a quarter of which survive)
            const uint32_t pos = base + wb + lanesBelow(bal[k]);
            s_a[pos]           = a[k];
            s_b[pos]           = pb[k];
            // (the opacity alone, a 4-byte LDS store: a 16-byte one costs the wave 13 cycles of the LDS path however few of its lanes
            //  are survivors — composite -3 us; rgb is the shading phase's)
            reinterpret_cast<float*>(&s_c[pos])[3] = al[k];
            s_g[pos]           = g[k];
          }
          base += m;
          used = k + 1;
        }
      }
      statStaged += base - fill;
      fill                    = base;
      const uint32_t consumed = min(avail, (uint32_t)used * 256u);
      statScanned += consumed;
      prefValid               = (used == cEnt);
      hi -= consumed;
      if(used < cEnt)
        break;  // batch full: blend, then rescan the unconsumed sub-groups
    }
    __syncthreads();
    MGS_TRACE_PHASE(traceA)
    // ---- shading: the SH sum of the staged splats (mesh.slang:243), one thread per record ---------------------
    // Deferred from the projection: only splats that reach an unsaturated region are ever shaded (a quarter of
    // the frustum survivors on the garden-sized bench), and their 192-byte SH records are the bulk of a splat.
    // ---- finish the staged records (dense lanes, one thread per record): the blend walk's operands and the quarter masks ----
    auto finishRecord = [&](uint32_t pos) {
      const float4 ak = s_a[pos], pbk = s_b[pos];  // (cx, cy, ex, ey), (p1, p2)
      const float  alk = s_c[pos].w;
        const float4   sb  = make_float4(pbk.x * kSqrtLog2e, pbk.y * kSqrtLog2e, pbk.z * kSqrtLog2e, pbk.w * kSqrtLog2e);
        // (s, u) = (d.p1, d.p2) with d = pixel - centre, rewritten around the region centre: s = lx*p1x + (ly*p1y + k1),
        // k1 = -((cx - bcx)*p1x + (cy - bcy)*p1y), lx/ly = the pixel relative to the region centre (|.| <= 15.5):
        // two instructions per pixel pair instead of three, and small operands (no 1000-px coordinates in the products)
        const float rx = ak.x - bcx, ry = ak.y - bcy;
        s_a[pos]       = make_float4(-(rx * sb.x + ry * sb.y), -(rx * sb.z + ry * sb.w), 0.0f, 0.0f);  // .z: the fragment cutoff, below
        // (saturated Σα walk: the y terms of (s, u) are ONE packed fma when p1y, p2y are neighbours — as k1, k2 are)
        s_b[pos]           = (sumWalk || fold) ? make_float4(sb.x, sb.z, sb.y, sb.w) : sb;
        // quarter (qx,qy): pixel centres x in [bcx-15.5,bcx-0.5] / [bcx+0.5,bcx+15.5], y in [bcy-7.5,bcy-0.5] / [bcy+0.5,bcy+7.5].
        // Footprint box first, then a bound in the ellipse's own frame: over the quarter (centre m, half
        // extents 7.5 x 3.5) s = d.p1 stays within |s_m| -+ (7.5|p1x| + 3.5|p1y|), likewise u = d.p2, so
        // q >= max(0,|s_m|-rs)^2 + max(0,|u_m|-ru)^2; a quarter whose bound exceeds what the alpha
        // threshold lets through (a 2^-q > 1/255) cannot receive a fragment.  43 % of the wave-level
        // evaluations were empty before this test (corner overlaps of slanted ellipses).
        const bool  xl = ak.x - ak.z <= bcx - 0.5f, xr = ak.x + ak.z >= bcx + 0.5f;
        const bool  yt = ak.y - ak.w <= bcy - 0.5f, yb = ak.y + ak.w >= bcy + 0.5f;
        const float rc = alk;
        // Fragment rule of frag.slang:242-262 — discard A > 8 (q > kQMax here), discard alpha <= 1/255 — as ONE compare per
        // pixel: alpha = a 2^-q > 1/255  <=>  q < log2(255 a).  (Two compares + the saturation test per pixel were a
        // quarter of the blend loop's instructions.)  A fragment whose alpha is within rounding of 1/255 may fall on
        // the other side than with the exp-then-compare order: a contribution of <= 0.4 % of one splat's colour.
        const float qCut = noGauss ? kQMax : fminf(kQMax, __log2f(fmaxf(rc * 255.0f, 1.0f)));
        if constexpr(sumWalk || fold)
        {  // saturated waves: alpha = 2^(l2a - q), kept iff q <= qCut  <=>  l2a - q >= l2a - qCut =: c, evaluated as
           // clamp((l2a - q) * 2^30 - c * 2^30) in {0, 1}
          const float l2a = noGauss ? 0.0f : __log2f(fmaxf(rc, 1.0e-30f));
          const float tb  = (qCut - l2a) * kSumBig;
          if(sumWalk && MGS_SUM_POLY && allSat)
          {  // every wave of the region is saturated: this batch is only ever summed, by the polynomial walk (stage B) —
             // nq = l2a - q as a quadratic in the pixel's offset (lx, ly) from the region centre,
             // q = (lx p1x + ly p1y + k1)^2 + (lx p2x + ly p2y + k2)^2:
             // nq = (nA lx + (nB ly + nD)) lx + ((nC ly + nE) ly + F'), coefficients per record, computed here once
            const float k1 = -(rx * sb.x + ry * sb.y), k2 = -(rx * sb.z + ry * sb.w);
            s_a[pos] = make_float4(-2.0f * (sb.x * sb.y + sb.z * sb.w), -(sb.y * sb.y + sb.w * sb.w),
                                   -2.0f * (k1 * sb.x + k2 * sb.z), -2.0f * (k1 * sb.y + k2 * sb.w));
            s_b[pos] = make_float4(-(sb.x * sb.x + sb.z * sb.z), l2a - (k1 * k1 + k2 * k2), tb, 0.0f);
          }
          else
          {
            s_a[pos].z = l2a;
            s_a[pos].w = tb;
            if constexpr(!fold)
              s_t[pos] = qCut;
          }
        }
        else
        {
          s_a[pos].z = qCut;
          s_a[pos].w = rc;
        }
        const float qLim = qCut * 1.001f + 1e-3f;
        const float rs = 7.5f * fabsf(sb.x) + 3.5f * fabsf(sb.y), ru = 7.5f * fabsf(sb.z) + 3.5f * fabsf(sb.w);
        uint32_t    qm = 0;
#pragma unroll
        for(int qd = 0; qd < 4; ++qd)
        {
          const float mx = bcx + ((qd & 1) ? 8.0f : -8.0f) - ak.x, my = bcy + ((qd & 2) ? 4.0f : -4.0f) - ak.y;
          const float ds = fmaxf(fabsf(mx * sb.x + my * sb.y) - rs, 0.0f), du = fmaxf(fabsf(mx * sb.z + my * sb.w) - ru, 0.0f);
          const bool  box = ((qd & 1) ? xr : xl) && ((qd & 2) ? yb : yt);
          qm |= (box && (ds * ds + du * du <= qLim || F.looseMask)) ? (1u << qd) : 0u;
        }
        //  a record with a non-finite axis or centre takes no part: the folded fragment rule is clamp(fma) * exp2(nq), and a
        // NaN nq would pass the clamp as 0 and then poison colour and T through 0 * NaN — the select it replaced mapped NaN to 0.
        // k1 + k2 is non-finite whenever any of sb, rx, ry is (0 * Inf = NaN included); two instructions per STAGED record.
        if(!(fabsf((rx * sb.x + ry * sb.y) + (rx * sb.z + ry * sb.w)) < 3.0e38f))
          qm = 0u;
        s_m[pos] = (uint8_t)qm;
    };
    for(uint32_t j = t; j < fill; j += 256)
    {
      finishRecord(j);
      if(allSat)
        continue;  // (a region whose four waves are saturated sums alpha only: nothing to shade)
      const uint32_t gid = s_g[j];
      CompositeArgs::Inst I = F.inst[0];
      int                 instIdx = 0;
      if(F.nInstances <= kMaxInlineInstances)
      {
        for(int i = 1; i < F.nInstances; ++i)
          if(gid >= F.inst[i].globalOffset)
          {
            I       = F.inst[i];
            instIdx = i;
          }
      }
      else
      {  // many instances: binary search in the device table (ascending global offsets)
        int lo = 0, hi2 = F.nInstances;
        while(hi2 - lo > 1)
        {
          const int mid = (lo + hi2) >> 1;
          if(gid >= F.instTable[mid].globalOffset)
            lo = mid;
          else
            hi2 = mid;
        }
        I       = F.instTable[lo];
        instIdx = lo;
      }
      const uint32_t li = gid - I.globalOffset;
      // base colour (mesh.slang:162,205-207) and the view direction in model space (mesh.slang:240-241); the
      // camera position of the instance is per-frame data, read through the frame-argument pointer.
      // (Fetching the first part of the SH record together with these loads — it does not depend on the direction —
      // was measured: the extra live registers spill, 0.140 -> 0.179 ms.)
      const float4 col = I.rgba[li];  // fetchColor, dequantised at commit
      const float  cpx = I.centers[3 * (size_t)li], cpy = I.centers[3 * (size_t)li + 1], cpz = I.centers[3 * (size_t)li + 2];
      const float* cam = Ap->inst[instIdx].camModel;
      float        dx = cpx - cam[0], dy = cpy - cam[1], dz = cpz - cam[2];
      const float  dl = rsqrtf(dx * dx + dy * dy + dz * dz);
      dx *= dl;
      dy *= dl;
      dz *= dl;
      float4 c = s_c[j];
      c.x = F.shOnly ? 0.5f : col.x;
      c.y = F.shOnly ? 0.5f : col.y;
      c.z = F.shOnly ? 0.5f : col.z;
      const int deg = (I.sh == nullptr) ? 0 : min(I.shDegree, F.shDegree);
      if(deg > 0)
        addShRadiance<SHF>(I.sh, li, deg, dx, dy, dz, c.x, c.y, c.z);
      s_c[j] = c;
      if constexpr(surf)
      {
        const InstanceConst& IC = Ap->inst[instIdx];
        s_n[j] = splatWorldNormal(Ap->f, IC, li, Ap->f.quantizeNormals != 0);
        // fragCoord.z of the splat's quad: clip.z / clip.w of the centre (mesh.slang:175-178,276-289)
        const float* MV = IC.modelView;
        const float* P  = Ap->f.proj;
        const float  tx = MV[0] * cpx + MV[4] * cpy + MV[8] * cpz + MV[12];
        const float  ty = MV[1] * cpx + MV[5] * cpy + MV[9] * cpz + MV[13];
        const float  tz = MV[2] * cpx + MV[6] * cpy + MV[10] * cpz + MV[14];
        const float  tw = MV[3] * cpx + MV[7] * cpy + MV[11] * cpz + MV[15];
        const float  cz = P[2] * tx + P[6] * ty + P[10] * tz + P[14] * tw;
        const float  cw = P[3] * tx + P[7] * ty + P[11] * tz + P[15] * tw;
        s_z[j]          = cz * (1.0f / cw);
      }
    }
    __syncthreads();
    MGS_TRACE_PHASE(traceS)
    // ---- stage B: blend the batch ------------------------------------------------------------------------
    // 64 records at a time: a ballot over the quarter masks gives this wave's hit set; the hits are walked
    // with scalar bit tricks (no per-record branch) and the per-pixel discards are predicated.
    if(!waveDone)
    {
      // (the wave's quarter bit as a scalar: derived from the thread index it was a vector register the compiler spilled and
      //  re-loaded from scratch at the head of every 64-record chunk)
      const uint32_t wbit = 1u << (uint32_t)__builtin_amdgcn_readfirstlane(w);
      const v2f      sumBig2 = {kSumBig, kSumBig};
      for(uint32_t j0 = 0; j0 < fill; j0 += 64)
      {
        const uint32_t jl   = j0 + (uint32_t)lane;
        const bool     mine = jl < fill && (s_m[jl] & wbit) != 0u;
        uint64_t       hits = __ballot(mine);
        if constexpr(sumWalk)
        {
          if(MGS_SUM_POLY && allSat)
          {  // the batch was staged for the polynomial walk (stage A): 5 packed fmas, one plain one, two v_exp per record
            uint64_t hs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(hits >> 32)) << 32)
                          | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hits);
            auto polyTerm = [&](const float4 a1, const float4 b1, v2f& m, v2f& e) {  // (nB, nC, nD, nE), (nA, F', cutoff term, -)
              const v2f    bc = {a1.x, a1.y}, de = {a1.z, a1.w}, tz = {b1.z, b1.w};
              const v2f    hg = ly * bc + de;          // (nB ly + nD, nC ly + nE)
              const float  g  = hg.y * ly + b1.y;
              const v2f    nq = (lx * b1.x + hg.x) * lx + g;
              asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0] clamp" : "=v"(m) : "v"(nq), "s"(sumBig2), "v"(tz));
              if(noGauss)
                e = (v2f){1.0f, 1.0f};
              else
                e = (v2f){__builtin_amdgcn_exp2f(nq.x), __builtin_amdgcn_exp2f(nq.y)};
            };
#if MGS_SUM_UNROLL >= 2
            // several records per trip, their LDS quads requested together: the walk of ONE wave is a latency chain per record
            // (LDS round trip, four dependent packed fmas, v_exp), and the region with the longest list is the kernel's span
            while(__builtin_popcountll(hs) >= MGS_SUM_UNROLL)
            {
              float4 qa[MGS_SUM_UNROLL], qb[MGS_SUM_UNROLL];
#pragma unroll
              for(int u = 0; u < MGS_SUM_UNROLL; ++u)
              {
                const uint32_t j = j0 + (uint32_t)__builtin_ctzll(hs);
                hs &= hs - 1ull;
                qa[u] = s_a[j];
                qb[u] = s_b[j];
              }
              v2f mm[MGS_SUM_UNROLL], ee[MGS_SUM_UNROLL];
#pragma unroll
              for(int u = 0; u < MGS_SUM_UNROLL; ++u)
                polyTerm(qa[u], qb[u], mm[u], ee[u]);
#pragma unroll
              for(int u = 0; u < MGS_SUM_UNROLL; ++u)
                asum = ee[u] * mm[u] + asum;  // in list order
            }
#endif
            while(hs != 0ull)
            {
              const uint32_t j = j0 + (uint32_t)__builtin_ctzll(hs);
              hs &= hs - 1ull;
              v2f m, e;
              polyTerm(s_a[j], s_b[j], m, e);
              asum = e * m + asum;
            }
            continue;
          }
        }
        // MGS_ALPHA_SUM: a saturated wave walks its hits in a loop of its own (below); the switch happens per record
        while(hits != 0ull && !(sumWalk && waveSat))
        {
          const uint32_t j = j0 + (uint32_t)__builtin_ctzll(hits);
          hits &= hits - 1ull;
#if MGS_CMP_FOLD
          // alpha = a 2^-q as 2^(log2 a - q); the fragment rule of frag.slang:242-262 (discard A > 8, discard alpha <= 1/255: one
          // cutoff per record, stage A) as clamp((log2 a - q) * 2^30 + cutoff term) in {0, 1} — a packed fma instead of two
          // compares and two selects; the y terms of (s, u) packed.  No test of T: a saturated pixel of a live wave keeps taking
          // fragments (each weighs < 1e-4) until the WAVE retires, which is checked per record and depends only on the
          // sequence of records — the same in a strip and in the full frame
          const float4 a1 = s_a[j], b1 = s_b[j];  // (k1, k2, log2 a, cutoff term), (p1x, p2x, p1y, p2y)
          const v2f    k12 = {a1.x, a1.y}, py12 = {b1.z, b1.w}, zw = {a1.z, a1.w};
          const v2f    yt  = ly * py12 + k12;
          const v2f    s1 = lx * b1.x + yt.x, u1 = lx * b1.y + yt.y;
          const v2f    nq = -(u1 * u1) + (-(s1 * s1) + a1.z);  // log2 alpha;  q == (A/2) * log2 e of frag.slang:236
          float4       c1 = s_c[j];
          {  // keep (b, a) a register pair: the whole quad in one ds_read_b128 (4 LDS cycles; the b96 the compiler picks when
             // .w is unused takes 8) and blue broadcast through op_sel instead of a move
            v2f czw = {c1.z, c1.w};
            asm("" : "+v"(czw));
            c1.z = czw.x;
            c1.w = czw.y;
          }
          v2f          ah;
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1] clamp" : "=v"(ah) : "v"(nq), "s"(sumBig2), "v"(zw));
          if(!noGauss)  // frag.slang:248-254
          {
            const v2f e = {__builtin_amdgcn_exp2f(nq.x), __builtin_amdgcn_exp2f(nq.y)};
            ah          = e * ah;
          }
#else
          const float4 a1 = s_a[j], b0 = s_b[j];
          const float4 b1 = sumWalk ? make_float4(b0.x, b0.z, b0.y, b0.w) : b0;  // (p1x, p1y, p2x, p2y)
          const v2f    s1 = lx * b1.x + (ly * b1.y + a1.x), u1 = lx * b1.z + (ly * b1.w + a1.y);
          const v2f    q  = s1 * s1 + u1 * u1;  // == (A/2) * log2 e of frag.slang:236
          const float  qc = sumWalk ? s_t[j] : a1.z;
          const float4 c1 = s_c[j];
          v2f          al = {1.0f, 1.0f};
          if(!noGauss)  // frag.slang:248-254
          {
            const v2f e = {__builtin_amdgcn_exp2f(-q.x), __builtin_amdgcn_exp2f(-q.y)};
            al          = e * c1.w;
          }
          v2f ah;  // frag.slang:242-245,258-262, predicated (cutoff per record, see stage A)
#ifdef MGS_CMP_EXACT_PRED
          ah.x = (q.x <= kQMax && al.x > (1.0f / 255.0f) && T.x >= tMin) ? al.x : 0.0f;
          ah.y = (q.y <= kQMax && al.y > (1.0f / 255.0f) && T.y >= tMin) ? al.y : 0.0f;
#else
          // no test of T: a saturated pixel of a live wave keeps taking fragments (each weighs < 1e-4) until the WAVE retires,
          // which is checked per record and depends only on the sequence of records — the same in a strip and in the full frame
          ah.x = (q.x <= qc) ? al.x : 0.0f;
          ah.y = (q.y <= qc) ? al.y : 0.0f;
#endif
#endif
          if constexpr(stoch)
          {  // frag.slang:272-276: seed = xxhash32(uint3(seed, splatId, primitiveID)); accept iff rand(seed) < opacity.
            // primitiveID = 2 * (index of the splat in its mesh workgroup of 32) + triangle; the quad (-1,-1),(1,-1),(1,1),(-1,1)
            // is split along (-1,-1)-(1,1): triangle 0 = (0,2,1) holds u > v (mesh.slang:158-159,193).  The reference's index
            // comes from an unsorted, atomically compacted list (not reproducible); ours is the global id modulo 32.
            const uint32_t gid = s_g[j], prim = 2u * (gid & 31u);
            uint32_t       h0 = rngXxhash32(seedPx0, gid, prim + (s1.x > u1.x ? 0u : 1u));
            uint32_t       h1 = rngXxhash32(seedPx1, gid, prim + (s1.y > u1.y ? 0u : 1u));
            ah.x = (ah.x > 0.0f && rngRand(h0) < ah.x) ? 1.0f : 0.0f;
            ah.y = (ah.y > 0.0f && rngRand(h1) < ah.y) ? 1.0f : 0.0f;
          }
          const v2f wgt = ah * T;
          cr += wgt * c1.x;
          cg += wgt * c1.y;
          cb += wgt * c1.z;
          if(!early)
            asum += ah;
          T -= wgt;
          if constexpr(surf)
          {  // normal attachment: "under" blend of (normal * opacity, opacity) (gaussian_splatting.cpp:2095-2107)
            const float4 n1 = s_n[j];
            nx += wgt * n1.x;
            ny += wgt * n1.y;
            nz += wgt * n1.z;
            // transmittance *= (1 - opacity); if(depth == 0 && transmittance < threshold) depth = fragCoord.z
            const float    zr  = s_z[j];
            const uint32_t gid = s_g[j];
            if(ah.x > 0.0f && pickZ.x == 0.0f && T.x < F.depthIsoThreshold)
            {
              pickZ.x = zr;
              pickId0 = gid;
            }
            if(ah.y > 0.0f && pickZ.y == 0.0f && T.y < F.depthIsoThreshold)
            {
              pickZ.y = zr;
              pickId1 = gid;
            }
          }
          // checked per record (the compares are the predicate's): a 64-record chunk used to run to its end
          // after the last pixel had saturated
          if(early && __ballot(T.x >= tMin || T.y >= tMin) == 0ull)
          {
            waveDone = true;
#ifdef MGS_CMP_TRACE
            traceNeed = j + 1u;  // this wave needed the batch's records up to here
#endif
            break;
          }
          if(!early && __ballot(T.x >= 1.0e-4f || T.y >= 1.0e-4f) == 0ull)
            waveSat = true;
        }
        if constexpr(sumWalk)
        {
          // Saturated wave in MGS_ALPHA_SUM mode: the fragment only adds its alpha.  (Not with surface outputs: the depth / id
          // pick fires when T crosses depth_iso_threshold, which may lie below 1e-4 — T has to keep falling there, as in the
          // reference, the oracle and the default mode's full path.)  Two LDS quads per record; the opacity is folded into the
          // exponent (nq = log2 a - q, alpha = 2^nq) and the fragment rule (q <= qCut) is one packed fma with the clamp
          // modifier instead of two compares + two selects: seven packed fmas and two v_exp per record and 128 pixels — this
          // walk IS bound by VALU issue.
          // (the hit mask is wave-uniform; said explicitly, or the walk's bit tricks land on the vector unit)
          uint64_t hs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(hits >> 32)) << 32)
                        | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hits);
          // one record's fragments for this lane's two pixels: weight m in {0, 1} (the fragment rule) and alpha e
          auto sumTerm = [&](uint32_t j, v2f& m, v2f& e) {
            const float4 a1 = s_a[j], b1 = s_b[j];  // (k1, k2, log2 a, cutoff term), (p1x, p2x, p1y, p2y)
            const v2f    k12 = {a1.x, a1.y}, py12 = {b1.z, b1.w}, zw = {a1.z, a1.w};
            const v2f    yt  = ly * py12 + k12;
            const v2f    s1 = lx * b1.x + yt.x, u1 = lx * b1.y + yt.y;
            const v2f    nq = -(u1 * u1) + (-(s1 * s1) + a1.z);
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1] clamp" : "=v"(m) : "v"(nq), "s"(sumBig2), "v"(zw));
            if(noGauss)
              e = (v2f){1.0f, 1.0f};
            else
              e = (v2f){__builtin_amdgcn_exp2f(nq.x), __builtin_amdgcn_exp2f(nq.y)};
          };
#if MGS_SUM_UNROLL == 2
          // two records per trip: two independent chains (LDS round trip, six dependent packed fmas, v_exp) in flight per wave;
          // the sums are added in list order, as in the one-record form
          while((hs & (hs - 1ull)) != 0ull)
          {
            const uint32_t ja = j0 + (uint32_t)__builtin_ctzll(hs);
            hs &= hs - 1ull;
            const uint32_t jb = j0 + (uint32_t)__builtin_ctzll(hs);
            hs &= hs - 1ull;
            v2f ma, ea, mb, eb;
            sumTerm(ja, ma, ea);
            sumTerm(jb, mb, eb);
            asum = ea * ma + asum;
            asum = eb * mb + asum;
          }
#endif
          while(hs != 0ull)
          {
            const uint32_t j = j0 + (uint32_t)__builtin_ctzll(hs);
            hs &= hs - 1ull;
            v2f m, e;
            sumTerm(j, m, e);
            asum = e * m + asum;
          }
        }
        if(waveDone)
          break;
      }
    }
#ifdef MGS_CMP_TRACE
    {  // how many of the batch's (shaded) records did the region need?  max over the waves that blended this batch
      if(!waveDone)
        traceNeed = fill;
      if(t == 0) s_traceNeed = 0u;
      __syncthreads();
      if(lane == 0 && traceNeed) atomicMax(&s_traceNeed, traceNeed);
      __syncthreads();
      traceNeeded += s_traceNeed;
      traceNeed = 0u;
    }
    ++traceIters;
#endif
    fill = 0;
    const int allFlag = __syncthreads_and((early ? waveDone : waveSat) ? 1 : 0);
    MGS_TRACE_PHASE(traceB)
    const bool allDone = early && allFlag != 0;
    allSat             = !early && !surf && allFlag != 0;  // (surface outputs: the records' depth and normal stay needed)
    if(allDone || hi <= range.x)
      break;
  }

#ifdef MGS_CMP_TRACE
  if(t == 0 && F.trace)
  {
    uint64_t* o = F.trace + (size_t)blockIdx.x * 10;
    o[0] = traceT0; o[1] = wall_clock64(); o[2] = statScanned; o[3] = statStaged; o[4] = traceIters | ((uint64_t)traceNeeded << 32); o[5] = traceRounds;
    o[6] = traceA; o[7] = traceS; o[8] = traceB; o[9] = ((uint64_t)range.y - range.x) | ((uint64_t)(ty * colsX + cx2) << 32);
  }
#endif
  if(t == 0)
  {  // frame statistics (mgs_frame_stats): two fire-and-forget adds per workgroup
    uint32_t* stat = frameStatSlot(plan, blockIdx.x >> 3);  // (sort_plan.h: one 128-byte line per slot)
    atomicAdd(&stat[0], statStaged);
    atomicAdd(&stat[1], statScanned);
    // the bin's longest region, for the next frame of this context: regions that never saturate take twice as long as the
    // others, and starting them late is the kernel's tail; which ones they are is the same from one frame to the next
    if(F.binCost != nullptr)
      atomicMax(&F.binCost[bin], (uint32_t)(wall_clock64() - costT0) + 1u);
  }
  const v2f aout = early ? (v2f){1.0f - T.x, 1.0f - T.y} : asum;
#pragma unroll
  for(int h = 0; h < 2; ++h)
  {
    if(h ? in1 : in0)
    {
      const float  r = h ? cr.y : cr.x, g = h ? cg.y : cg.x, b = h ? cb.y : cb.x, ao = h ? aout.y : aout.x;
      const size_t o = (size_t)py * (size_t)F.width + (size_t)(px + 8 * h);
      if(halfOut == 2)
      {  // RGBA8 UNORM: clamp, scale, round to nearest
        auto q8 = [](float v) { return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f); };
        reinterpret_cast<uint32_t*>(outImage)[o] = q8(r) | (q8(g) << 8) | (q8(b) << 16) | (q8(ao) << 24);
      }
      else if(halfOut == 1)
      {
        const __half2 lo = __floats2half2_rn(r, g), hi2 = __floats2half2_rn(b, ao);
        uint2         pk;
        pk.x = *reinterpret_cast<const uint32_t*>(&lo);
        pk.y = *reinterpret_cast<const uint32_t*>(&hi2);
        reinterpret_cast<uint2*>(outImage)[o] = pk;
      }
      else
        reinterpret_cast<float4*>(outImage)[o] = make_float4(r, g, b, ao);
      if constexpr(surf)
      {
        outDepth[o]   = h ? pickZ.y : pickZ.x;
        outSplatId[o] = h ? pickId1 : pickId0;
        outNormal[o]  = h ? make_float4(nx.y, ny.y, nz.y, 1.0f - T.y) : make_float4(nx.x, ny.x, nz.x, 1.0f - T.x);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
void launchFrameInit(hipStream_t stream, uint2* ranges, uint32_t nTiles)
{
  const uint32_t blocks = (nTiles + 255u) / 256u < 4u ? 4u : min((nTiles + 255u) / 256u, 256u);
  hipLaunchKernelGGL(k_frame_init, dim3(blocks), dim3(256), 0, stream, ranges, nTiles);
}

void launchBinning(hipStream_t stream, const uint32_t* idsX, const uint32_t* idsY, const SortPlan* planKeys,
                   const uint32_t* rect, uint32_t* blockCount, uint32_t maxBlocks, FrameCounters* ctr, uint32_t* sortedRect,
                   uint32_t* splatOffset, uint32_t* chunkStart, uint32_t* pairKey, uint32_t* pairVal, uint32_t capacity,
                   int binsX, bool gatherRects)
{
  if(maxBlocks == 0)
    return;
  hipLaunchKernelGGL(k_bin_count, dim3(maxBlocks), dim3(kBinThreads), 0, stream, idsX, idsY, planKeys, rect, sortedRect,
                     blockCount, gatherRects ? 1 : 0);
  hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(256), 0, stream, planKeys, blockCount, ctr, capacity);
  const uint32_t chunks = (capacity + kChunk - 1) / kChunk;
  hipLaunchKernelGGL(k_bin_offsets, dim3(maxBlocks), dim3(kBinThreads), 0, stream, planKeys, sortedRect, blockCount,
                     splatOffset, chunkStart, chunks + 1);
  hipLaunchKernelGGL(k_bin_expand, dim3(chunks), dim3(kBinThreads), 0, stream, idsX, idsY, planKeys, ctr, sortedRect,
                     splatOffset, chunkStart, pairKey, pairVal, binsX);
}

bool directBinningSupported(int binsX, int binsY)
{
  return binsX <= kDbMaxDim && binsY <= kDbMaxDim && binsX * binsY <= 256 && binsX + binsY <= kDbMaxSum;
}

void launchDirectBinning(hipStream_t stream, const uint32_t* idsX, const uint32_t* idsY, const SortPlan* planKeys,
                         const uint32_t* rect, const uint16_t* sortedCode16, uint64_t* maskBuf, uint32_t maxSplats, uint32_t* binHist, uint32_t pStride,
                         uint32_t* binTotal, uint32_t* binList, uint2* ranges, FrameCounters* ctr, uint32_t capacity,
                         int binsX, int binsY, uint32_t* binCost)
{
  const uint32_t maxChunks = (maxSplats + kDbChunk - 1) / kDbChunk;
  if(maxChunks == 0)
    return;
  // MGS_DB_TRANSPOSE=0: the rounds' masks by ballots everywhere (A/B switch of the transpose path; the masks are the same)
  static const int kTranspose = [] { const char* e = std::getenv("MGS_DB_TRANSPOSE"); return e ? std::atoi(e) : 1; }();
  hipLaunchKernelGGL(k_dbin_count, dim3((maxChunks + kDbCntMul - 1) / kDbCntMul), dim3(256), 0, stream, idsX, idsY, planKeys, rect, sortedCode16, maskBuf,
                     binHist, pStride, binsX, binsY, kTranspose);
  hipLaunchKernelGGL(k_dbin_scan, dim3(binsX * binsY), dim3(256), 0, stream, planKeys, binHist, pStride, binTotal);
#ifdef MGS_DB_TRACE
  static uint64_t* traceBuf = nullptr;
  const char*      tracePath = std::getenv("MGS_DB_TRACE_FILE");
  if(tracePath)
  {
    if(!traceBuf)
    {
      (void)hipMalloc(&traceBuf, (size_t)maxChunks * 64);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbTrace), &traceBuf, sizeof(traceBuf));
    }
    (void)hipMemsetAsync(traceBuf, 0, (size_t)maxChunks * 64, stream);
  }
#endif
  hipLaunchKernelGGL(k_dbin_emit, dim3(maxChunks), dim3(256), 0, stream, idsX, idsY, planKeys, maskBuf, binHist, pStride,
                     binTotal, binList, ranges, ctr, capacity, binsX, binsY, binTotal + 256, sortedCode16, binCost);
#ifdef MGS_DB_TRACE
  if(tracePath)
  {
    (void)hipStreamSynchronize(stream);
    std::vector<uint64_t> h((size_t)maxChunks * 8);
    (void)hipMemcpy(h.data(), traceBuf, h.size() * 8, hipMemcpyDeviceToHost);
    if(FILE* fp = std::fopen(tracePath, "wb"))
    {
      std::fwrite(h.data(), 8, h.size(), fp);
      std::fclose(fp);
    }
  }
#endif
}

void launchTileRanges(hipStream_t stream, const uint32_t* keyX, const uint32_t* keyY, const SortPlan* planPairs,
                      uint2* ranges)
{
  hipLaunchKernelGGL(k_tile_ranges, dim3(4096), dim3(256), 0, stream, keyX, keyY, planPairs, ranges);
}

void launchComposite(hipStream_t stream, const FrameArgs& A, const uint2* ranges, const uint32_t* valX,
                     const uint32_t* valY, const SortPlan* planPairs, const SplatRec* rec, void* image, int halfOut,
                     int shFormat, FrameCounters* ctr, float* outDepth, uint32_t* outSplatId,
                     const void* instTable, const FrameArgs* dArgs, float4* outNormal, uint32_t* binCost)
{
  const FrameConst& F = A.f;
  if(F.stripRow1 <= F.stripRow0)
    return;
  // all bins of the frame are enumerated (the bin order of the binning stage is over the whole frame; bins outside
  // a strip have empty regions that exit at once)
  const int nBins   = F.binsX * F.binsY;
  const int per     = ((nBins + 7) / 8) * (1 << (F.binShiftX - 1 + F.binShiftY));  // workgroups per XCD
  int mode = (F.alphaMode != 0 ? 1 : 0) | ((F.debugFlags & 4) ? 2 : 0) | (F.surfaceOutputs ? 4 : 0);
  // stochastic splats: opaque writes, so the alpha mode is irrelevant; with the opacity gaussian disabled every fragment is
  // accepted (alpha 1) and the plain path already yields the nearest fragment
  if(F.stochastic && !(F.debugFlags & 4))
    mode = 8 | (F.surfaceOutputs ? 4 : 0);
  CompositeArgs C;
  std::memset(&C, 0, sizeof(C));
  C.width = F.width; C.height = F.height; C.tilesX = F.tilesX;
  C.binShiftX = F.binShiftX; C.binShiftY = F.binShiftY; C.binsX = F.binsX; C.binsY = F.binsY;
  C.stripRow0 = F.stripRow0; C.stripRow1 = F.stripRow1;
  C.nInstances = F.nInstances; C.shDegree = F.shDegree; C.looseMask = (F.debugFlags & 256) ? 1 : 0;
  C.depthIsoThreshold = F.depthIsoThreshold;
  C.shOnly     = (F.debugFlags & 2) ? 1 : 0;
  C.binCost    = binCost;
  for(int i = 0; i < F.nInstances && i < kMaxInlineInstances; ++i)
  {
    C.inst[i].sh           = A.inst[i].sh;
    C.inst[i].rgba         = reinterpret_cast<const float4*>(A.inst[i].rgbaF32);
    C.inst[i].centers      = A.inst[i].centers;
    C.inst[i].globalOffset = A.inst[i].globalOffset;
    C.inst[i].shDegree     = A.inst[i].shDegree;
  }
  C.instTable = static_cast<const CompositeArgs::Inst*>(instTable);
#ifdef MGS_CMP_TRACE
  static uint64_t* traceBuf = nullptr;
  const char*      tracePath = std::getenv("MGS_CMP_TRACE_FILE");
  const size_t     traceN = (size_t)per * 8 * 10;
  if(tracePath)
  {
    if(!traceBuf)
      (void)hipMalloc(&traceBuf, (size_t)1 << 24);
    (void)hipMemsetAsync(traceBuf, 0, traceN * 8, stream);
    C.trace = traceBuf;
  }
#endif
#define MGS_CMP(M, S)                                                                                                  \
  hipLaunchKernelGGL((k_composite<M, S>), dim3(per * 8), dim3(256), 0, stream, C, ranges, valX, valY, planPairs, rec, image, \
                     halfOut, ctr, outDepth, outSplatId, dArgs, outNormal)
#define MGS_CMP_FMT(M)                                                                                                 \
  switch(shFormat)                                                                                                     \
  {                                                                                                                    \
    case 0: MGS_CMP(M, 0); break;                                                                                      \
    case 1: MGS_CMP(M, 1); break;                                                                                      \
    default: MGS_CMP(M, 2); break;                                                                                     \
  }
  switch(mode)
  {
    case 0: MGS_CMP_FMT(0); break;
    case 1: MGS_CMP_FMT(1); break;
    case 2: MGS_CMP_FMT(2); break;
    case 3: MGS_CMP_FMT(3); break;
    case 4: MGS_CMP_FMT(4); break;
    case 5: MGS_CMP_FMT(5); break;
    case 6: MGS_CMP_FMT(6); break;
    case 7: MGS_CMP_FMT(7); break;
    case 8: MGS_CMP_FMT(8); break;
    default: MGS_CMP_FMT(12); break;
  }
#undef MGS_CMP_FMT
#undef MGS_CMP
#ifdef MGS_CMP_TRACE
  if(tracePath)
  {
    (void)hipStreamSynchronize(stream);
    std::vector<uint64_t> h(traceN);
    (void)hipMemcpy(h.data(), traceBuf, traceN * 8, hipMemcpyDeviceToHost);
    if(FILE* fp = std::fopen(tracePath, "wb"))
    {
      std::fwrite(h.data(), 8, traceN, fp);
      std::fclose(fp);
    }
  }
#endif
}

}  // namespace mgs
