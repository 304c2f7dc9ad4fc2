// microbenchmark: what resolving a partition's digit prefixes INSIDE a radix pass costs on MI355X at the key sort's size, by
// scheme.  A pass has P partitions x 256 digits; partition p needs, per digit d, the number of keys with digit d in the
// partitions before it.  Reduce-then-scan pays a histogram kernel + a scan kernel + two kernel boundaries for it (~17 us per
// pass on the garden-sized frame); the schemes below pay a hand-off inside the scatter kernel instead:
//   serial   one predecessor per dependent load (the classic decoupled look-back, one thread per digit)        [round 2]
//   window W thread d issues W independent loads of predecessors p-1 .. p-W, folds them in order, repeats until it
//            meets an inclusive prefix (onesweep's look-back with a W-deep window per thread, no extra lanes)
//   grouped  two levels, fan-in 32: member m of a group sums the aggregates of the m members before it (one batch of
//            independent loads); the group's last member publishes the group total, resolves the group's base by a
//            windowed look-back over the GROUP totals and publishes the inclusive group prefix; every member then needs one
//            more word.  Latency ~3 round trips whatever P is; no chain over partitions.
// Every workgroup takes its partition from a ticket, so a predecessor has always started (no deadlock under any dispatch
// order, also when P exceeds what is resident).  Status words carry flag + value in ONE 32-bit word (relaxed agent-scope
// atomics = sc1 accesses: the data is the flag, nothing to order).  Spins are bounded; a timeout is reported, not hung on.
// Between publishing its counts and needing the prefix a real pass ranks its keys; `work` emulates that gap, `stream`
// makes every workgroup read its 32 KB of keys/values first, as the pass does.  Every output word is checked on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)
constexpr uint32_t kAgg = 1u << 30, kInc = 2u << 30, kMask = (1u << 30) - 1u;
constexpr int      kG   = 32;        // group size of the grouped scheme
constexpr uint32_t kSpinMax = 1u << 22;

__device__ __forceinline__ uint32_t ldAgent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stAgent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t localCount(uint32_t p, uint32_t d) { return (p * 7u + d * 13u) % 31u + 1u; }

// scheme: 0 none (baseline: ticket + stream + work only), 1 serial, 2 window, 3 grouped
template <int SCHEME, int W>
__global__ __launch_bounds__(256) void k_pass(uint32_t* __restrict__ status, uint32_t* __restrict__ gstat, uint32_t* __restrict__ ticket,
                                              uint32_t* __restrict__ out, uint32_t* __restrict__ err, const uint4* __restrict__ keys,
                                              int work, int stream)
{
  __shared__ uint32_t s_p;
  if(threadIdx.x == 0)
    s_p = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t p = s_p, d = threadIdx.x;
  uint32_t       c = localCount(p, d);
  uint32_t       acc = 0;
  if(stream)
  {  // the pass's own loads: 4096 keys + 4096 values per workgroup
    uint4 v[8];
#pragma unroll
    for(int i = 0; i < 8; ++i)
      v[i] = keys[((size_t)p * 8 + i) * 256 + d];
#pragma unroll
    for(int i = 0; i < 8; ++i)
      acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if(SCHEME != 0)
    stAgent(&status[(size_t)p * 256 + d], kAgg | c);
  for(int i = 0; i < work; ++i)  // the ranking between "counts known" and "prefix needed"
    acc = acc * 1664525u + 1013904223u;
  uint32_t sum = 0, spins = 0;
  bool     bad = false;
  if constexpr(SCHEME == 1)
  {
    for(int q = (int)p - 1; q >= 0;)
    {
      const uint32_t v = ldAgent(&status[(size_t)q * 256 + d]);
      if((v >> 30) == 0u)
      {
        if(++spins > kSpinMax) { bad = true; break; }
        continue;
      }
      sum += v & kMask;
      if((v >> 30) == 2u)
        break;
      --q;
    }
    stAgent(&status[(size_t)p * 256 + d], kInc | (sum + c));
  }
  else if constexpr(SCHEME == 2)
  {
    int q = (int)p - 1;
    while(q >= 0 && !bad)
    {
      uint32_t v[W];
#pragma unroll
      for(int j = 0; j < W; ++j)
        v[j] = (q - j >= 0) ? ldAgent(&status[(size_t)(q - j) * 256 + d]) : kInc;  // before partition 0: inclusive 0
      bool done = false;
#pragma unroll
      for(int j = 0; j < W; ++j)
      {
        if(done)
          continue;
        if((v[j] >> 30) == 0u)
        {  // not there yet: resume the window at this predecessor
          done = true;
          q -= j;
          if(++spins > kSpinMax) bad = true;
          continue;
        }
        sum += v[j] & kMask;
        if((v[j] >> 30) == 2u)
        {
          done = true;
          q    = -1;
        }
        else if(j == W - 1)
        {
          done = true;
          q -= W;
        }
      }
    }
    stAgent(&status[(size_t)p * 256 + d], kInc | (sum + c));
  }
  else if constexpr(SCHEME == 3)
  {
    const uint32_t g = p / kG, m = p % kG;
    // level 1: the aggregates of the members before me in my group, all loads of a round in flight together
    uint32_t intra = 0;
    {
      uint32_t need = (m == 0) ? 0u : (0xFFFFFFFFu >> (32 - m));  // bit j: member j still missing
      while(need && !bad)
      {
        uint32_t v[kG - 1];
#pragma unroll
        for(int j = 0; j < kG - 1; ++j)
          v[j] = ((need >> j) & 1u) ? ldAgent(&status[(size_t)(g * kG + j) * 256 + d]) : 0u;
#pragma unroll
        for(int j = 0; j < kG - 1; ++j)
          if(((need >> j) & 1u) && (v[j] >> 30) != 0u)
          {
            intra += v[j] & kMask;
            need &= ~(1u << j);
          }
        if(need && ++spins > kSpinMax) bad = true;
      }
    }
    if(m == kG - 1)
    {  // group total -> aggregate; then the base of the group by a windowed look-back over the groups, -> inclusive
      const uint32_t total = intra + c;
      stAgent(&gstat[(size_t)g * 256 + d], kAgg | total);
      uint32_t base = 0;
      int      q    = (int)g - 1;
      while(q >= 0 && !bad)
      {
        uint32_t v[W];
#pragma unroll
        for(int j = 0; j < W; ++j)
          v[j] = (q - j >= 0) ? ldAgent(&gstat[(size_t)(q - j) * 256 + d]) : kInc;
        bool done = false;
#pragma unroll
        for(int j = 0; j < W; ++j)
        {
          if(done)
            continue;
          if((v[j] >> 30) == 0u)
          {
            done = true;
            q -= j;
            if(++spins > kSpinMax) bad = true;
            continue;
          }
          base += v[j] & kMask;
          if((v[j] >> 30) == 2u)
          {
            done = true;
            q    = -1;
          }
          else if(j == W - 1)
          {
            done = true;
            q -= W;
          }
        }
      }
      stAgent(&gstat[(size_t)g * 256 + d], kInc | (base + total));
      sum = base + intra;
    }
    else
    {  // everybody else: the inclusive prefix of the previous group is one word
      uint32_t base = 0;
      if(g > 0)
      {
        uint32_t v;
        while(((v = ldAgent(&gstat[(size_t)(g - 1) * 256 + d])) >> 30) != 2u)
          if(++spins > kSpinMax) { bad = true; break; }
        base = v & kMask;
      }
      sum = base + intra;
    }
  }
  if(bad)
    atomicOr(err, 1u);
  out[(size_t)p * 256 + d] = sum + (acc == 0x12345u ? 1u : 0u);
}

template <int SCHEME, int W>
static int run(const char* name, uint32_t P, int work, int stream, uint32_t* st, uint32_t* gs, uint32_t* tk, uint32_t* out, uint32_t* err, const uint4* keys)
{
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  float total = 0, best = 1e9f;
  const int reps = 22;
  for(int it = 0; it < reps; ++it)
  {
    CHK(hipMemsetAsync(st, 0, (size_t)P * 256 * 4)); CHK(hipMemsetAsync(gs, 0, (size_t)(P / kG + 1) * 256 * 4));
    CHK(hipMemsetAsync(tk, 0, 4)); CHK(hipMemsetAsync(err, 0, 4));
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((k_pass<SCHEME, W>), dim3(P), dim3(256), 0, 0, st, gs, tk, out, err, keys, work, stream);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    if(it >= 2) { total += ms; best = ms < best ? ms : best; }
  }
  std::vector<uint32_t> h((size_t)P * 256);
  uint32_t herr = 0;
  CHK(hipMemcpy(h.data(), out, (size_t)P * 256 * 4, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  size_t wrong = 0;
  if(SCHEME != 0)
  {
    std::vector<uint32_t> run(256, 0);
    for(uint32_t p = 0; p < P; ++p)
      for(uint32_t d = 0; d < 256; ++d)
      {
        if(h[(size_t)p * 256 + d] != run[d]) ++wrong;
        run[d] += (p * 7u + d * 13u) % 31u + 1u;
      }
  }
  std::printf("%-12s P %5u work %5d stream %d: mean %7.2f us  best %7.2f us  %s%s\n", name, P, work, stream, total * 1000.f / (reps - 2), best * 1000.f,
              wrong ? "WRONG " : (SCHEME ? "checked" : "-"), herr ? " SPIN-TIMEOUT" : "");
  return (wrong || herr) ? 2 : 0;
}

int main(int argc, char** argv)
{
  const uint32_t Pmax = 12288;
  uint32_t *st, *gs, *tk, *out, *err;
  uint4*    keys;
  CHK(hipMalloc(&st, (size_t)Pmax * 256 * 4)); CHK(hipMalloc(&gs, (size_t)(Pmax / kG + 1) * 256 * 4)); CHK(hipMalloc(&tk, 4));
  CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&out, (size_t)Pmax * 256 * 4)); CHK(hipMalloc(&keys, (size_t)Pmax * 8 * 256 * 16));
  CHK(hipMemset(keys, 1, (size_t)Pmax * 8 * 256 * 16));
  int rc = 0;
  for(uint32_t P : {512u, 1024u, 2048u, 11392u})
    for(int stream : {0, 1})
      for(int work : {0, 2000})
      {
        rc |= run<0, 1>("none", P, work, stream, st, gs, tk, out, err, keys);
        if(P <= 1024 && work == 0 && stream == 0)
          rc |= run<1, 1>("serial", P, work, stream, st, gs, tk, out, err, keys);
        rc |= run<2, 8>("window8", P, work, stream, st, gs, tk, out, err, keys);
        rc |= run<2, 16>("window16", P, work, stream, st, gs, tk, out, err, keys);
        rc |= run<2, 32>("window32", P, work, stream, st, gs, tk, out, err, keys);
        rc |= run<3, 16>("grouped/16", P, work, stream, st, gs, tk, out, err, keys);
        rc |= run<3, 32>("grouped/32", P, work, stream, st, gs, tk, out, err, keys);
      }
  return rc;
}
