// microbenchmark: device-scope atomicAdd throughput on MI355X for the access pattern a fused "next pass histogram" would have:
// n = 4.2 M adds spread over [digit][partition] counters (256 x 1019), issued by n/4096 workgroups whose keys go to 256
// runs -> each workgroup touches ~4096 distinct-ish counters.  Prints us per launch for: no atomics (baseline: the address
// computation only), atomics without return, with LDS pre-aggregation per (digit, partition) pair inside the workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

__global__ __launch_bounds__(256) void k_atomic(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pos, uint32_t* __restrict__ hist,
                                                uint32_t n, uint32_t stride, int mode)
{
  const uint32_t base = blockIdx.x * 4096u;
  uint32_t       acc = 0;
#pragma unroll
  for(int i = 0; i < 16; ++i)
  {
    const uint32_t e = base + i * 256u + threadIdx.x;
    if(e < n)
    {
      const uint32_t k = keys[e], p = pos[e];
      const uint32_t a = ((k >> 8) & 255u) * stride + (p >> 12);
      if(mode == 0)
        acc += a;
      else
        atomicAdd(&hist[a], 1u);
    }
  }
  if(mode == 0 && acc == 0x12345u)
    hist[0] = acc;
}

int main()
{
  const uint32_t n = 4174912, parts = (n + 4095) / 4096, stride = 1024;
  std::vector<uint32_t> hk(n), hp(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  // pos: what a scatter produces — the keys of workgroup w go to 256 runs, run d at (d * n/256 + w * 16 + j)
  for(uint32_t e = 0; e < n; ++e)
  {
    hk[e] = rnd();
    const uint32_t w = e / 4096, j = e % 4096, d = j / 16;
    hp[e] = (uint32_t)(((uint64_t)d * n) / 256 + (uint64_t)w * 16 + (j % 16));
    if(hp[e] >= n) hp[e] = n - 1;
  }
  uint32_t *dk, *dp, *dh;
  CHK(hipMalloc(&dk, n * 4)); CHK(hipMalloc(&dp, n * 4)); CHK(hipMalloc(&dh, 256 * stride * 4));
  CHK(hipMemcpy(dk, hk.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dp, hp.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for(int mode = 0; mode < 2; ++mode)
  {
    CHK(hipMemset(dh, 0, 256 * stride * 4));
    for(int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_atomic, dim3(parts), dim3(256), 0, 0, dk, dp, dh, n, stride, mode);
    CHK(hipEventRecord(a));
    for(int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_atomic, dim3(parts), dim3(256), 0, 0, dk, dp, dh, n, stride, mode);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    std::printf("mode %d (%s): %.2f us per launch, %u adds over %u x %u counters\n", mode, mode ? "atomicAdd" : "loads only", ms * 1000.f / 20, n, 256u, parts);
  }
  return 0;
}
