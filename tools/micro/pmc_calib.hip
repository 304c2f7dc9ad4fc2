// pmc_calib.hip — what do FETCH_SIZE / WRITE_SIZE count per byte actually moved, for the access widths the frame's kernels use?
// MI355X_MICROARCH.md calibrates the gfx950 correction (FETCH_SIZE tallies 16 B/lane streaming reads at half their bytes) for
// 16-byte lanes only; k_project reads 4-byte lanes at a 12-byte pitch (centres), 4 / 8 / 16-byte planar lanes (opacity, cov6)
// and writes 8 / 16-byte lanes.  Each kernel here streams a buffer of known size once with ONE lane width; run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (separate passes)
// and tools/micro/pmc_calib.py divides the counted KB by the bytes moved.  Buffers are 1 GiB (>> the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while(0)

template <typename T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ src, uint32_t* __restrict__ sink, size_t n)
{
  uint32_t acc = 0;
  for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    const T v = src[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    for(int k = 0; k < (int)(sizeof(T) / 4); ++k)
      acc ^= w[k];
  }
  if(acc == 0x12345678u)
    sink[0] = acc;
}
// three 4-byte loads per lane at a 12-byte pitch: the centres' pattern
__global__ __launch_bounds__(256) void k_read12(const float* __restrict__ src, uint32_t* __restrict__ sink, size_t n)
{
  float acc = 0.f;
  for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += src[3 * i] + src[3 * i + 1] + src[3 * i + 2];
  if(acc == 12345.678f)
    sink[0] = 1u;
}
template <typename T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ dst, size_t n, uint32_t seed)
{
  for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    T         v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    for(int k = 0; k < (int)(sizeof(T) / 4); ++k)
      w[k] = seed + (uint32_t)i + k;
    dst[i] = v;
  }
}
// 32-byte records, two 16-byte lanes per record (k_project's record stores)
__global__ __launch_bounds__(256) void k_write32(uint4* __restrict__ dst, size_t n, uint32_t seed)
{
  for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (size_t)gridDim.x * 256)
    dst[i] = make_uint4(seed, (uint32_t)i, seed, (uint32_t)i);
}

int main()
{
  const size_t bytes = (size_t)1 << 30;
  void *a = nullptr, *sink = nullptr;
  CHK(hipMalloc(&a, bytes + 64));
  CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(a, 1, bytes));
  CHK(hipDeviceSynchronize());
  const dim3 g(8192), b(256);
  for(int rep = 0; rep < 3; ++rep)
  {
    hipLaunchKernelGGL(k_read<uint32_t>, g, b, 0, 0, (const uint32_t*)a, (uint32_t*)sink, bytes / 4);
    hipLaunchKernelGGL(k_read<uint2>, g, b, 0, 0, (const uint2*)a, (uint32_t*)sink, bytes / 8);
    hipLaunchKernelGGL(k_read12, g, b, 0, 0, (const float*)a, (uint32_t*)sink, bytes / 12);
    hipLaunchKernelGGL(k_read<uint4>, g, b, 0, 0, (const uint4*)a, (uint32_t*)sink, bytes / 16);
    hipLaunchKernelGGL(k_write<uint32_t>, g, b, 0, 0, (uint32_t*)a, bytes / 4, (uint32_t)rep);
    hipLaunchKernelGGL(k_write<uint2>, g, b, 0, 0, (uint2*)a, bytes / 8, (uint32_t)rep);
    hipLaunchKernelGGL(k_write<uint4>, g, b, 0, 0, (uint4*)a, bytes / 16, (uint32_t)rep);
    hipLaunchKernelGGL(k_write32, g, b, 0, 0, (uint4*)a, bytes / 32, (uint32_t)rep);
    CHK(hipDeviceSynchronize());
  }
  std::printf("pmc_calib: every kernel moved %zu bytes per launch\n", bytes);
  return 0;
}
