// cu_hog.hip — test helper (NOT part of libmgs): a kernel that does nothing but hold compute-unit resources for a while, so that
// a test can make the key sort's single-kernel passes run OVERSUBSCRIBED: with LDS-heavy workgroups of this kernel resident on
// every CU, only part of a pass's workgroups find a slot at a time and the rest start as the hogs retire — the situation in
// which the look-back of k_os_pass leans on "a 1-D grid is dispatched in index order" (k_osort.hip header).  The test asserts a
// correct order (or kErrSpinTimeout reported as an error), never a hang.
// Built in-tree by __graft_entry__.build() into tests/helpers/libcuhog.so; loaded with ctypes.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_cu_hog(uint32_t spinUsBase, uint32_t spinUsStep, uint32_t* sink)
{
  extern __shared__ uint32_t s_hold[];
  s_hold[threadIdx.x] = threadIdx.x;  // the dynamic LDS must be "used" to be allocated
  __syncthreads();
  // staggered retirement: workgroup b holds its slot for base + (b % 16) * step microseconds (100 MHz wall clock)
  const uint64_t ticks = (uint64_t)(spinUsBase + (blockIdx.x & 15u) * spinUsStep) * 100ull;
  const uint64_t t0    = wall_clock64();
  uint32_t       acc   = s_hold[(threadIdx.x * 7u) % blockDim.x];
  while(wall_clock64() - t0 < ticks)
    acc = acc * 1664525u + 1013904223u;
  if(acc == 0x12345u && sink != nullptr)
    sink[0] = acc;
}

extern "C" int cu_hog_launch(void* stream, uint32_t blocks, uint32_t threads, uint32_t ldsBytes, uint32_t spinUsBase, uint32_t spinUsStep)
{
  if(ldsBytes > 64u * 1024u)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_cu_hog), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
  hipLaunchKernelGGL(k_cu_hog, dim3(blocks), dim3(threads), ldsBytes, (hipStream_t)stream, spinUsBase, spinUsStep, (uint32_t*)nullptr);
  return (int)hipGetLastError();
}
