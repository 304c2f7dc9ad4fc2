#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float* out)
{
  const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
  h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
  // row i: a one at (half = i / 8 % 2, v = i % 8)
  if(((row >> 3) & 1) == half) a[row & 7] = (_Float16)1.0f;
  for(int v = 0; v < 8; ++v) b[v] = (_Float16)(float)(100 * half + v + 1 + 1000 * (lane & 31 ? 1 : 0));
  v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if((lane & 31) == 0)
    for(int r = 0; r < 16; ++r)
      out[8 * (r / 4) + 4 * half + (r % 4)] = acc[r];
}
int main()
{
  float o[32], *d; hipMalloc(&d, sizeof(o));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
  for(int i = 0; i < 16; ++i) printf("row %d (half %d, v %d): %g  expect %d\n", i, (i >> 3) & 1, i & 7, o[i], 100 * ((i >> 3) & 1) + (i & 7) + 1);
  return 0;
}
