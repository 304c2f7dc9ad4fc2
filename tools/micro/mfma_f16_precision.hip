// How exactly does v_mfma_f32_32x32x16_f16 add its 16 products?  (k_alpha_tail feeds it cancelling terms.)
// Row 0 of A holds (big, -big, 2^-s, 0...) against B = 1: the exact sum is 2^-s; then (big, -2^-s): big - 2^-s.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfp tools/micro/mfma_f16_precision.hip && /tmp/mfp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float* in, float* out)
{
  const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
  h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for(int v = 0; v < 8; ++v)
  {
    a[v] = (_Float16)in[(row * 2 + half) * 8 + v];
    b[v] = (_Float16)1.0f;
  }
  v16f acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  // D[i][j]: lane holds j = lane & 31, i = 8 (r / 4) + 4 half + r % 4
  if((lane & 31) == 0)
    for(int r = 0; r < 16; ++r)
      out[8 * (r / 4) + 4 * half + (r % 4)] = acc[r];
}
int main()
{
  float h[32 * 16] = {0}, o[32];
  // rows 0..15: (2048, -2048, 2^-s) s = row;  rows 16..31: (2048, -2^-(s-16)) ; all in the first half's k
  for(int i = 0; i < 16; ++i) { h[i * 16 + 0] = 2048.f; h[i * 16 + 1] = -2048.f; h[i * 16 + 2] = ldexpf(1.f, -i); }
  for(int i = 16; i < 32; ++i) { h[i * 16 + 0] = 2048.f; h[i * 16 + 8] = -ldexpf(1.f, -(i - 16)); }  // second term in the OTHER half-wave's k
  float *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for(int i = 0; i < 16; ++i) printf("2048 - 2048 + 2^-%d = %.10g (exact %.10g)\n", i, o[i], ldexp(1.0, -i));
  for(int i = 16; i < 32; ++i) printf("2048 - 2^-%d = %.10f (exact %.10f)\n", i - 16, o[i], 2048.0 - ldexp(1.0, -(i - 16)));
  return 0;
}
