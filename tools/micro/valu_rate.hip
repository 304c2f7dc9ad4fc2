// valu_rate.hip — how many cycles does a wave64 VALU instruction occupy its SIMD on gfx950?  (round 5: calibrates what
// "VALU-busy" derived from SQ_ACTIVE_INST_VALU means for k_project / k_composite.)
// Every thread runs ITER iterations of 8 independent chains of one instruction kind; 1024 workgroups x 256 threads at
// 8 waves / SIMD keep every SIMD saturated.  Prints wave-instructions per SIMD per microsecond and, at the clock the chip
// reports, cycles per instruction.   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256, 8) void k(float* out, int iters, float seed)
{
  float a[8];
  v2f   p[8];
#pragma unroll
  for(int i = 0; i < 8; ++i)
  {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = v2f{a[i], a[i] + 1.f};
  }
  const float m = 1.0001f, c = 1e-6f;
  const v2f   pm = {m, m}, pc = {c, c};
  for(int it = 0; it < iters; ++it)
  {
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
      if(KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if(KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
      if(KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if(KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if(KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
      if(KIND == 5) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p[i]) : "v"(pm));
      if(KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
      if(KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if(KIND == 8) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
      if(KIND == 9) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : "v"(m) : "vcc");  // counted as ONE below: the pair
      if(KIND == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(p[i]) : "v"(pm), "v"(pc));
      if(KIND == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if(KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
      if(KIND == 13) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
      if(KIND == 14) asm volatile("v_cmp_le_f32 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, 0, %0, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
    }
  }
  float s = 0.f;
#pragma unroll
  for(int i = 0; i < 8; ++i)
    s += a[i] + p[i].x + p[i].y;
  if(s == 12345.678f)
    out[0] = s;
}
template <int KIND>
void run(const char* name, float* d, double ghz)
{
  const int iters = 4096, wgs = 2048;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, 64, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)wgs * 4 * iters * 8;  // wave-instructions
  const double perSimdPerUs = inst / 1024.0 / (ms * 1e3);
  std::printf("%-26s %8.3f ms  %7.1f wave-instr / SIMD / us  -> %.2f cycles per instruction at %.2f GHz\n", name, ms, perSimdPerUs,
              ghz * 1e3 / perSimdPerUs, ghz);
}
int main()
{
  float* d;
  (void)hipMalloc(&d, 4096);
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const double ghz = pr.clockRate * 1e-6;
  std::printf("%s, %d CUs, clockRate %.2f GHz\n", pr.gcnArchName, pr.multiProcessorCount, ghz);
  run<0>("v_fma_f32", d, ghz);
  run<1>("v_pk_fma_f32", d, ghz);
  run<6>("v_pk_mul_f32", d, ghz);
  run<3>("v_add_u32", d, ghz);
  run<4>("v_cndmask_b32", d, ghz);
  run<5>("v_lshl_add_u64", d, ghz);
  run<2>("v_exp_f32", d, ghz);
  run<7>("v_rcp_f32", d, ghz);
  run<8>("v_cmp_le_f32 vcc", d, ghz);
  run<9>("cmp+cndmask vcc (pair)", d, ghz);
  run<14>("cmp+cndmask sgpr (pair)", d, ghz);
  run<10>("v_pk_fma_f32 clamp", d, ghz);
  run<11>("v_max_f32", d, ghz);
  run<12>("v_cndmask e64 sgpr", d, ghz);
  run<13>("v_mov_b32", d, ghz);
  return 0;
}
