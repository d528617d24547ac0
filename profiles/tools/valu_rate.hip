#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2048
template <int MODE> __global__ void k(double* out, double a0, float f0) {
  double a[8]; float f[8];
  for (int i = 0; i < 8; i++) { a[i] = a0 + i + threadIdx.x; f[i] = f0 + i + threadIdx.x; }
  double c = a0 * 3; float cf = f0 * 3;
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_min_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
      if (MODE == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f[i]) : "v"(f[i]), "v"(cf));
      if (MODE == 2) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(f[i]) : "v"(f[i]), "v"(cf), "v"(f[(i + 1) & 7]));
      if (MODE == 3) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(c), "v"(a[i]));
      if (MODE == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(f[i]), "v"(cf));
      if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
      if (MODE == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
      if (MODE == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[i]), "v"(cf) : "vcc");
    }
  }
  double s = 0; for (int i = 0; i < 8; i++) s += a[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  double* out; hipMalloc(&out, 256 * 4 * 1024 * 64 * sizeof(double));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd; const int blocks = 256 * (256 * waves_per_simd / threads);
  k<MODE><<<blocks, threads>>>(out, 1.5, 2.5f); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, threads>>>(out, 1.5, 2.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = 8.0 * N_IT;
  printf("%-14s waves/SIMD %d: %.3f ms -> %.2f ns per instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 4}) {
    if (w == 1) { run<0>("v_min_f64", 1); run<1>("v_max_f32", 1); run<2>("v_med3_f32", 1); run<3>("v_fma_f64", 1); run<4>("v_cndmask_b32", 1); run<5>("v_pk_add_f32", 1); run<6>("v_cmp_lt_f64", 1); run<7>("v_cmp_lt_f32", 1); }
    else { run<0>("v_min_f64", 4); run<1>("v_max_f32", 4); run<2>("v_med3_f32", 4); run<3>("v_fma_f64", 4); run<4>("v_cndmask_b32", 4); run<5>("v_pk_add_f32", 4); run<6>("v_cmp_lt_f64", 4); run<7>("v_cmp_lt_f32", 4); }
  }
  return 0;
}
