// Micro-benchmark: issue rate of candidate VALU ops on gfx950 (cycles per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/op_rates.hip -o tools/ubench/op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define REPS 8      // 64 instructions per loop trip: loop overhead < 5 %
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a[CHAINS];
  unsigned long long u[CHAINS];
  unsigned w[CHAINS];
  __attribute__((ext_vector_type(4))) unsigned q4[2];
  for (int c = 0; c < CHAINS; c++) { a[c] = seed * (threadIdx.x + c + 1); u[c] = (unsigned long long)(threadIdx.x * 977 + c) << 20; w[c] = threadIdx.x * 31 + c; }
  double x = seed * 1.0001;
  unsigned long long ux = 12345ull + threadIdx.x;
  unsigned wx = 777u + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < REPS; rep++)
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(x));
      if (OP == 1) asm volatile("v_max_f64 %0, %0, |%1|" : "+v"(a[c]) : "v"(x));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(x));
      if (OP == 3) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(w[c]) : "v"(a[c]), "v"(x), "v"(wx) : "vcc");
      if (OP == 4) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(u[c]), "v"(ux) : "vcc");
      if (OP == 5) asm volatile("v_max_u32 %0, %0, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 6) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 7) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[c]), "v"(x) : "vcc");
      if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(x));
      if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[c]) : "v"(wx));
      if (OP == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(w[c]) : "v"(wx));
      if (OP == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 18) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(w[c]) : "v"(wx));
      if (OP == 19) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(w[c]) : "v"(wx));
      if (OP == 20) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(w[c]) : "v"(a[c]));
      if (OP == 21) asm volatile("v_cmp_gt_f64 vcc, |%0|, %1" : : "v"(a[c]), "v"(x) : "vcc");
      if (OP == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 23) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 24) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(ux));
      if (OP == 25) asm volatile("ds_read_b64 %0, %1" : "=v"(u[c]) : "v"(wx & 0xff8));
      if (OP == 26) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[c & 1]) : "v"(wx & 0xff0));
      if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[c]) : "v"(wx));
      if (OP == 11) asm volatile("v_min_f64 %0, %0, |%1|" : "+v"(a[c]) : "v"(x));
      if (OP == 12) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(w[c]) : "v"(a[c]));
      if (OP == 13) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(w[c]) : "v"(wx));
      if (OP == 14) asm volatile("v_max_f32 %0, %0, |%1|" : "+v"(w[c]) : "v"(wx));
      if (OP == 15) asm volatile("v_mov_b64 %0, %1" : "=v"(u[c]) : "v"(ux));
    }
  }
  double s = 0;
  for (int c = 0; c < CHAINS; c++) s += a[c] + (double)u[c] + (double)w[c];
  asm volatile("s_waitcnt lgkmcnt(0)");
  s += q4[0].x + q4[1].y;
  if (s == 1.2345) out[0] = s;
}

template <int OP>
void run(const char* name, double* d) {
  const int iters = 512, grid = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 16, 1.5);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, 1.5);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_simd = (double)iters * CHAINS * REPS * 8;     // 8 waves per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
  printf("%-28s %8.3f ms  ~%.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc);
}

int main() {
  double* d;
  hipMalloc(&d, 64);
  run<0>("v_fma_f64", d);
  run<8>("v_mul_f64", d);
  run<2>("v_add_f64", d);
  run<1>("v_max_f64 |x|", d);
  run<11>("v_min_f64 |x|", d);
  run<7>("v_cmp_gt_f64", d);
  run<3>("v_cmp_gt_f64 + 1 cndmask", d);
  run<4>("v_cmp_gt_u64", d);
  run<5>("v_max_u32", d);
  run<6>("v_max3_u32", d);
  run<9>("v_cndmask_b32", d);
  run<10>("v_and_b32", d);
  run<12>("v_cvt_f32_f64", d);
  run<13>("v_mov_b32_dpp quad_perm", d);
  run<14>("v_max_f32 |x|", d);
  run<15>("v_mov_b64", d);
  run<16>("v_mov_b32", d);
  run<17>("v_add_u32", d);
  run<18>("v_lshlrev_b32", d);
  run<19>("v_lshl_or_b32", d);
  run<20>("v_cvt_i32_f64", d);
  run<21>("v_cmp_gt_f64 |x|", d);
  run<22>("v_xor_b32", d);
  run<23>("v_fma_f32", d);
  run<24>("v_pk_fma_f32", d);
  run<25>("ds_read_b64 (no wait)", d);
  run<26>("ds_read_b128 (no wait)", d);
  return 0;
}
