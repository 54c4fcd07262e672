// valu_rates.hip -- issue cost of vector instructions on gfx950 as k_compress uses them: cycles per wave64 instruction for
// v_fma_f64 / v_add_f64 / v_fma_f32 / v_pk_fma_f32 / v_add_u32 / v_cvt_f32_f64 / v_floor_f64 at 1, 2, 3, 4 waves per SIMD
// (independent chains of 8).   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int OP>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, double seed) {
  double a[8]; float f[8]; unsigned u[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; u[i] = (unsigned)(i + threadIdx.x); p[i] = f2{f[i], f[i] + 1.f}; }
  const double c = seed * 0.5, d = seed * 0.25;
  const float cf = (float)c, df = (float)d;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __builtin_fma(a[i], c, d);
      if (OP == 1) a[i] = a[i] + c;
      if (OP == 2) f[i] = __builtin_fmaf(f[i], cf, df);
      if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], f2{cf, cf}, f2{df, df});
      if (OP == 4) u[i] = u[i] * 3u + 1u;                 // v_mad_u32_u24 / mul_lo + add
      if (OP == 5) { f[i] = (float)a[i]; a[i] = a[i] + (double)1.0; }   // v_cvt_f32_f64 + v_add_f64
      if (OP == 6) a[i] = __builtin_floor(a[i] * 1.000001);             // v_floor_f64 + v_mul_f64
      if (OP == 7) u[i] = (u[i] << 1) ^ (u[i] >> 3);      // two 32-bit ops (shift, xor-shift)
      if (OP == 8) a[i] = a[i] * c;                        // v_mul_f64
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < 8; i++) s += a[i] + f[i] + u[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> int run(const char* name, int ops_per_iter, double* out, unsigned long long* cyc) {
  for (int wps = 1; wps <= 4; wps++) {
    const int grid = 256 * 4 * wps;                       // single-wave workgroups: wps per SIMD when evenly placed
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, out, cyc, 1.000001);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid);
    CHK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto v : h) m += (double)v; m /= grid;
    printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction in a wave's own time, %.2f per SIMD\n", name, wps,
           m / ((double)ITERS * ops_per_iter), m / ((double)ITERS * ops_per_iter) / wps);
  }
  return 0;
}
int main() {
  double* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, 8 * 64 * 4096)); CHK(hipMalloc(&cyc, 8 * 4096));
  run<0>("v_fma_f64", 8, out, cyc); run<1>("v_add_f64", 8, out, cyc); run<8>("v_mul_f64", 8, out, cyc); run<2>("v_fma_f32", 8, out, cyc);
  run<3>("v_pk_fma_f32", 8, out, cyc); run<4>("u32 mul+add", 8, out, cyc); run<5>("v_cvt_f32_f64 + v_add_f64", 16, out, cyc);
  run<6>("v_floor_f64 + v_mul_f64", 16, out, cyc); run<7>("3 x 32-bit shift/xor", 24, out, cyc);
  return 0;
}
