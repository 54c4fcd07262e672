// Round-2 hardware probes (gfx950 / MI355X).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/probe_r2.hip -o tools/ubench/probe_r2
// 1. store-data hazard: buffer_store_dwordx4 (MUBUF) with an SGPR soffset, followed at once by a VALU write /
//    an LDS read into its data VGPRs -- LLVM's hazard recognizer inserts the wait states only when soffset is
//    NOT a register (GCNHazardRecognizer::createsVALUHazard); does the hardware agree?
// 2. v_cvt_pk_u8_f32: rounding and saturation.
// 3. global_load_lds_dwordx4: where do the 16 bytes of lane l land (M0 base + 16 l)?
// 4. fp64 issue rate of ONE wave per SIMD (independent chains vs one dependent chain).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- 1. hazard --
// Every wave stores `iters` rows of 8 x 1 KiB.  Data = (row id, lane, vector, 0xA5A5A5A5); right after each
// store the data registers are overwritten with POISON by the instruction form under test.
#define POISON 0xDEADBEEFu
template <int VARIANT>
__global__ __launch_bounds__(64) void k_hazard(unsigned* out, int iters, unsigned per_wave_dwords) {
  __shared__ u32x4 lds[64];
  lds[threadIdx.x] = u32x4{POISON, POISON, POISON, POISON};
  __syncthreads();
  const unsigned lane = threadIdx.x;
  const unsigned wave = blockIdx.x;
  unsigned* base = out + (size_t)wave * per_wave_dwords;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(per_wave_dwords * 4u), 0x00020000);
  const unsigned voff = lane * 16u;
  const unsigned ldsaddr = lane * 16u;
  const unsigned poison = POISON;
  for (int it = 0; it < iters; it++) {
    const unsigned row = (unsigned)it;
#pragma unroll
    for (int v = 0; v < 8; v++) {
      const unsigned soff = (row * 8u + (unsigned)v) * 1024u;
      const unsigned tag = (row << 8) | (unsigned)v;
      // v20..v23 hold the data; the overwrite follows the store with NO instruction in between
      if (VARIANT == 0)        // SGPR soffset, VALU overwrite, 0 wait states
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, %5 offen\n"
                     "v_mov_b32 v20, %6\n v_mov_b32 v21, %6\n v_mov_b32 v22, %6\n v_mov_b32 v23, %6\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(voff), "s"(rsrc), "s"(soff), "v"(poison) : "v20", "v21", "v22", "v23", "memory");
      if (VARIANT == 1)        // SGPR soffset, 1 wait state
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, %5 offen\n s_nop 0\n"
                     "v_mov_b32 v20, %6\n v_mov_b32 v21, %6\n v_mov_b32 v22, %6\n v_mov_b32 v23, %6\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(voff), "s"(rsrc), "s"(soff), "v"(poison) : "v20", "v21", "v22", "v23", "memory");
      if (VARIANT == 2)        // SGPR soffset, 2 wait states
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, %5 offen\n s_nop 1\n"
                     "v_mov_b32 v20, %6\n v_mov_b32 v21, %6\n v_mov_b32 v22, %6\n v_mov_b32 v23, %6\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(voff), "s"(rsrc), "s"(soff), "v"(poison) : "v20", "v21", "v22", "v23", "memory");
      if (VARIANT == 3) {      // soffset = 0 (offset in the VGPR): the documented hazard, 0 wait states
        const unsigned vo = voff + soff;
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, 0 offen\n"
                     "v_mov_b32 v20, %5\n v_mov_b32 v21, %5\n v_mov_b32 v22, %5\n v_mov_b32 v23, %5\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(vo), "s"(rsrc), "v"(poison) : "v20", "v21", "v22", "v23", "memory");
      }
      if (VARIANT == 4)        // SGPR soffset, LDS read into the data registers right behind the store
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, %5 offen\n"
                     "ds_read_b128 v[20:23], %6\n s_waitcnt lgkmcnt(0)\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(voff), "s"(rsrc), "s"(soff), "v"(ldsaddr) : "v20", "v21", "v22", "v23", "memory");
      if (VARIANT == 5)        // SGPR soffset + nt, fp64 VALU write (v_mul_f64) right behind: the shape the old store_tile_buf had
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %1\n v_mov_b32 v22, %2\n v_mov_b32 v23, 0xA5A5A5A5\n s_nop 4\n"
                     "buffer_store_dwordx4 v[20:23], %3, %4, %5 offen nt\n"
                     "v_mul_f64 v[20:21], v[24:25], v[24:25]\n v_mul_f64 v[22:23], v[24:25], v[24:25]\n"
                     :: "v"(tag), "v"(lane), "v"(wave), "v"(voff), "s"(rsrc), "s"(soff) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
    }
  }
}

template <int VARIANT>
static int run_hazard(const char* name) {
  const int waves = 256 * 12, iters = 16;
  const unsigned per_wave_dwords = (unsigned)iters * 8u * 256u;      // iters * 8 KiB
  const size_t total = (size_t)waves * per_wave_dwords;
  unsigned* d = nullptr;
  CK(hipMalloc(&d, total * 4));
  std::vector<unsigned> h(total);
  size_t bad = 0, runs = 0;
  for (int rep = 0; rep < 20; rep++) {
    CK(hipMemset(d, 0, total * 4));
    hipLaunchKernelGGL(k_hazard<VARIANT>, dim3(waves), dim3(64), 0, 0, d, iters, per_wave_dwords);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, total * 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < waves; w++)
      for (int it = 0; it < iters; it++)
        for (int v = 0; v < 8; v++)
          for (int l = 0; l < 64; l++) {
            const unsigned* q = &h[(size_t)w * per_wave_dwords + ((size_t)it * 8 + v) * 256 + (size_t)l * 4];
            const unsigned tag = ((unsigned)it << 8) | (unsigned)v;
            if (q[0] != tag || q[1] != (unsigned)l || q[2] != (unsigned)w || q[3] != 0xA5A5A5A5u) {
              if (bad < 5) printf("    bad: wave %d row %d vec %d lane %d: %08x %08x %08x %08x\n", w, it, v, l, q[0], q[1], q[2], q[3]);
              bad++;
            }
            runs++;
          }
  }
  printf("hazard %-62s corrupted 16-byte stores: %zu of %zu\n", name, bad, runs);
  CK(hipFree(d));
  return 0;
}

// ------------------------------------------------------------ 2. cvt_pk_u8 --
__global__ void k_cvt(const float* x, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0x11223344u);
}

// -------------------------------------------------------------- 3. LDS DMA --
__global__ __launch_bounds__(64) void k_dma(const u32x4* __restrict__ g, u32x4* out) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[128];
  lds[threadIdx.x] = u32x4{0, 0, 0, 0};
  lds[threadIdx.x + 64] = u32x4{0, 0, 0, 0};
  __syncthreads();
  // lane l fetches g[63 - l]; LDS destination = base (&lds[32]) + 16 l ?
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (63 - threadIdx.x)),
                                   (__attribute__((address_space(3))) void*)(lds + 32), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = lds[threadIdx.x];
  out[threadIdx.x + 64] = lds[threadIdx.x + 64];
}

// ------------------------------------------------------------ 4. issue rate --
template <int MODE>   // 0: 8 independent fma chains, 1: one dependent chain, 2: 8 independent add chains, 3: 4 chains
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double seed, long long* cyc) {
  extern __shared__ double pad[];
  double a[8];
  for (int c = 0; c < 8; c++) a[c] = seed * (threadIdx.x + c + 1);
  const double x = seed * 1.0001;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < 8; rep++)
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(x));
        if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[0]) : "v"(x));
        if (MODE == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(x));
        if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c & 1]) : "v"(x));
        if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c & 3]) : "v"(x));
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < 8; c++) s += a[c];
  if (s == 1.2345) out[0] = s + pad[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static int run_rate(const char* name, int wg_per_cu_lds) {
  double* d; long long* c;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&c, 64));
  const int iters = 4000;
  // dynamic LDS sized so that exactly `wg_per_cu_lds` 256-thread workgroups fit a CU
  const size_t lds = (size_t)(160 * 1024 / wg_per_cu_lds) - 256;
  CK(hipFuncSetAttribute((const void*)k_rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * wg_per_cu_lds), dim3(256), lds, 0, d, iters, 1.000001, c);
  CK(hipDeviceSynchronize());
  long long cy = 0;
  CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
  printf("rate %-40s %d wave(s)/SIMD: %.2f shader cycles per v_*_f64 (one wave's view), %.2f per SIMD slot\n", name, wg_per_cu_lds,
         (double)cy / (iters * 64.0), (double)cy / (iters * 64.0) / wg_per_cu_lds);
  CK(hipFree(d)); CK(hipFree(c));
  return 0;
}

int main() {
  printf("== 1. buffer_store_dwordx4 data-register overwrite ==\n");
  if (run_hazard<0>("SGPR soffset, VALU overwrite, 0 wait states")) return 1;
  if (run_hazard<1>("SGPR soffset, VALU overwrite, s_nop 0 (1 wait state)")) return 1;
  if (run_hazard<2>("SGPR soffset, VALU overwrite, s_nop 1 (2 wait states)")) return 1;
  if (run_hazard<3>("soffset = 0 (VGPR offset), VALU overwrite, 0 wait states")) return 1;
  if (run_hazard<4>("SGPR soffset, ds_read_b128 into the data registers")) return 1;
  if (run_hazard<5>("SGPR soffset nt, v_mul_f64 into the data registers, 0 wait")) return 1;

  printf("== 2. v_cvt_pk_u8_f32 (byte 1 of 0x11223344) ==\n");
  {
    const float xs[] = {-1e9f, -1.f, -0.6f, -0.5f, -0.f, 0.f, 0.4f, 0.5f, 0.6f, 1.f, 1.5f, 2.5f, 3.5f, 127.5f, 254.f, 254.4f, 254.5f, 254.6f,
                        255.f, 255.5f, 256.f, 300.f, 1e9f, INFINITY, -INFINITY, NAN};
    const int n = sizeof(xs) / sizeof(xs[0]);
    float* dx; unsigned* dout;
    CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dout, n * 4));
    CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, dout, n);
    CK(hipDeviceSynchronize());
    unsigned ho[64];
    CK(hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) printf("  %12g -> byte %3u (word %08x)\n", xs[i], (ho[i] >> 8) & 255u, ho[i]);
  }

  printf("== 3. global_load_lds_dwordx4 ==\n");
  {
    u32x4 hg[64], ho[128];
    for (int i = 0; i < 64; i++) hg[i] = u32x4{(unsigned)i, 100u + i, 200u + i, 300u + i};
    u32x4 *dg, *dout;
    CK(hipMalloc(&dg, sizeof(hg))); CK(hipMalloc(&dout, sizeof(ho)));
    CK(hipMemcpy(dg, hg, sizeof(hg), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, dg, dout);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int s = 0; s < 128; s++) {
      const int l = s - 32;                                  // expected: slot 32 + l holds what lane l fetched = g[63 - l]
      const unsigned want = (l >= 0 && l < 64) ? (unsigned)(63 - l) : 0u;
      if (ho[s].x != want || ((l >= 0 && l < 64) && ho[s].w != 300u + want)) { ok = 0; printf("  slot %d: %u %u %u %u (want %u)\n", s, ho[s].x, ho[s].y, ho[s].z, ho[s].w, want); }
    }
    printf("  LDS slot of lane l = base + 16 l, per-lane global address honoured: %s\n", ok ? "yes" : "NO");
  }

  printf("== 4. fp64 issue rate ==\n");
  if (run_rate<0>("8 independent v_fma_f64 chains", 1)) return 1;
  if (run_rate<4>("4 independent v_fma_f64 chains", 1)) return 1;
  if (run_rate<3>("2 independent v_fma_f64 chains", 1)) return 1;
  if (run_rate<1>("1 dependent v_fma_f64 chain", 1)) return 1;
  if (run_rate<2>("8 independent v_add_f64 chains", 1)) return 1;
  if (run_rate<0>("8 independent v_fma_f64 chains", 2)) return 1;
  if (run_rate<1>("1 dependent v_fma_f64 chain", 2)) return 1;
  if (run_rate<0>("8 independent v_fma_f64 chains", 4)) return 1;
  return 0;
}
