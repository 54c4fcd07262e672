// Micro-benchmark: which cache-policy bits make a 1 GiB read / write stream fastest on gfx950?
// buffer_load/store_dwordx4 through a descriptor, 12 single-wave workgroups per CU walking contiguous
// ranges (the k_compress / k_decompress shape).  aux: bit0 sc0, bit1 nt, bit4 sc1 (gfx940 encoding).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_policy.hip -o tools/ubench/stream_policy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX, bool WRITE>
__global__ __launch_bounds__(64) void k(char* x, size_t bytes, unsigned* out) {
  const size_t per = bytes / gridDim.x;                      // multiple of 8 KiB for our sizes
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(x + per * blockIdx.x, 0, (int)per, 0x00020000);
  unsigned acc = 0;
  for (size_t off = 0; off < per; off += 8192) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (WRITE) { v[i] = u32x4{(unsigned)off, 1u, 2u, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v[i], r, threadIdx.x * 16 + (i & 3) * 1024, (int)off + (i >> 2) * 4096, AUX); }
      else v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16 + (i & 3) * 1024, (int)off + (i >> 2) * 4096, AUX);
    }
    if (!WRITE) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc += v[i].x ^ v[i].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int AUX, bool WRITE>
void run(char* x, size_t bytes, unsigned* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<AUX, WRITE>), dim3(256 * 12), dim3(64), 0, 0, x, bytes, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  printf("%s aux=%2d (%s%s%s)  %7.3f ms  %6.2f TB/s\n", WRITE ? "store" : "load ", AUX, (AUX & 1) ? "sc0 " : "", (AUX & 2) ? "nt " : "",
         (AUX & 16) ? "sc1" : "", best, bytes / (best * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)3072 * 8192 * 42;             // ~1.03 GB, whole 8 KiB tiles per workgroup
  char* x; unsigned* out;
  (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(x, 1, bytes);
  (void)hipDeviceSynchronize();
  run<0, false>(x, bytes, out);  run<1, false>(x, bytes, out);  run<2, false>(x, bytes, out);  run<3, false>(x, bytes, out);
  run<16, false>(x, bytes, out); run<17, false>(x, bytes, out); run<18, false>(x, bytes, out); run<19, false>(x, bytes, out);
  run<0, true>(x, bytes, out);   run<1, true>(x, bytes, out);   run<2, true>(x, bytes, out);   run<3, true>(x, bytes, out);
  run<16, true>(x, bytes, out);  run<17, true>(x, bytes, out);  run<18, true>(x, bytes, out);  run<19, true>(x, bytes, out);
  return 0;
}
