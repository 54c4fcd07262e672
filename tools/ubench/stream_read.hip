// Micro-benchmark: how fast can gfx950 stream 1 GiB from HBM into registers, as a function of
// the launch shape and of the bytes each wave keeps in flight?  (What k_stats / k_compress can hope for.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_read.hip -o tools/ubench/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef double V __attribute__((ext_vector_type(2)));   // 16 bytes per lane

// persistent waves; each wave walks a CONTIGUOUS range in steps of `TILE` bytes (64 lanes x 16 B x NV),
// NV loads in flight, DEPTH tiles in flight (software pipeline in registers)
template <int NV, int DEPTH, bool NT>
__global__ __launch_bounds__(64) void k_range(const V* __restrict__ x, size_t nvec, double* out) {
  const size_t per_tile = (size_t)64 * NV;
  const size_t ntiles = nvec / per_tile;
  const size_t lo = ntiles * blockIdx.x / gridDim.x, hi = ntiles * (blockIdx.x + 1) / gridDim.x;
  double acc = 0;
  V v[DEPTH][NV];
  auto issue = [&](int slot, size_t tile) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const V* p = x + tile * per_tile + (size_t)i * 64 + threadIdx.x;
      v[slot][i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++) if (lo + d < hi) issue(d, lo + d);
  size_t t = lo;
  for (; t + DEPTH <= hi + 0 && t < hi; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const size_t nxt = t + d + DEPTH - 1;
      if (nxt < hi) issue((d + DEPTH - 1) % DEPTH, nxt);
      if (t + d < hi) {
#pragma unroll
        for (int i = 0; i < NV; i++) acc += v[d][i].x + v[d][i].y;
      }
    }
  }
  for (; t < hi; t++) { issue(0, t);
#pragma unroll
    for (int i = 0; i < NV; i++) acc += v[0][i].x + v[0][i].y; }
  if (acc == 1.2345e300) out[0] = acc;
}

// grid-stride, 256-thread workgroups, UN vectors per thread per trip (k_stats shape)
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_stride(const V* __restrict__ x, size_t nvec, double* out) {
  double acc = 0;
  for (size_t i0 = (size_t)blockIdx.x * 256 * UN + threadIdx.x; i0 < nvec; i0 += (size_t)gridDim.x * 256 * UN) {
    V v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) { const V* p = x + i0 + (size_t)u * 256; v[u] = (i0 + (size_t)u * 256 < nvec) ? (NT ? __builtin_nontemporal_load(p) : *p) : V{0, 0}; }
#pragma unroll
    for (int u = 0; u < UN; u++) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <typename F>
void timeit(const char* name, size_t bytes, F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; r++) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-56s %7.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)1 << 30, nvec = bytes / 16;
  V* x; double* out;
  (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(x, 1, bytes);
  (void)hipDeviceSynchronize();
#define RANGE(NV, DEPTH, NT, WGS) timeit("range NV=" #NV " depth=" #DEPTH " nt=" #NT " waves/CU=" #WGS, bytes, [&] { \
    hipLaunchKernelGGL((k_range<NV, DEPTH, NT>), dim3(256 * WGS), dim3(64), 0, 0, x, nvec, out); })
  RANGE(8, 1, false, 12);
  RANGE(8, 2, false, 12);
  RANGE(8, 3, false, 12);
  RANGE(8, 1, false, 16);
  RANGE(8, 1, false, 24);
  RANGE(8, 1, false, 32);
  RANGE(8, 2, false, 16);
  RANGE(8, 2, false, 32);
  RANGE(16, 1, false, 12);
  RANGE(4, 1, false, 32);
  RANGE(8, 1, true, 12);
  RANGE(8, 2, true, 12);
  RANGE(8, 2, true, 32);
#define STRIDE(UN, NT, G) timeit("stride UN=" #UN " nt=" #NT " grid=" #G, bytes, [&] { \
    hipLaunchKernelGGL((k_stride<UN, NT>), dim3(G), dim3(256), 0, 0, x, nvec, out); })
  STRIDE(4, false, 2048);
  STRIDE(4, false, 1024);
  STRIDE(8, false, 2048);
  STRIDE(8, false, 1024);
  STRIDE(4, true, 2048);
  STRIDE(8, true, 2048);
  STRIDE(2, false, 4096);
  return 0;
}
