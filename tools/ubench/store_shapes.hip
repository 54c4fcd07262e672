// store_shapes.hip -- does the SHAPE of a 1 GiB store stream matter?  k_decompress gives every single-wave workgroup a
// contiguous range of tiles (1024 ranges of 1 MiB walked side by side); the alternative is tile-interleaved (workgroup b
// takes tiles b, b + G, b + 2G, ...: at any moment the grid writes one contiguous window).  Rows of 1 KiB per instruction,
// 16 bytes per lane, like the kernels' row stores.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_shapes.hip -o tools/ubench/store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: a contiguous range per workgroup; 1: tile-interleaved; 2: a contiguous range walked from a rotated start
// (workgroup b begins at tile b mod len of its range and wraps: at any moment the workgroups are at different offsets of
// their ranges instead of all at the same one)
template <int AUX, int SHAPE, bool READ = false>
__global__ __launch_bounds__(64) void k(char* x, size_t bytes, unsigned tile_bytes, unsigned* sink = nullptr) {
  const size_t ntiles = bytes / tile_bytes;
  const unsigned G = gridDim.x, b = blockIdx.x;
  const size_t per = ntiles / G;
  unsigned acc = 0;
  for (size_t r = 0; r < per; r++) {
    const size_t tile = SHAPE == 1 ? (size_t)b + r * G : (SHAPE == 2 ? (size_t)b * per + (r + b) % per : (size_t)b * per + r);
    char* base = x + tile * tile_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)tile_bytes, 0x00020000);
    for (unsigned off = 0; off < tile_bytes; off += 8192) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (READ) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16 + i * 1024, (int)off, AUX); acc += v.x ^ v.w; }
        else {
          const u32x4 v = u32x4{(unsigned)off, (unsigned)r, 2u, (unsigned)i};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16 + i * 1024, (int)off, AUX);
        }
      }
    }
  }
  if (READ && acc == 0x12345678u && sink) sink[0] = acc;
}

template <int AUX, int IL, bool READ = false>
static void run(const char* name, char* x, size_t bytes, unsigned grid, unsigned tile_bytes) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f, sum = 0;
  for (int r = 0; r < 8; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<AUX, IL, READ>), dim3(grid), dim3(64), 0, 0, x, bytes, tile_bytes, (unsigned*)nullptr);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  printf("%-44s grid %5u tile %6u B: mean %.3f ms = %.2f TB/s (best %.2f)\n", name, grid, tile_bytes, sum / 6, bytes / (sum / 6) / 1e9, bytes / best / 1e9);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* x;
  if (hipMalloc(&x, bytes) != hipSuccess) return 1;
  (void)hipMemset(x, 0, bytes);
  for (unsigned grid : {1024u, 2048u}) {
    for (unsigned tb : {32768u, 16384u}) {
      run<2, 0>("store: contiguous range per workgroup, nt", x, bytes, grid, tb);
      run<2, 2>("store: contiguous range, rotated start, nt", x, bytes, grid, tb);
      run<2, 1>("store: tile-interleaved, nt", x, bytes, grid, tb);
    }
  }
  for (unsigned grid : {2048u, 3072u}) {
    for (unsigned tb : {32768u, 16384u}) {
      run<2, 0, true>("read: contiguous range per workgroup, nt", x, bytes, grid, tb);
      run<2, 2, true>("read: contiguous range, rotated start, nt", x, bytes, grid, tb);
      run<2, 1, true>("read: tile-interleaved, nt", x, bytes, grid, tb);
    }
  }
  (void)hipFree(x);
  return 0;
}
