// dctz_device.h -- types shared by the kernels (dctz_kernels.hip) and the C-ABI
// shim (dctz_shim.hip): per-dtype traits, the per-call control block, kernel
// parameter blocks and the launcher prototypes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dctz_hip.h"
#include "dct64_lane.h"

namespace dctz {

#ifndef DCTZ_PITCH
#define DCTZ_PITCH 66      // LDS elements per block (fp64); fp32 uses DCTZ_PITCH + 2 when even
#endif
#ifndef DCTZ_TILE_BLKS
#define DCTZ_TILE_BLKS 16      // 16 blocks = one wavefront per workgroup: barriers cost nothing, 12 workgroups per CU (fp64)
#endif
constexpr int TILE_BLKS = DCTZ_TILE_BLKS; // blocks per tile (one quad of lanes per block)
constexpr int TILE_ELEMS = TILE_BLKS * 64;
constexpr int WG = TILE_BLKS * 4;         // threads per workgroup
constexpr int SWG = 256;                  // threads per workgroup of the streaming helpers (stats, count, compact, ...)

template <typename T> struct Traits;
template <> struct Traits<double> {
  using Vec = double2;
  using Bits = unsigned long long;
  static constexpr int EPV = 2;           // elements per 16-byte vector
  static constexpr int PITCH = DCTZ_PITCH;   // LDS elements per block
  __host__ __device__ static Vec zero() { return make_double2(0.0, 0.0); }
  __device__ static void div(Vec& v, double s) { v.x = v.x / s; v.y = v.y / s; }
  __device__ static void mul(Vec& v, double s) { v.x = v.x * s; v.y = v.y * s; }
  __device__ static void unpack(const Vec& v, double* e) { e[0] = v.x; e[1] = v.y; }
  __device__ static Vec pack(const double* e) { return make_double2(e[0], e[1]); }
  __device__ static double huge() { return 1.79769313486231570815e308; }
  __device__ static double from_bits(Bits b) { return __longlong_as_double((long long)b); }
};
template <> struct Traits<float> {
  using Vec = float4;
  using Bits = unsigned int;
  static constexpr int EPV = 4;
  static constexpr int PITCH = (DCTZ_PITCH % 2) ? DCTZ_PITCH : DCTZ_PITCH + 2;
  __host__ __device__ static Vec zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ static void div(Vec& v, float s) { v.x = v.x / s; v.y = v.y / s; v.z = v.z / s; v.w = v.w / s; }
  __device__ static void mul(Vec& v, float s) { v.x = v.x * s; v.y = v.y * s; v.z = v.z * s; v.w = v.w * s; }
  __device__ static void unpack(const Vec& v, float* e) { e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
  __device__ static Vec pack(const float* e) { return make_float4(e[0], e[1], e[2], e[3]); }
  __device__ static float huge() { return 3.40282346638528859812e38f; }
  __device__ static float from_bits(Bits b) { return __uint_as_float(b); }
};

// Per-call control block in device memory; zeroed by ONE hipMemsetAsync before
// every call (ticket, look-back total, watchdog flag, QT table accumulators).
struct Ctl {
  unsigned ticket;                 // next tile to hand out
  unsigned cnt_total;              // exceptions emitted / consumed so far
  unsigned error;                  // 1: look-back watchdog, 2: AC_exact underrun
  unsigned pad0;
  unsigned long long qraw[64];     // max |coef| per position, raw bits of T (dctz-comp-lib.c:371-372)
  unsigned long long q0;           // bits of the last block's DC (qtable[0], :355-360)
  unsigned long long pad1;
  unsigned gticket[8 * 32];        // per-group tile tickets, one 128-byte line each
  unsigned long long dbg[8];       // F_STAMP builds: summed phase cycles of thread 0 of every workgroup
};
static_assert(sizeof(Ctl) % 16 == 0, "memset block must be a multiple of 16 bytes");

// Result mailbox in fine-grained pinned HOST memory: the last kernel of a phase writes
// the few words the host needs and then a sequence number, the host spins on that word
// instead of paying a D2H copy kernel + hipStreamSynchronize wake-up per hand-off.
struct HostBox {
  volatile unsigned long long seq_stats;   // == call sequence number once stats[] is valid
  double stats[3];                         // max|x|, min|x|, sum of the statistics pass / the sample
  volatile unsigned long long seq_done;    // == call sequence number once everything below is valid
  unsigned cnt_total, error;
  unsigned long long q0;
  unsigned long long qraw[64];
  double fstats[3];                        // statistics fused into k_compress (F_STATS)
};

enum : int { F_LOOKBACK = 1, F_GROUP = 2, F_STAMP = 4, F_STATS = 8 };   // kernel feature bits (dctz_kernels.hip)

template <typename T>
struct FwdParams {
  const T* x;                      // input
  uint8_t* bin;                    // bin_index out
  float* dc;                       // DC out
  float* ac;                       // AC_exact out
  T* scaled;                       // optional x/sf out
  T* coef;                         // optional coefficient tap
  float* ac_tmp;                   // two-level scheme: tile-local AC_exact lists, slot = tile * 4096
  unsigned* tile_cnt;              // two-level scheme: list lengths (NULL selects the single-pass kernels' rules)
  const unsigned* tile_off;        // two-level scheme: exclusive prefix of tile_cnt (k_scan_tiles)
  T* qt_item;                      // QT scratch: flagged coefficients, full precision
  uint8_t* qt_j;                   // QT scratch: their position j
  const T* tab;                    // TAB_* block (device)
  const T* rtab;                   // RTAB_* block (device), remainder block only
  Ctl* ctl;
  unsigned long long* desc;        // look-back descriptors, one per tile
  double* stat_part;               // F_STATS: {max|x|, min|x|, sum} per workgroup (+1 slot for the remainder block), else NULL
  unsigned nfull;                  // number of full 64-element blocks
  unsigned ntiles;
  unsigned last_is_full;           // N % 64 == 0
  unsigned fast_sf, fast_bw;       // divisor inside FastDiv's exponent window (host check)
  unsigned ngroups;                // ticket groups, min(8, grid)
  unsigned nlists_main;            // two-level scheme: number of workgroup lists = grid of k_compress
  T sf, bin_width, range_min, range_max;
};

template <typename T>
struct InvParams {
  const uint8_t* bin;
  const float* dc;
  const float* ac;
  T* out;
  const T* tab;
  const T* rtab;
  const T* qtab;                   // QT: clamped table (device)
  const unsigned* tile_off;        // two-level scheme: exclusive prefix of per-tile flag counts (else NULL)
  Ctl* ctl;
  unsigned long long* desc;
  unsigned nfull, ntiles, ac_count;
  unsigned ngroups;                // ticket groups, min(8, grid)
  unsigned nlists_main;            // two-level scheme: number of workgroup ranges = grid of k_decompress
  T sf, bin_width, range_min, range_max;
  double eb;
};

template <typename T> void launch_stats(const T* x, size_t n, double* part, int nparts, double* out, hipStream_t s,
                                        HostBox* box = nullptr, unsigned long long seq = 0);
template <typename T> void launch_stats_sample(const T* x, size_t n, unsigned group, double* part, int nparts, double* out, hipStream_t s,
                                               HostBox* box = nullptr, unsigned long long seq = 0);
void launch_stats_final(const double* part, int nparts, double* out, hipStream_t s);
void launch_finish(Ctl* ctl, const double* part, int nparts, HostBox* box, unsigned long long seq, hipStream_t s);
template <typename T> void launch_debug_divide(const T* x, size_t n, T d, int ok, T* fast, T* ref, hipStream_t s);
template <typename T> void launch_serial_sum(const T* x, size_t n, double* out, hipStream_t s);
template <typename T> void launch_scale(T* x, size_t n, T sf, int grid, hipStream_t s);
template <typename T> void launch_compress(const FwdParams<T>& p, int mode, bool scale, int grid, int feat, hipStream_t s);
template <typename T> void launch_compress_rem(const FwdParams<T>& p, int mode, bool scale, int l, hipStream_t s);
template <typename T> void launch_qt_finish(const FwdParams<T>& p, double eb, int grid, hipStream_t s);
template <typename T> void launch_compact_ac(const FwdParams<T>& p, int mode, double eb, unsigned nlists, int grid, hipStream_t s);
void launch_scan_tiles(const unsigned* cnt, unsigned* off, unsigned n, Ctl* ctl, hipStream_t s);
void launch_count_tiles(const uint8_t* bin, unsigned nfull, unsigned ntiles, unsigned* tile_cnt, int grid, hipStream_t s);
template <typename T> void launch_decompress(const InvParams<T>& p, int mode, bool scale, int grid, int feat, hipStream_t s);
template <typename T> void launch_decompress_rem(const InvParams<T>& p, int mode, bool scale, int l, hipStream_t s);
template <typename T> void launch_dct_blocks(const T* x, T* out, const T* gtab, const T* rtab, size_t n,
                                             bool inverse, int grid, hipStream_t s);

}  // namespace dctz
