// dctz_kernel_common.h -- device helpers shared by the two kernel files: streaming accesses, the wave scan, the exact
// division by a launch constant (FastDiv), the tile range of a workgroup, the tile image in LDS (TileMap, LDS-DMA issue,
// LDS <-> registers), the statistics accumulator, the QT (de-)normalisation, the mailbox publish, small reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dct64_block.h"
#include "dct_nd_block.h"
#include "dct64_block_pk.h"
#include "dctz_device.h"

namespace dctz {


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// The twiddle block is written once at context creation and never by a kernel: reading it through the constant
// address space lets the compiler use scalar loads (the table indices are compile-time constants, the base is a
// kernel argument: wave-uniform).  A plain global pointer gets VECTOR loads here, because the kernels also store to
// global memory and nothing tells the compiler that the table is not among the targets.
template <typename T> using CTab = const __attribute__((address_space(4))) T*;
template <typename T> __device__ __forceinline__ CTab<T> as_ctab(const T* p) { return (CTab<T>)(p); }
// LDS-DMA: 16 bytes per lane, HBM -> LDS (lane l lands at lds + 16 l), through a buffer descriptor (range-checked: zeros
// beyond the end).  Device pass only (the host pass of hipcc does not know the builtin).
#if defined(__HIP_DEVICE_COMPILE__)
#define DMA16(rsrc, lds, voff, soff, aux) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds), 16, voff, soff, 0, aux)
#else
#define DMA16(rsrc, lds, voff, soff, aux) ((void)0)
#endif

// LDS stores the compiler does not see as such.  A pending LDS-DMA (the next tile on its way into its image) makes the
// compiler put `s_waitcnt vmcnt(0)` in front of every LDS STORE it cannot prove disjoint from the DMA's target -- and
// it proves that for loads only (the disassembly of round 2's k_compress had such a wait in front of every parked
// group: the DMA that was meant to land under the arithmetic was waited for at the first store behind its issue).  The
// staging buffers these helpers write are arrays of their own, never a DMA target.  LDS operations of a wave execute
// in order, so a compiler-visible load behind such a store sees its data.
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(size_t)LDS_PTR(p); }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void lds_store_b8(unsigned at, unsigned v) { asm volatile("ds_write_b8 %0, %1" :: "v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_b32(unsigned at, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_b64(unsigned at, u32x2 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_store_b128(unsigned at, u32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(at), "v"(v) : "memory"); }
#else
__device__ __forceinline__ void lds_store_b8(unsigned, unsigned) {}
__device__ __forceinline__ void lds_store_b32(unsigned, unsigned) {}
__device__ __forceinline__ void lds_store_b64(unsigned, u32x2) {}
__device__ __forceinline__ void lds_store_b128(unsigned, u32x4) {}
#endif
// (the same for LDS atomics without a return value: a store as far as the wait insertion is concerned)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void lds_max_u32(unsigned at, unsigned v) { asm volatile("ds_max_u32 %0, %1" :: "v"(at), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_max_u64(unsigned at, unsigned long long v) { asm volatile("ds_max_u64 %0, %1" :: "v"(at), "v"(v) : "memory"); }
#else
__device__ __forceinline__ void lds_max_u32(unsigned, unsigned) {}
__device__ __forceinline__ void lds_max_u64(unsigned, unsigned long long) {}
#endif
__device__ __forceinline__ void lds_max_bits(unsigned at, unsigned v) { lds_max_u32(at, v); }
__device__ __forceinline__ void lds_max_bits(unsigned at, unsigned long long v) { lds_max_u64(at, v); }
__device__ __forceinline__ void lds_store_item(unsigned at, float v) { lds_store_b32(at, __builtin_bit_cast(unsigned, v)); }
__device__ __forceinline__ void lds_store_item(unsigned at, double v) { lds_store_b64(at, __builtin_bit_cast(u32x2, v)); }

// ------------------------------------------------------------------ helpers --
// Streaming (read-once / write-once) 16-byte accesses: the `nt` policy.  A pure 1 GiB read stream
// runs at 6.8-7.1 TB/s with nt loads against 6.0-6.3 TB/s with plain ones (tools/ubench/stream_read.hip).
template <typename V>
__device__ __forceinline__ V load_stream(const V* p) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
  const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  V v;
  __builtin_memcpy(&v, &r, 16);
  return v;
}
template <typename V>
__device__ __forceinline__ void store_stream(V* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
  __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
}

// Inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts / row broadcasts
// (six dependent VALU steps instead of six ds_bpermute round trips through the LDS pipe).
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}

__device__ __forceinline__ unsigned long long to_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned int to_bits(float v) { return __float_as_uint(v); }

// ----------------------------------------------- division by a kernel constant --
// x / d with d uniform over the launch (the scaling factor, the bin width).
// hipcc expands an IEEE division into: v_div_scale x2, v_rcp, two (f64) / one (f32)
// Newton steps on the reciprocal, q = x*y, r = fma(-d, q, x), fma(r, y, q) [f32:
// one more residual step], v_div_fmas, v_div_fixup.  Everything up to the
// reciprocal y depends on d alone, and the scale/fixup steps are the identity
// while the exponents of x, d and x/d stay away from the overflow / denormal
// ends.  So: y is computed once per thread with the very same instructions, x is
// checked against a conservative exponent window, and inside it the remaining
// 3 (f64) / 5 (f32) operations give bit-for-bit what `x / d` gives.  Outside the
// window (and for zeros, whose sign v_div_fixup restores) the full division runs.
// tests/test_gpu_parity.py::test_fast_division_is_exact checks the identity on
// the GPU against the compiler's own division.
template <typename T> struct FastDiv;
template <> struct FastDiv<double> {
  double d, y;
  bool ok;                       // host: |d| in [2^-250, 2^250]
  __device__ __forceinline__ void init(double dd, bool okk) {
    d = dd; ok = okk;
    double r = __builtin_amdgcn_rcp(dd);
    double e = fma(-dd, r, 1.0); r = fma(r, e, r);
    e = fma(-dd, r, 1.0); r = fma(r, e, r);
    // the divisor is a kernel argument, so y is wave-uniform: keep it in SGPRs
    y = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r)),
                         __builtin_amdgcn_readfirstlane(__double2loint(r)));
  }
  __device__ __forceinline__ double core(double x) const {
    const double q = x * y;
    const double r = fma(-d, q, x);
    return fma(r, y, q);
  }
  __device__ __forceinline__ double div(double x) const {          // any x
    const unsigned ex = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    if (ok && (ex - 523u) <= 1000u) return core(x);                // |x| in [2^-500, 2^501)
    if (ok && x == 0.0) return x * y;                              // signed zero
    return x / d;
  }
};
template <> struct FastDiv<float> {
  float d, y;
  bool ok;                       // host: |d| in [2^-30, 2^30]
  __device__ __forceinline__ void init(float dd, bool okk) {
    d = dd; ok = okk;
    const float r = __builtin_amdgcn_rcpf(dd);
    const float e = fmaf(-dd, r, 1.0f);
    y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(fmaf(e, r, r))));   // wave-uniform -> SGPR
  }
  __device__ __forceinline__ float core(float x) const {
    const float q = x * y;
    const float r = fmaf(-d, q, x);
    const float q2 = fmaf(r, y, q);
    const float r2 = fmaf(-d, q2, x);
    return fmaf(r2, y, q2);
  }
  // two quotients at once with the packed fp32 instructions of gfx950 (v_pk_mul_f32 / v_pk_fma_f32: the same IEEE
  // operations per component, half the issue slots)
  __device__ __forceinline__ f32x2 core2(f32x2 x) const {
    const f32x2 dd = {d, d}, yy = {y, y};
    const f32x2 q = x * yy;
    const f32x2 r = __builtin_elementwise_fma(-dd, q, x);
    const f32x2 q2 = __builtin_elementwise_fma(r, yy, q);
    const f32x2 r2 = __builtin_elementwise_fma(-dd, q2, x);
    return __builtin_elementwise_fma(r2, yy, q2);
  }
  __device__ __forceinline__ float div(float x) const {
    const unsigned ex = (__float_as_uint(x) >> 23) & 0xffu;
    if (ok && (ex - 64u) <= 126u) return core(x);                  // |x| in [2^-63, 2^64)
    if (ok && x == 0.0f) return x * y;
    return x / d;
  }
};

// (call sites shared by the fp64 instantiations, where the packed form does not exist and the branch is compiled out)
__device__ __forceinline__ f32x2 fastdiv_core2(const FastDiv<float>& d, f32x2 x) { return d.core2(x); }
__device__ __forceinline__ f32x2 fastdiv_core2(const FastDiv<double>&, f32x2 x) { return x; }

// Workgroup b of G owns the contiguous tiles [lo, hi) -- the same partition in k_compress / k_compact_ac
// and in k_decompress.
struct TileRange { unsigned lo, hi; };
__host__ __device__ __forceinline__ TileRange tile_range(unsigned b, unsigned G, unsigned ntiles) {
  const unsigned q = ntiles / G, r = ntiles % G;
  TileRange tr;
  tr.lo = b * q + (b < r ? b : r);
  tr.hi = tr.lo + q + (b < r ? 1u : 0u);
  return tr;
}

// ---------------------------------------------------------- the tile in LDS --
// A tile's image in LDS is made of 1 KiB ROWS; row (jg, s) holds segment s (128 bytes) of the 8 blocks
// 8 jg .. 8 jg + 7, and inside a block's 128 bytes the 16-byte chunks are XOR-swizzled with f(block).  A row is what ONE
// LDS-DMA instruction writes (lane l -> bytes [16 l, 16 l + 16) of the row) and what one 16-byte-per-lane store
// instruction reads back, and lane l's share of a row is a piece of a whole 128-byte line in HBM.  Lane b = block b
// moves chunk ch of its block with ds_read_b128 / ds_write_b128 at lds_a[ch & 7] + (ch >> 3) * 1024.
// The two directions bank differently on gfx950 (MI355X guide, LDS table):
//   * ds_read_b128: groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... ; bank = (a / 4) mod 64.  Lane b's
//     address mod 256 is (b & 1) * 128 + (g ^ f) * 16, so f has to differ between the lanes of equal parity of a group;
//   * ds_write_b128: groups of 8 CONTIGUOUS lanes; bank = (a / 4) mod 32, i.e. only (g ^ f) * 16 counts: f has to take
//     8 different values over b & 7.
// f = (b & 7) ^ ((b >> 4) & 1) does both.  Rounds 1-3 used f = (b >> 1) & 7 (DCTZ_SWZ = 0), which is right for the reads
// and 2-way conflicted for every write of the image: k_decompress's 32 writes per tile made 9.07 M of its 22.05 M
// LDS-array cycles conflict cycles (41 %, profiles/r03_pmc.txt).
#ifndef DCTZ_SWZ
#define DCTZ_SWZ 1
#endif
template <typename T, int PH>
struct TileMap {
  using G = Geo<T, PH>;
  int lds_a[8];        // LDS byte offset of chunk class g = ch & 7 of this lane's block (row part of segment 0)
  int g_even, g_odd;   // HBM byte offset (inside a tile) of this lane's 16 bytes of row (jg, 0), for odd_row(jg) false / true
  // which of the two a row takes: the part of f that comes from the row's number
  static __host__ __device__ constexpr bool odd_row(int jg) { return DCTZ_SWZ ? (jg & 2) != 0 : (jg & 1) != 0; }
  __device__ __forceinline__ int g_of(int jg) const { return odd_row(jg) ? g_odd : g_even; }
  __device__ __forceinline__ void init(int lane) {
    const int f = DCTZ_SWZ ? ((lane & 7) ^ ((lane >> 4) & 1)) : ((lane >> 1) & 7);
#pragma unroll
    for (int g = 0; g < 8; g++) lds_a[g] = (lane >> 3) * G::SEGP * 1024 + (lane & 7) * 128 + ((g ^ f) * 16);
    // as lane l of a ROW instruction: slot l of the row = chunk class (l & 7) ^ f(block) of block 8 jg + (l >> 3)
    const int beta = lane >> 3, gam = lane & 7;
    g_even = beta * G::BLKB + ((gam ^ (DCTZ_SWZ ? beta : (beta >> 1))) * 16);
    g_odd = g_even ^ (DCTZ_SWZ ? 16 : 64);
  }
};
// (the same for the multi-dimensional forms, which compute their chunk per row: chunk class of row-instruction lane
// (beta, gam) in row jg)
__device__ __forceinline__ int swz_row_chunk(int beta, int gam, int jg) {
  return DCTZ_SWZ ? (gam ^ beta ^ ((jg >> 1) & 1)) : (gam ^ (beta >> 1) ^ ((jg & 1) << 2));
}

// HBM -> LDS, one phase of a tile, no registers.  rsrc covers the workgroup's input range; the range check
// zero-fills whatever lies beyond the last whole block.
// (JG0, JG1: the rows of block groups [JG0, JG1) only -- k_compress issues a phase in pieces, between its other work)
template <typename T, int PH, int JG0 = 0, int JG1 = 8>
__device__ __forceinline__ void issue_phase_dma(__amdgpu_buffer_rsrc_t rsrc, unsigned rel, int phase, unsigned char* tilebuf, const TileMap<T, PH>& tm) {
  using G = Geo<T, PH>;
  const int base = (int)(rel * (unsigned)G::TILEB) + phase * G::SEGP * 128;
#pragma unroll
  for (int jg = JG0; jg < JG1; jg++)
#pragma unroll
    for (int s = 0; s < G::SEGP; s++)
      DMA16(rsrc, tilebuf + (jg * G::SEGP + s) * 1024, tm.g_of(jg), base + jg * 8 * G::BLKB + s * 128, 2 /* nt */);
}

// ---- multi-dimensional blocks straight from / to the array (NdDirect) ----
// Byte offset of the origin of block B of the tile grid, or an offset beyond any descriptor range for B >= nblk
// (loads return zeros there, stores are dropped).
template <typename T>
__device__ __forceinline__ unsigned nd_block_origin(const NdDirect& nd, unsigned B) {
  auto divmod = [](unsigned a, unsigned d, unsigned m, unsigned& r) {
    unsigned q = __umulhi(a, m);                     // m = floor(2^32 / d): q is the quotient or one short of it
    r = a - q * d;
    if (r >= d) { q++; r -= d; }
    return q;
  };
  unsigned bx, elem;
  const unsigned t = divmod(B, nd.nbx, nd.mx, bx);
  if (nd.nd == 2) {
    elem = t * 8u * nd.dx + bx * 8u;
  } else {
    unsigned by;
    const unsigned bz = divmod(t, nd.nby, nd.my, by);
    elem = (bz * 4u * nd.dy + by * 4u) * nd.dx + bx * 4u;
  }
  return B < nd.nblk ? elem * (unsigned)sizeof(T) : 0xFFFFFFF0u;
}
// Byte offset, inside its block's footprint in the array, of 16-byte chunk ch of the block (chunk ch = elements
// [ch * EPV, ch * EPV + EPV) of the row-major tile: always inside one row of the tile)
template <typename T>
__device__ __forceinline__ unsigned nd_chunk_offset(const NdDirect& nd, int ch) {
  const unsigned j0 = (unsigned)ch * (unsigned)Traits<T>::EPV;
  if (nd.nd == 2) return ((j0 >> 3) * nd.dx + (j0 & 7u)) * (unsigned)sizeof(T);
  return (((j0 >> 4) * nd.dy + ((j0 >> 2) & 3u)) * nd.dx + (j0 & 3u)) * (unsigned)sizeof(T);
}
// HBM -> LDS, one phase of a tile of a multi-dimensional array: same image in LDS as issue_phase_dma builds for the
// flat layout (row (jg, s) = segment s of blocks 8 jg .. 8 jg + 7, chunks XOR-swizzled), other addresses in HBM.
template <typename T, int PH>
__device__ __forceinline__ void issue_phase_dma_nd(__amdgpu_buffer_rsrc_t rsrc, const NdDirect& nd, unsigned tile, int phase, unsigned char* tilebuf, int lane) {
  using G = Geo<T, PH>;
  const int beta = lane >> 3, gam = lane & 7;
#pragma unroll
  for (int jg = 0; jg < 8; jg++) {
    const unsigned org = nd_block_origin<T>(nd, tile * (unsigned)TILE_BLKS + (unsigned)(8 * jg + beta));
    const int cg = swz_row_chunk(beta, gam, jg);                       // chunk of the segment this lane moves (TileMap's swizzle)
#pragma unroll
    for (int s = 0; s < G::SEGP; s++) {
      const unsigned off = org + nd_chunk_offset<T>(nd, 8 * (phase * G::SEGP + s) + cg);
      DMA16(rsrc, tilebuf + (jg * G::SEGP + s) * 1024, (int)(org >= 0xFFFFFFF0u ? org : off), 0, 2 /* nt */);
    }
  }
}

// LDS image of phase PHASE -> this lane's elements [PHASE * 64 / PH, (PHASE + 1) * 64 / PH) of its block
template <typename T, int PH, int PHASE>
__device__ __forceinline__ void read_phase(T (&x)[64], const unsigned char* tilebuf, const TileMap<T, PH>& tm) {
  using Vec = typename Traits<T>::Vec;
  using G = Geo<T, PH>;
  constexpr int EPV = Traits<T>::EPV;
#pragma unroll
  for (int ch = 0; ch < G::CHP; ch++) {
    const Vec v = *reinterpret_cast<const Vec*>(tilebuf + tm.lds_a[ch & 7] + (ch >> 3) * 1024);
    Traits<T>::unpack(v, &x[(PHASE * G::CHP + ch) * EPV]);
  }
}

template <typename T, int PH, int PHASE>
__device__ __forceinline__ void write_phase(const T (&x)[64], unsigned char* tilebuf, const TileMap<T, PH>& tm) {
  using Vec = typename Traits<T>::Vec;
  using G = Geo<T, PH>;
  constexpr int EPV = Traits<T>::EPV;
#pragma unroll
  for (int ch = 0; ch < G::CHP; ch++)
    *reinterpret_cast<Vec*>(tilebuf + tm.lds_a[ch & 7] + (ch >> 3) * 1024) = Traits<T>::pack(&x[(PHASE * G::CHP + ch) * EPV]);
}

// ---------------------------------------------------- statistics on the fly --
// calc_data_stat's reductions (util.c:18-25 / :31-38).
template <typename T>
struct StatAcc {
  T mx, mn;
  double sum;                                      // raw-domain sum (or correction term)
  double dcs;                                      // fused path: sum of the blocks' DC coefficients (see k_compress)
  __device__ __forceinline__ void init() { mx = T(0); mn = Traits<T>::huge(); sum = 0.0; dcs = 0.0; }
  // one v_max / v_min with the |x| source modifier each (a NaN operand is skipped, like `a > mx ? a : mx`)
  __device__ __forceinline__ void add(T e, bool in_sum) {
    minmax(e);
    if (in_sum) sum += (double)e;
  }
  __device__ __forceinline__ void minmax(T e) {
    if constexpr (sizeof(T) == 8) {
      asm("v_max_f64 %0, %1, |%2|" : "=v"(mx) : "v"(mx), "v"(e));
      asm("v_min_f64 %0, %1, |%2|" : "=v"(mn) : "v"(mn), "v"(e));
    } else {
      asm("v_max_f32 %0, %1, |%2|" : "=v"(mx) : "v"(mx), "v"(e));
      asm("v_min_f32 %0, %1, |%2|" : "=v"(mn) : "v"(mn), "v"(e));
    }
  }
  // workgroup reduction -> part[3*slot .. 3*slot+2]; `s` is scratch for 3 * (threads/64) doubles
  // dc_scale: raw-domain value of one unit of DC (8 * sf for 64-element orthonormal blocks)
  __device__ __forceinline__ void flush(double* part, unsigned slot, double* s, int nwaves, double dc_scale = 0.0) {
    double dmx = (double)mx, dmn = (double)mn, sm = sum + dcs * dc_scale;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      dmx = fmax(dmx, __shfl_down(dmx, d));
      dmn = fmin(dmn, __shfl_down(dmn, d));
      sm += __shfl_down(sm, d);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s[wave] = dmx; s[nwaves + wave] = dmn; s[2 * nwaves + wave] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < nwaves; w++) { dmx = fmax(dmx, s[w]); dmn = fmin(dmn, s[nwaves + w]); sm += s[2 * nwaves + w]; }
      part[3 * slot + 0] = dmx;
      part[3 * slot + 1] = dmn;
      part[3 * slot + 2] = sm;
    }
  }
};

// QT normalisation of an out-of-range coefficient (dctz-comp-lib.c:488-492 /
// :514-518); error_bound is a double there, so f32 evaluates product and sum in
// double and rounds once.
__device__ __forceinline__ double qt_normalise(double item, double q, double eb, double qf, double rmin, double rmax) {
  if (item < rmin) return (item / q) * eb * qf + rmin;
  if (item > rmax) return (item / q) * eb * qf + rmax;
  return item;
}
__device__ __forceinline__ float qt_normalise(float item, float q, double eb, float qf, float rmin, float rmax) {
  if (item < rmin) return (float)((double)(item / q) * eb * (double)qf + (double)rmin);
  if (item > rmax) return (float)((double)(item / q) * eb * (double)qf + (double)rmax);
  return item;
}
// QT de-normalisation on decode (dctz-decomp-lib.c:404-409 / :450-454)
__device__ __forceinline__ double qt_restore(double v, double q, double eb, double qf, double rmin, double rmax) {
  return (v > 0) ? ((v - rmax) / (eb * qf)) * q : ((v - rmin) / (eb * qf)) * q;
}
__device__ __forceinline__ float qt_restore(float v, float q, double eb, float qf, float rmin, float rmax) {
  return (v > 0) ? (float)(((double)(v - rmax) / (eb * (double)qf)) * (double)q)
                 : (float)(((double)(v - rmin) / (eb * (double)qf)) * (double)q);
}

// The QT table on decode: lane j of a register (pair) holds qtable[j]; entry j -- j a compile-time constant at every use --
// comes out with v_readlane into scalar registers.  (Rounds 2-3 kept the table in LDS: every flagged position of a tile
// then was an LDS round trip inside its branch, with one wave per SIMD nothing to hide it behind -- the QT decoder waited
// 76 % longer than its EC twin for 4 % FEWER vector instructions, profiles/r04_pmc_qt.txt.)
template <typename T> struct QtLanes;
template <> struct QtLanes<double> {
  int lo, hi;
  __device__ __forceinline__ void load(const double* qtab, int lane) { const double v = qtab[lane]; lo = __double2loint(v); hi = __double2hiint(v); }
  __device__ __forceinline__ double at(int j) const { return __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j)); }
};
template <> struct QtLanes<float> {
  int w;
  __device__ __forceinline__ void load(const float* qtab, int lane) { w = __float_as_int(qtab[lane]); }
  __device__ __forceinline__ float at(int j) const { return __int_as_float(__builtin_amdgcn_readlane(w, j)); }
};

// ------------------------------------------------------------- host hand-off --
// System-scope release of a sequence number into the HostBox (fine-grained pinned host
// memory): everything this thread (and, after a barrier, its workgroup) wrote to the
// box before is visible to the polling host thread once it reads the number.
__device__ __forceinline__ void box_publish(volatile unsigned long long* flag, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store(const_cast<unsigned long long*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// reduction of {max, min, sum} partials by one workgroup (any size up to 256 threads); thread 0 returns with the result
__device__ __forceinline__ void reduce_parts(const double* __restrict__ part, int nparts, double& dmx, double& dmn, double& sum) {
  dmx = 0.0; dmn = 1.79769313486231570815e308; sum = 0.0;
  for (int i0 = threadIdx.x; i0 < nparts; i0 += 4 * (int)blockDim.x) {      // four entries in flight per thread
    double a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * (int)blockDim.x;
      const bool in = i < nparts;
      a[u] = in ? part[3 * i] : 0.0; b[u] = in ? part[3 * i + 1] : 1.79769313486231570815e308; c[u] = in ? part[3 * i + 2] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { dmx = fmax(dmx, a[u]); dmn = fmin(dmn, b[u]); sum += c[u]; }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmx = fmax(dmx, __shfl_down(dmx, d));
    dmn = fmin(dmn, __shfl_down(dmn, d));
    sum += __shfl_down(sum, d);
  }
  __shared__ double s[3][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  if (lane == 0) { s[0][wave] = dmx; s[1][wave] = dmn; s[2][wave] = sum; }
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < nw; w++) { dmx = fmax(dmx, s[0][w]); dmn = fmin(dmn, s[1][w]); sum += s[2][w]; }
}

// Sum over a workgroup of one unsigned per thread (every thread gets the total); `sh`: one word per wave
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  __syncthreads();                                   // sh[] of an earlier call is consumed
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  unsigned tot = 0;
  for (int w = 0; w < nw; w++) tot += sh[w];
  return tot;
}

// ------------------------------------------------- binning / bin centres --
// Binning of one coefficient (:363-414; conv_tbl :27-43) in floating point:
//   q = (item - range_min) / bin_width  (the reference's own expression, exact division),  f = floor(q)
//   (== the (t_bin_id) truncation for q >= 0),  conv_tbl[f] = |254.5 - 2 f| - 0.5  for f = 0..254, and
//   >= 255 for f >= 255 or f <= -1, which v_cvt_pk_u8_f32 saturates to 255 = "stored exactly" while it
//   packs the byte.  SAFE = false relies on  item > range_max  =>  q >= 255, which the host verifies
//   for the launch constants ((2 range_max) / bin_width >= 255 in T arithmetic); otherwise the
//   reference's range test is applied explicitly (one compare + one select more per coefficient).
template <typename T, bool SAFE>
__device__ __forceinline__ float bin_value(T item, T q, T range_max) {
  const T f = floor(q);
  const T g = fma_(T(-2), f, T(254.5));
  float h = (float)(fabs(g) - T(0.5));
  if (SAFE) h = (fabs(item) > range_max) ? 255.0f : h;       // == (item < range_min || item > range_max): range_min = -range_max
  return h;
}


// bin_center[b] of gen_bins / gen_bins_f (binning.c:17-23 / :37-43) = (T)(b odd ? b/2 + 1 : -(b/2)) * bin_width, computed
// instead of looked up, in the fp64 kernel (a table in LDS is 63 reads per block with bank conflicts wherever the bin ids
// of a position spread over the 64 blocks of a tile: 36 % of the kernel's LDS cycles, profiles/r02_pmc.txt, r03_pmc.txt).  `w1` holds the four
// magnitudes (b + 1) >> 1 of a dword of bin ids as bytes, `nw` the dword's complement (bit 0 of a byte set <=> b even
// <=> the centre is negative).  The magnitude goes byte -> float in one instruction, the sign is or-ed in, and the
// product passes through "+ (+0)": -0 * bin_width + 0 = +0, the table's value for b = 0.
template <typename T>
__device__ __forceinline__ T bin_centre(const unsigned w1, const unsigned nw, const int i, const T bin_width) {
  const float mag = (float)((w1 >> (8 * i)) & 255u);
  const unsigned sgn = (nw << (31 - 8 * i)) & 0x80000000u;
  const float t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, mag) | sgn);
  if constexpr (sizeof(T) == 8) return __builtin_fma((double)t, bin_width, 0.0);
  else return __builtin_fmaf(t, bin_width, 0.0f);
}


// FastDiv's windows on the host's terms (dctz_shim.hip: divisor_in_window / value_in_window): unbiased exponent of a
// finite non-zero double in [lo, hi)
__device__ __forceinline__ bool exp_in(double v, int lo, int hi) {
  const int e = (int)(((unsigned)__double2hiint(v) >> 20) & 0x7ffu) - 1023;     // subnormal / zero: -1023, inf / NaN: 1024
  return e >= lo && e < hi;
}

}  // namespace dctz
