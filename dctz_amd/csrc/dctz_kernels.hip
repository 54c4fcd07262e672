// dctz_kernels.hip -- gfx950 (MI355X) kernels of the DCTZ hot path.
//
// Work decomposition (both directions):
//   * a TILE is 16 consecutive 64-element blocks (1024 elements, 8 KiB fp64) = the work of
//     ONE wavefront; workgroups of the two big kernels are single wavefronts, the grid is
//     persistent (12 workgroups per CU for fp64, 16 for fp32 = what registers and LDS admit)
//     and workgroup b owns the contiguous tile range [b*ntiles/G, (b+1)*ntiles/G);
//   * the tile is staged in LDS once; HBM is touched with 16-byte-per-lane, fully coalesced
//     accesses only, through buffer descriptors over the workgroup's range (one VGPR of
//     addressing, hardware range check instead of predicates), nt policy on the read-once /
//     write-once streams;
//   * inside the tile a QUAD of lanes owns a block and runs the 64-point DCT of
//     dct64_lane.h in registers, exchanging partners with DPP quad_perm moves;
//   * the ordered stream of "stored exactly" coefficients (AC_exact) is placed by the
//     TWO-LEVEL scheme: every workgroup appends the exceptions of its tiles to its own list
//     and leaves a count, k_scan_tiles turns counts into offsets, k_compact_ac moves the
//     lists; the big kernels have no inter-workgroup traffic.  A single-pass variant
//     (tickets + decoupled look-back, FEAT & F_LOOKBACK) is kept and is byte-identical;
//   * calc_data_stat rides inside k_compress (F_STATS) behind a sampled, verified guess of sf.
//
// Reference code replaced: see include/dctz_hip.h (per entry point) and the
// comment on each kernel.  Built with -ffp-contract=off: the arithmetic that
// the reference does unfused (gcc, baseline x86-64, reference Makefile:2) is
// unfused here too.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dct64_lane.h"
#include "dctz_device.h"

// minimum waves per SIMD the register allocator must leave room for in the two
// big kernels (single-wave workgroups: N waves/SIMD <=> 4 N workgroups per CU)
#ifndef DCTZ_MINWAVES
#define DCTZ_MINWAVES 3
#endif
// Scheduling fence between the stages of the in-register transform: keeps the
// compiler from hoisting the next stage's LDS / DPP operands over the current one
// (which costs tens of VGPRs and, with them, a wave per SIMD).
#ifndef DCTZ_NO_SCHED_FENCE
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() ((void)0)
#endif
// The forward transform is better off WITHOUT them since the descriptor loads freed ~30 VGPRs
// (k_compress 0.281 -> 0.276 ms, still no spills); the inverse keeps them (0.280 vs 0.283 ms).
#ifdef DCTZ_FWD_FENCE
#define FWD_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FWD_FENCE() ((void)0)
#endif

// Exactly the occupancy the LDS budget allows (12 single-wave workgroups per CU for fp64, 16 for
// fp32): with only a lower bound the compiler, seeing 127 VGPRs within reach, trades ILP for a
// fourth wave that the LDS cannot host (k_compress 0.281 -> 0.313 ms).
#define DCTZ_WAVES_PER_EU(T) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? DCTZ_MINWAVES : 4, sizeof(T) == 8 ? DCTZ_MINWAVES : 4)))

namespace dctz {

// ------------------------------------------------------------------ helpers --
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);   // bound_ctrl: no "old" value to initialise (saves a v_mov per move)
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Streaming (read-once / write-once) 16-byte accesses: the `nt` policy.  A pure 1 GiB read stream
// runs at 6.8-7.1 TB/s with nt loads against 6.0-6.3 TB/s with plain ones (tools/ubench/stream_read.hip).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V load_stream(const V* p) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifndef DCTZ_NO_NT
  const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  V v;
  __builtin_memcpy(&v, &r, 16);
  return v;
#else
  return *p;
#endif
}
template <typename V>
__device__ __forceinline__ void store_stream(V* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vectors only");
#ifndef DCTZ_NO_NT_STORE
  u32x4 r;
  __builtin_memcpy(&r, &v, 16);
  __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
#else
  *p = v;
#endif
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

// quad_perm control words: lane i reads from lane perm[i] of its quad
constexpr int QP_XOR1 = 0xB1;     // [1,0,3,2]
constexpr int QP_XOR2 = 0x4E;     // [2,3,0,1]
constexpr int QP_MIRROR = 0x1B;   // [3,2,1,0]
constexpr int QP_0132 = 0xB4;     // [0,1,3,2]

__device__ __forceinline__ unsigned long long to_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned int to_bits(float v) { return __float_as_uint(v); }

template <typename T>
__device__ __forceinline__ int tile_idx(int e) { return (e >> 6) * Traits<T>::PITCH + (e & 63); }

// conv_tbl of dctz-comp-lib.c:27-43 as arithmetic (sign-interleave of t-127)
// t <= 127 ? 254 - 2t : 2t - 255  ==  zigzag(127 - t): (u << 1) ^ (u >> 31), u = 127 - t
__device__ __forceinline__ unsigned conv_bin(unsigned t) {
  const int u = 127 - (int)t;
  return (unsigned)((u << 1) ^ (u >> 31));
}
// The same from nt = -t, as the kernels get it for free from the conversion's negate modifier:
// 2u = 2 nt + 254 is one shift-add, and 2u has the sign of u  (|u| <= 128).
__device__ __forceinline__ unsigned conv_bin_neg(int nt) {
  const int u2 = (nt << 1) + 254;
  return (unsigned)(u2 ^ (u2 >> 31));
}

// Pass-1 binning of one coefficient (dctz-comp-lib.c:363-414).  Returns the bin
// id; *out_of_range tells whether the QT table must see it (:367-373).
template <typename T, typename DIV>
__device__ __forceinline__ unsigned bin_of(T item, T range_min, T range_max, const DIV& bw, bool* out_of_range) {
  // range_min == -range_max exactly (both are +-255 eb rounded once), so the
  // reference's (item < range_min || item > range_max) is one |item| compare
  (void)range_min;
  const bool out = fabs(item) > range_max;
  // in range: 0 <= item - range_min <= 510 eb, far inside the fast window; the
  // quotient of an out-of-range item is never used
  const int nt = (int)(-bw.div_small(item - range_min));  // -(t_bin_id) cast: trunc toward 0 is symmetric
  *out_of_range = out;
  return out ? 255u : conv_bin_neg(nt);              // in range: 0 <= t <= 255 (t = 255 only for item == range_max: bin 255)
}

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
  __device__ __forceinline__ bool in_window(double x) const {
    return ((((unsigned)__double2hiint(x) >> 20) & 0x7ffu) - 523u) <= 1000u;
  }
  __device__ __forceinline__ double slow(double x) const {         // outside the window
    return (ok && x == 0.0) ? x * y : x / d;                       // signed zero / full division
  }
  __device__ __forceinline__ double div(double x) const {          // any x
    const unsigned ex = ((unsigned)__double2hiint(x) >> 20) & 0x7ffu;
    if (ok && (ex - 523u) <= 1000u) return core(x);                // |x| in [2^-500, 2^501)
    if (ok && x == 0.0) return x * y;                              // signed zero
    return x / d;
  }
  // x is zero or inside the window by construction (binning: 0 <= x <= 510 eb)
  __device__ __forceinline__ double div_small(double x) const { return ok ? core(x) : x / d; }
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
  __device__ __forceinline__ bool in_window(float x) const {
    return (((__float_as_uint(x) >> 23) & 0xffu) - 64u) <= 126u;
  }
  __device__ __forceinline__ float slow(float x) const { return (ok && x == 0.0f) ? x * y : x / d; }
  __device__ __forceinline__ float div(float x) const {
    const unsigned ex = (__float_as_uint(x) >> 23) & 0xffu;
    if (ok && (ex - 64u) <= 126u) return core(x);                  // |x| in [2^-63, 2^64)
    if (ok && x == 0.0f) return x * y;
    return x / d;
  }
  __device__ __forceinline__ float div_small(float x) const { return ok ? core(x) : x / d; }
};

template <typename T>
__device__ __forceinline__ unsigned bin_from_quotient(T item, T range_max, T q, bool* out_of_range) {
  const bool out = fabs(item) > range_max;          // == (item < range_min || item > range_max): range_min = -range_max
  const int nt = (int)(-q);                         // -(t_bin_id) cast: trunc toward 0 is symmetric
  *out_of_range = out;
  return out ? 255u : conv_bin_neg(nt);             // in range: 0 <= t <= 255 (t = 255 only for item == range_max: bin 255)
}

// diagnostic phase timers (F_STAMP builds only)
struct Stamps {
  unsigned long long last, acc[8];
  __device__ __forceinline__ void start() { for (int i = 0; i < 8; i++) acc[i] = 0; last = clock64(); }
  __device__ __forceinline__ void mark(int i) { const unsigned long long n = clock64(); acc[i] += n - last; last = n; }
  __device__ __forceinline__ void flush(Ctl* ctl) {
    for (int i = 0; i < 8; i++) atomicAdd(&ctl->dbg[i], acc[i]);
  }
};

// ------------------------------------------------- decoupled look-back scan --
// One 64-bit word per tile: status in the top 2 bits, value in the low 32.
// Single-word relaxed agent-scope accesses need no fences (the datum IS the flag).
constexpr unsigned long long ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = 3ull << 62;
constexpr unsigned SPIN_LIMIT = 1u << 22;

// Called by all 64 lanes of ONE wavefront.  Lane l inspects predecessor
// tile-1-l (then the next 64 further back, ...): the walk to the nearest tile
// whose inclusive prefix is known costs one memory round trip per 64 tiles
// instead of one per tile.  Returns the exclusive prefix (wave-uniform).
// publish_agg = false: the tile's aggregate is already out (software-pipelined
// kernels publish it one iteration before they resolve it); have_first: the
// caller loaded the first window (lane l: desc[tile-1-l]) ahead of time.
__device__ __forceinline__ unsigned lookback(unsigned long long* desc, unsigned tile, unsigned total,
                                             unsigned* err, bool publish_agg = true, bool have_first = false,
                                             unsigned long long d_first = 0) {
  const int lane = threadIdx.x & 63;
  if (publish_agg) {
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&desc[0], ST_PREFIX | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0;
    }
    if (lane == 0) __hip_atomic_store(&desc[tile], ST_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned excl = 0, spins = 0;
  int base = (int)tile;                            // window = tiles [base-64, base-1]
  for (;;) {
    const int idx = base - 1 - lane;
    unsigned long long d = ST_PREFIX;              // before tile 0: prefix 0
    if (have_first) { d = d_first; have_first = false; }
    else if (idx >= 0) d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long ready = __ballot((d & ST_MASK) != 0);
    const unsigned long long pref = __ballot((d & ST_MASK) == ST_PREFIX);
    unsigned long long need = ~0ull;               // lanes whose value we must add
    if (pref) {
      const int f = __ffsll((long long)pref) - 1;  // nearest tile with a known prefix
      need = (f == 63) ? ~0ull : ((2ull << f) - 1ull);
    }
    if ((ready & need) == need) {
      unsigned v = ((need >> lane) & 1ull) ? (unsigned)d : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      excl += v;
      if (pref) break;
      base -= 64;
      spins = 0;
    } else {
      if (++spins >= SPIN_LIMIT) {                 // watchdog: never hang the GPU
        if (lane == 0) atomicExch(err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (lane == 0)
    __hip_atomic_store(&desc[tile], ST_PREFIX | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// Exclusive scan of one count per thread over the 256-thread workgroup, chained
// over tiles by look-back.  Returns this thread's global offset; every thread
// must call it (two barriers inside).  sc: 8 words of LDS scratch.
__device__ __forceinline__ unsigned tile_scan(unsigned cnt, unsigned tile, unsigned ntiles, unsigned* sc,
                                              unsigned long long* desc, Ctl* ctl, bool publish = false,
                                              unsigned publish_value = 0, Stamps* st = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (publish && t == 0) sc[6] = publish_value;     // visible to the workgroup after the first barrier
  unsigned incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) sc[wave] = incl;
  __syncthreads();
  if (wave == 0) {                                   // wave-uniform branch: all 64 lanes look back
    if (st && t == 0) st->mark(4);                   // wave scan + first barrier
    unsigned total = 0;
#pragma unroll
    for (int w = 0; w < WG / 64; w++) total += sc[w];
    const unsigned excl = lookback(desc, tile, total, &ctl->error);
    if (t == 0) {
      sc[4] = excl;
      if (tile == ntiles - 1) ctl->cnt_total = excl + total;
      if (st) st->mark(5);                           // look-back (incl. drain of this wave's VMEM)
    }
  }
  __syncthreads();
  unsigned off = sc[4] + (incl - cnt);
#pragma unroll
  for (int w = 0; w < WG / 64 - 1; w++)
    if (wave > w) off += sc[w];
  return off;
}

// Two-level scheme: workgroup b of G owns the contiguous tiles [lo, hi) -- the same
// partition in k_compress / k_compact_ac and in k_count_tiles / k_decompress.
struct TileRange { unsigned lo, hi; };
__host__ __device__ __forceinline__ TileRange tile_range(unsigned b, unsigned G, unsigned ntiles) {
  const unsigned q = ntiles / G, r = ntiles % G;
  TileRange tr;
  tr.lo = b * q + (b < r ? b : r);
  tr.hi = tr.lo + q + (b < r ? 1u : 0u);
  return tr;
}

// Intra-tile exclusive scan only (two-level scheme): returns this thread's offset
// inside the tile's exception list and the tile total.  One barrier.
__device__ __forceinline__ unsigned tile_scan_local(unsigned cnt, unsigned* sc, unsigned* total) {
  if constexpr (WG == 64) {                        // one wavefront: no LDS, no barrier
    (void)sc;
    const unsigned incl = wave_incl_scan(cnt);
    *total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    return incl - cnt;
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) sc[wave] = incl;
  __syncthreads();
  unsigned off = incl - cnt, sum = 0;
#pragma unroll
  for (int w = 0; w < WG / 64; w++) {
    if (wave > w) off += sc[w];
    sum += sc[w];
  }
  *total = sum;
  return off;
}

// ------------------------------------------------------------ tile tickets --
// Tiles are handed out in increasing order so that the look-back of tile j only
// ever waits for tiles that some running workgroup already owns.  One global
// counter saturates at ~90 tickets/us on MI355X (MI355X_MICROARCH.md, row
// "dequeue"), i.e. ~0.37 ms for the 32 Ki tiles of a 1 GiB shard -- more than the
// whole kernel should take.  F_GROUP therefore splits the counter into `ngroups`
// (<= 8) counters on separate 128-byte lines: workgroup b serves group
// b % ngroups, group g owns tiles g, g + ngroups, g + 2 ngroups, ...  Correct for
// any placement: every group has at least one workgroup (ngroups <= grid), each
// group hands its tiles out in increasing order, and a workgroup never waits
// for a higher tile, so the lowest unfinished tile is always owned or claimable.
// F_LOOKBACK: single-pass kernels (tickets + decoupled look-back); default (0) is the
// two-level scheme: static tiles, tile-local exception lists, tiny scan, compaction.
// F_GROUP: per-group ticket counters (look-back kernels).  F_STAMP: diagnostic phase
// timers into Ctl::dbg (look-back kernels).
// F_STATS: k_compress also computes calc_data_stat's max|x|, min|x| and sum of the RAW input on the way (two-level
// scheme only) -- the host launched it with a scaling factor guessed from a sample and verifies the guess afterwards.

template <int FEAT>
__device__ __forceinline__ unsigned take_ticket(Ctl* ctl, unsigned ngroups) {
  if (FEAT & F_GROUP) {
    const unsigned g = blockIdx.x % ngroups;
    const unsigned k = __hip_atomic_fetch_add(&ctl->gticket[g * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return k * ngroups + g;
  }
  return __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte vector <-> LDS.  With an odd pitch (bank-conflict-free quad accesses,
// see fwd_quad_block) a block base is only element-aligned, so the vector moves
// as individual elements (the compiler pairs them into ds_*2_b64 / ds_*2_b32).
template <typename T>
__device__ __forceinline__ void lds_store_vec(T* p, const typename Traits<T>::Vec& v) {
  if constexpr ((Traits<T>::PITCH * sizeof(T)) % 16 == 0) {
    *reinterpret_cast<typename Traits<T>::Vec*>(p) = v;
  } else {
    T el[Traits<T>::EPV];
    Traits<T>::unpack(v, el);
#pragma unroll
    for (int k = 0; k < Traits<T>::EPV; k++) p[k] = el[k];
  }
}
template <typename T>
__device__ __forceinline__ typename Traits<T>::Vec lds_load_vec(const T* p) {
  if constexpr ((Traits<T>::PITCH * sizeof(T)) % 16 == 0) {
    return *reinterpret_cast<const typename Traits<T>::Vec*>(p);
  } else {
    T el[Traits<T>::EPV];
#pragma unroll
    for (int k = 0; k < Traits<T>::EPV; k++) el[k] = p[k];
    return Traits<T>::pack(el);
  }
}

// --------------------------------------------------------- tile load / store --
// Split form used by the software-pipelined kernels: issue the 16-byte loads of a
// tile into registers (they stay in flight across the compute phase of the
// previous tile), stage them into LDS later.
template <typename T>
__device__ __forceinline__ void issue_tile_loads(typename Traits<T>::Vec (&v)[TILE_ELEMS / Traits<T>::EPV / WG],
                                                 const T* __restrict__ x, unsigned tile_id, unsigned ntiles,
                                                 unsigned nfull) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int t = threadIdx.x;
  if (tile_id >= ntiles) return;
  const unsigned valid = min((unsigned)TILE_BLKS, nfull - tile_id * TILE_BLKS) * 64u;
  const Vec* src = reinterpret_cast<const Vec*>(x + (size_t)tile_id * TILE_ELEMS);
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const unsigned e = (unsigned)(i * WG + t) * EPV;
    if (e < valid) v[i] = load_stream(&src[i * WG + t]);
    else v[i] = Traits<T>::zero();
  }
}

// The same through a buffer descriptor (two-level kernels): one VGPR (lane * 16) addresses all
// vectors of a tile, the tile's offset inside the workgroup's range rides in an SGPR, and the
// range check of the descriptor zero-fills whatever lies beyond the last whole block -- no
// 64-bit per-vector pointers to keep (or spill) across the loop, no per-vector predicate.
template <typename T>
__device__ __forceinline__ void issue_tile_loads_buf(typename Traits<T>::Vec (&v)[TILE_ELEMS / Traits<T>::EPV / WG],
                                                     __amdgpu_buffer_rsrc_t rsrc, unsigned tile_rel) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int voff = (int)threadIdx.x * 16;
  const int soff = (int)(tile_rel * (unsigned)(TILE_ELEMS * sizeof(T)));
#pragma unroll
  for (int i = 0; i < NV; i++) {
    constexpr int STEP = WG * 16;                    // bytes between a lane's consecutive vectors
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ((i * STEP) & 4095), soff + ((i * STEP) & ~4095), 2 /* nt */);
    __builtin_memcpy(&v[i], &r, 16);
  }
}

// ---------------------------------------------------- statistics on the fly --
// calc_data_stat's three reductions (util.c:18-25 / :31-38) over the vectors a
// thread has just loaded for a tile; `skip0`: the vector holds x[0], which the
// reference's loop (i = 1 ...) never adds to the sum.
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

// LEVEL 2: the host saw min|x| and max|x| of the whole array inside FastDiv's
// window (k_stats), so every element takes the 3-operation path, no test.
// LEVEL 1: window unknown (zeros, extreme exponents possible): per-element test,
// exact fallback.  LEVEL 0: divisor outside the window: plain IEEE division.
// STATS: calc_data_stat's max|x| / min|x| over the raw vectors on the way (speculative launch);
// invalid vectors are guarded by `valid`.  The sum costs nothing here: an orthonormal 64-point
// DCT has DC = (sum of the block)/8, so sum(x) = 8 sf * sum(DC) (emit_tile adds the DCs up; the
// sum is tree-order in either path, only its decimal digits go into the header's `mean`).
// skip0: v[0] of thread 0 starts with x[0], which util.c:22 never adds.
template <typename T, bool SCALE, int LEVEL, bool STATS, bool FULL = false>
__device__ __forceinline__ void stage_tile_l(T* tile, typename Traits<T>::Vec (&v)[TILE_ELEMS / Traits<T>::EPV / WG],
                                             size_t ebase, unsigned valid, const FastDiv<T>& sfd, T* scaled,
                                             StatAcc<T>* acc, bool skip0) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const unsigned e = (unsigned)(i * WG + t) * EPV;
    Vec a = v[i];
    if (STATS) {
      T el[EPV];
      Traits<T>::unpack(a, el);
      if (FULL || e < valid) {                     // FULL: a whole tile (all but the array's last), no per-vector branch
#pragma unroll
        for (int k = 0; k < EPV; k++) acc->minmax(el[k]);
      }
      if (i == 0 && skip0) acc->sum -= (double)el[0];
    }
    if (SCALE) {
      T el[EPV];
      Traits<T>::unpack(a, el);
#pragma unroll
      for (int k = 0; k < EPV; k++)                                // dctz-comp-lib.c:197-199 / :212-214
        el[k] = (LEVEL == 2) ? sfd.core(el[k]) : (LEVEL == 1) ? sfd.div(el[k]) : el[k] / sfd.d;
      a = Traits<T>::pack(el);
      if (scaled != nullptr && e < valid) reinterpret_cast<Vec*>(scaled + ebase)[i * WG + t] = a;
    }
    // element (i*WG + t)*EPV lives at block i*(WG*EPV/64) + (t*EPV >> 6): one base
    // address per thread plus a compile-time stride (immediate offsets)
    lds_store_vec<T>(tile + tile_idx<T>(t * EPV) + i * (WG * EPV / 64) * Traits<T>::PITCH, a);
  }
}

template <typename T, bool SCALE, bool STATS = false>
__device__ __forceinline__ void stage_tile(T* tile, typename Traits<T>::Vec (&v)[TILE_ELEMS / Traits<T>::EPV / WG],
                                           size_t ebase, unsigned valid, const FastDiv<T>& sfd, T* scaled, unsigned level,
                                           StatAcc<T>* acc = nullptr, bool skip0 = false) {
  if (STATS && level == 2 && valid == (unsigned)TILE_ELEMS) stage_tile_l<T, SCALE, 2, STATS, true>(tile, v, ebase, valid, sfd, scaled, acc, skip0);
  else if (level == 2) stage_tile_l<T, SCALE, 2, STATS>(tile, v, ebase, valid, sfd, scaled, acc, skip0);
  else if (level == 1) stage_tile_l<T, SCALE, 1, STATS>(tile, v, ebase, valid, sfd, scaled, acc, skip0);
  else stage_tile_l<T, SCALE, 0, STATS>(tile, v, ebase, valid, sfd, scaled, acc, skip0);
}

template <typename T, bool SCALE>
__device__ __forceinline__ void load_tile(T* tile, const T* __restrict__ x, size_t ebase, unsigned valid,
                                          T sf, T* scaled) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int t = threadIdx.x;
  const Vec* src = reinterpret_cast<const Vec*>(x + ebase);
  Vec v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const unsigned e = (unsigned)(i * WG + t) * EPV;
    if (e < valid) v[i] = src[i * WG + t];
    else v[i] = Traits<T>::zero();
  }
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const unsigned e = (unsigned)(i * WG + t) * EPV;
    if (SCALE) {
      Traits<T>::div(v[i], sf);
      if (scaled != nullptr && e < valid) reinterpret_cast<Vec*>(scaled + ebase)[i * WG + t] = v[i];
    }
    lds_store_vec<T>(tile + tile_idx<T>(t * EPV) + i * (WG * EPV / 64) * Traits<T>::PITCH, v[i]);
  }
}

template <typename T, bool SCALE>
__device__ __forceinline__ void store_tile(const T* tile, T* __restrict__ out, size_t ebase, unsigned valid, T sf) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int t = threadIdx.x;
  Vec* dst = reinterpret_cast<Vec*>(out + ebase);
  // every vector in its own registers before the first store issues: with several 16-byte stores
  // queued, an LDS read that reused the registers of a store two back was seen to land before that
  // store had read its data (k_decompress, QT mode) -- see store_tile_buf
  Vec v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    v[i] = lds_load_vec<T>(tile + tile_idx<T>(t * EPV) + i * (WG * EPV / 64) * Traits<T>::PITCH);
    if (SCALE) Traits<T>::mul(v[i], sf);          // dctz-decomp-lib.c:494-511
  }
  SCHED_FENCE();
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const unsigned e = (unsigned)(i * WG + t) * EPV;
    if (e < valid) store_stream(&dst[i * WG + t], v[i]);
  }
  SCHED_FENCE();
}

// store_tile through a buffer descriptor over the workgroup's output range: nt policy, and the
// range check drops whatever lies beyond the last whole block (no predicate, no 64-bit pointers).
template <typename T, bool SCALE>
__device__ __forceinline__ void store_tile_buf(const T* tile, __amdgpu_buffer_rsrc_t rsrc, unsigned tile_rel, T sf) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  const int t = threadIdx.x;
  const int voff = t * 16;
  const int soff = (int)(tile_rel * (unsigned)(TILE_ELEMS * sizeof(T)));
  Vec v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    v[i] = lds_load_vec<T>(tile + tile_idx<T>(t * EPV) + i * (WG * EPV / 64) * Traits<T>::PITCH);
    if (SCALE) Traits<T>::mul(v[i], sf);          // dctz-decomp-lib.c:494-511
  }
  SCHED_FENCE();                                  // every vector in its own registers before the first store issues
#pragma unroll
  for (int i = 0; i < NV; i++) {
    constexpr int STEP = WG * 16;
    u32x4 r;
    __builtin_memcpy(&r, &v[i], 16);
    __builtin_amdgcn_raw_buffer_store_b128(r, rsrc, voff + ((i * STEP) & 4095), soff + ((i * STEP) & ~4095), 2 /* nt */);
  }
  SCHED_FENCE();
}

// ------------------------------------------------------- in-tile transforms --
// Forward DCT-II of the 64 blocks of the tile, in place in LDS (dct.c:55-103).
// Two workgroup barriers inside (after reads, after writes).
// Quad -> block map of the forward transform.  With an odd pitch a quad's reads
// hit elements blk + 4*lane + const (mod 32 banks / bank pairs); giving the 8 quads
// of a half-wave the blocks {0,1,2,3,16,17,18,19} + 4*(h&3) + 32*(h>>2) tiles all
// 32.  (The inverse reads 8q' + k1, for which consecutive blocks already tile.)
__device__ __forceinline__ int fwd_quad_block(int quad) {
#if (DCTZ_PITCH % 2) == 1
  const int h = quad >> 3, k = quad & 7;
  return (k & 3) | ((h & 3) << 2) | ((k >> 2) << 4) | ((h >> 2) << 5);
#else
  return quad;
#endif
}

template <typename T>
__device__ __forceinline__ void tile_dct_fwd(T* tile, const T* tab) {
  const int t = threadIdx.x, blk = fwd_quad_block(t >> 2), lane = t & 3;
  T* b = tile + blk * Traits<T>::PITCH;
  T yr[8], yi[8];
#pragma unroll
  for (int n1 = 0; n1 < 8; n1++) {
    yr[n1] = b[pack_pos(4 * n1 + lane, 0)];
    yi[n1] = b[pack_pos(4 * n1 + lane, 1)];
  }
  __syncthreads();                                 // every lane has read its inputs
  fwd_stage_lane<T>(yr, yi, lane, tab);
  FWD_FENCE();
  {
    const T s = (lane & 2) ? T(-1) : T(1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const T r = bfly_s(yr[k], dpp<QP_XOR2>(yr[k]), s), i = bfly_s(yi[k], dpp<QP_XOR2>(yi[k]), s);
      if (lane == 3) { yr[k] = i; yi[k] = -r; }    // times -i
      else { yr[k] = r; yi[k] = i; }
    }
  }
  FWD_FENCE();
  {
    const T s = (lane & 1) ? T(-1) : T(1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      yr[k] = bfly_s(yr[k], dpp<QP_XOR1>(yr[k]), s);
      yi[k] = bfly_s(yi[k], dpp<QP_XOR1>(yi[k]), s);
    }
  }
  FWD_FENCE();
  const int q = lane_q(lane);
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {                 // split + twiddle, results straight to LDS
    T pr, pi, lo, hi;
    if (k1 == 0) { pr = dpp<QP_0132>(yr[0]); pi = dpp<QP_0132>(yi[0]); }
    else { pr = dpp<QP_MIRROR>(yr[8 - k1]); pi = dpp<QP_MIRROR>(yi[8 - k1]); }
    fwd_split_one<T>(k1, yr[k1], yi[k1], pr, pi, lane, tab, lo, hi);
    b[8 * q + k1] = lo;
    b[(k1 == 0 && lane == 0) ? 32 : 64 - (8 * q + k1)] = hi;
    FWD_FENCE();
  }
  __syncthreads();
}

// Inverse DCT-III of the 64 blocks of the tile, in place in LDS (dct.c:115-205).
template <typename T>
__device__ __forceinline__ void tile_dct_inv(T* tile, const T* tab) {
  const int t = threadIdx.x, blk = t >> 2, lane = t & 3;
  T* b = tile + blk * Traits<T>::PITCH;
  const int q = lane_q(lane);
  T lo[8], hi[8], gr[8], gi[8], zr[8], zi[8], g32r, g32i;
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    lo[k1] = b[8 * q + k1];
    hi[k1] = b[(k1 == 0 && lane == 0) ? 32 : 64 - (8 * q + k1)];
  }
  __syncthreads();                                 // every lane has read its inputs
  inv_prepare<T>(lo, hi, lane, tab, gr, gi, g32r, g32i);
  SCHED_FENCE();
  {
    T pr[8], pi[8];
    pr[0] = dpp<QP_0132>(gr[0]); pi[0] = dpp<QP_0132>(gi[0]);
    if (lane == 0) { pr[0] = g32r; pi[0] = g32i; }
#pragma unroll
    for (int k = 1; k < 8; k++) { pr[k] = dpp<QP_MIRROR>(gr[8 - k]); pi[k] = dpp<QP_MIRROR>(gi[8 - k]); }
    inv_merge<T>(gr, gi, pr, pi, lane, tab, zr, zi);
  }
  SCHED_FENCE();
  {
    const T s = (lane & 1) ? T(-1) : T(1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const T r = bfly_s(zr[k], dpp<QP_XOR1>(zr[k]), s), i = bfly_s(zi[k], dpp<QP_XOR1>(zi[k]), s);
      if (lane == 3) { zr[k] = -i; zi[k] = r; }    // times +i
      else { zr[k] = r; zi[k] = i; }
    }
  }
  SCHED_FENCE();
  {
    const T s = (lane & 2) ? T(-1) : T(1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      zr[k] = bfly_s(zr[k], dpp<QP_XOR2>(zr[k]), s);
      zi[k] = bfly_s(zi[k], dpp<QP_XOR2>(zi[k]), s);
    }
  }
  SCHED_FENCE();
  inv_stage_lane<T>(zr, zi, lane, tab);
  SCHED_FENCE();
#pragma unroll
  for (int n1 = 0; n1 < 8; n1++) {
    b[pack_pos(4 * n1 + lane, 0)] = zr[n1];
    b[pack_pos(4 * n1 + lane, 1)] = zi[n1];
  }
  __syncthreads();
}

template <typename T>
__device__ __forceinline__ void load_tab(T* tab, const T* __restrict__ gtab) {
  for (int i = threadIdx.x; i < TAB_SIZE; i += WG) tab[i] = gtab[i];
}

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

// ------------------------------------------------------------- host hand-off --
// System-scope release of a sequence number into the HostBox (fine-grained pinned host
// memory): everything this thread (and, after a barrier, its workgroup) wrote to the
// box before is visible to the polling host thread once it reads the number.
__device__ __forceinline__ void box_publish(volatile unsigned long long* flag, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store(const_cast<unsigned long long*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// =============================================================== statistics ==
// calc_data_stat (util.c:12-44): max|x|, min|x| and sum (x[0] is never added,
// util.c:22 starts at i = 1).  Tree order: `sum` is NOT the reference's serial
// order (it is never used by the codec; the host wrapper recomputes it
// serially for the header).
template <typename T>
__global__ __launch_bounds__(SWG) void k_stats(const T* __restrict__ x, size_t n, double* __restrict__ part) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nvec = n / EPV;
  const Vec* src = reinterpret_cast<const Vec*>(x);
  T mx = T(0), mn = Traits<T>::huge();
  double sum = 0.0;
  constexpr int UN = 4;                            // each workgroup streams 16 KiB contiguous per trip
  for (size_t i0 = (size_t)blockIdx.x * SWG * UN + threadIdx.x; i0 < nvec; i0 += (size_t)gridDim.x * SWG * UN) {
    Vec v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const size_t i = i0 + (size_t)u * SWG;
      v[u] = load_stream((i < nvec) ? &src[i] : &src[i0]);   // a repeated vector changes neither max nor min
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const size_t i = i0 + (size_t)u * SWG;
      T e[EPV];
      Traits<T>::unpack(v[u], e);
#pragma unroll
      for (int k = 0; k < EPV; k++) {
        const T a = fabs(e[k]);
        mx = a > mx ? a : mx;
        mn = a < mn ? a : mn;
        if (i < nvec && (i != 0 || k != 0)) sum += (double)e[k];
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = nvec * EPV; i < n; i++) {
      const T a = fabs(x[i]);
      mx = a > mx ? a : mx;
      mn = a < mn ? a : mn;
      if (i != 0) sum += (double)x[i];
    }
  double dmx = (double)mx, dmn = (double)mn;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmx = fmax(dmx, __shfl_down(dmx, d));
    dmn = fmin(dmn, __shfl_down(dmn, d));
    sum += __shfl_down(sum, d);
  }
  __shared__ double s[3][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = dmx; s[1][wave] = dmn; s[2][wave] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SWG / 64; w++) { dmx = fmax(dmx, s[0][w]); dmn = fmin(dmn, s[1][w]); sum += s[2][w]; }
    part[3 * blockIdx.x + 0] = dmx;
    part[3 * blockIdx.x + 1] = dmn;
    part[3 * blockIdx.x + 2] = sum;
  }
}

__global__ __launch_bounds__(SWG) void k_stats_final(const double* __restrict__ part, int nparts, double* __restrict__ out,
                                                    HostBox* box, unsigned long long seq) {
  double dmx = 0.0, dmn = 1.79769313486231570815e308, sum = 0.0;
  for (int i = threadIdx.x; i < nparts; i += SWG) {
    dmx = fmax(dmx, part[3 * i]); dmn = fmin(dmn, part[3 * i + 1]); sum += part[3 * i + 2];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmx = fmax(dmx, __shfl_down(dmx, d));
    dmn = fmin(dmn, __shfl_down(dmn, d));
    sum += __shfl_down(sum, d);
  }
  __shared__ double s[3][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = dmx; s[1][wave] = dmn; s[2][wave] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SWG / 64; w++) { dmx = fmax(dmx, s[0][w]); dmn = fmin(dmn, s[1][w]); sum += s[2][w]; }
    out[0] = dmx; out[1] = dmn; out[2] = sum;
    if (box != nullptr) {                            // hand the three numbers straight to the polling host thread
      box->stats[0] = dmx; box->stats[1] = dmn; box->stats[2] = sum;
      box_publish(&box->seq_stats, seq);
    }
  }
}

// Last kernel of a compress / decompress call (two-level scheme): final reduction of the
// fused statistics (nparts > 0), results -> host box, control block back to all-zero for
// the next call (replaces its hipMemsetAsync), then the sequence number.
__global__ __launch_bounds__(SWG) void k_finish(Ctl* ctl, const double* __restrict__ part, int nparts, HostBox* box,
                                               unsigned long long seq) {
  const int t = threadIdx.x;
  if (nparts > 0) {
    double dmx = 0.0, dmn = 1.79769313486231570815e308, sum = 0.0;
    for (int i = t; i < nparts; i += SWG) {
      dmx = fmax(dmx, part[3 * i]); dmn = fmin(dmn, part[3 * i + 1]); sum += part[3 * i + 2];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      dmx = fmax(dmx, __shfl_down(dmx, d));
      dmn = fmin(dmn, __shfl_down(dmn, d));
      sum += __shfl_down(sum, d);
    }
    __shared__ double s[3][SWG / 64];
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) { s[0][wave] = dmx; s[1][wave] = dmn; s[2][wave] = sum; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < SWG / 64; w++) { dmx = fmax(dmx, s[0][w]); dmn = fmin(dmn, s[1][w]); sum += s[2][w]; }
      box->fstats[0] = dmx; box->fstats[1] = dmn; box->fstats[2] = sum;
    }
  }
  if (t < 64) box->qraw[t] = ctl->qraw[t];
  if (t == 0) { box->cnt_total = ctl->cnt_total; box->error = ctl->error; box->q0 = ctl->q0; }
  __threadfence_system();
  __syncthreads();                                   // all box writes issued and fenced; all ctl reads done
  unsigned long long* w = reinterpret_cast<unsigned long long*>(ctl);
  for (int i = t; i < (int)(sizeof(Ctl) / 8); i += SWG) w[i] = 0ull;
  if (t == 0) box_publish(&box->seq_done, seq);
}

// Sampled statistics for the speculative path: one 4 KiB chunk out of every group of
// `group` chunks, at a hashed position inside the group (a fixed stride would alias with
// the row structure of power-of-two volumes).  Same partials layout as k_stats.
template <typename T>
__global__ __launch_bounds__(SWG) void k_stats_sample(const T* __restrict__ x, size_t n, unsigned group,
                                                       double* __restrict__ part) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nchunks = n / ((size_t)SWG * EPV);                // whole chunks only; the tail is never sampled
  const size_t ngroups = nchunks / group;
  const Vec* src = reinterpret_cast<const Vec*>(x);
  StatAcc<T> acc;
  acc.init();
  for (size_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const unsigned h = ((unsigned)g * 2654435761u) >> 8;
    const size_t chunk = g * group + h % group;
    T e[EPV];
    Traits<T>::unpack(load_stream(&src[chunk * SWG + threadIdx.x]), e);
#pragma unroll
    for (int k = 0; k < EPV; k++) acc.add(e[k], true);
  }
  __shared__ double ss[3 * (SWG / 64)];
  acc.flush(part, blockIdx.x, ss, SWG / 64);
}

// Serial-order sum for the header's `mean` (util.c:18-28 / :31-41): the reference
// adds x[1..N-1] one after the other in the data type, and a tree reduction
// cannot reproduce those roundings.  One wavefront: all lanes stage a chunk in
// LDS with coalesced loads, lane 0 adds it up in index order.  Slow by design
// (one dependent add per element) and OFF the critical path: the host wrapper
// runs it on a side stream underneath the zlib tail.
template <typename T>
__global__ __launch_bounds__(64) void k_serial_sum(const T* __restrict__ x, size_t n, double* __restrict__ out) {
  constexpr int CH = 4096;
  __shared__ T buf[CH];
  const int lane = threadIdx.x;
  T sum = T(0);
  for (size_t base = 0; base < n; base += CH) {
    const size_t m = (n - base < (size_t)CH) ? n - base : (size_t)CH;
    for (size_t i = lane; i < m; i += 64) buf[i] = x[base + i];
    __syncthreads();
    if (lane == 0) {
      size_t i = (base == 0) ? 1 : 0;          // util.c:22: the loop starts at i = 1
      for (; i < m; i++) sum += buf[i];
    }
    __syncthreads();
  }
  if (lane == 0) out[0] = (double)sum;
}

// x[i] /= sf in place (dctz-comp-lib.c:193-216), for callers that need the
// reference's in-place side effect on their own buffer.
template <typename T>
__global__ __launch_bounds__(SWG) void k_scale(T* __restrict__ x, size_t n, T sf) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nvec = n / EPV;
  Vec* v = reinterpret_cast<Vec*>(x);
  for (size_t i = (size_t)blockIdx.x * SWG + threadIdx.x; i < nvec; i += (size_t)gridDim.x * SWG) {
    Vec a = v[i];
    Traits<T>::div(a, sf);
    v[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = nvec * EPV; i < n; i++) x[i] = x[i] / sf;
}

// ================================================================= compress ==
// Fused: scale (dctz-comp-lib.c:193-216) -> DCT-II per block (:337-340, dct.c:55-103)
// -> DC (:350-351) -> pass-1 binning (:361-414) -> ordered exception stream
// (:478-544) [-> QT per-position max (:371-372)], full 64-element blocks only.
// emit phase of one tile: thread t owns elements [16t, 16t+16)
// Two-level kernels: the outputs of a workgroup's tile range behind buffer descriptors (range
// checks instead of predicates, 32-bit offsets instead of 64-bit pointers per store).
struct EmitBufs {
  __amdgpu_buffer_rsrc_t bin, dc, ac, qi, qj;      // bin_index / DC of the range; the workgroup's exception list(s)
  unsigned tile_rel;                               // tile index inside the range
};

template <typename T, int MODE, int FEAT>
__device__ __forceinline__ void emit_tile(const FwdParams<T>& p, const T* tile, typename Traits<T>::Bits* qmax,
                                          unsigned* sc, const FastDiv<T>& bwd, unsigned tile_id, unsigned blks_here,
                                          bool publish, unsigned publish_value, Stamps* st, unsigned list_base = 0,
                                          unsigned* run = nullptr, double* dcs = nullptr, const EmitBufs* eb = nullptr) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const int t = threadIdx.x;
  const size_t ebase = (size_t)tile_id * TILE_ELEMS;
  const int blk = t >> 2, j0 = (t & 3) * 16;
  const bool active = (unsigned)blk < blks_here;
  T c[16];
#pragma unroll
  for (int i = 0; i < 16 / EPV; i++) {
    const Vec cv = lds_load_vec<T>(&tile[blk * Traits<T>::PITCH + j0 + i * EPV]);
    Traits<T>::unpack(cv, &c[i * EPV]);
  }
  unsigned w[4] = {0, 0, 0, 0};
  unsigned mask = 0;
  auto bin_loop = [&](auto fast) {                   // divisor test hoisted: a launch constant
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const T u = c[i] - p.range_min;                // 0 <= u <= 510 eb whenever the bin is used
      const T q = decltype(fast)::value ? bwd.core(u) : u / bwd.d;
      bool out;
      unsigned b = bin_from_quotient<T>(c[i], p.range_max, q, &out);
      const int j = j0 + i;
      if (j == 0) { b = 255u; out = false; }       // :361 DC slot
      else if (b == 255u) mask |= 1u << i;
      if (MODE == DCTZHIP_QT && out && active) atomicMax(&qmax[j], to_bits(fabs(c[i])));
      w[i >> 2] |= b << (8 * (i & 3));
    }
  };
  if (bwd.ok) bin_loop(std::true_type{}); else bin_loop(std::false_type{});
  if (!active) mask = 0;
  if ((FEAT & F_STAMP) && st && t == 0) st->mark(3);   // binning + bin store issue
  unsigned r;
  if (FEAT & F_LOOKBACK) {
    r = tile_scan((unsigned)__popc(mask), tile_id, p.ntiles, sc, p.desc, p.ctl, publish, publish_value,
                  (FEAT & F_STAMP) ? st : nullptr);
  } else {                                           // workgroup-local list; k_compact_ac places it later
    unsigned total;
    r = *run + tile_scan_local((unsigned)__popc(mask), sc, &total);     // index inside the workgroup's list
    *run += total;
    // stores through the range descriptors (inactive blocks fall outside them and are dropped)
    u32x4 wq = {w[0], w[1], w[2], w[3]};
    __builtin_amdgcn_raw_buffer_store_b128(wq, eb->bin, t * 16, (int)(eb->tile_rel * (unsigned)TILE_ELEMS), 0);
    if (p.coef != nullptr && active) {
#pragma unroll
      for (int i = 0; i < 16 / EPV; i++)
        reinterpret_cast<Vec*>(p.coef + ebase + (size_t)t * 16)[i] = Traits<T>::pack(&c[i * EPV]);
    }
    if (j0 == 0) {
      const unsigned gblk = tile_id * TILE_BLKS + blk;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)c[0]), eb->dc, blk * 4,
                                            (int)(eb->tile_rel * (unsigned)(TILE_BLKS * 4)), 0);     // :350-351 USE_TRUNCATE
      if (FEAT & F_STATS) { if (active) *dcs += (double)c[0]; }
      if (active && p.last_is_full && gblk == p.nfull - 1) p.ctl->q0 = (unsigned long long)to_bits(c[0]);   // :355-360
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (mask & (1u << i)) {
        if (MODE == DCTZHIP_EC) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)c[i]), eb->ac, (int)(r * 4u), 0, 0);   // :535-537
        } else {
          if constexpr (sizeof(T) == 8) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, c[i]), eb->qi, (int)(r * 8u), 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[i]), eb->qi, (int)(r * 4u), 0, 0);
          }
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(j0 + i), eb->qj, (int)r, 0, 0);
        }
        r++;
      }
    }
    return;
  }
  // all global stores of the tile go out AFTER the look-back, so that the polling
  // wave's vmcnt(0) waits never sit behind its own stores
  if (active) {
    reinterpret_cast<uint4*>(p.bin + ebase)[t] = make_uint4(w[0], w[1], w[2], w[3]);
    if (p.coef != nullptr) {
#pragma unroll
      for (int i = 0; i < 16 / EPV; i++)
        reinterpret_cast<Vec*>(p.coef + ebase + (size_t)t * 16)[i] = Traits<T>::pack(&c[i * EPV]);
    }
    if (j0 == 0) {
      const unsigned gblk = tile_id * TILE_BLKS + blk;
      p.dc[gblk] = (float)c[0];                  // :350-351 USE_TRUNCATE
      if (FEAT & F_STATS) *dcs += (double)c[0];
      if (p.last_is_full && gblk == p.nfull - 1) p.ctl->q0 = (unsigned long long)to_bits(c[0]);   // :355-360
    }
  }
  float* acdst = (FEAT & F_LOOKBACK) ? p.ac : p.ac_tmp;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (mask & (1u << i)) {
      if (MODE == DCTZHIP_EC) acdst[r] = (float)c[i];           // :535-537
      else { p.qt_item[r] = c[i]; p.qt_j[r] = (uint8_t)(j0 + i); }
      r++;
    }
  }
}

template <typename T, int MODE, bool SCALE, int FEAT>
__global__ __launch_bounds__(WG) DCTZ_WAVES_PER_EU(T) void k_compress(FwdParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Vec = typename Traits<T>::Vec;
  using Bits = typename Traits<T>::Bits;
  constexpr int EPV = Traits<T>::EPV, NV = TILE_ELEMS / EPV / WG;
  T* tile = reinterpret_cast<T*>(smem);
  T* tab = tile + TILE_BLKS * Traits<T>::PITCH;
  Bits* qmax = reinterpret_cast<Bits*>(tab + TAB_SIZE);
  unsigned* sc = reinterpret_cast<unsigned*>(qmax + 64);
  const int t = threadIdx.x;
  load_tab<T>(tab, p.tab);
  if (MODE == DCTZHIP_QT && t < 64) qmax[t] = 0;
  FastDiv<T> sfd, bwd;
  sfd.init(p.sf, p.fast_sf != 0);
  bwd.init(p.bin_width, p.fast_bw != 0);

  if constexpr ((FEAT & F_LOOKBACK) == 0) {
    // two-level scheme: tiles are assigned statically (no inter-workgroup traffic
    // at all in this kernel); every tile leaves its exceptions as a tile-local
    // list + a count, k_scan_tiles / k_compact_ac stitch them into AC_exact[]
    const TileRange tr = tile_range(blockIdx.x, gridDim.x, p.ntiles);
    const unsigned list_base = tr.lo * TILE_ELEMS;   // this workgroup's exception list lives in its tiles' slots
    unsigned run = 0;                                // its length so far (uniform over the workgroup)
    StatAcc<T> acc;
    if (FEAT & F_STATS) acc.init();
    // The input of the workgroup's tile range sits behind one buffer descriptor; the loads of tile
    // k+1 are issued as soon as tile k has been staged, so they have the whole transform + emit phase
    // to land.  With descriptor addressing the 32 extra live VGPRs fit (164-168, no spills) and the
    // prefetch is worth 1-2 %; with 64-bit pointers it spilled and serialised the loads behind scratch
    // reloads (-5 %), issued only before the emit phase it was a wash.  DESIGN.md section 6.
    const size_t first_el = (size_t)tr.lo * TILE_ELEMS;
    const size_t end_el = min((size_t)p.nfull * 64, (size_t)tr.hi * TILE_ELEMS);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(p.x + first_el), 0, tr.lo < tr.hi ? (int)((end_el - first_el) * sizeof(T)) : 0, 0x00020000);
    EmitBufs eb;
    {
      const int range_el = tr.lo < tr.hi ? (int)(end_el - first_el) : 0;     // whole blocks only
      eb.bin = __builtin_amdgcn_make_buffer_rsrc(p.bin + first_el, 0, range_el, 0x00020000);
      eb.dc = __builtin_amdgcn_make_buffer_rsrc(p.dc + first_el / 64, 0, range_el / 64 * 4, 0x00020000);
      // the workgroup's list lives in the slots of its own tiles (cannot outgrow them: <= 63 per block)
      const int list_el = (int)((size_t)(tr.hi - tr.lo) * TILE_ELEMS);
      eb.ac = __builtin_amdgcn_make_buffer_rsrc(p.ac_tmp + list_base, 0, list_el * 4, 0x00020000);
      eb.qi = __builtin_amdgcn_make_buffer_rsrc(p.qt_item + list_base, 0, MODE == DCTZHIP_QT ? list_el * (int)sizeof(T) : 0, 0x00020000);
      eb.qj = __builtin_amdgcn_make_buffer_rsrc(p.qt_j + list_base, 0, MODE == DCTZHIP_QT ? list_el : 0, 0x00020000);
    }
    Vec v[NV];
    if (tr.lo < tr.hi) issue_tile_loads_buf<T>(v, rsrc, 0u);
    for (unsigned tile_id = tr.lo; tile_id < tr.hi; tile_id++) {
      const size_t ebase = (size_t)tile_id * TILE_ELEMS;
      const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile_id * TILE_BLKS);
      __syncthreads();                               // previous tile's LDS reads are done
      stage_tile<T, SCALE, (FEAT & F_STATS) != 0>(tile, v, ebase, blks_here * 64u, sfd, p.scaled, p.fast_sf, &acc,
                                                  tile_id == 0 && t == 0);
      __syncthreads();
      SCHED_FENCE();
      if (tile_id + 1 < tr.hi) issue_tile_loads_buf<T>(v, rsrc, tile_id + 1 - tr.lo);
      SCHED_FENCE();
      tile_dct_fwd<T>(tile, tab);
      eb.tile_rel = tile_id - tr.lo;
      emit_tile<T, MODE, FEAT>(p, tile, qmax, sc, bwd, tile_id, blks_here, false, 0u, nullptr, list_base, &run, &acc.dcs, &eb);
    }
    if (t == 0) p.tile_cnt[blockIdx.x] = run;
    if (FEAT & F_STATS) {
      __syncthreads();                               // the tile buffer is free: scratch for the reduction
      acc.flush(p.stat_part, blockIdx.x, reinterpret_cast<double*>(tile), WG / 64, SCALE ? 8.0 * (double)p.sf : 8.0);
    }
  } else {
    Stamps st;
    if (FEAT & F_STAMP) st.start();
    for (;;) {
      __syncthreads();                               // tile + sc[] free for reuse
      if (t == 0) sc[5] = take_ticket<FEAT>(p.ctl, p.ngroups);
      __syncthreads();
      if ((FEAT & F_STAMP) && t == 0) st.mark(0);    // ticket (+ drain of own stores)
      const unsigned tile_id = sc[5];
      if (tile_id >= p.ntiles) break;
      const size_t ebase = (size_t)tile_id * TILE_ELEMS;
      const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile_id * TILE_BLKS);
      Vec v[NV];
      issue_tile_loads<T>(v, p.x, tile_id, p.ntiles, p.nfull);
      stage_tile<T, SCALE>(tile, v, ebase, blks_here * 64u, sfd, p.scaled, p.fast_sf);
      __syncthreads();
      if ((FEAT & F_STAMP) && t == 0) st.mark(1);    // load + stage
      tile_dct_fwd<T>(tile, tab);
      if ((FEAT & F_STAMP) && t == 0) st.mark(2);    // DCT
      emit_tile<T, MODE, FEAT>(p, tile, qmax, sc, bwd, tile_id, blks_here, false, 0u, &st);
      if ((FEAT & F_STAMP) && t == 0) st.mark(6);    // AC writes
    }
    if ((FEAT & F_STAMP) && t == 0) st.flush(p.ctl);
  }
  if (MODE == DCTZHIP_QT) {
    __syncthreads();
    if (t < 64 && qmax[t] != 0) atomicMax(&p.ctl->qraw[t], (unsigned long long)qmax[t]);
  }
}

// The last, short block (length l = N % 64): the reference re-plans a length-l
// (l even) or 2l (l odd) FFT for it (dctz-comp-lib.c:326-336, dct.c:59-72).
// One wavefront, definition-order DFT with host-built roots.
template <typename T, int MODE, bool SCALE>
__global__ __launch_bounds__(64) void k_compress_rem(FwdParams<T> p, int l) {
  __shared__ T v[128];
  const int k = threadIdx.x;
  const size_t base = (size_t)p.nfull * 64;
  const T* rt = p.rtab;
  const int N = (l & 1) ? 2 * l : l;
  FastDiv<T> sfd, bwd;
  sfd.init(p.sf, p.fast_sf != 0);
  bwd.init(p.bin_width, p.fast_bw != 0);
  if (p.stat_part != nullptr) {                                    // speculative launch: raw-input statistics of this block
    __shared__ double ss[3];
    StatAcc<T> acc;
    acc.init();
    if (k < l) acc.add(p.x[base + k], base + k != 0);
    acc.flush(p.stat_part, p.nlists_main, ss, 1);
  }
  if (k < l) {
    T a = p.x[base + k];
    if (SCALE) { a = sfd.div(a); if (p.scaled != nullptr) p.scaled[base + k] = a; }
    if (l & 1) { v[k] = a; v[l + (l - 1 - k)] = a; }               // dct.c:61-64
    else if (k & 1) v[l - 1 - (k >> 1)] = a;                       // dct.c:75-83
    else v[k >> 1] = a;
  }
  __syncthreads();
  T coef = T(0);
  if (k < l) {
    T sr = T(0), si = T(0);
    for (int j = 0; j < N; j++) {
      const int tt = (j * k) % N;
      sr = sr + v[j] * rt[RTAB_WR + tt];
      si = si + v[j] * rt[RTAB_WI + tt];
    }
    coef = rt[RTAB_AS + k] * sr + rt[RTAB_AX + k] * si;            // dct.c:100-102 (Im V = -si)
  }
  bool out = false;
  unsigned b = bin_of<T>(coef, p.range_min, p.range_max, bwd, &out);
  bool exc = false;
  if (k == 0) { b = 255u; out = false; } else exc = (b == 255u);
  if (k >= l) { exc = false; out = false; }
  const unsigned long long m = __ballot(exc);
  const unsigned rank = (unsigned)__popcll(m & ((1ull << k) - 1ull));
  // single-pass: append after the full blocks; two-level: this block is list #ntiles
  const unsigned start = p.tile_cnt ? p.ntiles * TILE_ELEMS : p.ctl->cnt_total;
  float* acdst = p.tile_cnt ? p.ac_tmp : p.ac;
  if (k < l) {
    p.bin[base + k] = (uint8_t)b;
    if (p.coef != nullptr) p.coef[base + k] = coef;
    if (k == 0) { p.dc[p.nfull] = (float)coef; p.ctl->q0 = (unsigned long long)to_bits(coef); }
    if (MODE == DCTZHIP_QT && out) atomicMax(&p.ctl->qraw[k], (unsigned long long)to_bits(fabs(coef)));
    if (exc) {
      if (MODE == DCTZHIP_EC) acdst[start + rank] = (float)coef;
      else { p.qt_item[start + rank] = coef; p.qt_j[start + rank] = (uint8_t)k; }
    }
  }
  __syncthreads();
  if (k == 0) {
    if (p.tile_cnt) p.tile_cnt[p.nlists_main] = (unsigned)__popcll(m);
    else p.ctl->cnt_total = start + (unsigned)__popcll(m);
  }
}

// QT pass 2 for the single-pass kernels (dctz-comp-lib.c:450-461 clamp, :478-533
// normalise + append): the flagged coefficients are already in global order.
template <typename T>
__global__ __launch_bounds__(SWG) void k_qt_finish(FwdParams<T> p, double eb) {
  using Bits = typename Traits<T>::Bits;
  __shared__ T q[64];
  if (threadIdx.x < 64) {
    T v = Traits<T>::from_bits((Bits)p.ctl->qraw[threadIdx.x]);
    if (v < T(1)) v = T(1);                                        // :450-461
    q[threadIdx.x] = v;
  }
  __syncthreads();
  const unsigned cnt = p.ctl->cnt_total;
  for (unsigned i = blockIdx.x * SWG + threadIdx.x; i < cnt; i += gridDim.x * SWG) {
    const T item = p.qt_item[i];
    const int j = p.qt_j[i];
    // The in-range else-branch of :502-506 cannot fire for finite data and
    // stores nothing; every flagged coefficient is appended.
    p.ac[i] = (float)qt_normalise(item, q[j], eb, T(10), p.range_min, p.range_max);
  }
}

// Two-level scheme, step 2: exclusive prefix over the per-tile exception counts
// (one workgroup; n <= 2^19 entries for the largest legal input).  off[n] = total.
__global__ __launch_bounds__(1024) void k_scan_tiles(const unsigned* __restrict__ cnt, unsigned* __restrict__ off,
                                                     unsigned n, Ctl* ctl) {
  // Segments of 8192 entries go through LDS so that global accesses are coalesced
  // (a strided walk through one CU's address path took 34 us for 32 Ki entries);
  // in LDS thread t owns 8 consecutive entries at pitch 9 (conflict-free).
  constexpr unsigned SEG = 8192, PER = SEG / 1024;
  __shared__ unsigned buf[SEG + SEG / PER];
  __shared__ unsigned part[1024 / 64];
  __shared__ unsigned carry_s;
  const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t == 0) carry_s = 0;
  for (unsigned base = 0; base < n; base += SEG) {
    const unsigned m = min(SEG, n - base);
    for (unsigned i = t; i < m; i += 1024) buf[i + i / PER] = cnt[base + i];
    __syncthreads();
    unsigned vals[PER], sum = 0;
#pragma unroll
    for (unsigned k = 0; k < PER; k++) {
      vals[k] = (t * PER + k < m) ? buf[t * (PER + 1) + k] : 0u;
      sum += vals[k];
    }
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned o = __shfl_up(incl, d);
      if (lane >= (unsigned)d) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    unsigned run = carry_s + incl - sum;
    for (unsigned w = 0; w < wave; w++) run += part[w];
#pragma unroll
    for (unsigned k = 0; k < PER; k++) {
      buf[t * (PER + 1) + k] = run;
      run += vals[k];
    }
    __syncthreads();
    for (unsigned i = t; i < m; i += 1024) off[base + i] = buf[i + i / PER];
    if (t == 1023) carry_s = run;                   // inclusive total through this segment
    __syncthreads();
  }
  if (t == 0) { off[n] = carry_s; ctl->cnt_total = carry_s; }
}

// Two-level scheme, step 3: move every tile-local list to its place in AC_exact[]
// (dctz-comp-lib.c:478-544 order: lists are already block-major, j ascending).
// QT: clamp the table (:450-461) and normalise on the way (:488-518).
template <typename T, int MODE>
__global__ __launch_bounds__(SWG) void k_compact_ac(FwdParams<T> p, double eb, unsigned nlists) {
  using Bits = typename Traits<T>::Bits;
  __shared__ T q[64];
  if (MODE == DCTZHIP_QT) {
    if (threadIdx.x < 64) {
      T v = Traits<T>::from_bits((Bits)p.ctl->qraw[threadIdx.x]);
      if (v < T(1)) v = T(1);
      q[threadIdx.x] = v;
    }
    __syncthreads();
  }
  // list l < G belongs to workgroup l of k_compress (slots of its tile range); list G is
  // the remainder block's.  One workgroup per list, 16 bytes per lane where aligned.
  const unsigned G = p.nlists_main;
  for (unsigned l = blockIdx.x; l < nlists; l += gridDim.x) {
    const unsigned n = p.tile_cnt[l], dst = p.tile_off[l];
    const size_t src = (size_t)(l < G ? tile_range(l, G, p.ntiles).lo : p.ntiles) * TILE_ELEMS;
    for (unsigned i = threadIdx.x; i < n; i += SWG) {
      if (MODE == DCTZHIP_EC) p.ac[dst + i] = p.ac_tmp[src + i];
      else p.ac[dst + i] = (float)qt_normalise(p.qt_item[src + i], q[p.qt_j[src + i]], eb, T(10), p.range_min, p.range_max);
    }
  }
}

// =============================================================== decompress ==
// Fused: de-quantise (dctz-decomp-lib.c:389-417 / :438-463; gen_bins
// binning.c:12-50) -> DCT-III per block (:428, dct.c:115-205) -> de-scale (:494-511).
// Two-level scheme, decode side: per-tile count of "stored exactly" flags
// (bin id 255 at j != 0, dctz-decomp-lib.c:400 / :446), 1 byte per element read.
__global__ __launch_bounds__(SWG) void k_count_tiles(const uint8_t* __restrict__ bin, unsigned nfull, unsigned ntiles,
                                                     unsigned* __restrict__ tile_cnt) {
  __shared__ unsigned part[SWG / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const TileRange tr = tile_range(blockIdx.x, gridDim.x, ntiles);     // the range k_decompress's workgroup b owns
  // the range is one contiguous string of whole 64-byte blocks; 16 bytes per thread and trip
  const size_t first = (size_t)tr.lo * TILE_ELEMS;
  const size_t last = (size_t)min(nfull, tr.hi * (unsigned)TILE_BLKS) * 64;
  unsigned c = 0;
  for (size_t o = first + (size_t)t * 16; o < last; o += (size_t)SWG * 16) {
    const uint4 wv = *reinterpret_cast<const uint4*>(bin + o);
    const unsigned w[4] = {wv.x, wv.y, wv.z, wv.w};
    const bool head = (o & 63) == 0;                                   // byte 0 of these 16 is a block's DC slot
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (((w[i >> 2] >> (8 * (i & 3))) & 255u) == 255u && !(head && i == 0)) c++;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
  if (lane == 0) part[wave] = c;
  __syncthreads();
  if (t == 0) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < SWG / 64; w++) sum += part[w];
    tile_cnt[blockIdx.x] = sum;
  }
}

// Loop order per tile k: ticket(k) -> load bins(k) -> count + scan + look-back(k)
// -> store tile k-1 (its IDCT output is still in LDS) -> gather coefficients(k)
// -> IDCT(k).  The 32 KiB of stores of tile k-1 are issued AFTER the look-back of
// tile k, so the polling wave never waits behind them, and they drain under the
// gather + IDCT of tile k.
template <typename T, int MODE, bool SCALE, int FEAT>
__global__ __launch_bounds__(WG) DCTZ_WAVES_PER_EU(T) void k_decompress(InvParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  T* tile = reinterpret_cast<T*>(smem);
  T* tab = tile + TILE_BLKS * Traits<T>::PITCH;
  T* qt = tab + TAB_SIZE;
  unsigned* sc = reinterpret_cast<unsigned*>(qt + 64);
  const int t = threadIdx.x;
  const int blk = t >> 2, j0 = (t & 3) * 16;
  load_tab<T>(tab, p.tab);
  if (MODE == DCTZHIP_QT && t < 64) qt[t] = p.qtab[t];

  bool pending = false;                  // tile `prev_id` sits in LDS, not yet stored
  unsigned prev_id = 0;

  // this thread's 16 bin ids and (quad lane 0) its block's DC for a given tile
  auto fetch = [&](unsigned id, uint4& wv, float& dcv) {
    wv = make_uint4(0, 0, 0, 0);
    dcv = 0.f;
    if (id < p.ntiles && (unsigned)blk < min((unsigned)TILE_BLKS, p.nfull - id * TILE_BLKS)) {
      wv = reinterpret_cast<const uint4*>(p.bin + (size_t)id * TILE_ELEMS)[t];
      if (j0 == 0) dcv = p.dc[id * TILE_BLKS + blk];
    }
  };
  // coefficient of position j from its bin id / fetched exact value / DC
  auto dequant = [&](unsigned b, int j, bool exc, float exact, float dcv) -> T {
    if (j == 0) return (T)dcv;                                     // :392 / :438
    if (exc) {
      T v = (T)exact;
      if (MODE == DCTZHIP_QT) v = qt_restore(v, qt[j], p.eb, T(10), p.range_min, p.range_max);
      return v;
    }
    const int ti = (b & 1u) ? (int)(b >> 1) + 1 : -(int)(b >> 1);  // binning.c:20 / :40
    return (T)ti * p.bin_width;                                    // :416 / :462
  };

  if constexpr ((FEAT & F_LOOKBACK) == 0) {
    // Two-level scheme: static tiles, exception offsets from k_scan_tiles.  Order per
    // tile k: flags + local scan -> issue the AC_exact gathers of k -> prefetch the
    // bin ids / DC of tile k+G -> store tile k-G (still in LDS) while the gathers
    // fly -> coefficients(k) to LDS -> IDCT(k).
    const TileRange tr = tile_range(blockIdx.x, gridDim.x, p.ntiles);
    unsigned tile_id = tr.lo;
    unsigned run = p.tile_off[blockIdx.x];   // global index of this workgroup's first exact coefficient
    // descriptors over this workgroup's ranges of out[] / bin[] / dc[]: hardware range checks replace
    // the per-vector predicates, one VGPR of addressing each
    const size_t first_el = (size_t)tr.lo * TILE_ELEMS;
    const size_t end_el = min((size_t)p.nfull * 64, (size_t)tr.hi * TILE_ELEMS);
    const int range_el = tr.lo < tr.hi ? (int)(end_el - first_el) : 0;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + first_el, 0, range_el * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bin + first_el), 0, range_el, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dc + first_el / 64), 0, range_el / 64 * 4, 0x00020000);
    auto fetch_buf = [&](unsigned rel, uint4& wv, float& dcv) {   // rel beyond the range: zeros
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(r_bin, t * 16, (int)(rel * (unsigned)TILE_ELEMS), 0);
      wv = make_uint4(r.x, r.y, r.z, r.w);
      dcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_dc, blk * 4, (int)(rel * (unsigned)(TILE_BLKS * 4)), 0));
    };
    uint4 wv, wv_n;
    float dcv, dcv_n;
    fetch_buf(0u, wv, dcv);
    bool underrun = false;
    while (tile_id < tr.hi) {
      const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile_id * TILE_BLKS);
      const bool active = (unsigned)blk < blks_here;
      const unsigned w[4] = {wv.x, wv.y, wv.z, wv.w};
      unsigned mask = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const unsigned b = (w[i >> 2] >> (8 * (i & 3))) & 255u;
        if (b == 255u && (j0 + i) != 0) mask |= 1u << i;           // :400 / :446
      }
      if (!active) mask = 0;
      __syncthreads();                   // sc[] of the previous tile is consumed
      unsigned total;
      unsigned r = run + tile_scan_local((unsigned)__popc(mask), sc, &total);
      run += total;
      float av[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {     // gathers first, uses later
        av[i] = 0.f;
        if (mask & (1u << i)) {
          if (r < p.ac_count) av[i] = p.ac[r]; else underrun = true;
          r++;
        }
      }
      const unsigned next_id = tile_id + 1;
      fetch_buf(next_id - tr.lo, wv_n, dcv_n);
      if (pending) {                     // flush the previous tile (uniform branch)
        store_tile_buf<T, SCALE>(tile, r_out, prev_id - tr.lo, p.sf);
        __syncthreads();                 // LDS tile free for the next coefficients
      }
      T c[16];
#pragma unroll
      for (int i = 0; i < 16; i++)
        c[i] = dequant((w[i >> 2] >> (8 * (i & 3))) & 255u, j0 + i, (mask >> i) & 1u, av[i], dcv);
#pragma unroll
      for (int i = 0; i < 16 / EPV; i++)
        lds_store_vec<T>(&tile[blk * Traits<T>::PITCH + j0 + i * EPV], Traits<T>::pack(&c[i * EPV]));
      __syncthreads();
      tile_dct_inv<T>(tile, tab);
      pending = true;
      prev_id = tile_id;
      tile_id = next_id; wv = wv_n; dcv = dcv_n;
    }
    if (underrun) atomicExch(&p.ctl->error, 2u);
    if (pending) store_tile_buf<T, SCALE>(tile, r_out, prev_id - tr.lo, p.sf);
    pending = false;
  } else {
    Stamps st;
    if (FEAT & F_STAMP) st.start();
    for (;;) {
      if (t == 0) sc[5] = take_ticket<FEAT>(p.ctl, p.ngroups);
      __syncthreads();                   // also: every lane is done with sc[] of the previous tile
      if ((FEAT & F_STAMP) && t == 0) st.mark(0);
      const unsigned tile_id = sc[5];
      if (tile_id >= p.ntiles) break;
      const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile_id * TILE_BLKS);
      const bool active = (unsigned)blk < blks_here;
      uint4 wv;
      float dcv;
      fetch(tile_id, wv, dcv);
      const unsigned w[4] = {wv.x, wv.y, wv.z, wv.w};
      unsigned mask = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const unsigned b = (w[i >> 2] >> (8 * (i & 3))) & 255u;
        if (b == 255u && (j0 + i) != 0) mask |= 1u << i;           // :400 / :446
      }
      if (!active) mask = 0;
      if ((FEAT & F_STAMP) && t == 0) st.mark(1);                  // bins load
      unsigned r = tile_scan((unsigned)__popc(mask), tile_id, p.ntiles, sc, p.desc, p.ctl, false, 0u,
                             (FEAT & F_STAMP) ? &st : nullptr);
      if ((FEAT & F_STAMP) && t == 0) st.mark(2);
      if (pending) {                     // now flush the previous tile (uniform branch)
        const unsigned pv = min((unsigned)TILE_BLKS, p.nfull - prev_id * TILE_BLKS) * 64u;
        store_tile<T, SCALE>(tile, p.out, (size_t)prev_id * TILE_ELEMS, pv, p.sf);
        __syncthreads();                 // LDS tile free for the next coefficients
      }
      if ((FEAT & F_STAMP) && t == 0) st.mark(3);                  // store of the previous tile
      T c[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const bool exc = (mask >> i) & 1u;
        float exact = 0.f;
        if (exc) {
          if (r < p.ac_count) exact = p.ac[r]; else atomicExch(&p.ctl->error, 2u);
          r++;
        }
        c[i] = dequant((w[i >> 2] >> (8 * (i & 3))) & 255u, j0 + i, exc, exact, dcv);
      }
#pragma unroll
      for (int i = 0; i < 16 / EPV; i++)
        lds_store_vec<T>(&tile[blk * Traits<T>::PITCH + j0 + i * EPV], Traits<T>::pack(&c[i * EPV]));
      __syncthreads();
      if ((FEAT & F_STAMP) && t == 0) st.mark(6);                  // gather + de-quantise
      tile_dct_inv<T>(tile, tab);
      if ((FEAT & F_STAMP) && t == 0) st.mark(7);                  // IDCT
      pending = true;
      prev_id = tile_id;
    }
    if ((FEAT & F_STAMP) && t == 0) st.flush(p.ctl);
  }
  if (pending) {
    const unsigned pv = min((unsigned)TILE_BLKS, p.nfull - prev_id * TILE_BLKS) * 64u;
    store_tile<T, SCALE>(tile, p.out, (size_t)prev_id * TILE_ELEMS, pv, p.sf);
  }
}

// Last, short block on decode (dctz-decomp-lib.c:423-428, dct.c:144-199).
template <typename T, int MODE, bool SCALE>
__global__ __launch_bounds__(64) void k_decompress_rem(InvParams<T> p, int l) {
  __shared__ T a[64];
  __shared__ T cr[128];
  __shared__ T ci[128];
  const int k = threadIdx.x;
  const size_t base = (size_t)p.nfull * 64;
  const T* rt = p.rtab;
  const int N = (l & 1) ? 2 * l : l;
  unsigned b = 0;
  if (k < l) b = p.bin[base + k];
  const bool exc = (k < l) && (k != 0) && (b == 255u);
  const unsigned long long m = __ballot(exc);
  const unsigned rank = (unsigned)__popcll(m & ((1ull << k) - 1ull));
  const unsigned start = p.tile_off ? p.tile_off[p.nlists_main] : p.ctl->cnt_total;
  cr[k] = T(0); ci[k] = T(0); cr[k + 64] = T(0); ci[k + 64] = T(0);
  if (k < l) {
    T val;
    if (k == 0) val = (T)p.dc[p.nfull];
    else if (exc) {
      T v = T(0);
      if (start + rank < p.ac_count) v = (T)p.ac[start + rank]; else atomicExch(&p.ctl->error, 2u);
      if (MODE == DCTZHIP_QT) v = qt_restore(v, p.qtab[k], p.eb, T(10), p.range_min, p.range_max);
      val = v;
    } else {
      const int ti = (b & 1u) ? (int)(b >> 1) + 1 : -(int)(b >> 1);
      val = (T)ti * p.bin_width;
    }
    a[k] = val;
  }
  __syncthreads();
  if (k < l) {
    cr[k] = rt[RTAB_IAS + k] * a[k];                               // dct.c:146-151 / :166-172
    ci[k] = rt[RTAB_IAX + k] * a[k];
    if ((l & 1) && k >= 1) {                                       // dct.c:152-153
      cr[l + k] = rt[RTAB_IAX + k] * a[l - k];
      ci[l + k] = -(rt[RTAB_IAS + k] * a[l - k]);
    }
  }
  __syncthreads();
  if (k < l) {
    const int s = (l & 1) ? k : ((k & 1) ? l - 1 - (k >> 1) : (k >> 1));   // dct.c:189-199
    T acc = T(0);
    for (int j = 0; j < N; j++) {
      const int tt = (s * j) % N;
      acc = acc + (cr[j] * rt[RTAB_WR + tt] - ci[j] * rt[RTAB_WI + tt]);
    }
    T val = (l & 1) ? (acc / (T)l) / T(2) : acc / (T)l;            // dct.c:163 / :185
    if (SCALE) val = val * p.sf;
    p.out[base + k] = val;
  }
}

// Diagnostics: FastDiv against the compiler's own division, element by element.
template <typename T>
__global__ __launch_bounds__(SWG) void k_debug_divide(const T* __restrict__ x, size_t n, T d, int ok,
                                                     T* __restrict__ fast, T* __restrict__ ref) {
  FastDiv<T> fd;
  fd.init(d, ok != 0);
  for (size_t i = (size_t)blockIdx.x * SWG + threadIdx.x; i < n; i += (size_t)gridDim.x * SWG) {
    fast[i] = fd.div(x[i]);
    ref[i] = x[i] / d;
  }
}

// ============================================================ transform only ==
// Batched dct_fftw / ifft_idct over all full blocks (dct.h:17-27; dct-test.c:81-89,144-152).
template <typename T, bool INVERSE>
__global__ __launch_bounds__(WG) void k_dct_blocks(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ gtab,
                                                   unsigned nfull, unsigned ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* tile = reinterpret_cast<T*>(smem);
  T* tab = tile + TILE_BLKS * Traits<T>::PITCH;
  load_tab<T>(tab, gtab);
  for (unsigned tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {
    __syncthreads();
    const size_t ebase = (size_t)tile_id * TILE_ELEMS;
    const unsigned valid = min((unsigned)TILE_BLKS, nfull - tile_id * TILE_BLKS) * 64u;
    load_tile<T, false>(tile, x, ebase, valid, T(1), nullptr);
    __syncthreads();
    if (INVERSE) tile_dct_inv<T>(tile, tab); else tile_dct_fwd<T>(tile, tab);
    store_tile<T, false>(tile, out, ebase, valid, T(1));
  }
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(64) void k_dct_rem(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ rt, int l) {
  __shared__ T v[128];
  __shared__ T w[128];
  const int k = threadIdx.x;
  const int N = (l & 1) ? 2 * l : l;
  v[k] = T(0); v[k + 64] = T(0); w[k] = T(0); w[k + 64] = T(0);
  __syncthreads();
  if (!INVERSE) {
    if (k < l) {
      const T a = x[k];
      if (l & 1) { v[k] = a; v[l + (l - 1 - k)] = a; }
      else if (k & 1) v[l - 1 - (k >> 1)] = a;
      else v[k >> 1] = a;
    }
    __syncthreads();
    if (k < l) {
      T sr = T(0), si = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (j * k) % N;
        sr = sr + v[j] * rt[RTAB_WR + tt];
        si = si + v[j] * rt[RTAB_WI + tt];
      }
      out[k] = rt[RTAB_AS + k] * sr + rt[RTAB_AX + k] * si;
    }
  } else {
    if (k < l) {
      v[k] = rt[RTAB_IAS + k] * x[k];
      w[k] = rt[RTAB_IAX + k] * x[k];
      if ((l & 1) && k >= 1) {
        v[l + k] = rt[RTAB_IAX + k] * x[l - k];
        w[l + k] = -(rt[RTAB_IAS + k] * x[l - k]);
      }
    }
    __syncthreads();
    if (k < l) {
      const int s = (l & 1) ? k : ((k & 1) ? l - 1 - (k >> 1) : (k >> 1));
      T acc = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (s * j) % N;
        acc = acc + (v[j] * rt[RTAB_WR + tt] - w[j] * rt[RTAB_WI + tt]);
      }
      out[k] = (l & 1) ? (acc / (T)l) / T(2) : acc / (T)l;
    }
  }
}

// ================================================================= launchers ==
template <typename T>
static size_t fwd_smem(bool pipe) {
  return sizeof(T) * (TILE_BLKS * Traits<T>::PITCH + TAB_SIZE) + 64 * sizeof(typename Traits<T>::Bits) + 64 +
         (pipe ? TILE_ELEMS * sizeof(float) : 0);       // + AC_exact park of one tile
}
template <typename T>
static size_t inv_smem() {
  return sizeof(T) * (TILE_BLKS * Traits<T>::PITCH + TAB_SIZE + 64) + 32;
}
template <typename T>
static size_t dct_smem() {
  return sizeof(T) * (TILE_BLKS * Traits<T>::PITCH + TAB_SIZE);
}

template <typename T>
void launch_stats(const T* x, size_t n, double* part, int nparts, double* out, hipStream_t s, HostBox* box, unsigned long long seq) {
  hipLaunchKernelGGL(k_stats<T>, dim3(nparts), dim3(SWG), 0, s, x, n, part);
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, box, seq);
}

template <typename T>
void launch_stats_sample(const T* x, size_t n, unsigned group, double* part, int nparts, double* out, hipStream_t s,
                         HostBox* box, unsigned long long seq) {
  hipLaunchKernelGGL(k_stats_sample<T>, dim3(nparts), dim3(SWG), 0, s, x, n, group, part);
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, box, seq);
}
void launch_stats_final(const double* part, int nparts, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, part, nparts, out, (HostBox*)nullptr, 0ull);
}
void launch_finish(Ctl* ctl, const double* part, int nparts, HostBox* box, unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(SWG), 0, s, ctl, part, nparts, box, seq);
}

template <typename T>
void launch_debug_divide(const T* x, size_t n, T d, int ok, T* fast, T* ref, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_divide<T>, dim3(1024), dim3(SWG), 0, s, x, n, d, ok, fast, ref);
}

template <typename T>
void launch_serial_sum(const T* x, size_t n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_serial_sum<T>, dim3(1), dim3(64), 0, s, x, n, out);
}

template <typename T>
void launch_scale(T* x, size_t n, T sf, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_scale<T>, dim3(grid), dim3(SWG), 0, s, x, n, sf);
}

template <typename T, int FEAT>
static void launch_compress_f(const FwdParams<T>& p, int mode, bool scale, int grid, hipStream_t s) {
  const size_t sm = fwd_smem<T>(false);
  if (mode == DCTZHIP_EC) {
    if (scale) hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, true, FEAT>), dim3(grid), dim3(WG), sm, s, p);
    else hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, false, FEAT>), dim3(grid), dim3(WG), sm, s, p);
  } else {
    if (scale) hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, true, FEAT>), dim3(grid), dim3(WG), sm, s, p);
    else hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, false, FEAT>), dim3(grid), dim3(WG), sm, s, p);
  }
}
template <typename T>
void launch_compress(const FwdParams<T>& p, int mode, bool scale, int grid, int feat, hipStream_t s) {
  if (!(feat & F_LOOKBACK)) {
    if (feat & F_STATS) launch_compress_f<T, F_STATS>(p, mode, scale, grid, s);
    else launch_compress_f<T, 0>(p, mode, scale, grid, s);
    return;
  }
  if (feat & F_STAMP) launch_compress_f<T, F_LOOKBACK | F_STAMP>(p, mode, scale, grid, s);
  else if (feat & F_GROUP) launch_compress_f<T, F_LOOKBACK | F_GROUP>(p, mode, scale, grid, s);
  else launch_compress_f<T, F_LOOKBACK>(p, mode, scale, grid, s);
}

template <typename T>
void launch_compress_rem(const FwdParams<T>& p, int mode, bool scale, int l, hipStream_t s) {
  if (mode == DCTZHIP_EC) {
    if (scale) hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_EC, true>), dim3(1), dim3(64), 0, s, p, l);
    else hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_EC, false>), dim3(1), dim3(64), 0, s, p, l);
  } else {
    if (scale) hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_QT, true>), dim3(1), dim3(64), 0, s, p, l);
    else hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_QT, false>), dim3(1), dim3(64), 0, s, p, l);
  }
}

template <typename T>
void launch_qt_finish(const FwdParams<T>& p, double eb, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_qt_finish<T>, dim3(grid), dim3(SWG), 0, s, p, eb);
}

void launch_scan_tiles(const unsigned* cnt, unsigned* off, unsigned n, Ctl* ctl, hipStream_t s) {
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, cnt, off, n, ctl);
}

void launch_count_tiles(const uint8_t* bin, unsigned nfull, unsigned ntiles, unsigned* tile_cnt, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_count_tiles, dim3(grid), dim3(SWG), 0, s, bin, nfull, ntiles, tile_cnt);
}

template <typename T>
void launch_compact_ac(const FwdParams<T>& p, int mode, double eb, unsigned nlists, int grid, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_compact_ac<T, DCTZHIP_EC>), dim3(grid), dim3(SWG), 0, s, p, eb, nlists);
  else hipLaunchKernelGGL((k_compact_ac<T, DCTZHIP_QT>), dim3(grid), dim3(SWG), 0, s, p, eb, nlists);
}

template <typename T, int FEAT>
static void launch_decompress_f(const InvParams<T>& p, int mode, bool scale, int grid, hipStream_t s) {
  const size_t sm = inv_smem<T>();
  if (mode == DCTZHIP_EC) {
    if (scale) hipLaunchKernelGGL((k_decompress<T, DCTZHIP_EC, true, FEAT>), dim3(grid), dim3(WG), sm, s, p);
    else hipLaunchKernelGGL((k_decompress<T, DCTZHIP_EC, false, FEAT>), dim3(grid), dim3(WG), sm, s, p);
  } else {
    if (scale) hipLaunchKernelGGL((k_decompress<T, DCTZHIP_QT, true, FEAT>), dim3(grid), dim3(WG), sm, s, p);
    else hipLaunchKernelGGL((k_decompress<T, DCTZHIP_QT, false, FEAT>), dim3(grid), dim3(WG), sm, s, p);
  }
}
template <typename T>
void launch_decompress(const InvParams<T>& p, int mode, bool scale, int grid, int feat, hipStream_t s) {
  if (!(feat & F_LOOKBACK)) { launch_decompress_f<T, 0>(p, mode, scale, grid, s); return; }
  if (feat & F_STAMP) launch_decompress_f<T, F_LOOKBACK | F_STAMP>(p, mode, scale, grid, s);
  else if (feat & F_GROUP) launch_decompress_f<T, F_LOOKBACK | F_GROUP>(p, mode, scale, grid, s);
  else launch_decompress_f<T, F_LOOKBACK>(p, mode, scale, grid, s);
}

template <typename T>
void launch_decompress_rem(const InvParams<T>& p, int mode, bool scale, int l, hipStream_t s) {
  if (mode == DCTZHIP_EC) {
    if (scale) hipLaunchKernelGGL((k_decompress_rem<T, DCTZHIP_EC, true>), dim3(1), dim3(64), 0, s, p, l);
    else hipLaunchKernelGGL((k_decompress_rem<T, DCTZHIP_EC, false>), dim3(1), dim3(64), 0, s, p, l);
  } else {
    if (scale) hipLaunchKernelGGL((k_decompress_rem<T, DCTZHIP_QT, true>), dim3(1), dim3(64), 0, s, p, l);
    else hipLaunchKernelGGL((k_decompress_rem<T, DCTZHIP_QT, false>), dim3(1), dim3(64), 0, s, p, l);
  }
}

template <typename T>
void launch_dct_blocks(const T* x, T* out, const T* gtab, const T* rtab, size_t n, bool inverse, int grid,
                       hipStream_t s) {
  const unsigned nfull = (unsigned)(n / 64), ntiles = (nfull + TILE_BLKS - 1) / TILE_BLKS;
  const int l = (int)(n % 64);
  if (nfull) {
    const int g = (int)min((unsigned)grid, ntiles);
    if (inverse) hipLaunchKernelGGL((k_dct_blocks<T, true>), dim3(g), dim3(WG), dct_smem<T>(), s, x, out, gtab, nfull, ntiles);
    else hipLaunchKernelGGL((k_dct_blocks<T, false>), dim3(g), dim3(WG), dct_smem<T>(), s, x, out, gtab, nfull, ntiles);
  }
  if (l) {
    const T* xr = x + (size_t)nfull * 64;
    T* orr = out + (size_t)nfull * 64;
    if (inverse) hipLaunchKernelGGL((k_dct_rem<T, true>), dim3(1), dim3(64), 0, s, xr, orr, rtab, l);
    else hipLaunchKernelGGL((k_dct_rem<T, false>), dim3(1), dim3(64), 0, s, xr, orr, rtab, l);
  }
}

// explicit instantiations used by dctz_shim.hip
#define INST(T)                                                                                         \
  template void launch_stats<T>(const T*, size_t, double*, int, double*, hipStream_t, HostBox*, unsigned long long); \
  template void launch_stats_sample<T>(const T*, size_t, unsigned, double*, int, double*, hipStream_t, HostBox*, unsigned long long); \
  template void launch_debug_divide<T>(const T*, size_t, T, int, T*, T*, hipStream_t);                  \
  template void launch_serial_sum<T>(const T*, size_t, double*, hipStream_t);                           \
  template void launch_scale<T>(T*, size_t, T, int, hipStream_t);                                       \
  template void launch_compress<T>(const FwdParams<T>&, int, bool, int, int, hipStream_t);              \
  template void launch_compress_rem<T>(const FwdParams<T>&, int, bool, int, hipStream_t);               \
  template void launch_qt_finish<T>(const FwdParams<T>&, double, int, hipStream_t);                     \
  template void launch_compact_ac<T>(const FwdParams<T>&, int, double, unsigned, int, hipStream_t);     \
  template void launch_decompress<T>(const InvParams<T>&, int, bool, int, int, hipStream_t);            \
  template void launch_decompress_rem<T>(const InvParams<T>&, int, bool, int, hipStream_t);             \
  template void launch_dct_blocks<T>(const T*, T*, const T*, const T*, size_t, bool, int, hipStream_t);
INST(double)
INST(float)

}  // namespace dctz
