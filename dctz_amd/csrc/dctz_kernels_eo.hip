// dctz_kernels_eo.hip -- k_compress_eo: the compress kernel for large fp64 arrays with every block shared by TWO lanes
// (gfx950, MI355X).  Same outputs as k_compress (dctz_kernels.hip), bit for bit; same place in the chain of kernels.
//
// Why.  k_compress gives a 64-element block to ONE lane: 64 fp64 values = 128 registers of data, 256 in all, two waves per
// SIMD -- and at two waves per SIMD the vector unit issues 54 % of the time (profiles/r04_pmc.txt: neither the ALU nor HBM
// is saturated, the kernel waits).  A third wave needs <= 168 registers, i.e. half a block per lane.
//
// How the block is cut.  Not inside a wavefront: two lanes of one wave that took different parts of a transform would
// need per-lane constants (the split constants and twiddles are scalar operands today) and an exchange of half the block
// at the radix boundary (EXPERIMENTS section 13.1: ~1150 issue slots per block against ~700).  The cut is between two
// WAVEFRONTS of a workgroup instead: lane b of the "even" wave and lane b of the "odd" wave share block b of the tile.
// dct64_block_eo.h shows that the operations of the pinned 64-point flow fall into two disjoint sets -- those behind the
// even-numbered coefficients, which are functions of the 32 sums a[e] + a[63 - e] alone, and those behind the
// odd-numbered ones, functions of the 32 differences -- so the two waves share NO arithmetic but the scaling x / sf
// (dctz-comp-lib.c:193-216) and nothing crosses between them inside the transform; every constant stays wave-uniform.
// A lane carries 32 values.  What the waves do exchange, through LDS: the tile's image (both read all of it), the flag
// masks of their blocks (the exact coefficients of a block interleave between the two lanes: position j belongs to the
// even wave for j even), and the odd wave's bin ids.
//
// Per tile (64 blocks, 32 KiB), both waves:
//   phase 0: segments 0 and 3 of every block (elements 0..15, 48..63: the pairs (e, 63 - e) stay inside a phase) come in
//            by LDS-DMA (each wave issues half of the rows), 16 KiB; both waves read all 32 elements of their lane's
//            block, take max|x| / min|x| of half of them each (calc_data_stat, util.c:18-25), scale, and form their 16
//            sums (even wave) or differences (odd wave);
//   phase 1: segments 1 and 2 likewise; the DMA of the NEXT tile's phase 0 is issued as soon as the image is read out and
//            lands under the transform;
//   transform half (dct64_block_eo.h; dct.c:55-103), pass-1 binning of the lane's 32 coefficients (:363-414);
//   the exact coefficients (:478-544) are put INTO THE REFERENCE'S ORDER inside LDS: with both masks of a block a lane
//   knows the rank of each of its flagged positions, a wave scan of the blocks' counts gives the block's place; the tile's
//   piece leaves in whole rows for the workgroup's list, which therefore IS in order (LIST_IN_ORDER: k_compact_ac only
//   moves it to its place); the bin ids are interleaved by the even wave and leave in 1 KiB rows, as in k_compress.
//   The stores of a tile's outputs are issued in the NEXT tile, right behind the DMA issue of its second phase: the wait
//   that follows is the wait for that DMA, which takes as long as they do (vmcnt counts loads and stores together).
//
// Replaces, like k_compress: [calc_data_stat util.c:12-44 ->] scale (dctz-comp-lib.c:193-216) -> DCT-II (:337-340,
// dct.c:55-103) -> DC (:350-351) -> pass-1 binning (:361-414) -> ordered exact coefficients (:478-544), full blocks only.
#include "dctz_kernel_common.h"
#include "dct64_block_eo.h"

namespace dctz {

#ifndef DCTZ_EO_WAVES
#define DCTZ_EO_WAVES 3            /* waves per SIMD the register allocation aims at */
#endif
#ifndef DCTZ_EO_PREFETCH
#define DCTZ_EO_PREFETCH 0
#endif
#ifndef DCTZ_EO_CAP
#define DCTZ_EO_CAP 1024           /* exact coefficients of a tile staged per round, EC */
#endif
#ifndef DCTZ_EO_CAP_QT
#define DCTZ_EO_CAP_QT 448         /* ... QT (8-byte items + a position byte each) */
#endif
template <int I> using IC2 = std::integral_constant<int, I>;
constexpr int EO_WG = 128;         // threads per workgroup: the even wave and the odd wave of a tile

// Phase geometry: 128-byte segment of the block that slot s (0 / 1) of phase ph holds; raw element e -> its phase and
// its index among the 32 elements a lane reads per phase.
struct EoMap {
  static __host__ __device__ constexpr int seg(int ph, int s) { return ph == 0 ? (s ? 3 : 0) : (s ? 2 : 1); }
  static __host__ __device__ constexpr int phase_of(int e) { return ((e >> 4) == 0 || (e >> 4) == 3) ? 0 : 1; }
  static __host__ __device__ constexpr int idx_of(int e) { return ((e >> 4) >> 1) * 16 + (e & 15); }
};
static_assert(EoMap::phase_of(eo_lhs(0, 0)) == EoMap::phase_of(eo_rhs(0, 0)) && EoMap::phase_of(eo_lhs(5, 1)) == EoMap::phase_of(eo_rhs(5, 1)) &&
              EoMap::phase_of(eo_lhs(9, 0)) == EoMap::phase_of(eo_rhs(9, 0)) && EoMap::phase_of(eo_lhs(15, 1)) == EoMap::phase_of(eo_rhs(15, 1)),
              "both elements of a first butterfly lie in one phase");

template <int MODE> struct EoStage {
  using Item = typename Sub<double, MODE>::Item;     // float (EC) | double (QT: normalised later, by k_compact_ac)
  static constexpr int CAP = MODE == DCTZHIP_EC ? DCTZ_EO_CAP : DCTZ_EO_CAP_QT;
  static constexpr int SLOTS = CAP + 128;            // + a dump slot per lane of either wave
  static constexpr int ITEM_BYTES = SLOTS * (int)sizeof(Item);
  static constexpr int POS_BYTES = MODE == DCTZHIP_QT ? SLOTS : 0;
  static constexpr int BYTES = ITEM_BYTES + POS_BYTES;
};
constexpr int EO_TILE_LDS = Geo<double, 2>::PHB;     // 16 KiB: one phase of the tile
template <int MODE> constexpr size_t eo_lds_bytes() { return (size_t)EO_TILE_LDS + 4096 + EoStage<MODE>::BYTES + 512 + 512 + 64; }

// OR over the 64 lanes of a wavefront (DPP row shifts / broadcasts, as wave_incl_scan); the result is uniform
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void eo_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (LDS pointers in their own address space: handed over as generic pointers, every use pays a null test of a 64-bit
// pointer that has to be kept -- sixteen DMA targets alone were 32 spilled scalar registers)
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) const double2 lds_cd2;
typedef __attribute__((address_space(3))) const u32x4 lds_cu4;
typedef __attribute__((address_space(3))) const u32x2 lds_cu2;
template <typename P> __device__ __forceinline__ unsigned lds_at(P* p) { return (unsigned)(size_t)p; }
typedef __attribute__((address_space(3))) const double lds_cf64;
struct EoLds {
  lds_u8* tile;             // the phase image (DMA target)
  lds_u8* bins;             // the odd wave's bin ids on their way to the even wave, then the tile's bin ids on their way out
  lds_u8* stage;            // the tile's exact coefficients, in order, on their way out
  lds_u32* xmask;                 // [2][64] the waves' flag masks
  lds_u64* qmax;        // QT: per-position maxima of the workgroup
  lds_f64* stat;                    // [2][3]
};

template <int MODE, bool STATS, int ROLE>
__device__ __forceinline__ void compress_eo_role(const FwdParams<double>& p, const unsigned wg, const unsigned nwg, const EoLds& L) {
  using T = double;
  using G = Geo<T, 2>;
  using ST = EoStage<MODE>;
  using Item = typename ST::Item;
  constexpr unsigned CAP = (unsigned)ST::CAP;
  lds_u8* const tilebuf = L.tile;
  lds_u8* const binbuf = L.bins;
  lds_u8* const stagebuf = L.stage;
  const int lane = threadIdx.x & 63;
  const TileRange tr = tile_range(wg, nwg, p.ntiles);
  const unsigned list_base = tr.lo * TILE_ELEMS;
  const size_t first_el = (size_t)tr.lo * TILE_ELEMS;
  const size_t end_el = min((size_t)p.nfull * 64, (size_t)tr.hi * TILE_ELEMS);
  const int range_el = tr.lo < tr.hi ? (int)(end_el - first_el) : 0;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.x + first_el), 0, range_el * (int)sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(p.bin + first_el, 0, range_el, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(p.dc + first_el / 64, 0, range_el / 64 * 4, 0x00020000);
  const int list_slots = (int)((tr.hi - tr.lo) * (unsigned)TILE_ELEMS);
  const __amdgpu_buffer_rsrc_t r_list = (MODE == DCTZHIP_EC)
      ? __builtin_amdgcn_make_buffer_rsrc(p.ac_tmp + list_base, 0, list_slots * 4, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(p.qt_item + list_base, 0, list_slots * (int)sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_listj = __builtin_amdgcn_make_buffer_rsrc(p.qt_j + (MODE == DCTZHIP_QT ? list_base : 0u), 0, MODE == DCTZHIP_QT ? list_slots : 0, 0x00020000);
  const T sf = p.guess ? (T)p.guess->sf : p.sf;
  const unsigned fast_sf = p.guess ? p.guess->fast_sf : p.fast_sf;
  FastDiv<T> sfd, bwd;
  sfd.init(sf, fast_sf != 0);
  bwd.init(p.bin_width, (p.fast_bw & 1u) != 0);
  const bool scale = (sf != T(1));                   // dctz-comp-lib.c:193 / :208
  const T rmin = p.range_min, rmax = p.range_max;
  const CTab<T> tab = as_ctab<T>(p.tab);
  StatAcc<T> acc;
  acc.init();
  TileMap<T, 2> tm;
  tm.init(lane);
  const unsigned stage_at = lds_at(stagebuf), bins_at = lds_at(binbuf), xmask_at = lds_at(L.xmask);
  const unsigned qmax_at = lds_at(L.qmax);
  if (MODE == DCTZHIP_QT && ROLE == EO_EVEN) L.qmax[lane] = 0ull;

  // this wave's half of a phase's rows, HBM -> LDS
  auto issue = [&](unsigned rel, auto phase) {
    constexpr int PHASE = decltype(phase)::value;
    // (ONE scalar the compiler cannot see through: left alone, it keeps an induction variable per DMA instruction --
    // sixteen of them, spilled, a v_readlane / v_writelane pair each per trip)
    int base = (int)(rel * (unsigned)G::TILEB);
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int jg = 4 * ROLE; jg < 4 * ROLE + 4; jg++)
#pragma unroll
      for (int s = 0; s < 2; s++)
        DMA16(r_in, tilebuf + (jg * 2 + s) * 1024, tm.g_of(jg), base + jg * 8 * G::BLKB + EoMap::seg(PHASE, s) * 128, 2 /* nt */);
  };

  // the outputs of the tile before, on their way out (see the head of the file)
  bool pend = false;
  unsigned p_rel = 0, p_run = 0, p_cnt = 0;
  float p_dc = 0.f;
  unsigned run = 0;                                  // length of the workgroup's list so far (uniform, the same in both waves)

  // rows [0, cnt) of the staged piece -> the workgroup's list at `at0`; the two waves take every other row
  // (lane numbers through a register the compiler cannot see through wherever per-lane addresses are made of them outside
  // the arithmetic: kept alive across the loop they are spilled to scratch, and a scratch reload waits behind the DMA in flight)
  auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto store_rows = [&](unsigned at0, unsigned cnt) {
    const int ln = lane_now();
    for (unsigned r = (unsigned)ROLE; r * 64u < cnt; r += 2u) {
      const unsigned e = r * 64u + (unsigned)ln;
      const bool in = e < cnt;
      const int at = (int)(at0 + e);
      if (MODE == DCTZHIP_EC) {
        const unsigned v = *(const lds_u32*)(stagebuf + e * 4u);
        __builtin_amdgcn_raw_buffer_store_b32(v, r_list, in ? at * 4 : 0x7FFFFFF0, 0, 0);
      } else {
        const u32x2 v = *(lds_cu2*)(stagebuf + e * 8u);
        const unsigned char jj = stagebuf[ST::ITEM_BYTES + e];
        __builtin_amdgcn_raw_buffer_store_b64(v, r_list, in ? at * 8 : 0x7FFFFFF0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8(jj, r_listj, in ? at : 0x7FFFFFF0, 0, 0);
        if (in) {                                    // :371-372 / :396-397
          const T a = fabs(__builtin_bit_cast(double, v));
          if (a > rmax) lds_max_u64(qmax_at + (unsigned)jj * 8u, to_bits(a));
        }
      }
    }
  };
  auto flush_prev = [&]() {
    store_rows(p_run, p_cnt);
    if (ROLE == EO_EVEN) {
      // bin ids: (through their staging buffer) 1 KiB rows of 16 consecutive blocks; DC (:350-351 USE_TRUNCATE)
      const int lane = lane_now();
      const int bin_goff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) * 16);
      const int voff = (int)(p_rel * (unsigned)TILE_ELEMS) + bin_goff;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32x4 v = *(lds_cu4*)(binbuf + i * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, r_bin, voff + i * 1024, 0, 0);
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p_dc), r_dc, (int)(p_rel * 64u + (unsigned)lane) * 4, 0, 0);
    }
  };

  T pr[16], pi[16];                                  // z[m] + z[m + 16] (even wave) / z[m] - z[m + 16] (odd wave) of the scaled block
  // the lane's 32 elements of the phase in the image -> registers
  auto read32 = [&](T (&r)[32]) {
#pragma unroll
    for (int ch = 0; ch < 16; ch++) {
      typedef double f64x2 __attribute__((ext_vector_type(2)));
      const f64x2 v = *(__attribute__((address_space(3))) const f64x2*)(tilebuf + tm.lds_a[ch & 7] + (ch >> 3) * 1024);
      r[2 * ch] = v.x; r[2 * ch + 1] = v.y;
    }
  };
  // statistics over the raw values (this wave's half of them), scaling, first butterflies of the phase's eight points --
  // chunk pair by chunk pair: the 16-byte chunk ch of the phase's first segment and its mirror image 15 - ch in the second
  // hold (a[e], a[e + 1]) and (a[62 - e], a[63 - e]), i.e. both operands of two first butterflies, (a[e], a[63 - e]) and
  // (a[62 - e], a[e + 1]); the four raw registers die there and then (scheduled as ONE block the phase keeps 32 raw values,
  // their quotients and the results alive at once and spills)
  auto phase_math = [&](T (&r)[32], auto phase, bool active, bool first, auto lvl) {
    constexpr int PHASE = decltype(phase)::value, LVL = decltype(lvl)::value;
    // util.c:22 starts at i = 1: x[0] never enters the sum; the sum is 8 sf * (sum of the DCs), so x[0] leaves it as
    // x[0] / (8 sf) DC units (as in k_compress)
    if (STATS && PHASE == 0 && ROLE == EO_EVEN && first && lane == 0) acc.dcs -= (double)r[0] / (scale ? 8.0 * (double)sf : 8.0);
#pragma unroll
    for (int ch = 0; ch < 8; ch++) {
      T v[4] = {r[2 * ch], r[2 * ch + 1], r[2 * (15 - ch)], r[2 * (15 - ch) + 1]};
      if (STATS && active) { acc.minmax(v[2 * ROLE]); acc.minmax(v[2 * ROLE + 1]); }     // (this wave's half of the elements)
      if (LVL >= 0) {                                // dctz-comp-lib.c:197-199
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = LVL == 2 ? sfd.core(v[i]) : (LVL == 1 ? sfd.div(v[i]) : v[i] / sfd.d);
      }
      // element numbers: slot 0 of the phase starts at element e0, slot 1 (the mirror) ends at 63 - e0
      constexpr int E0 = PHASE == 0 ? 0 : 16;
      const int e = E0 + 2 * ch;                     // v[0] = a[e], v[1] = a[e + 1], v[2] = a[62 - e], v[3] = a[63 - e]
      // a[e] is the left operand of point m = e / 4's real (e % 4 == 0) or imaginary (e % 4 == 2) butterfly, a[62 - e] of
      // point 15 - m's imaginary resp. real one
      const int m = e / 4, mm = 15 - m;
      const T s0 = ROLE == EO_EVEN ? v[0] + v[3] : v[0] - v[3];
      const T s1 = ROLE == EO_EVEN ? v[2] + v[1] : v[2] - v[1];
      if (e % 4 == 0) { pr[m] = s0; pi[mm] = s1; } else { pi[m] = s0; pr[mm] = s1; }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto phase_any = [&](T (&r)[32], auto phase, bool active, bool first) {
#ifdef DCTZ_EO_HOT
    phase_math(r, phase, active, first, std::integral_constant<int, 2>{}); return;
#endif
    if (!scale) phase_math(r, phase, active, first, std::integral_constant<int, -1>{});
    else if (fast_sf == 2) phase_math(r, phase, active, first, std::integral_constant<int, 2>{});
    else if (fast_sf == 1) phase_math(r, phase, active, first, std::integral_constant<int, 1>{});
    else phase_math(r, phase, active, first, std::integral_constant<int, 0>{});
  };

  // ---- the tile loop ------------------------------------------------------------------------------------------------------
  // One tile per trip.  (Two software-pipelined forms were built and measured in round 5 -- the binning and ordering of tile
  // k - 1 under tile k's second DMA flight, carrying the coefficients, resp. their float images and the bin ids, across the
  // trip: both need the previous tile's state AND a phase's 32 raw values in registers at once, spill at 168 registers,
  // and a scratch reload waits behind the DMA in flight: 353 and 484 us against 268, EXPERIMENTS.)
  const unsigned my_tiles = tr.hi - tr.lo;
  unsigned pf_sink = 0;
  if (my_tiles) issue(0u, IC2<0>{});
  for (unsigned it = 0; it < my_tiles; it++) {
    const unsigned tile = tr.lo + it, rel = it;
    const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile * TILE_BLKS);
    const bool active = (unsigned)lane < blks_here;
    {
      T r[32];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows of phase 0 have landed (and everything older is done)
      eo_barrier();                                  // ... and the other wave's
      read32(r);
      eo_barrier();                                  // both waves have the phase in registers: the image is free
      issue(rel, IC2<1>{});
      if (pend) { flush_prev(); pend = false; }
      phase_any(r, IC2<0>{}, active, tile == 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      T r[32];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      eo_barrier();
      read32(r);
      eo_barrier();
      if (it + 1u < my_tiles) {
        issue(rel + 1u, IC2<0>{});
#if DCTZ_EO_PREFETCH
        // ... and the lines of its SECOND phase are asked for as well, one dword of each 128-byte line into a register
        // nothing reads: that DMA is issued with ~110 instructions in front of its wait (the image is one phase large), and
        // finds its data in the caches instead of in HBM
        {
          const int ln = lane_now();
          const int line = ROLE * 64 + ln;                                    // 128 lines = the 16 KiB of the phase: (block, slot)
          const int off = (int)((rel + 1u) * (unsigned)G::TILEB) + (line >> 1) * G::BLKB + EoMap::seg(1, line & 1) * 128;
          pf_sink = __builtin_amdgcn_raw_buffer_load_b32(r_in, off, 0, 0);
          asm volatile("" :: "v"(pf_sink));
        }
#endif
      }
      phase_any(r, IC2<1>{}, active, false);
    }
    __builtin_amdgcn_sched_barrier(0);
    T c[32];                                         // coefficient 2 i + ROLE of the block
    {
      // (the table's address through a register the compiler cannot see through: left alone, it hoists the scalar loads of
      // the transform's ~140 constants out of the tile loop, spills them, and every use is a v_readlane)
      CTab<T> tabl = tab;
      asm volatile("" : "+s"(tabl));
      dct64_fwd_half<T, ROLE, CTab<T>, true>(pr, pi, c, tabl);
    }
    if (p.coef != nullptr && active) {               // test tap: the coefficients as computed
#pragma unroll
      for (int i = 0; i < 32; i++) p.coef[((size_t)tile * TILE_BLKS + lane) * 64 + 2 * i + ROLE] = c[i];
    }
    if (ROLE == EO_EVEN && active) {
      if (STATS) acc.dcs += (double)c[0];            // orthonormal 64-point DCT: DC = (sum of the block) / 8
      if (p.last_is_full && tile * TILE_BLKS + lane == p.nfull - 1) p.ctl->q0 = (unsigned long long)to_bits(c[0]);   // :355-360
    }
    const float dc_here = (float)c[0];               // (even wave)
    __builtin_amdgcn_sched_barrier(0);
    // pass-1 binning (:363-414) of the lane's 32 coefficients, four = one dword of bin ids at a time (as k_compress)
    unsigned w[8];
    unsigned m = 0;                                  // bit i: coefficient 2 i + ROLE of this block is stored exactly
    auto bin_all = [&](auto fast, auto safe) {
#pragma unroll
      for (int g = 0; g < 8; g++) {
        float h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const T u = c[4 * g + i] - rmin;           // :377 / :402
          const T q = decltype(fast)::value ? bwd.core(u) : u / bwd.d;
          h[i] = bin_value<T, decltype(safe)::value>(c[4 * g + i], q, rmax);
        }
        if (ROLE == EO_EVEN && g == 0) h[0] = 0.0f;  // j = 0 is the DC slot (:361): never stored exactly, its id is set below
        unsigned wgd = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) wgd = __builtin_amdgcn_cvt_pk_u8_f32(h[i], i, wgd);
        asm volatile("" : "+v"(wgd));
        w[g] = wgd;
        // "stored exactly" = id 255 = a bin value of 255 or more (whole numbers, or beyond the byte's range); a group is
        // only looked at further when SOME lane of the wave has one
        const float hm = fmaxf(fmaxf(h[0], h[1]), fmaxf(h[2], h[3]));
        if (__builtin_amdgcn_ballot_w64(hm >= 255.0f))
          m |= ((h[0] >= 255.0f ? 1u : 0u) | (h[1] >= 255.0f ? 2u : 0u) | (h[2] >= 255.0f ? 4u : 0u) | (h[3] >= 255.0f ? 8u : 0u)) << (4 * g);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#ifdef DCTZ_EO_HOT
    bin_all(std::true_type{}, std::false_type{});
#else
    if (bwd.ok) { if (p.fast_bw & 2u) bin_all(std::true_type{}, std::false_type{}); else bin_all(std::true_type{}, std::true_type{}); }
    else bin_all(std::false_type{}, std::true_type{});
#endif
    if (ROLE == EO_EVEN) w[0] |= 0xFFu;              // :361 DC slot
    if (!active) m = 0;
    // ---- what the two waves tell each other: their masks, and the odd wave's bin ids
    const int lane = lane_now();
    lds_store_b32(xmask_at + (unsigned)(ROLE * 64 + lane) * 4u, m);
    if (ROLE == EO_ODD) {
      lds_store_b128(bins_at + (unsigned)lane * 64u, u32x4{w[0], w[1], w[2], w[3]});
      lds_store_b128(bins_at + (unsigned)lane * 64u + 16u, u32x4{w[4], w[5], w[6], w[7]});
    }
    eo_barrier();
    const unsigned mp = L.xmask[(ROLE ^ 1) * 64 + lane];
    if (ROLE == EO_EVEN) {
      unsigned pw[16];
      const u32x4 o0 = *(lds_cu4*)(binbuf + lane * 64), o1 = *(lds_cu4*)(binbuf + lane * 64 + 16);
      const unsigned wo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
      for (int g = 0; g < 8; g++) {                  // bytes: even positions from this wave, odd ones from the other
        pw[2 * g] = __builtin_amdgcn_perm(wo[g], w[g], 0x05010400u);
        pw[2 * g + 1] = __builtin_amdgcn_perm(wo[g], w[g], 0x07030602u);
      }
      // the tile's bin ids, 64 bytes per lane, into the layout their row stores read (k_compress's flush); the lane's own
      // 64 bytes of the buffer, which it has just read out
      const int f2 = (lane >> 1) & 3;
#pragma unroll
      for (int i = 0; i < 4; i++)
        lds_store_b128(bins_at + (unsigned)((lane * 4 + (i ^ f2)) * 16), u32x4{pw[4 * i], pw[4 * i + 1], pw[4 * i + 2], pw[4 * i + 3]});
    }
    const unsigned n = (unsigned)(__popc(m) + __popc(mp));
    const unsigned incl = wave_incl_scan(n);
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    const unsigned base = incl - n;                  // this block's place in the tile's piece
    const bool single = tot <= CAP;
    // position j = 2 i + ROLE of the block is item number popc(own & below(i)) + popc(other & below(i + ROLE)) of the block
    unsigned jv = (unsigned)ROLE;
    if (MODE == DCTZHIP_QT) asm volatile("" : "+v"(jv));
    for (unsigned lo = 0; lo < tot; lo += CAP) {
      if (lo != 0u) eo_barrier();                    // both waves have taken the rows of the round before out of the buffer
      const unsigned b0 = base - lo;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (__builtin_amdgcn_ballot_w64(((m >> (4 * g)) & 0xFu) != 0u)) {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            constexpr unsigned ALL = 0xFFFFFFFFu;
            const int i = 4 * g + k;
            const unsigned own_below = i == 0 ? 0u : (ALL >> (32 - i));
            const unsigned oth_below = (i + ROLE) == 0 ? 0u : (ALL >> (32 - (i + ROLE)));
            const bool f = ((m >> i) & 1u) != 0u;
            const unsigned pos = b0 + (unsigned)__popc(m & own_below) + (unsigned)__popc(mp & oth_below);
            const unsigned at = (f && pos < CAP) ? pos : CAP + (unsigned)(ROLE * 64 + lane);
            lds_store_item(stage_at + at * (unsigned)sizeof(Item), (Item)c[i]);     // :496-497 / :535-537 USE_TRUNCATE (EC)
            if (MODE == DCTZHIP_QT) lds_store_b8(stage_at + (unsigned)ST::ITEM_BYTES + at, jv + 2u * (unsigned)i);
          }
        }
      }
      if (!single) {                                 // a dense tile: round by round, at once
        eo_barrier();
        store_rows(run + lo, min(tot - lo, CAP));
      }
    }
    pend = true; p_rel = rel; p_run = run; p_cnt = single ? tot : 0u; p_dc = dc_here;
    run += tot;
  }
  if (pend) { eo_barrier(); flush_prev(); }
  if (ROLE == EO_EVEN && lane == 0) p.tile_cnt[wg] = run | LIST_IN_ORDER;
  if (MODE == DCTZHIP_QT) {
    eo_barrier();                                    // every ds_max of both waves is in
    if (ROLE == EO_EVEN) {
      const unsigned long long mq = L.qmax[lane];
      if (mq != 0ull) atomicMax(&p.ctl->qraw[lane], mq);
    }
  }
  if (STATS) {
    const double dc_scale = scale ? 8.0 * (double)sf : 8.0;
    double dmx = (double)acc.mx, dmn = (double)acc.mn, sm = acc.sum + acc.dcs * dc_scale;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      dmx = fmax(dmx, __shfl_down(dmx, d));
      dmn = fmin(dmn, __shfl_down(dmn, d));
      sm += __shfl_down(sm, d);
    }
    if (lane == 0) { L.stat[3 * ROLE] = dmx; L.stat[3 * ROLE + 1] = dmn; L.stat[3 * ROLE + 2] = sm; }
    eo_barrier();
    if (ROLE == EO_EVEN && lane == 0) {
      p.stat_part[3 * wg + 0] = fmax(L.stat[0], L.stat[3]);
      p.stat_part[3 * wg + 1] = fmin(L.stat[1], L.stat[4]);
      p.stat_part[3 * wg + 2] = L.stat[2] + L.stat[5];
    }
  }
}

template <int MODE, bool STATS>
__global__ __launch_bounds__(EO_WG) __attribute__((amdgpu_waves_per_eu(DCTZ_EO_WAVES))) void k_compress_eo(FwdParams<double> p) {
  __shared__ __attribute__((aligned(1024))) unsigned char tilebuf[EO_TILE_LDS];
  __shared__ __attribute__((aligned(16))) unsigned char binbuf[4096];
  __shared__ __attribute__((aligned(16))) unsigned char stagebuf[EoStage<MODE>::BYTES];
  __shared__ unsigned xmask[128];
  __shared__ unsigned long long qmax[MODE == DCTZHIP_QT ? 64 : 1];
  __shared__ double stat[6];
  const EoLds L = {(lds_u8*)tilebuf, (lds_u8*)binbuf, (lds_u8*)stagebuf, (lds_u32*)xmask, (lds_u64*)qmax, (lds_f64*)stat};
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (role == 0) compress_eo_role<MODE, STATS, EO_EVEN>(p, blockIdx.x, gridDim.x, L);
  else compress_eo_role<MODE, STATS, EO_ODD>(p, blockIdx.x, gridDim.x, L);
}

void launch_compress_eo(const FwdParams<double>& p, int mode, bool stats, int grid, hipStream_t s) {
  if (mode == DCTZHIP_EC) {
    if (stats) hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, true>), dim3(grid), dim3(EO_WG), 0, s, p);
    else hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, false>), dim3(grid), dim3(EO_WG), 0, s, p);
  } else {
    if (stats) hipLaunchKernelGGL((k_compress_eo<DCTZHIP_QT, true>), dim3(grid), dim3(EO_WG), 0, s, p);
    else hipLaunchKernelGGL((k_compress_eo<DCTZHIP_QT, false>), dim3(grid), dim3(EO_WG), 0, s, p);
  }
}

// resident workgroups (of two waves) per CU: registers and LDS
int compress_eo_occupancy(int mode, bool stats) {
  int n = 0;
  hipError_t e;
  if (mode == DCTZHIP_EC) e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, true>, EO_WG, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, false>, EO_WG, 0);
  else e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_QT, true>, EO_WG, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_QT, false>, EO_WG, 0);
  if (e != hipSuccess || n <= 0) n = (int)((size_t)160 * 1024 / (mode == DCTZHIP_EC ? eo_lds_bytes<DCTZHIP_EC>() : eo_lds_bytes<DCTZHIP_QT>()));
  return n;
}

}  // namespace dctz
