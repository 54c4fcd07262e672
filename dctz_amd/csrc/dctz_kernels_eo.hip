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

// ---- single-pass placement of AC_exact (EC): decoupled look-back over the tiles' counts ------------------------------------
// With the tile's exact coefficients in the reference's order in LDS, all that the tile still needs to write them at their
// FINAL place is the running tot_AC_exact_count of dctz-comp-lib.c:478-544 at its first block = the counts of all tiles in
// front of it.  Tiles are handed out in order by ticket counters (whichever workgroup is running takes the next tile, so
// a workgroup that is not resident holds nothing anybody waits for: see the loop), every tile posts its count
// ("aggregate", A) as soon as it is known, and later -- when its stores are due, half a tile on -- looks back over the
// descriptors in front of it: 64 of them per load, summing counts until it meets one that already carries its prefix
// (P), and posts its own inclusive prefix.  Descriptors are 8-byte granules {epoch of the call << 2 | state, value},
// written by one agent-scope store and read with agent-scope loads (nothing to clear between calls: another call's tags do
// not match).  No workgroup-local lists, no k_compact_ac: every exact coefficient is written once.
constexpr unsigned long long EO_SPIN_TICKS = 2000000ull;   // 20 ms of the 100 MHz clock: a look-back that sees no progress gives up (Ctl::error)
constexpr unsigned EO_ERR_LOOKBACK = 5u;
typedef __attribute__((address_space(1))) unsigned long long eo_gu64;
__device__ __forceinline__ unsigned long long eo_ld_agent(const unsigned long long* p) {
  return __hip_atomic_load((eo_gu64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void eo_st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store((eo_gu64*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long eo_desc(unsigned epoch, unsigned state, unsigned value) {
  return ((unsigned long long)((epoch << 2) | state) << 32) | value;
}
// the exclusive prefix of tile t (the number of exact coefficients in front of it); every lane gets it.
// 1: there it is; 0: (block == false only) a tile in front has not posted its count yet; -1: gave up waiting.
__device__ __forceinline__ int eo_look_back(const unsigned long long* desc, unsigned epoch, unsigned t, unsigned& excl_out, bool block = true) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned excl = 0;
  unsigned pos = t;                                  // descriptors pos - 1, pos - 2, ... are looked at next
  unsigned long long t0 = 0;
  for (;;) {
    if (pos == 0u) break;
    const bool valid = lane < pos;
    const unsigned long long v = valid ? eo_ld_agent(desc + (pos - 1u - lane)) : eo_desc(epoch, 2u, 0u);   // (in front of tile 0: prefix 0)
    const unsigned tag = (unsigned)(v >> 32);
    const bool ready = (tag >> 2) == (epoch & 0x3FFFFFFFu) && (tag & 3u) != 0u;
    const unsigned long long mR = __builtin_amdgcn_ballot_w64(ready), mP = __builtin_amdgcn_ballot_w64(ready && (tag & 3u) == 2u);
    if (mP != 0ull) {
      const unsigned fp = (unsigned)__builtin_ctzll(mP);             // the closest tile that has its prefix
      const unsigned long long low = fp == 0u ? 0ull : (~0ull >> (64u - fp));
      if ((mR & low) == low) {                                       // ... and every tile between it and us has its count
        unsigned s = lane <= fp ? (unsigned)v : 0u;
        s = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(s), 63);
        excl += s;
        break;
      }
    } else if (mR == ~0ull) {                                        // 64 counts, no prefix among them: further back
      excl += (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan((unsigned)v), 63);
      pos -= 64u;                                                    // (valid for all 64 lanes, so pos >= 64)
      continue;
    }
    if (!block) { excl_out = 0u; return 0; }
    __builtin_amdgcn_s_sleep(2);
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0ull) t0 = now;
    else if (now - t0 > EO_SPIN_TICKS) { excl_out = 0u; return -1; }
  }
  excl_out = excl;
  return 1;
}

template <int MODE, bool STATS, int ROLE, bool DIRECT>
__device__ __forceinline__ void compress_eo_role(const FwdParams<double>& p, const unsigned wg, const unsigned nwg, const EoLds& L) {
  using T = double;
  using G = Geo<T, 2>;
  using ST = EoStage<MODE>;
  using Item = typename ST::Item;
  static_assert(!DIRECT || MODE == DCTZHIP_EC, "QT normalises with the table of the whole array: its items go through the lists");
  constexpr unsigned CAP = (unsigned)ST::CAP;
  lds_u8* const tilebuf = L.tile;
  lds_u8* const binbuf = L.bins;
  lds_u8* const stagebuf = L.stage;
  const int lane = threadIdx.x & 63;
  // lists (not DIRECT): workgroup b owns the contiguous tiles [lo, hi) and its list lives in their slots
  const TileRange tr = DIRECT ? TileRange{0u, 0u} : tile_range(wg, nwg, p.ntiles);
  const unsigned list_base = tr.lo * TILE_ELEMS;
  const int list_slots = (int)((tr.hi - tr.lo) * (unsigned)TILE_ELEMS);
  const __amdgpu_buffer_rsrc_t r_list = DIRECT ? __builtin_amdgcn_make_buffer_rsrc(p.ac, 0, 0, 0x00020000) : ((MODE == DCTZHIP_EC)
      ? __builtin_amdgcn_make_buffer_rsrc(p.ac_tmp + list_base, 0, list_slots * 4, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(p.qt_item + list_base, 0, list_slots * (int)sizeof(T), 0x00020000));
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
  auto blocks_of = [&](unsigned tile) { return min((unsigned)TILE_BLKS, p.nfull - tile * (unsigned)TILE_BLKS); };

  // this wave's half of a phase's rows of a tile, HBM -> LDS (a descriptor per tile: offsets are constants, and the range
  // check zero-fills whatever lies beyond the last whole block)
  auto issue = [&](unsigned tile, auto phase) {
    constexpr int PHASE = decltype(phase)::value;
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.x + (size_t)tile * TILE_ELEMS), 0, (int)(blocks_of(tile) * (unsigned)G::BLKB), 0x00020000);
#pragma unroll
    for (int jg = 4 * ROLE; jg < 4 * ROLE + 4; jg++)
#pragma unroll
      for (int s = 0; s < 2; s++)
        DMA16(r_in, tilebuf + (jg * 2 + s) * 1024, tm.g_of(jg), jg * 8 * G::BLKB + EoMap::seg(PHASE, s) * 128, 2 /* nt */);
  };

  // the outputs of the tile before, on their way out (see the head of the file)
  bool pend = false;
  unsigned p_tile = 0, p_run = 0, p_cnt = 0, p_tot = 0;
  float p_dc = 0.f;
  unsigned run = 0;                                  // lists: length of the workgroup's list so far (uniform, the same in both waves)
  bool gave_up = false;

  // (lane numbers through a register the compiler cannot see through wherever per-lane addresses are made of them outside
  // the arithmetic: kept alive across the loop they are spilled to scratch, and a scratch reload waits behind the DMA in flight)
  auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  // rows [0, cnt) of the staged piece -> the workgroup's list (DIRECT: AC_exact[]) at `at0`; the two waves take every other row
  auto store_rows = [&](unsigned at0, unsigned cnt, unsigned first_row = (unsigned)ROLE, unsigned row_step = 2u) {
    const int ln = lane_now();
    const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(p.ac + (DIRECT ? at0 : 0u), 0, DIRECT ? (int)(cnt * 4u) : 0, 0x00020000);
    for (unsigned r = first_row; r * 64u < cnt; r += row_step) {
      const unsigned e = r * 64u + (unsigned)ln;
      const bool in = e < cnt;
      const int at = (int)(at0 + e);
      if (MODE == DCTZHIP_EC) {
        const unsigned v = *(const lds_u32*)(stagebuf + e * 4u);
        if (DIRECT) __builtin_amdgcn_raw_buffer_store_b32(v, r_ac, (int)(e * 4u), 0, 0);      // (beyond cnt: outside the descriptor)
        else __builtin_amdgcn_raw_buffer_store_b32(v, r_list, in ? at * 4 : 0x7FFFFFF0, 0, 0);
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
  // DIRECT: where tile t's piece of AC_exact[] starts (waits for it); the even wave also posts the tile's inclusive prefix
  auto posted = [&](unsigned t, unsigned excl, unsigned tot) {
    if (ROLE == EO_EVEN && lane == 0) {
      eo_st_agent(p.lb_desc + t, eo_desc(p.lb_epoch, 2u, excl + tot));
      if (t == p.ntiles - 1u) p.ctl->cnt_total = excl + tot;       // tot_AC_exact_count of the full blocks (:478-544)
    }
  };
  auto place_of = [&](unsigned t, unsigned tot) -> unsigned {
    unsigned excl = 0;
    if (eo_look_back(p.lb_desc, p.lb_epoch, t, excl) < 0) gave_up = true;
    posted(t, excl, tot);
    return excl;
  };
  // DIRECT, even wave: the pending piece of AC_exact[] (a tile's exact coefficients, in order, in the staging buffer) leaves as
  // soon as every tile in front of it has posted its count; asked at several points of the next tile, waited for only at the
  // last one -- behind the next tile's OWN count, so that no workgroup ever keeps the others waiting for a count while it
  // waits itself (with the look-back waited for where the stores are first due, the tiles ran in convoys: 460 us)
  bool pend_rows = false;
  auto try_rows = [&](bool block) {
    unsigned excl = 0;
    const int st = eo_look_back(p.lb_desc, p.lb_epoch, p_tile, excl, block);
    if (st == 0) return;
    if (st < 0) gave_up = true;
    posted(p_tile, excl, p_tot);
    store_rows(excl, p_cnt, 0u, 1u);
    pend_rows = false;
  };
  auto flush_prev = [&]() {
    if (!DIRECT) store_rows(p_run, p_cnt);
    if (ROLE == EO_EVEN) {
      // bin ids: (through their staging buffer) 1 KiB rows of 16 consecutive blocks; DC (:350-351 USE_TRUNCATE)
      const int lane = lane_now();
      const unsigned nb = blocks_of(p_tile);
      const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(p.bin + (size_t)p_tile * TILE_ELEMS, 0, (int)(nb * 64u), 0x00020000);
      const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(p.dc + (size_t)p_tile * TILE_BLKS, 0, (int)(nb * 4u), 0x00020000);
      const int bin_goff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) * 16);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32x4 v = *(lds_cu4*)(binbuf + i * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, r_bin, bin_goff + i * 1024, 0, 0);
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p_dc), r_dc, lane * 4, 0, 0);
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

  // ---- one tile ------------------------------------------------------------------------------------------------------------
  // Phase 0 of `tile` is in flight when this is entered; phase 0 of `next_tile` (if has_next) when it is left.
  // (Two software-pipelined forms were built and measured in round 5 -- the binning and ordering of tile k - 1 under tile k's
  // second DMA flight, carrying the coefficients, resp. their float images and the bin ids, across the trip: both need the
  // previous tile's state AND a phase's 32 raw values in registers at once, spill at 168 registers, and a scratch reload
  // waits behind the DMA in flight: 353 and 484 us against 268, EXPERIMENTS.)
  // Hooks (the ticket draw of the DIRECT loop): after_first_issue -- behind the DMA issue of the second phase;
  // after_second_wait -- behind the wait for that DMA, in front of the barrier; next_of(has_next, next_tile) -- behind the
  // barrier that frees the image for the next tile's first phase.
  auto do_tile = [&](const unsigned tile, auto&& after_first_issue, auto&& after_second_wait, auto&& next_of) {
    const unsigned blks_here = blocks_of(tile);
    const bool active = (unsigned)lane < blks_here;
    {
      T r[32];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows of phase 0 have landed (and everything older is done)
      eo_barrier();                                  // ... and the other wave's
      read32(r);
      eo_barrier();                                  // both waves have the phase in registers: the image is free
      issue(tile, IC2<1>{});
      after_first_issue();
      if (pend) { flush_prev(); pend = false; }
      if (DIRECT && ROLE == EO_EVEN && pend_rows) try_rows(false);
      phase_any(r, IC2<0>{}, active, tile == 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      T r[32];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      after_second_wait();
      eo_barrier();
      read32(r);
      eo_barrier();
      bool has_next = false;
      unsigned next_tile = 0;
      next_of(has_next, next_tile);
      if (has_next) issue(next_tile, IC2<0>{});
      if (DIRECT && ROLE == EO_EVEN && pend_rows) try_rows(false);
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
    if (DIRECT && ROLE == EO_EVEN && lane == 0) L.xmask[131] = pend_rows ? 1u : 0u;     // (the piece before is still in the staging buffer)
    eo_barrier();
    const unsigned mp = L.xmask[(ROLE ^ 1) * 64 + lane];
    const bool late = DIRECT && __builtin_amdgcn_readfirstlane((int)L.xmask[131]) != 0;
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
    unsigned at_dense = 0;
    if (DIRECT) {
      // the tile's count goes on the board at once (tile 0 has its prefix with it); its place is looked up when its stores
      // are due -- now, if the piece leaves in several rounds
      if (ROLE == EO_EVEN && lane == 0) eo_st_agent(p.lb_desc + tile, eo_desc(p.lb_epoch, 1u, tot));
      if (late) {                                    // now it is waited for (this tile's count is out); the odd wave keeps off the buffer meanwhile
        if (ROLE == EO_EVEN) try_rows(true);
        eo_barrier();
      }
      if (!single) at_dense = place_of(tile, tot);
      if (p.direct == 2u && tile == 1u) gave_up = true;      // (tests: a look-back that gave up, DCTZHIP_EO_LB_FAIL)
    }
    // position j = 2 i + ROLE of the block is item number popc(own & below(i)) + popc(other & below(i + ROLE)) of the block
    unsigned jv = (unsigned)ROLE;
    if (MODE == DCTZHIP_QT) asm volatile("" : "+v"(jv));
    // (sparse tiles -- one round --: the masks go through registers the compiler cannot see through, inside each group's
    // branch.  The ranks are pure arithmetic, and left alone all 32 of them are computed in front of the first group: 130
    // instructions a tile whose flags sit in one or two groups never needs.  Dense tiles -- several rounds, every group
    // entered --: left alone, so that the ranks are computed once for all rounds: 405 -> 384 us at p = 0.69.)
    auto scatter = [&](const unsigned lo, auto sparse) {
      const unsigned b0 = base - lo;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (__builtin_amdgcn_ballot_w64(((m >> (4 * g)) & 0xFu) != 0u)) {
          unsigned mg = m, mpg = mp;
          if (decltype(sparse)::value) asm volatile("" : "+v"(mg), "+v"(mpg));
#pragma unroll
          for (int k = 0; k < 4; k++) {
            constexpr unsigned ALL = 0xFFFFFFFFu;
            const int i = 4 * g + k;
            const unsigned own_below = i == 0 ? 0u : (ALL >> (32 - i));
            const unsigned oth_below = (i + ROLE) == 0 ? 0u : (ALL >> (32 - (i + ROLE)));
            const bool f = ((mg >> i) & 1u) != 0u;
            const unsigned pos = b0 + (unsigned)__popc(mg & own_below) + (unsigned)__popc(mpg & oth_below);
            const unsigned at = (f && pos < CAP) ? pos : CAP + (unsigned)(ROLE * 64 + lane);
            lds_store_item(stage_at + at * (unsigned)sizeof(Item), (Item)c[i]);     // :496-497 / :535-537 USE_TRUNCATE (EC)
            if (MODE == DCTZHIP_QT) lds_store_b8(stage_at + (unsigned)ST::ITEM_BYTES + at, jv + 2u * (unsigned)i);
          }
        }
      }
    };
    if (single) { if (tot != 0u) scatter(0u, std::true_type{}); }
    else for (unsigned lo = 0; lo < tot; lo += CAP) {
      if (lo != 0u) eo_barrier();                    // both waves have taken the rows of the round before out of the buffer
      scatter(lo, std::false_type{});
      eo_barrier();                                  // a dense tile: round by round, at once
      store_rows((DIRECT ? at_dense : run) + lo, min(tot - lo, CAP));
    }
    pend = true; p_tile = tile; p_run = run; p_cnt = single ? tot : 0u; p_tot = tot; p_dc = dc_here;
    pend_rows = DIRECT && single;                    // (a piece that left in rounds is placed already)
    run += tot;
  };

  if (!DIRECT) {
    if (tr.lo < tr.hi) issue(tr.lo, IC2<0>{});
    for (unsigned tile = tr.lo; tile < tr.hi; tile++)
      do_tile(tile, []() {}, []() {}, [&](bool& has_next, unsigned& next_tile) { has_next = tile + 1u < tr.hi; next_tile = tile + 1u; });
  } else {
    // Tickets.  A tile is a ticket (with several tiles per ticket, a workgroup's later tiles post their counts only after its
    // first tile's look-back -- which waits for the LAST tiles of the tickets in front: the chunks run one after the other;
    // measured, round 5).  One counter would have to serve 130 draws per microsecond (a word saturates at ~88), so there are
    // eight, 64 bytes apart: counter x hands out the tiles x, x + 8, x + 16, ... and a workgroup draws from the counter of the
    // XCD it runs on (workgroups are dealt round-robin over the XCDs: eight streams of equal speed); when that one is
    // exhausted it goes on to the others, so every tile is drawn by a workgroup that is running, whatever the placement.
    // The even wave's first lane draws; the number crosses to the odd wave through LDS behind a barrier that is there anyway:
    // the NEXT tile is drawn behind the DMA issue of this tile's second phase (the wait for that DMA is the wait for the
    // number too), stored in front of the barrier that follows the wait, and read where the next tile's first DMA is issued.
    lds_u32* const tick = L.xmask + 128;             // [0], [1]: the draw made during a tile, by parity; [2]: a draw of its own
    unsigned shard;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(shard));     // HW_REG_XCC_ID (any value will do: speed only)
    shard &= 7u;
    unsigned tried = 0;                              // counters found exhausted
    auto draw_sync = [&]() -> unsigned {             // a tile of the first counter, from `shard` on, that still has one; else ~0
      for (; tried < 8u; tried++, shard = (shard + 1u) & 7u) {
        if (ROLE == EO_EVEN && lane == 0) tick[2] = atomicAdd(p.lb_ticket + shard * 16u, 1u);
        eo_barrier();
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)tick[2]) * 8u + shard;
        eo_barrier();                                // (read by both waves before the next draw overwrites it)
        if (t < p.ntiles) return t;
      }
      return ~0u;
    };
    unsigned tile = draw_sync();
    if (tile != ~0u) issue(tile, IC2<0>{});
    unsigned nd = 0;
    while (tile != ~0u) {
      const unsigned slot = nd & 1u;
      nd++;
      unsigned drawn = 0, next = ~0u;
      do_tile(tile,
              [&]() { if (ROLE == EO_EVEN && lane == 0) drawn = atomicAdd(p.lb_ticket + shard * 16u, 1u); },
              [&]() { if (ROLE == EO_EVEN && lane == 0) tick[slot] = drawn; },
              [&](bool& has_next, unsigned& next_tile) {
                next_tile = (unsigned)__builtin_amdgcn_readfirstlane((int)tick[slot]) * 8u + shard;
                has_next = next_tile < p.ntiles;
                if (has_next) next = next_tile;
              });
      if (next == ~0u) {                             // this counter is exhausted: the others (the tail of the launch)
        shard = (shard + 1u) & 7u; tried++;
        next = draw_sync();
        if (next != ~0u) issue(next, IC2<0>{});
      }
      tile = next;
    }
  }
  if (pend) { eo_barrier(); flush_prev(); }
  if (DIRECT && ROLE == EO_EVEN && pend_rows) try_rows(true);
  if (DIRECT && gave_up && lane == 0) atomicExch(&p.ctl->error, EO_ERR_LOOKBACK);
  if (!DIRECT && ROLE == EO_EVEN && lane == 0) p.tile_cnt[wg] = run | LIST_IN_ORDER;
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

template <int MODE, bool STATS, bool DIRECT>
__global__ __launch_bounds__(EO_WG) __attribute__((amdgpu_waves_per_eu(DCTZ_EO_WAVES))) void k_compress_eo(FwdParams<double> p) {
  __shared__ __attribute__((aligned(1024))) unsigned char tilebuf[EO_TILE_LDS];
  __shared__ __attribute__((aligned(16))) unsigned char binbuf[4096];
  __shared__ __attribute__((aligned(16))) unsigned char stagebuf[EoStage<MODE>::BYTES];
  __shared__ unsigned xmask[128 + 4];
  __shared__ unsigned long long qmax[MODE == DCTZHIP_QT ? 64 : 1];
  __shared__ double stat[6];
  const EoLds L = {(lds_u8*)tilebuf, (lds_u8*)binbuf, (lds_u8*)stagebuf, (lds_u32*)xmask, (lds_u64*)qmax, (lds_f64*)stat};
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (role == 0) compress_eo_role<MODE, STATS, EO_EVEN, DIRECT>(p, blockIdx.x, gridDim.x, L);
  else compress_eo_role<MODE, STATS, EO_ODD, DIRECT>(p, blockIdx.x, gridDim.x, L);
}

void launch_compress_eo(const FwdParams<double>& p, int mode, bool stats, int grid, hipStream_t s) {
  if (mode == DCTZHIP_EC && p.direct) {
    if (stats) hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, true, true>), dim3(grid), dim3(EO_WG), 0, s, p);
    else hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, false, true>), dim3(grid), dim3(EO_WG), 0, s, p);
  } else if (mode == DCTZHIP_EC) {
    if (stats) hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, true, false>), dim3(grid), dim3(EO_WG), 0, s, p);
    else hipLaunchKernelGGL((k_compress_eo<DCTZHIP_EC, false, false>), dim3(grid), dim3(EO_WG), 0, s, p);
  } else {
    if (stats) hipLaunchKernelGGL((k_compress_eo<DCTZHIP_QT, true, false>), dim3(grid), dim3(EO_WG), 0, s, p);
    else hipLaunchKernelGGL((k_compress_eo<DCTZHIP_QT, false, false>), dim3(grid), dim3(EO_WG), 0, s, p);
  }
}

// resident workgroups (of two waves) per CU: registers and LDS
int compress_eo_occupancy(int mode, bool stats, bool direct) {
  int n = 0;
  hipError_t e;
  if (mode == DCTZHIP_EC && direct) e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, true, true>, EO_WG, 0)
                                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, false, true>, EO_WG, 0);
  else if (mode == DCTZHIP_EC) e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, true, false>, EO_WG, 0)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_EC, false, false>, EO_WG, 0);
  else e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_QT, true, false>, EO_WG, 0)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_eo<DCTZHIP_QT, false, false>, EO_WG, 0);
  if (e != hipSuccess || n <= 0) n = (int)((size_t)160 * 1024 / (mode == DCTZHIP_EC ? eo_lds_bytes<DCTZHIP_EC>() : eo_lds_bytes<DCTZHIP_QT>()));
  return n;
}

}  // namespace dctz
