// dctz_kernels.hip -- gfx950 (MI355X) kernels of the DCTZ hot path: k_compress, k_decompress and what hangs on them
// (remainder blocks, the QT maxima, the list placement, the flag counts, the hand-off).  The statistics passes, scaling,
// batched transforms, gather / scatter and PSNR kernels live in dctz_kernels_aux.hip, the shared device helpers
// (FastDiv, TileMap, LDS-DMA issue, StatAcc, ...) in dctz_kernel_common.h.
//
// Work decomposition (both directions):
//   * a TILE is 64 consecutive 64-element blocks (4096 elements, 32 KiB fp64) = one loop trip of ONE
//     wavefront; LANE b OWNS BLOCK b of the tile and runs its whole 64-point transform in registers
//     (dct64_block.h): no cross-lane traffic, no twiddle lookups by lane index (the constants are
//     wave-uniform: scalar loads), every register index a compile-time constant;
//   * HBM is touched with 16 bytes per lane in whole 128-byte lines: the input tile goes HBM -> LDS by
//     LDS-DMA (buffer_load_dwordx4 ... lds, no VGPRs, nt policy), issued one whole tile ahead, and the
//     lane -> global-address map of every DMA row is chosen so that the image in LDS is the transposed,
//     bank-conflict-free one (TileMap); the reconstruction goes registers -> LDS (same image) -> 1 KiB rows;
//   * workgroups are single wavefronts, the grid is persistent (as many workgroups per CU as the LDS
//     admits: k_compress 8, k_decompress 4 for fp64 / 7 for fp32) and workgroup b owns the contiguous tile range
//     [b*ntiles/G, (b+1)*ntiles/G), so its exceptions form one contiguous piece of AC_exact[];
//   * the ordered stream of "stored exactly" coefficients (AC_exact) is placed in two levels: a lane
//     parks the exceptions of its block in a private LDS strip while it bins, a wave scan of the
//     counts gives every block its place in the workgroup's list, and k_compact_ac moves every list to its
//     place -- the sum of the lengths of the lists before it, added up by the list's own workgroup; on decode
//     k_count_tiles leaves the counts per tile and per workgroup of k_decompress.  No scan kernels, and the big
//     kernels have no inter-workgroup traffic;
//   * k_compress<double> moves a tile through LDS in two phases (two waves per SIMD cover each other); the outputs
//     of tile k are flushed after the next DMA of tile k+1 has been issued, so that the wait for tile k+1's data
//     never sits behind tile k's stores;
//   * calc_data_stat rides inside k_compress (STATS) behind a sampled guess of sf that the DEVICE turns into the
//     scaling factor (k_stats_final_sf, host-built decade tables) and the host verifies afterwards;
//   * what the host waits for at the end of a call is handed over by the first workgroup of the call's last big
//     kernel (finish_body), while the GPU drains;
//   * multi-dimensional blocks (8 x 8 / 4 x 4 x 4 tiles, dct_nd_block.h): the same kernels with the block transform
//     swapped (GEOM) and, where no tile is padded, the array addressed in place (NdDirect).
//
// Store-data hazard (gfx950, found in round 1 as rare corruption of QT reconstructions, reproduced in
// tools/ubench/probe_r2.hip): a buffer_store_dwordx4 whose soffset is an SGPR, followed with NO wait
// state by a VALU write of its data registers, stores the NEW register contents in ~0.7 % of the cases;
// LLVM's hazard recognizer only covers the form without a register soffset.  Every 16-byte buffer store
// here therefore passes soffset = 0 (offsets ride in the VGPR / the immediate), which the recognizer
// guards; tools/check_isa.py rejects any other form in the built code object.
//
// Reference code replaced: see include/dctz_hip.h (per entry point) and the comment on each kernel.
// Built with -ffp-contract=off: the arithmetic that the reference does unfused (gcc, baseline x86-64,
// reference Makefile:2) is unfused here too; the transform's fused operations are explicit.
#include <cstring>
#include "dctz_kernel_common.h"

#ifndef DCTZ_PART
#define DCTZ_PART 0          /* n > 0: this translation unit is the n-th hot kernel alone (see the end of the file) */
#endif

namespace dctz {

// Hand-off of a call's results to the host, by ONE workgroup: final reduction of the fused statistics (nparts > 0),
// results -> host box, then the sequence number.  Everything it reads was written by EARLIER kernels of the call, so it
// runs in the first workgroup of the call's last big kernel (k_compact_ac / k_decompress) as soon as that kernel
// starts: the host gets its few words while the GPU is still busy and has the next call's launches queued by the time
// the stream drains (results are complete in STREAM order, as with any asynchronous launch).  k_finish is the same
// hand-off as a kernel of its own, for the calls whose last kernel is a different one (remainder block on decode).
// (Folding it into the workgroup that FINISHES last was measured and dropped: the agent-scope release fence every
// workgroup then needs writes back the whole L2 of its XCD -- k_decompress 0.214 -> 0.272 ms, DESIGN.md.)
struct FinBody : FinArgs {
  bool cnt_known, err_known;       // the caller has computed tot_AC_exact_count / the error flag itself (else: from the control block)
  unsigned cnt_total, error;
};
template <bool WITH_STATS>
__device__ __forceinline__ void finish_body(const FinBody& f) {
  const int t = threadIdx.x;
  Ctl* ctl = f.ctl;
  HostBox* box = f.box;
  if (WITH_STATS && f.nparts > 0) {
    double dmx, dmn, sum;
    reduce_parts(f.part, f.nparts, dmx, dmn, sum);
    if (t == 0) { box->fstats[0] = dmx; box->fstats[1] = dmn; box->fstats[2] = sum; }
  }
  if (WITH_STATS) {                                  // (compress side: the QT table's raw maxima, the last block's DC)
    for (int i = t; i < 64; i += (int)blockDim.x) box->qraw[i] = ctl->qraw[i];
    if (t == 0) box->q0 = ctl->q0;
    if (t == 0 && f.guess != nullptr) { box->sf_used = f.guess->sf; box->fast_used = f.guess->fast_sf; }
  }
  if (t == 0) { box->cnt_total = f.cnt_known ? f.cnt_total : ctl->cnt_total; box->error = f.err_known ? f.error : ctl->error; }
  __threadfence_system();
  __syncthreads();                                   // all box writes issued and fenced
  if (t == 0) box_publish(&box->seq_done, f.seq);
}

#if DCTZ_PART == 0
__global__ __launch_bounds__(SWG) void k_finish(FinArgs a) {
  if (a.zero_words != nullptr) for (unsigned i = threadIdx.x; i < a.nzero; i += SWG) a.zero_words[i] = 0u;
  FinBody f;
  f.ctl = a.ctl; f.part = a.part; f.nparts = a.nparts; f.box = a.box; f.seq = a.seq; f.guess = a.guess;
  f.zero_words = nullptr; f.nzero = 0;
  f.cnt_known = false; f.err_known = false; f.cnt_total = 0; f.error = 0;
  finish_body<true>(f);
}
#endif

// ================================================================= compress ==
// Fused: [calc_data_stat util.c:12-44 ->] scale (dctz-comp-lib.c:193-216) -> DCT-II per block (:337-340,
// dct.c:55-103) -> DC (:350-351) -> pass-1 binning (:361-414) -> ordered exception stream (:478-544),
// full 64-element blocks only.
//
// (bin_value -- the binning of one coefficient in floating point -- lives in dctz_kernel_common.h: k_compress_one shares it)
// A list's entry in tile_cnt[]: its length, and LIST_IN_ORDER when every tile of the workgroup had items in its first
// sub-list only -- block-major over the first range of j is then the reference's order (a smooth field: what is stored
// exactly are the lowest frequencies), and k_compact_ac copies the list as it is.
// (LIST_IN_ORDER, LIST_LEN: dctz_device.h)
template <typename T>
size_t compress_lds_bytes(int mode) {              // tile image + sub-list staging (+ positions, QT); must match k_compress's static arrays
  using G = Geo<T, Phases<T>::C>;
  if (mode != DCTZHIP_QT) return (size_t)G::PHB + Sub<T, DCTZHIP_EC>::BYTES;
  return (size_t)G::PHB + Sub<T, DCTZHIP_QT>::BYTES + (Sub<T, DCTZHIP_QT>::PACKED ? 0 : 64 * sizeof(typename Traits<T>::Bits));   // + the per-position maxima
}

// PH = 1: the whole tile (fp32: 16 KiB) sits in LDS.  PH = 2 (fp64): half a tile at a time (16 KiB), eight single-wave
// workgroups per CU = two waves per SIMD that cover each other's waits.  The bin ids / DC / per-block counts of tile k
// are flushed only after the next DMA of tile k + 1 has been issued, so that the wait for tile k + 1's first phase
// never sits behind them.
// GEOM: what the 64 values of a block are -- the reference's 64 consecutive elements (GEOM_1D), or an 8 x 8 / 4 x 4 x 4
// tile of a multi-dimensional array that k_gather_nd has laid out block after block (dct_nd_block.h); only the
// transform differs.
// Waves per SIMD the register allocation of k_compress aims at: fp32 EC fits three (12 KiB of LDS per wave; a few
// registers over the 168 that allows are spilled: 0.166 against 0.180 ms at p = 17 %), everything else two.
#ifndef DCTZ_WPE32
#define DCTZ_WPE32 3
#endif
#ifndef DCTZ_DMA_SPREAD
#define DCTZ_DMA_SPREAD 1
#endif
#ifndef DCTZ_BIN_STORE_AUX
#define DCTZ_BIN_STORE_AUX 0     /* cache policy of k_compress's bin_index rows: plain (k_count_tiles / the entropy stage read them next) */
#endif
template <int I> using IC = std::integral_constant<int, I>;
template <typename T, int MODE, int PH> constexpr int compress_waves() { return (sizeof(T) == 4 && MODE == DCTZHIP_EC && DCTZ_WPE32) ? DCTZ_WPE32 : PH; }
// The body is shared by two launch shapes: k_compress (one array per launch: workgroup wg = blockIdx.x of nwg = gridDim.x)
// and k_compress_batch (many arrays per launch: the workgroup looks its array up and is workgroup wg of the nwg that array
// got).
// SC: the scaled values x / sf go back out as well (FwdParams::scaled: the reference's in-place division of the caller's
// array, dctz-comp-lib.c:193-216, which otherwise is a 2 s bytes / element pass of its own, k_scale): each half of a
// tile takes the way k_decompress's output takes -- registers -> the transposed image -> 1 KiB rows -- through the one
// image there is, in the window where it is free (the half is in registers, the next DMA not yet issued); that DMA
// therefore starts ~3 K cycles later than in the plain kernel, which is why this is a variant and not the kernel.
template <typename T, int MODE, bool STATS, int PH, int GEOM, bool SC = false>
__device__ __forceinline__ void compress_body(const FwdParams<T>& p, const unsigned wg, const unsigned nwg) {
  using G = Geo<T, PH>;
  using S = Sub<T, MODE>;
  using Item = typename S::Item;                     // what a "stored exactly" coefficient is on its way out (dctz_device.h: Sub)
  constexpr int QW = S::QW, NQ = S::NQ;
  // separate arrays, so that the compiler can tell the DMA target from the staging buffer (a pending LDS-DMA
  // forces a vmcnt(0) in front of every LDS read it may alias)
  __shared__ __attribute__((aligned(1024))) unsigned char tilebuf[G::PHB];
  __shared__ __attribute__((aligned(16))) unsigned char excbuf[S::BYTES];            // one sub-list; also: the tile's bin ids on their way out
  Item* const items = reinterpret_cast<Item*>(excbuf);
  unsigned char* const jbuf = excbuf + S::ITEM_BYTES;                                // QT: position j of every staged item
  const unsigned exc_at = lds_offset(excbuf);                                        // (stores into it: dctz_kernel_common.h, lds_store_*)
  // QT: per-position maximum |coef| over the out-of-range coefficients (dctz-comp-lib.c:371-372 / :396-397), kept per wave
  // while the sub-lists go out (one LDS atomic per item) and merged into Ctl::qraw at the end -- no pass over the lists
  // for it (round 2 needed one for fp64, k_qt_max, 24 us on 512^3: the strip flush had no register to spare; the row
  // loop of the sub-lists has).  fp32: an LDS array of its own for the whole kernel; fp64 (S::PACKED): the tail of the
  // staging buffer for the time of a tile's sub-lists, collected into lane j's register for position j at the tile's end
  // (the buffer serves the bin ids' way out between two tiles).
  using QBits = typename Traits<T>::Bits;
  constexpr bool QMAX_HERE = (MODE == DCTZHIP_QT);
  __shared__ QBits qmax_lds[(QMAX_HERE && !S::PACKED) ? 64 : 1];
  if (QMAX_HERE && !S::PACKED) qmax_lds[threadIdx.x] = 0;
  const unsigned qmax_at = S::PACKED ? exc_at + (unsigned)S::QMAX_AT : lds_offset(qmax_lds);
  QBits qreg = 0;
  const int lane = threadIdx.x;
  const TileRange tr = tile_range(wg, nwg, p.ntiles);
  const unsigned list_base = tr.lo * TILE_ELEMS;     // this workgroup's list lives in its tiles' slots
  const size_t first_el = (size_t)tr.lo * TILE_ELEMS;
  const size_t end_el = min((size_t)p.nfull * 64, (size_t)tr.hi * TILE_ELEMS);
  const int range_el = tr.lo < tr.hi ? (int)(end_el - first_el) : 0;                  // whole blocks only
  // input: the workgroup's own range of the flat / block-after-block layout -- or, for multi-dimensional blocks read in
  // place (p.nd.on), the whole array
  const bool nd_direct = (GEOM != GEOM_1D) && p.nd.on != 0u;
  const __amdgpu_buffer_rsrc_t r_in = nd_direct
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.x), 0, (int)p.nd.bytes, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.x + first_el), 0, range_el * (int)sizeof(T), 0x00020000);
  // The same in pieces (flat geometry): 16 (fp64) / 8 (fp32) DMA instructions issued back to back stall the wave for
  // ~270 cycles each behind the CU's full memory queue (4.3 K cycles per phase, stamped); spread over the work that
  // follows they find the queue drained.  SPREAD: build knob DCTZ_DMA_SPREAD.
  static_assert(!SC || (PH == 2 && GEOM == GEOM_1D), "the scaled write-back exists for the flat two-phase kernels");
  constexpr bool SPREAD = DCTZ_DMA_SPREAD != 0 && PH == 2 && GEOM == GEOM_1D && !SC;
  auto issue_rows = [&](unsigned rel, int phase, const TileMap<T, PH>& tmx, auto jg0, auto jg1) {
    issue_phase_dma<T, PH, decltype(jg0)::value, decltype(jg1)::value>(r_in, rel, phase, tilebuf, tmx);
  };
  auto issue_dma = [&](unsigned rel, int phase, const TileMap<T, PH>& tmx) {
    if (GEOM != GEOM_1D && nd_direct) {
      int l = lane;
      asm volatile("" : "+v"(l));                      // (re-derived per issue, like tile_map)
      issue_phase_dma_nd<T, PH>(r_in, p.nd, tr.lo + rel, phase, tilebuf, l);
    } else {
      issue_phase_dma<T, PH>(r_in, rel, phase, tilebuf, tmx);
    }
  };
  const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(p.bin + first_el, 0, range_el, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(p.dc + first_el / 64, 0, range_el / 64 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_qc = __builtin_amdgcn_make_buffer_rsrc(p.qcnt + first_el / 64, 0, range_el / 64 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc(SC ? p.scaled + first_el : nullptr, 0, SC ? range_el * (int)sizeof(T) : 0, 0x00020000);
  // SC: one phase of a tile, scaled, registers -> image -> HBM (the rows of k_decompress's store_rows); ends with the image
  // read out, i.e. free for the next DMA
  auto put_scaled = [&](const T (&v)[64], auto phase, unsigned rel_s, const TileMap<T, PH>& tmx) {
    constexpr int PHASE = decltype(phase)::value;
    write_phase<T, PH, PHASE>(v, tilebuf, tmx);
    const int vbase = (int)(rel_s * (unsigned)G::TILEB);
#pragma unroll
    for (int jg = 0; jg < 8; jg++) {
      const int vo = vbase + jg * 8 * G::BLKB + tmx.g_of(jg);
#pragma unroll
      for (int sg = 0; sg < G::SEGP; sg++) {
        const u32x4 r = *reinterpret_cast<const u32x4*>(tilebuf + (jg * G::SEGP + sg) * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(r, r_sc, vo + (PHASE * G::SEGP + sg) * 128, 0, 2 /* nt */);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // the workgroup's list(s) behind descriptors too: 32-bit offsets, no 64-bit pointers to keep alive (or spill)
  const int list_slots = (int)((tr.hi - tr.lo) * (unsigned)TILE_ELEMS);
  const __amdgpu_buffer_rsrc_t r_list = (MODE == DCTZHIP_EC)
      ? __builtin_amdgcn_make_buffer_rsrc(p.ac_tmp + list_base, 0, list_slots * 4, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(p.qt_item + list_base, 0, list_slots * (int)sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_listj = __builtin_amdgcn_make_buffer_rsrc(p.qt_j + (MODE == DCTZHIP_QT ? list_base : 0u), 0, MODE == DCTZHIP_QT ? list_slots : 0, 0x00020000);
  // the scaling factor: the host's, or (speculative call) the one k_stats_final_sf chose on the device
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
  unsigned run = 0;                                  // length of the workgroup's list so far (uniform)
  bool in_order = true;                              // every tile so far has items in its first sub-list only (see LIST_IN_ORDER)

  // bin ids, DC and per-block counts of a tile on their way out (those of the previous tile)
  bool pend = false;
  unsigned p_rel = 0, p_qc = 0;
  unsigned pw[16];
  float p_dc = 0.f;

  auto flush = [&]() {
    // bin ids: 64 bytes per lane -> (through the staging buffer, free between two tiles' sub-lists) 1 KiB rows of 16
    // consecutive blocks
    int lo = lane;
    asm volatile("" : "+v"(lo));                     // (re-derived per trip, see tile_map below)
    const int f2 = (lo >> 1) & 3;
#pragma unroll
    for (int i = 0; i < 4; i++)
      lds_store_b128(exc_at + (unsigned)((lo * 4 + (i ^ f2)) * 16), u32x4{pw[4 * i], pw[4 * i + 1], pw[4 * i + 2], pw[4 * i + 3]});
    const int bin_goff = (lo >> 2) * 64 + (((lo & 3) ^ ((lo >> 3) & 3)) * 16);
    const int voff = (int)(p_rel * (unsigned)TILE_ELEMS) + bin_goff;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(excbuf + i * 1024 + lo * 16);
#ifndef DCTZ_DBG_NO_BINS                              /* (timing experiment only: what the bin_index stream costs the kernel) */
      __builtin_amdgcn_raw_buffer_store_b128(v, r_bin, voff + i * 1024, 0, DCTZ_BIN_STORE_AUX);
#else
      asm volatile("" :: "v"(v));
#endif
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p_dc), r_dc, (int)(p_rel * 64u + (unsigned)lo) * 4, 0, 0);   // :350-351 USE_TRUNCATE
    __builtin_amdgcn_raw_buffer_store_b32(p_qc, r_qc, (int)(p_rel * 64u + (unsigned)lo) * 4, 0, 0);
  };

  // one phase of the block in registers: calc_data_stat's max|x| / min|x| over the raw values (util.c:18-25), then the
  // scaling (dctz-comp-lib.c:197-199 / :212-214)
  auto stats_scale = [&](T (&x)[64], auto phase, bool active, bool first) {
    constexpr int J0 = decltype(phase)::value * (64 / PH), J1 = J0 + 64 / PH;
    if (STATS) {
      if (active) {
#pragma unroll
        for (int j = J0; j < J1; j++) acc.minmax(x[j]);
      }
      // util.c:22 starts at i = 1: x[0] never enters the sum; here the sum is 8 sf * (sum of the DCs), so x[0] leaves it as
      // x[0] / (8 sf) DC units (tree-order sum either way: only the decimal digits of the header's `mean` come from it)
      if (J0 == 0 && first && lane == 0) acc.dcs -= (double)x[0] / (scale ? 8.0 * (double)sf : 8.0);
    }
    if (scale) {
      if (fast_sf == 2) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int j = J0; j < J1; j += 2) {
            const f32x2 v = fastdiv_core2(sfd, f32x2{(float)x[j], (float)x[j + 1]});
            x[j] = v.x; x[j + 1] = v.y;
          }
        } else {
#pragma unroll
          for (int j = J0; j < J1; j++) x[j] = sfd.core(x[j]);
        }
      } else if (fast_sf == 1) {
#pragma unroll
        for (int j = J0; j < J1; j++) x[j] = sfd.div(x[j]);
      } else {
#pragma unroll
        for (int j = J0; j < J1; j++) x[j] = x[j] / sfd.d;
      }
    }
  };

  // the lane's addresses inside the tile image are re-derived every trip from a value the compiler cannot see through:
  // kept alive across the loop they are spilled to scratch, and a scratch reload waits behind the DMA in flight
  auto tile_map = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    TileMap<T, PH> tm;
    tm.init(l);
    return tm;
  };

  // PH = 2: the first half of the NEXT tile is taken out of LDS in the middle of the binning of this one (the registers
  // of the coefficients already binned are free by then), so that the DMA of its second half can start early and
  // land under the rest of the tile; only what is left of that latency is exposed at the top of the loop
  T xn[64];
  if (tr.lo < tr.hi) {
    const TileMap<T, PH> tm0 = tile_map();
    issue_dma(0u, 0, tm0);
    if (PH == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      read_phase<T, PH, 0>(xn, tilebuf, tm0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (SC) {
        stats_scale(xn, std::integral_constant<int, 0>{}, (unsigned)lane < min((unsigned)TILE_BLKS, p.nfull - tr.lo * TILE_BLKS), tr.lo == 0);
        put_scaled(xn, IC<0>{}, 0u, tm0);
        issue_dma(0u, 1, tm0);
      } else {
        issue_dma(0u, 1, tm0);
        stats_scale(xn, std::integral_constant<int, 0>{}, (unsigned)lane < min((unsigned)TILE_BLKS, p.nfull - tr.lo * TILE_BLKS), tr.lo == 0);
      }
    }
  }
  for (unsigned tile = tr.lo; tile < tr.hi; tile++) {
    const unsigned rel = tile - tr.lo;
    const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile * TILE_BLKS);
    const bool active = (unsigned)lane < blks_here;
    const TileMap<T, PH> tm = tile_map();
    const bool more = tile + 1 < tr.hi;
    // (pieces are issued without a branch -- a branch inside the transform splits its scheduling regions --: behind the
    // last tile they ask for addresses beyond the descriptor's range, which move no data)
    TileMap<T, PH> tmn = tm;
    if (SPREAD && !more) tmn.g_even = tmn.g_odd = 0x7FFF0000;
    const unsigned reln = (SPREAD && !more) ? 0u : rel + 1;
    T x[64];
    if (PH == 2) {
#pragma unroll
      for (int j = 0; j < 32; j++) x[j] = xn[j];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the second half has landed (and everything older is done)
      read_phase<T, PH, PH - 1>(x, tilebuf, tm);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and is in registers: the buffer is free
      if constexpr (SPREAD) {
        issue_rows(reln, 0, tmn, IC<0>{}, IC<1>{});
        if (pend) flush();
        issue_rows(reln, 0, tmn, IC<1>{}, IC<2>{});
        stats_scale(x, std::integral_constant<int, PH - 1>{}, active, false);
        issue_rows(reln, 0, tmn, IC<2>{}, IC<3>{});
      } else if constexpr (SC) {
        stats_scale(x, std::integral_constant<int, PH - 1>{}, active, false);
        put_scaled(x, IC<PH - 1>{}, rel, tm);
        if (more) issue_dma(rel + 1, 0, tm);
        if (pend) flush();
      } else {
        if (more) issue_dma(rel + 1, 0, tm);
        if (pend) flush();
        stats_scale(x, std::integral_constant<int, PH - 1>{}, active, false);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA has landed (and everything older is done)
      read_phase<T, PH, 0>(x, tilebuf, tm);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (tile + 1 < tr.hi) issue_dma(rel + 1, 0, tm);
      if (pend) flush();
      stats_scale(x, std::integral_constant<int, 0>{}, active, tile == 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPREAD) {
      auto hook = [&](auto idx) {                      // rows 3 .. 7 at the transform's first five fences
        constexpr int I = decltype(idx)::value;
        if constexpr (I < 5) issue_rows(reln, 0, tmn, IC<3 + I>{}, IC<4 + I>{});
      };
      block_fwd<T, CTab<T>, GEOM, (PH > 1), decltype(hook)>(x, tab, hook);
    } else {
      block_fwd<T, CTab<T>, GEOM, (PH > 1)>(x, tab);
    }
    if (p.coef != nullptr && active) {               // test tap: the coefficients as computed
#pragma unroll
      for (int j = 0; j < 64; j++) p.coef[((size_t)tile * TILE_BLKS + lane) * 64 + j] = x[j];
    }
    if (active) {
      if (STATS) acc.dcs += (double)x[0];            // orthonormal 64-point DCT: DC = (sum of the block) / 8
      if (p.last_is_full && tile * TILE_BLKS + lane == p.nfull - 1) p.ctl->q0 = (unsigned long long)to_bits(x[0]);   // :355-360
    }

    if (S::PACKED) lds_store_b64(qmax_at + (unsigned)lane * 8u, u32x2{0u, 0u});     // this tile's maxima (the flush has been through the buffer)
    unsigned w[16];
    unsigned qc = 0;                                 // this block's counts, one field per sub-list
    unsigned ttot = 0;                               // "stored exactly" coefficients of the tile (uniform)
    __builtin_amdgcn_sched_barrier(0);               // the binning phase is scheduled on its own (the transform before it peaks in registers)
    // One sub-list = the coefficients [Q QW, (Q + 1) QW) of every block of the tile:
    //   * pass-1 binning (:363-414), four coefficients = one dword of bin ids at a time, stage by stage, so that the four
    //     dependent chains (subtract, divide, floor, map, convert, pack) interleave;
    //   * the coefficients that are stored exactly (:478-544) keep their converted value and a flag bit; a prefix sum of
    //     the lanes' counts gives every block its place in the sub-list, the lanes write their items there (the others
    //     go to the lane's dump slot) and the sub-list leaves in whole rows of 64 items.
    auto sub_list = [&](auto qi, auto fast, auto safe) {
      constexpr int Q = decltype(qi)::value, J0 = Q * QW;
      unsigned m = 0;                                // bit i: coefficient J0 + i of this block is stored exactly
#pragma unroll
      for (int gg = 0; gg < QW / 4; gg++) {
        const int g = J0 / 4 + gg;
        float h[4];
        if constexpr (sizeof(T) == 4 && decltype(fast)::value) {
#pragma unroll
          for (int i = 0; i < 4; i += 2) {           // fp32: subtract and divide two coefficients per instruction
            const int j = 4 * g + i;
            const f32x2 q = fastdiv_core2(bwd, f32x2{(float)x[j], (float)x[j + 1]} - f32x2{(float)rmin, (float)rmin});   // :377 / :402
            h[i] = bin_value<T, decltype(safe)::value>(x[j], q.x, rmax);
            h[i + 1] = bin_value<T, decltype(safe)::value>(x[j + 1], q.y, rmax);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int j = 4 * g + i;
            const T u = x[j] - rmin;                 // :377 / :402
            const T q = decltype(fast)::value ? bwd.core(u) : u / bwd.d;
            h[i] = bin_value<T, decltype(safe)::value>(x[j], q, rmax);
          }
        }
        if (g == 0) h[0] = 0.0f;                     // j = 0 is the DC slot (:361): never stored exactly, its id is set below
        unsigned wgd = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) wgd = __builtin_amdgcn_cvt_pk_u8_f32(h[i], i, wgd);
        asm volatile("" : "+v"(wgd));                // packed HERE: left alone, the compiler sinks all 64 conversions below the
        w[g] = wgd;                                  // last group and keeps 64 fp64 bin values alive (128 registers) until then
        // "stored exactly" = id 255: bit 7 of byte i of mm.  A group is only looked at further when SOME lane of the wave
        // has such a coefficient in it (the high-frequency groups of a smooth field never do)
        const unsigned nw = ~wgd;
        const unsigned mm = ~(((nw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nw) & 0x80808080u;
        if (__builtin_amdgcn_ballot_w64(mm != 0u))
          m |= (((mm >> 7) | (mm >> 14) | (mm >> 21) | (mm >> 28)) & 0xFu) << (4 * gg);
        __builtin_amdgcn_sched_barrier(0);           // keep the groups apart: hoisting all 64 quotients first costs 128 registers
      }
      if (!active) m = 0;
      const unsigned n = (unsigned)__popc(m);
      qc |= n << (Q * S::CBITS);
      const unsigned incl = wave_incl_scan(n);
      const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      const unsigned base = incl - n;                // this block's place in the sub-list
      for (unsigned lo = 0; lo < tot; lo += (unsigned)S::CAP) {      // (one round unless most coefficients are stored exactly)
        unsigned pos = base - lo;                    // (wraps for blocks in front of the round's window: never < CAP then)
        // The items are taken from the coefficient registers here (:496-497 / :535-537 USE_TRUNCATE for EC; QT: full
        // precision), four positions at a time and only where some lane of the wave has one: a smooth field pays for
        // its lowest frequencies, not for the width of the sub-list.
        // (QT: the positions J0 + i come from ONE register the compiler cannot see through -- as constants per
        // sub-list they are hoisted out of the tile loop and cost 64 registers, i.e. 60 spilled ones)
        unsigned jv = (unsigned)J0;
        if (MODE == DCTZHIP_QT) asm volatile("" : "+v"(jv));
#pragma unroll
        for (int gg = 0; gg < QW / 4; gg++) {
          if (__builtin_amdgcn_ballot_w64(((m >> (4 * gg)) & 0xFu) != 0u)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int i = 4 * gg + k;
              const bool f = ((m >> i) & 1u) != 0u;
              const unsigned at = (f && pos < (unsigned)S::CAP) ? pos : (unsigned)(S::CAP + lane);
              lds_store_item(exc_at + at * (unsigned)sizeof(Item), (Item)x[J0 + i]);
              if (MODE == DCTZHIP_QT) lds_store_b8(exc_at + (unsigned)S::ITEM_BYTES + at, jv + (unsigned)i);
              pos += f ? 1u : 0u;
            }
          }
        }
        const unsigned cnt = min(tot - lo, (unsigned)S::CAP);
        for (unsigned r = 0; r * 64u < cnt; r++) {   // whole rows -> the workgroup's list; lanes beyond the end fall outside the descriptor
          const unsigned e = r * 64u + (unsigned)lane;
          const Item v = items[e];
          const bool in = e < cnt;
          const int at = (int)(run + lo + e);
          if (MODE == DCTZHIP_EC) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r_list, in ? at * 4 : 0x7FFFFFF0, 0, 0);
          } else {
            const unsigned char jj = jbuf[e];
            if constexpr (sizeof(T) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (double)v), r_list, in ? at * 8 : 0x7FFFFFF0, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r_list, in ? at * 4 : 0x7FFFFFF0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8(jj, r_listj, in ? at : 0x7FFFFFF0, 0, 0);
            if (QMAX_HERE && in) {
              const T a = fabs((T)v);
              if (a > rmax) lds_max_bits(qmax_at + (unsigned)jj * (unsigned)sizeof(QBits), to_bits(a));   // positive values order like their bits
            }
          }
        }
      }
      run += tot;
      ttot += tot;
      if (Q != 0 && tot != 0u) in_order = false;
    };
    auto sub = [&](auto qi) {
      if (bwd.ok) { if (p.fast_bw & 2u) sub_list(qi, std::true_type{}, std::false_type{}); else sub_list(qi, std::true_type{}, std::true_type{}); }
      else sub_list(qi, std::false_type{}, std::true_type{});
    };
    sub(std::integral_constant<int, 0>{});
    if constexpr (NQ >= 4) sub(std::integral_constant<int, 1>{});
    if constexpr (NQ >= 8) { sub(std::integral_constant<int, 2>{}); sub(std::integral_constant<int, 3>{}); }
    if (PH == 2 && tile + 1 < tr.hi) {               // the next tile's first half (see above)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      read_phase<T, PH, 0>(xn, tilebuf, tm);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (SPREAD) {
        issue_rows(rel + 1, 1, tm, IC<0>{}, IC<2>{});
        stats_scale(xn, std::integral_constant<int, 0>{}, (unsigned)lane < min((unsigned)TILE_BLKS, p.nfull - (tile + 1) * TILE_BLKS), false);
        issue_rows(rel + 1, 1, tm, IC<2>{}, IC<4>{});
      } else if constexpr (SC) {
        stats_scale(xn, std::integral_constant<int, 0>{}, (unsigned)lane < min((unsigned)TILE_BLKS, p.nfull - (tile + 1) * TILE_BLKS), false);
        put_scaled(xn, IC<0>{}, rel + 1, tm);
        issue_dma(rel + 1, 1, tm);
      } else {
        issue_dma(rel + 1, 1, tm);
        stats_scale(xn, std::integral_constant<int, 0>{}, (unsigned)lane < min((unsigned)TILE_BLKS, p.nfull - (tile + 1) * TILE_BLKS), false);
      }
    }
    sub(std::integral_constant<int, NQ / 2>{});
    if constexpr (SPREAD) issue_rows(reln, 1, tmn, IC<4>{}, IC<6>{});
    if constexpr (NQ >= 4) sub(std::integral_constant<int, NQ / 2 + 1>{});
    if constexpr (SPREAD) issue_rows(reln, 1, tmn, IC<6>{}, IC<8>{});
    if constexpr (NQ >= 8) { sub(std::integral_constant<int, 6>{}); sub(std::integral_constant<int, 7>{}); }
    w[0] |= 0xFFu;                                   // :361 DC slot
    if (S::PACKED) {
      const QBits m = *reinterpret_cast<const QBits*>(excbuf + S::QMAX_AT + lane * (int)sizeof(QBits));
      qreg = m > qreg ? m : qreg;
    }
    if (lane == 0) p.ttot[tile] = ttot;
    pend = true; p_rel = rel; p_qc = qc; p_dc = (float)x[0];
#pragma unroll
    for (int i = 0; i < 16; i++) pw[i] = w[i];
  }
  if (pend) flush();
  if (lane == 0) p.tile_cnt[wg] = run | (in_order ? LIST_IN_ORDER : 0u);
  if (QMAX_HERE) {
    const QBits m = S::PACKED ? qreg : qmax_lds[lane];
    if (m != 0) atomicMax(&p.ctl->qraw[lane], (unsigned long long)m);
  }
  if (STATS) {
    __syncthreads();
    acc.flush(p.stat_part, wg, reinterpret_cast<double*>(tilebuf), 1, scale ? 8.0 * (double)sf : 8.0);
  }
}

template <typename T, int MODE, bool STATS, int PH, int GEOM, bool SC = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(compress_waves<T, MODE, PH>()))) void k_compress(FwdParams<T> p) {
  compress_body<T, MODE, STATS, PH, GEOM, SC>(p, blockIdx.x, gridDim.x);
}

// The last, short block (length l = N % 64): the reference re-plans a length-l
// (l even) or 2l (l odd) FFT for it (dctz-comp-lib.c:326-336, dct.c:59-72).
// One wavefront, definition-order DFT with host-built roots.
template <typename T, int MODE>
__device__ __forceinline__ void compress_rem_body(const FwdParams<T>& p, const int l) {
  __shared__ T v[128];
  const int k = threadIdx.x;
  const size_t base = (size_t)p.nfull * 64;
  const T* rt = p.rtab;
  const int N = (l & 1) ? 2 * l : l;
  const T sf = p.guess ? (T)p.guess->sf : p.sf;
  const bool SCALE = (sf != T(1));                                 // dctz-comp-lib.c:193 / :208
  FastDiv<T> sfd, bwd;
  sfd.init(sf, (p.guess ? p.guess->fast_sf : p.fast_sf) != 0);
  bwd.init(p.bin_width, (p.fast_bw & 1u) != 0);
  if (p.stat_part != nullptr) {                                    // speculative launch: raw-input statistics of this block
    __shared__ double ss[3];
    StatAcc<T> acc;
    acc.init();
    if (k < l) acc.add(p.x[base + k], base + k != 0);
    acc.flush(p.stat_part, p.nlists_main, ss, 1);
  }
  if (k < l) {
    T a = p.x[base + k];
    if (SCALE) a = sfd.div(a);
    if (p.scaled != nullptr) p.scaled[base + k] = a;               // (dctz-comp-lib.c:193-216, when k_compress writes the scaled copy)
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
  // pass-1 binning, the reference's own form (:363-414)
  const bool out = fabs(coef) > p.range_max;                       // == (item < range_min || item > range_max)
  const T u = coef - p.range_min;
  const T q = bwd.ok ? bwd.core(u) : u / bwd.d;
  const int t = (int)q;                                            // (t_bin_id) cast: truncation
  unsigned b = out ? 255u : (unsigned)(t <= 127 ? 254 - 2 * t : 2 * t - 255);   // conv_tbl :27-43 (t == 255 -> 255)
  bool exc = false;
  if (k == 0) b = 255u; else exc = (b == 255u);
  if (k >= l) exc = false;
  const unsigned long long m = __ballot(exc);
  const unsigned rank = (unsigned)__popcll(m & ((1ull << k) - 1ull));
  // (k_compress_eo with single-pass placement, EC: there are no lists -- the block's coefficients go behind the running
  // tot_AC_exact_count the tiles have left in the control block)
  const bool direct = MODE == DCTZHIP_EC && p.direct != 0u;
  const unsigned start = direct ? p.ctl->cnt_total : p.ntiles * TILE_ELEMS;   // else: this block is list #nlists_main, parked behind the tiles' slots
  if (k < l) {
    p.bin[base + k] = (uint8_t)b;
    if (p.coef != nullptr) p.coef[base + k] = coef;
    if (k == 0) { p.dc[p.nfull] = (float)coef; p.ctl->q0 = (unsigned long long)to_bits(coef); }
    if (exc) {
      if (MODE == DCTZHIP_EC) { if (direct) p.ac[start + rank] = (float)coef; else p.ac_tmp[start + rank] = (float)coef; }
      else {
        p.qt_item[start + rank] = coef; p.qt_j[start + rank] = (uint8_t)k;
        if (fabs(coef) > p.range_max) atomicMax(&p.ctl->qraw[k], (unsigned long long)to_bits(fabs(coef)));   // :371-372 / :396-397
      }
    }
  }
  __syncthreads();
  if (k == 0) { if (direct) p.ctl->cnt_total = start + (unsigned)__popcll(m); else p.tile_cnt[p.nlists_main] = (unsigned)__popcll(m); }
}
template <typename T, int MODE>
__global__ __launch_bounds__(64) void k_compress_rem(FwdParams<T> p, int l) { compress_rem_body<T, MODE>(p, l); }

// list l < G belongs to workgroup l of k_compress (slots of its tile range); list G is the remainder block's
__device__ __forceinline__ size_t list_slot(unsigned l, unsigned G, unsigned ntiles) {
  return (size_t)(l < G ? tile_range(l, G, ntiles).lo : ntiles) * TILE_ELEMS;
}

// Move every workgroup-local list to its place in AC_exact[], the sub-lists of every tile back in the reference's order
// (dctz-comp-lib.c:478-544: block after block, j ascending -- k_compress leaves a tile as NQ sub-lists, each block-major
// over a range of j; the remainder block's list is in order already).
// QT: clamp the table (:450-461) and normalise on the way (:488-518).
// The place of list l -- the running tot_AC_exact_count of :478-544 in front of it -- is the sum of the lengths of the
// lists before it: at most 2049 of them, summed by the workgroup itself (no scan kernel); the workgroup of the last
// list leaves the total.
// Grid: (list, chunk of COMPACT_TPW tiles of the list); a WAVE takes a tile, so that all tiles of the array are in flight
// at once (the work of a tile is a chain of dependent round trips).  Lane b reads block b's counts; prefix sums over
// the lanes give every (block, sub-list) run its place in the tile's piece of the list and every block its first place
// in the output.  Then the lanes walk the OUTPUT positions (coalesced stores): the owning block of a position comes
// from an owner map in LDS (every block marks its first position, a running maximum fills the gaps), the sub-list from
// the block's counts, the item from its run.
constexpr int COMPACT_TPW = SWG / 64;                // tiles per workgroup of k_compact_ac: one per wave
__device__ __forceinline__ unsigned wave_incl_max_scan(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));    // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));    // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));    // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));    // row_shr:8
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;
}
template <typename T, int MODE>
__device__ __forceinline__ void compact_ac_body(const FwdParams<T>& p, const double eb, const unsigned nlists, const unsigned l, const unsigned chunk,
                                                unsigned* sh) {
  using Bits = typename Traits<T>::Bits;
  using S = Sub<T, MODE>;
  constexpr int NQ = S::NQ, CB = S::CBITS, FB = S::FB, FPD = S::FPD, NPK = S::NPK;
  constexpr unsigned CMASK = (1u << CB) - 1u, FMASK = (1u << FB) - 1u;
  static_assert(NQ == 2 || NQ == 4 || NQ == 8, "table rows are laid out for 2, 4 or 8 sub-lists");
  constexpr int ROWW = NQ <= 4 ? 4 : 8;              // dwords of a block's table row (below)
  constexpr unsigned HALF = TILE_ELEMS / 2;          // output positions the owner map covers at a time
  __shared__ T qtab[64];
  // per wave, per block of its tile: [0] first output position, [1 ..] the ends of its sub-list runs inside the block
  // (a byte each), then per sub-list 16 bits: (where the run starts in the tile's piece of the list) - (where it starts
  // in the block) + 64 -- so that the item for the r-th position of the block is found with one table row
  __shared__ __attribute__((aligned(16))) unsigned tab[COMPACT_TPW][64][ROWW];
  __shared__ __attribute__((aligned(16))) unsigned char own[COMPACT_TPW][HALF];   // owner map: block that owns an output position
  if (MODE == DCTZHIP_QT) {
    if (threadIdx.x < 64) {
      T v = Traits<T>::from_bits((Bits)p.ctl->qraw[threadIdx.x]);
      if (v < T(1)) v = T(1);
      qtab[threadIdx.x] = v;
    }
    __syncthreads();
  }
  const unsigned G = p.nlists_main;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // The in-range else-branch of :502-506 cannot fire for finite data and stores nothing; every flagged coefficient is
  // appended.
  auto fetch = [&](size_t at) -> float {
    if (MODE == DCTZHIP_EC) return p.ac_tmp[at];
    return (float)qt_normalise(p.qt_item[at], qtab[p.qt_j[at]], eb, T(10), p.range_min, p.range_max);
  };
  // (the tile's own words are asked for first: they do not depend on the list's place, and a workgroup's life is a chain
  // of round trips -- list lengths, tile totals, block counts, items)
  const TileRange tr = tile_range(l < G ? l : 0u, G ? G : 1u, p.ntiles);
  // A chunk with fewer tiles than waves -- small arrays: a list is one or two tiles -- shares every tile among `share`
  // waves: each builds the tile's tables for itself (they are per wave anyway) and takes every share-th group of rows.
  // A dense single-tile list was one wave walking 63 rows (C1: k_compact_ac 23 us of a 62 us step).
  const unsigned first_t = tr.lo + chunk * (unsigned)COMPACT_TPW;
  const unsigned m_here = (l < G && first_t < tr.hi) ? min((unsigned)COMPACT_TPW, tr.hi - first_t) : 0u;
  const unsigned share = m_here == 1u ? 4u : (m_here == 2u ? 2u : 1u), part = wave % share;
  static_assert(COMPACT_TPW == 4, "the sharing above is written for four waves");
  const unsigned t = first_t + wave / share;
  const bool tile_here = wave / share < m_here;
  unsigned pre = 0, c = 0;                           // items of this list in front of tile t; block `lane`'s counts
  if (tile_here) {
    for (unsigned u = tr.lo + lane; u < t; u += 64u) pre += p.ttot[u];
    const unsigned blk = t * (unsigned)TILE_BLKS + lane;
    c = blk < p.nfull ? p.qcnt[blk] : 0u;
  }
  const unsigned nraw = p.tile_cnt[l], n = nraw & LIST_LEN;
  if ((l >= G || (nraw & LIST_IN_ORDER)) && chunk != 0) return;     // such a list is copied by its first workgroup alone
  unsigned before = 0;
  for (unsigned i = threadIdx.x; i < l; i += SWG) before += p.tile_cnt[i] & LIST_LEN;
  const unsigned dst = block_sum(before, sh);
  if (l == nlists - 1 && chunk == 0 && threadIdx.x == 0) p.ctl->cnt_total = dst + n;
  const size_t src = list_slot(l, G, p.ntiles);
  if (l >= G || (nraw & LIST_IN_ORDER)) {            // the remainder block's list, or a list that is in order as it is
    if (chunk == 0) {
      for (unsigned i0 = threadIdx.x; i0 < n; i0 += 4 * SWG) {          // (four independent items in flight per thread)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const unsigned i = i0 + (unsigned)u * SWG; v[u] = i < n ? fetch(src + i) : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const unsigned i = i0 + (unsigned)u * SWG; if (i < n) p.ac[dst + i] = v[u]; }
      }
    }
    return;
  }
  if (!tile_here) return;
  pre = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(pre), 63);
  if (!__builtin_amdgcn_ballot_w64((c >> CB) != 0u)) {
    // only the first sub-list of the tile has items (a smooth field: what is stored exactly are the lowest
    // frequencies): block-major over its range of j IS the reference's order -- the tile's piece is copied as it is
    const unsigned tt0 = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(c & CMASK), 63);
    for (unsigned o0 = part * 256u; o0 < tt0; o0 += 256u * share) {
      float v[4];
#pragma unroll
      for (unsigned u = 0; u < 4; u++) { const unsigned o = o0 + 64u * u + lane; v[u] = o < tt0 ? fetch(src + pre + o) : 0.f; }
#pragma unroll
      for (unsigned u = 0; u < 4; u++) { const unsigned o = o0 + 64u * u + lane; if (o < tt0) p.ac[dst + pre + o] = v[u]; }
    }
    return;
  }
  // prefix sums over the blocks, FPD sub-lists per dword
  unsigned ex[NPK], tot[NPK];
#pragma unroll
  for (int d = 0; d < NPK; d++) {
    unsigned v = 0;
#pragma unroll
    for (int f = 0; f < FPD; f++) if (d * FPD + f < NQ) v |= ((c >> ((d * FPD + f) * CB)) & CMASK) << (f * FB);
    const unsigned incl = wave_incl_scan(v);
    ex[d] = incl - v;
    tot[d] = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
  }
  unsigned rowbase = 0, cum = 0, acc_q = 0, row[ROWW];
#pragma unroll
  for (int i = 0; i < ROWW; i++) row[i] = 0;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const unsigned qbq = (ex[q / FPD] >> ((q % FPD) * FB)) & FMASK;            // start of the block's run in sub-list q
    const unsigned nq = (c >> (q * CB)) & CMASK;
    rowbase += qbq;
    const unsigned del = acc_q + qbq + 64u - cum;                               // acc_q: where sub-list q starts in the tile's piece (uniform)
    cum += nq;
    row[1 + q / 4] |= cum << (8 * (q % 4));
    row[(NQ <= 4 ? 2 : 4) + q / 2] |= del << (16 * (q % 2));
    acc_q += (tot[q / FPD] >> ((q % FPD) * FB)) & FMASK;
  }
  row[0] = rowbase;
  if (NQ == 2) row[1] |= 0x7F7F0000u;                // (bytes of sub-lists that do not exist: ends no position reaches)
  const unsigned nb = cum, tt = acc_q;               // the block's / the tile's items (tt == ttot[t])
  if (tt == 0) return;
  // (one wave writes and reads its own tables: LDS operations of a wave are in order)
#pragma unroll
  for (int i = 0; i < ROWW; i += 4) *reinterpret_cast<u32x4*>(&tab[wave][lane][i]) = u32x4{row[i], row[i + 1], row[i + 2], row[i + 3]};
  unsigned carry = 0;                                // owner of the last position handled so far
  unsigned grp = 0;                                  // row groups seen so far (this wave takes those with grp % share == part)
  for (unsigned h0 = 0; h0 < tt; h0 += HALF) {
    const unsigned hn = min(tt - h0, HALF);
    for (unsigned o = lane * 16u; o < hn; o += 1024u) *reinterpret_cast<u32x4*>(&own[wave][o]) = u32x4{0u, 0u, 0u, 0u};
    if (nb != 0 && rowbase - h0 < HALF) own[wave][rowbase - h0] = (unsigned char)lane;       // (unsigned: rowbase >= h0 too)
    // where the item for output position o (of this half) sits in the tile's piece of the list
    auto owner_of = [&](unsigned o) -> unsigned {      // (every wave of a shared tile walks ALL rows here: the running maximum is a chain)
      unsigned b = wave_incl_max_scan((unsigned)own[wave][o - h0]);
      b = max(b, carry) & 63u;                       // (& 63: lanes beyond the tile's last position read bytes nobody wrote)
      carry = (unsigned)__builtin_amdgcn_readlane((int)b, 63);
      return b;
    };
    auto source_of = [&](unsigned o, unsigned b) -> unsigned {
      const u32x4 t0 = *reinterpret_cast<const u32x4*>(&tab[wave][b][0]);
      const unsigned r = o - t0.x;                   // position inside the block (< 64 for a real position)
      const unsigned R = (r & 63u) * 0x01010101u;
      unsigned q = (unsigned)__popc(((R | 0x80808080u) - t0.y) & 0x80808080u);   // sub-lists of the block that end at or before r
      unsigned dw;
      if constexpr (NQ <= 4) {
        dw = q < 2u ? t0.z : t0.w;
      } else {
        q += (unsigned)__popc(((R | 0x80808080u) - t0.z) & 0x80808080u);
        const u32x4 t1 = *reinterpret_cast<const u32x4*>(&tab[wave][b][4]);
        const unsigned lo2 = (q & 2u) ? t1.y : t1.x, hi2 = (q & 2u) ? t1.w : t1.z;
        dw = (q & 4u) ? hi2 : lo2;
      }
      return r + ((dw >> ((q & 1u) * 16u)) & 0xFFFFu) - 64u;
    };
    // four rows of 64 positions at a time: their items are fetched side by side, then stored (one row at a time the loop
    // is a chain of load -> store -> load ...: the compiler cannot tell the list from AC_exact)
    constexpr unsigned RU = 4;
    for (unsigned o0 = h0; o0 < h0 + hn; o0 += 64u * RU, grp++) {
      unsigned at[RU];
      float v[RU];
#pragma unroll
      for (unsigned u = 0; u < RU; u++) {
        const unsigned o = o0 + 64u * u + lane;
        at[u] = (o0 + 64u * u < h0 + hn) ? owner_of(min(o, h0 + HALF - 1u)) : 0u;    // (whole rows: the max-scan wants every lane)
      }
      if (grp % share != part) continue;
#pragma unroll
      for (unsigned u = 0; u < RU; u++) {
        const unsigned o = o0 + 64u * u + lane;
        at[u] = (o0 + 64u * u < h0 + hn) ? source_of(min(o, h0 + HALF - 1u), at[u]) : 0u;
      }
#pragma unroll
      for (unsigned u = 0; u < RU; u++) {
        const unsigned o = o0 + 64u * u + lane;
        v[u] = (o < h0 + hn) ? fetch(src + pre + at[u]) : 0.f;
      }
#pragma unroll
      for (unsigned u = 0; u < RU; u++) {
        const unsigned o = o0 + 64u * u + lane;
        if (o < h0 + hn) p.ac[dst + pre + o] = v[u];
      }
    }
  }
}
template <typename T, int MODE>
__global__ __launch_bounds__(SWG) void k_compact_ac(FwdParams<T> p, double eb, unsigned nlists, FinArgs fin) {
  __shared__ unsigned sh[SWG / 64];
  if (fin.box != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
    unsigned all = 0;
    for (unsigned i = threadIdx.x; i < nlists; i += SWG) all += p.tile_cnt[i] & LIST_LEN;
    FinBody f;
    f.ctl = fin.ctl; f.part = fin.part; f.nparts = fin.nparts; f.box = fin.box; f.seq = fin.seq; f.guess = fin.guess;
    f.cnt_known = true; f.cnt_total = block_sum(all, sh);                // tot_AC_exact_count (:478-544)
    f.err_known = true; f.error = 0;
    finish_body<true>(f);
  }
  compact_ac_body<T, MODE>(p, eb, nlists, blockIdx.x, blockIdx.y, sh);
}

// =============================================================== decompress ==
// Decode side, step 1: per-TILE count of "stored exactly" flags (bin id 255 at j != 0,
// dctz-decomp-lib.c:400 / :446), 1 byte per element read, and the sum of the counts over the tile range of
// every workgroup of k_decompress (same partition: workgroup b here <-> workgroup b there), so that
// k_decompress finds the start of its piece of AC_exact by adding up at most a thousand words -- no scan kernel.
// A wave takes whole tiles (4 x 1 KiB coalesced rows, plain loads: k_decompress re-reads these lines from the
// Infinity Cache); no workgroup barrier per tile.
#ifndef DCTZ_COUNT_TF
#define DCTZ_COUNT_TF 2
#endif
__device__ __forceinline__ void count_tiles_body(const uint8_t* __restrict__ bin, unsigned nfull, unsigned ntiles, unsigned nwg,
                                                 unsigned* __restrict__ tile_cnt, unsigned* __restrict__ wg_cnt, const unsigned wg,
                                                 unsigned* __restrict__ tile_pre = nullptr) {
  __shared__ unsigned part[SWG / 64];
  constexpr unsigned PRE_MAX = 256;                                    // tiles of a range whose prefix is kept in LDS (tile_pre; the host checks the range)
  __shared__ unsigned tcs[PRE_MAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const TileRange tr = tile_range(wg, nwg, ntiles);
  const size_t end = (size_t)nfull * 64;
  unsigned acc = 0;                                                    // this wave's share of the workgroup's count (uniform)
  // (TF tiles of a wave in flight at once, 4 TF 16-byte loads per lane: the loop is a chain of round trips otherwise)
  constexpr unsigned NW = SWG / 64;
  constexpr int TF = DCTZ_COUNT_TF;
  for (unsigned tile0 = tr.lo + (unsigned)wave; tile0 < tr.hi; tile0 += TF * NW) {
    uint4 wv[TF][4];
#pragma unroll
    for (int h = 0; h < TF; h++) {
      const unsigned tile = tile0 + (unsigned)h * NW;
      const size_t o = (size_t)tile * TILE_ELEMS + (size_t)lane * 16;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        wv[h][i] = make_uint4(0u, 0u, 0u, 0u);                           // (a zero word has no 255 byte)
        if (tile < tr.hi && o + (size_t)i * 1024 < end) wv[h][i] = *reinterpret_cast<const uint4*>(bin + o + (size_t)i * 1024);
      }
    }
#pragma unroll
    for (int h = 0; h < TF; h++) {
      const unsigned tile = tile0 + (unsigned)h * NW;
      if (tile >= tr.hi) break;
      unsigned c = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned w[4] = {wv[h][i].x, wv[h][i].y, wv[h][i].z, wv[h][i].w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const unsigned v = ~w[k];                                      // a zero byte of v <=> bin id 255
          const unsigned z = ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v;      // bit 7 of a byte set <=> that byte of v is non-zero
          unsigned m = ~z & 0x80808080u;
          if (k == 0 && (lane & 3) == 0) m &= ~0x80u;                    // byte 0 of every 64: j = 0, the DC slot (:392 / :438)
          c += (unsigned)__popc(m);
        }
      }
      const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(c), 63);
      if (lane == 0) { tile_cnt[tile] = tot; if (tile_pre != nullptr && tile - tr.lo < PRE_MAX) tcs[tile - tr.lo] = tot; }
      acc += tot;
    }
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned sum = 0;
#pragma unroll
    for (int w = 0; w < SWG / 64; w++) sum += part[w];
    wg_cnt[wg] = sum;
  }
  // tile_pre[t]: the counts of the tiles of THIS range in front of t (k_decompress with tile-interleaved workgroups adds the
  // ranges in front: the running pos of dctz-decomp-lib.c:402-412 at any tile without a scan kernel)
  if (tile_pre != nullptr && wave == 0) {
    const unsigned nt = min(tr.hi - tr.lo, PRE_MAX);
    unsigned run = 0;
    for (unsigned i0 = 0; i0 < nt; i0 += 64u) {
      const unsigned i = i0 + (unsigned)lane;
      const unsigned v = i < nt ? tcs[i] : 0u;
      const unsigned incl = wave_incl_scan(v);
      if (i < nt) tile_pre[tr.lo + i] = run + incl - v;
      run += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
}
#if DCTZ_PART == 0
// (QT: the call's quantisation table rides in this kernel's arguments and is put where k_decompress -- the next kernel of
// the stream -- reads it: a 512-byte copy of its own in front of the call was a dispatch of 4 us)
__global__ __launch_bounds__(SWG) void k_count_tiles(const uint8_t* __restrict__ bin, unsigned nfull, unsigned ntiles, unsigned nwg,
                                                     unsigned* __restrict__ tile_cnt, unsigned* __restrict__ wg_cnt, QtabArg qt,
                                                     unsigned qwords, unsigned long long* __restrict__ qdst, unsigned* __restrict__ tile_pre) {
  if (blockIdx.x == 0 && threadIdx.x < qwords) qdst[threadIdx.x] = qt.w[threadIdx.x];
  count_tiles_body(bin, nfull, ntiles, nwg, tile_cnt, wg_cnt, blockIdx.x, tile_pre);
}
#endif

// Fused: gen_bins (binning.c:12-50) + de-quantise (dctz-decomp-lib.c:389-417 / :438-463) -> DCT-III per block
// (:428, dct.c:115-205) -> de-scale (:494-511).  Lane b rebuilds block b of the tile in registers; the tile's
// exact coefficients AC_exact[S, S + total) (S from the prefix over the per-tile counts) are staged in LDS by
// LDS-DMA one tile ahead, like the bin ids / DC (plain loads into registers).
template <typename T>
size_t decompress_lds_bytes() { return (size_t)Geo<T, Phases<T>::D>::PHB + 256 * sizeof(T) + DecStage<T>::CAP * 4 + 64 * sizeof(T); }

#ifndef DCTZ_WPED32
#define DCTZ_WPED32 0
#endif
// (body shared by k_decompress and k_decompress_batch, like compress_body; `handoff` runs where the single-array kernel
// hands the call's result to the host)
#ifndef DCTZ_BC_ARITH
#define DCTZ_BC_ARITH 1
#endif
#ifndef DCTZ_DEC_STORE_AUX
#define DCTZ_DEC_STORE_AUX 2     /* cache policy of k_decompress's row stores: 2 = nt */
#endif
// (bin_centre: dctz_kernel_common.h)
template <typename T, int MODE, int PH, int GEOM, typename Handoff>
__device__ __forceinline__ void decompress_body(const InvParams<T>& p, const unsigned wg, const unsigned nwg, Handoff&& handoff) {
  using G = Geo<T, PH>;
  // ONE array: the output image (a phase of the tile) and, behind it, the buffer the next tile's exact coefficients are
  // staged in; a dense tile's coefficients (up to 4032 floats) are brought into the front of the whole array when the
  // tile's turn comes (fp64: half of the image; fp32: the half-tile image and the staging buffer together)
  constexpr int DEC_CAP = DecStage<T>::CAP;
  __shared__ __attribute__((aligned(1024))) unsigned char io[G::PHB + DEC_CAP * 4];
  unsigned char* const outbuf = io;
  float* const excbuf = reinterpret_cast<float*>(io + G::PHB);
  static_assert(G::PHB + DEC_CAP * 4 >= TILE_ELEMS * 4, "a dense tile's coefficients fit the array");
  // fp64 (one wave per SIMD, every LDS round trip exposed): computed, 0.233 -> 0.230 ms at p = 5 %, 0.267 -> 0.262 at
  // p = 17 %; fp32 (VALU-bound, two waves per SIMD hide the reads): the table, computing measured 6 % slower
  constexpr bool BC_ARITH = DCTZ_BC_ARITH != 0 && sizeof(T) == 8;
  __shared__ __attribute__((aligned(16))) T bctab[BC_ARITH ? 1 : 256]; // bin_center[] of gen_bins
  const int lane = threadIdx.x;
  const TileRange tr = tile_range(wg, nwg, p.ntiles);
  const size_t first_el = (size_t)tr.lo * TILE_ELEMS;
  const size_t end_el = min((size_t)p.nfull * 64, (size_t)tr.hi * TILE_ELEMS);
  const int range_el = tr.lo < tr.hi ? (int)(end_el - first_el) : 0;
  const bool nd_direct = (GEOM != GEOM_1D) && p.nd.on != 0u;           // multi-dimensional blocks written in place: the whole array
  const __amdgpu_buffer_rsrc_t r_out = nd_direct
      ? __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.nd.bytes, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(p.out + first_el, 0, range_el * (int)sizeof(T), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bin + first_el), 0, range_el, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dc + first_el / 64), 0, range_el / 64 * 4, 0x00020000);
  // AC_exact behind a descriptor based at this workgroup's first exact coefficient (32-bit offsets stay small for any N):
  // the running pos of dctz-decomp-lib.c:402-412 at the workgroup's first tile = the counts of all workgroups before it
  unsigned before = 0;
  for (unsigned i = (unsigned)lane; i < wg; i += WG) before += p.wg_cnt[i];
  const unsigned S_wg = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(before), 63);
  handoff();
  const size_t ac_left = S_wg < p.ac_count ? (size_t)(p.ac_count - S_wg) * 4 : 0;
  const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ac + (ac_left ? S_wg : 0u)), 0, (int)min(ac_left, (size_t)0x7ffffffc), 0x00020000);
  TileMap<T, PH> tm;
  tm.init(lane);
  const CTab<T> tab = as_ctab<T>(p.tab);
  // gen_bins / gen_bins_f (binning.c:17-23 / :37-43): bin_center[b] = (b odd ? b/2 + 1 : -(b/2)) * bin_width
  if (!BC_ARITH)
    for (int b = lane; b < 256; b += WG) {
      const int ti = (b & 1) ? (b >> 1) + 1 : -(b >> 1);
      bctab[b] = (T)ti * p.bin_width;
    }
  QtLanes<T> qtl{};
  if (MODE == DCTZHIP_QT) qtl.load(p.qtab, lane);
  const bool scale = (p.sf != T(1));                 // dctz-decomp-lib.c:496 / :505
  bool underrun = false;

  // inputs of a tile, fetched one tile ahead: 64 bin ids + DC of the lane's block into registers, the tile's exact
  // coefficients into LDS
  u32x4 bw[4];
  float dcv = 0.f;
  unsigned S = 0, total = 0;                          // first exact coefficient of the tile / how many (uniform)
  unsigned cnts = (tr.lo + (unsigned)lane < tr.hi) ? p.tile_cnt[tr.lo + (unsigned)lane] : 0u;   // k_count_tiles' counts of the first 64 tiles
  unsigned S_next = S_wg;                             // ... of the tile to be prefetched next
  auto prefetch = [&](unsigned tile) {
    const unsigned rel = tile - tr.lo;
    S = S_next;
    // (the counts of 64 tiles at a time sit in a register, lane i = tile i of the batch: a load per tile here was a
    // round trip to wait for in every trip of the loop)
    if ((rel & 63u) == 0u && rel != 0u) cnts = (tile + (unsigned)lane < tr.hi) ? p.tile_cnt[tile + (unsigned)lane] : 0u;
    total = (unsigned)__builtin_amdgcn_readlane((int)cnts, (int)(rel & 63u));
    S_next = S + total;
    // (the DMA first: the loads into registers behind it are what the loop waits for -- `landed` below -- and vector
    // memory reads come back in the order they were issued)
    if (total <= (unsigned)DEC_CAP) {
#pragma unroll
      for (int i = 0; i < DEC_CAP / 256; i++)
        if ((unsigned)(i * 256) < total)
          DMA16(r_ac, excbuf + i * 256, lane * 16, (int)((S - S_wg + (unsigned)(i * 256)) * 4u), 0);
    }
    const int vo = (int)(rel * (unsigned)TILE_ELEMS) + lane * 64;
#pragma unroll
    for (int i = 0; i < 4; i++) bw[i] = __builtin_amdgcn_raw_buffer_load_b128(r_bin, vo + i * 16, 0, 0);
    dcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_dc, (int)(rel * 64u + (unsigned)lane) * 4, 0, 0));
  };
  // The next tile's inputs are waited for BEFORE this tile's row stores go out (a use of the registers the compiler can
  // see: it places the wait here).  Reads and writes share one counter and may complete out of order with each other, so
  // a wait for a read issued in front of stores is a wait for the stores as well: at the top of the next trip -- where
  // the inputs are needed -- it would sit behind 32 row stores just issued; here the reads have had the whole inverse
  // transform to land, and the stores in flight are the previous tile's, a tile old.
  // (fp64 only: in the fp32 kernel -- two waves per SIMD cover each other -- the statement made the compiler keep half of
  // the block in scratch across it)
  auto landed = [&]() {
    if constexpr (sizeof(T) == 8) asm volatile("" :: "v"(dcv));      // (the youngest of the reads: the others came back before it)
  };

  // The image's rows -> HBM, one 1 KiB row (8 whole 128-byte lines) per instruction; blocks beyond the end fall outside
  // r_out.  (Sending the second half of a tile's rows one stage later, behind the next tile's de-quantisation, so that
  // the 32 stores -- 7 K cycles at the issue port -- come in two bursts, measured slower: 0.287 against 0.260 ms at
  // p = 17 %, no gain at 5 %.)
  auto store_rows = [&](auto phase, unsigned tile_s) {
    constexpr int PHASE = decltype(phase)::value;
    const int vbase = (int)((tile_s - tr.lo) * (unsigned)G::TILEB);
#pragma unroll
    for (int jg = 0; jg < 8; jg++) {
      unsigned org = 0;
      int cg = 0;
      if (GEOM != GEOM_1D && nd_direct) {              // this lane's piece of row (jg, s): a chunk of block 8 jg + beta of the tile
        const int beta = lane >> 3, gam = lane & 7;
        org = nd_block_origin<T>(p.nd, tile_s * (unsigned)TILE_BLKS + (unsigned)(8 * jg + beta));
        cg = swz_row_chunk(beta, gam, jg);
      }
      const int vo = vbase + jg * 8 * G::BLKB + tm.g_of(jg);
#pragma unroll
      for (int s = 0; s < G::SEGP; s++) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(outbuf + (jg * G::SEGP + s) * 1024 + lane * 16);
        int at = vo + (PHASE * G::SEGP + s) * 128;
        if (GEOM != GEOM_1D && nd_direct) at = (int)(org >= 0xFFFFFFF0u ? org : org + nd_chunk_offset<T>(p.nd, 8 * (PHASE * G::SEGP + s) + cg));
        __builtin_amdgcn_raw_buffer_store_b128(v, r_out, at, 0, DCTZ_DEC_STORE_AUX);
      }
    }
  };
  if (tr.lo < tr.hi) { prefetch(tr.lo); landed(); }
  for (unsigned tile = tr.lo; tile < tr.hi; tile++) {
    const unsigned rel = tile - tr.lo;
    const unsigned blks_here = min((unsigned)TILE_BLKS, p.nfull - tile * TILE_BLKS);
    const bool active = (unsigned)lane < blks_here;
    const unsigned S_t = S, total_t = total;
    // The tile's exact coefficients AC_exact[S_t, S_t + total_t) are read out of LDS: staged one tile ahead in their own
    // buffer when they are few (DecStage::CAP: prefetch), and for a dense tile brought into the output image now -- the
    // image is free until this tile's own rows are written, and up to 4032 floats are half of it.  (Round 2 gathered a
    // dense tile's coefficients one by one from global memory: 0.52 ms against 0.22 ms for the decode of 512^3 at eb 1e-5.)
    const bool staged = total_t <= (unsigned)DEC_CAP;
    const float* const stage = reinterpret_cast<const float*>(io) + (staged ? (unsigned)(G::PHB / 4) : 0u);
    const unsigned stage_last = staged ? (unsigned)DEC_CAP - 1u : (unsigned)TILE_ELEMS - 1u;
    if (!staged) {
      for (unsigned i = 0; i * 256u < total_t; i++)
        DMA16(r_ac, io + i * 1024u, lane * 16, (int)((S_t - S_wg + i * 256u) * 4u), 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned w[16] = {bw[0].x, bw[0].y, bw[0].z, bw[0].w, bw[1].x, bw[1].y, bw[1].z, bw[1].w,
                      bw[2].x, bw[2].y, bw[2].z, bw[2].w, bw[3].x, bw[3].y, bw[3].z, bw[3].w};
    const float dc_t = dcv;
    // where this lane's exact coefficients start: count the flags of the block (byte == 255, j != 0), scan over the wave
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const unsigned v = ~w[i];                                        // a zero byte of v <=> bin id 255
      const unsigned z = ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v;        // bit 7 of a byte set <=> that byte of v is non-zero
      unsigned m = ~z & 0x80808080u;
      if (i == 0) m &= ~0x80u;                                         // j = 0 is the DC slot (:392 / :438)
      n += (unsigned)__popc(m);
    }
    if (!active) n = 0;
    unsigned ptr = wave_incl_scan(n) - n;                              // index inside the tile's piece of AC_exact
    if (S_t + total_t > p.ac_count) underrun = true;                  // the stream promises more than the caller provides
    T x[64];
    if constexpr (sizeof(T) == 8) {
    // Four coefficients = one dword of bin ids at a time: the (up to four) exact coefficients of the group are fetched
    // together -- their places follow from the flag bits alone -- so that a tile pays one LDS round trip per GROUP that
    // has a flag somewhere in the wave (16 at most), not one per flagged POSITION (63 on noisy data: at p = 17 % the
    // serialised round trips were a third of the kernel; 512^3 fp64: 0.33 -> 0.28 ms there, 0.245 -> 0.235 at p = 5 %).
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const unsigned wg = w[g];
      const unsigned nv = ~wg;                                         // a zero byte of nv <=> bin id 255
      unsigned m = ~(((nv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nv) & 0x80808080u;
      if (g == 0) m &= ~0x80u;                                         // j = 0 is the DC slot (:392 / :438)
      const unsigned w1 = ((wg >> 1) & 0x7F7F7F7Fu) + (wg & 0x01010101u);   // four magnitudes (b + 1) >> 1
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(m != 0u)) {                      // :400 / :446 somewhere in the wave
        unsigned at[4];
        at[0] = ptr;
        at[1] = at[0] + ((m >> 7) & 1u);
        at[2] = at[1] + ((m >> 15) & 1u);
        at[3] = at[2] + ((m >> 23) & 1u);
        ptr = at[3] + (m >> 31);
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = stage[min(at[i], stage_last)];      // (all lanes read: predicating the reads on the flag measured slower)
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int j = 4 * g + i;
        if (j == 0) { x[0] = (T)dc_t; continue; }                      // :392 / :438
        T v;
        if constexpr (BC_ARITH) v = bin_centre<T>(w1, nv, i, p.bin_width);
        else v = bctab[(wg >> (8 * i)) & 255u];                        // :416 / :462
        if ((m >> (8 * i + 7)) & 1u) v = (T)e[i];
        x[j] = v;
      }
    }
    if constexpr (MODE == DCTZHIP_QT) {
      // dctz-decomp-lib.c:404-409 in a pass of its own over the flagged positions: inside the loop above every division sat
      // right behind the LDS read of its coefficient, one exposed round trip per flagged group (with one wave per SIMD the
      // QT decoder spent 76 % more cycles waiting than its EC twin for FEWER vector instructions, profiles/r04_pmc_qt.txt);
      // here the coefficients are in registers already and the reads of all sixteen groups overlap as they do in EC mode
#pragma unroll
      for (int g = 0; g < 16; g++) {
        const unsigned nv = ~w[g];
        unsigned m = ~(((nv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nv) & 0x80808080u;
        if (g == 0) m &= ~0x80u;
        if (__builtin_amdgcn_ballot_w64(m != 0u)) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int j = 4 * g + i;
            if (j != 0 && ((m >> (8 * i + 7)) & 1u)) x[j] = qt_restore(x[j], qtl.at(j), p.eb, T(10), p.range_min, p.range_max);
          }
        }
      }
    }
    } else {
    // fp32 (seven waves per CU hide the round trips; the grouped form measured 10 % slower here): position by position
    x[0] = (T)dc_t;                                                    // :392 / :438
#pragma unroll
    for (int j = 1; j < 64; j++) {
      const unsigned b = (w[j >> 2] >> (8 * (j & 3))) & 255u;
      T v;
      if constexpr (BC_ARITH) {
        const unsigned wj = w[j >> 2];
        v = bin_centre<T>(((wj >> 1) & 0x7F7F7F7Fu) + (wj & 0x01010101u), ~wj, j & 3, p.bin_width);
      } else v = bctab[b];                                             // :416 / :462
      if (b == 255u) {                                                 // :400 / :446
        const float e = stage[min(ptr, stage_last)];
        ptr++;
        v = (T)e;
        if (MODE == DCTZHIP_QT) v = qt_restore(v, qtl.at(j), p.eb, T(10), p.range_min, p.range_max);
      }
      x[j] = v;
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the staged coefficients are consumed: the strip is free
    if (tile + 1 < tr.hi) prefetch(tile + 1);
    block_inv<T, CTab<T>, GEOM, (PH > 1)>(x, tab);
    if (scale) {
#pragma unroll
      for (int j = 0; j < 64; j++) x[j] = x[j] * p.sf;                 // dctz-decomp-lib.c:494-511
    }
    if (tile + 1 < tr.hi) landed();
    // registers -> LDS image -> HBM, one 1 KiB row (8 whole 128-byte lines) per instruction, a phase at a time; blocks
    // beyond the end fall outside r_out
    auto store_phase = [&](auto phase) {
      constexpr int PHASE = decltype(phase)::value;
      write_phase<T, PH, PHASE>(x, outbuf, tm);
      store_rows(phase, tile);
    };
    store_phase(std::integral_constant<int, 0>{});
    if (PH == 2) store_phase(std::integral_constant<int, PH - 1>{});
  }
  if (underrun) atomicExch(&p.ctl->error, 2u);
}


template <typename T, int MODE, int PH, int GEOM>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : PH, (sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : PH))) void k_decompress(InvParams<T> p, FinArgs fin) {
  decompress_body<T, MODE, PH, GEOM>(p, blockIdx.x, gridDim.x, [&]() {
    if (fin.box != nullptr && blockIdx.x == 0) {
      // the one thing the host waits for on decode: does the stream promise more exact coefficients than the caller
      // provides (all counts are in: k_count_tiles)?  Known before the first block is rebuilt -> hand it over now.
      const int lane = threadIdx.x;
      unsigned all = 0;
      for (unsigned i = (unsigned)lane; i < p.nwg; i += WG) all += p.wg_cnt[i];
      all = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(all), 63);
      FinBody f;
      f.ctl = fin.ctl; f.part = nullptr; f.nparts = 0; f.box = fin.box; f.seq = fin.seq; f.guess = nullptr;
      f.cnt_known = true; f.cnt_total = all;
      f.err_known = true; f.error = all > p.ac_count ? 2u : 0u;
      finish_body<false>(f);
    }
  });
}

// ---- the same kernel with TILE-INTERLEAVED workgroups (flat blocks, one array) -------------------------------------------
// Workgroup b of G takes tiles b, b + G, b + 2 G, ...: at any moment the grid writes ONE contiguous window of the output
// instead of G ranges side by side.  A 1 GiB stream of 1 KiB row stores from 1024 single-wave workgroups runs at 5.0 TB/s
// in the range-per-workgroup shape and at 5.4 in this one (tools/ubench/store_shapes.hip), and the reconstruction is
// seven eighths of what k_decompress moves.  The running pos of dctz-decomp-lib.c:402-412 at a tile = the counts of the
// k_count_tiles RANGES in front of the tile's range (their exclusive prefix: built once per workgroup in the LDS array that
// is still free then) + the counts of the range's tiles in front of it (tile_pre, left by k_count_tiles); a workgroup keeps
// the start and the count of its next 64 tiles in a register pair, lane r = its r-th tile.
template <typename T, int MODE, int PH, typename Handoff>
__device__ __forceinline__ void decompress_il_body(const InvParams<T>& p, const unsigned wg, const unsigned nwg, Handoff&& handoff) {
  using G = Geo<T, PH>;
  constexpr int DEC_CAP = DecStage<T>::CAP;
  __shared__ __attribute__((aligned(1024))) unsigned char io[G::PHB + DEC_CAP * 4];
  unsigned char* const outbuf = io;
  float* const excbuf = reinterpret_cast<float*>(io + G::PHB);
  static_assert(G::PHB + DEC_CAP * 4 >= TILE_ELEMS * 4, "a dense tile's coefficients fit the array");
  static_assert(G::PHB + DEC_CAP * 4 >= 4096 * 4, "the ranges' prefix (at most 4096 of them) fits the array before the first tile");
  constexpr bool BC_ARITH = DCTZ_BC_ARITH != 0 && sizeof(T) == 8;
  __shared__ __attribute__((aligned(16))) T bctab[BC_ARITH ? 1 : 256]; // bin_center[] of gen_bins
  const int lane = threadIdx.x;
  // the ranges' exclusive prefix -> LDS (p.nwg ranges, the partition k_count_tiles counted in)
  // (every workgroup does this before its first tile: the counts come in with eight independent loads per lane and trip --
  // one load per trip was sixteen to twenty-eight dependent round trips, 8-14 us -- and are scanned out of LDS, 64 entries
  // = one conflict-free row per wave scan)
  unsigned* const rpre = reinterpret_cast<unsigned*>(io);
  {
    for (unsigned i0 = 0; i0 < p.nwg; i0 += 512u) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const unsigned i = i0 + (unsigned)u * 64u + (unsigned)lane; v[u] = i < p.nwg ? p.wg_cnt[i] : 0u; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const unsigned i = i0 + (unsigned)u * 64u + (unsigned)lane; if (i < 4096u) rpre[i] = v[u]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned run = 0;
    for (unsigned i0 = 0; i0 < p.nwg; i0 += 64u) {
      const unsigned v = rpre[i0 + (unsigned)lane];
      const unsigned incl = wave_incl_scan(v);
      rpre[i0 + (unsigned)lane] = run + incl - v;
      run += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  handoff();
  // the inverse of tile_range(): which range a tile lies in
  const unsigned rq = p.ntiles / p.nwg, rr = p.ntiles % p.nwg;
  auto range_of = [&](unsigned t) -> unsigned { return t < rr * (rq + 1u) ? t / (rq + 1u) : rr + (t - rr * (rq + 1u)) / rq; };
  const unsigned my_tiles = wg < p.ntiles ? (p.ntiles - wg + nwg - 1u) / nwg : 0u;      // <= 64 (the host checks)
  unsigned starts = 0, cnts = 0;                     // lane r: first exact coefficient / count of this workgroup's r-th tile
  if ((unsigned)lane < my_tiles) {
    const unsigned t = wg + (unsigned)lane * nwg;
    starts = rpre[range_of(t)] + p.tile_pre[t];
    cnts = p.tile_cnt[t];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (rpre is read: the array may be written now)
  TileMap<T, PH> tm;
  tm.init(lane);
  const CTab<T> tab = as_ctab<T>(p.tab);
  if (!BC_ARITH)
    for (int b = lane; b < 256; b += WG) {
      const int ti = (b & 1) ? (b >> 1) + 1 : -(b >> 1);
      bctab[b] = (T)ti * p.bin_width;
    }
  QtLanes<T> qtl{};
  if (MODE == DCTZHIP_QT) qtl.load(p.qtab, lane);
  const bool scale = (p.sf != T(1));                 // dctz-decomp-lib.c:496 / :505
  bool underrun = false;

  // a tile's buffers behind descriptors based at the tile (32-bit offsets stay small for any N)
  auto blocks_of = [&](unsigned tile) { return min((unsigned)TILE_BLKS, p.nfull - tile * (unsigned)TILE_BLKS); };
  u32x4 bw[4];
  float dcv = 0.f;
  unsigned S = 0, total = 0;                          // first exact coefficient of the prefetched tile / how many (uniform)
  auto prefetch = [&](unsigned tile, unsigned r) {
    S = (unsigned)__builtin_amdgcn_readlane((int)starts, (int)r);
    total = (unsigned)__builtin_amdgcn_readlane((int)cnts, (int)r);
    const unsigned nb = blocks_of(tile);
    const size_t ac_left = S < p.ac_count ? (size_t)(p.ac_count - S) * 4 : 0;
    const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ac + (ac_left ? S : 0u)), 0, (int)min(ac_left, (size_t)0x7ffffffc), 0x00020000);
    if (total <= (unsigned)DEC_CAP) {
#pragma unroll
      for (int i = 0; i < DEC_CAP / 256; i++)
        if ((unsigned)(i * 256) < total) DMA16(r_ac, excbuf + i * 256, lane * 16, i * 1024, 0);
    }
    const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bin + (size_t)tile * TILE_ELEMS), 0, (int)(nb * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dc + (size_t)tile * TILE_BLKS), 0, (int)(nb * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; i++) bw[i] = __builtin_amdgcn_raw_buffer_load_b128(r_bin, lane * 64 + i * 16, 0, 0);
    dcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_dc, lane * 4, 0, 0));
  };
  auto landed = [&]() {
    if constexpr (sizeof(T) == 8) asm volatile("" :: "v"(dcv));
  };
  if (my_tiles) { prefetch(wg, 0u); landed(); }
  for (unsigned r = 0; r < my_tiles; r++) {
    const unsigned tile = wg + r * nwg;
    const unsigned blks_here = blocks_of(tile);
    const bool active = (unsigned)lane < blks_here;
    const unsigned S_t = S, total_t = total;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)tile * TILE_ELEMS, 0, (int)(blks_here * 64u * (unsigned)sizeof(T)), 0x00020000);
    const bool staged = total_t <= (unsigned)DEC_CAP;
    const float* const stage = reinterpret_cast<const float*>(io) + (staged ? (unsigned)(G::PHB / 4) : 0u);
    const unsigned stage_last = staged ? (unsigned)DEC_CAP - 1u : (unsigned)TILE_ELEMS - 1u;
    if (!staged) {
      const size_t ac_left = S_t < p.ac_count ? (size_t)(p.ac_count - S_t) * 4 : 0;
      const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ac + (ac_left ? S_t : 0u)), 0, (int)min(ac_left, (size_t)0x7ffffffc), 0x00020000);
      for (unsigned i = 0; i * 256u < total_t; i++) DMA16(r_ac, io + i * 1024u, lane * 16, (int)(i * 1024u), 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned w[16] = {bw[0].x, bw[0].y, bw[0].z, bw[0].w, bw[1].x, bw[1].y, bw[1].z, bw[1].w,
                      bw[2].x, bw[2].y, bw[2].z, bw[2].w, bw[3].x, bw[3].y, bw[3].z, bw[3].w};
    const float dc_t = dcv;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const unsigned v = ~w[i];                                        // a zero byte of v <=> bin id 255
      const unsigned z = ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v;        // bit 7 of a byte set <=> that byte of v is non-zero
      unsigned m = ~z & 0x80808080u;
      if (i == 0) m &= ~0x80u;                                         // j = 0 is the DC slot (:392 / :438)
      n += (unsigned)__popc(m);
    }
    if (!active) n = 0;
    unsigned ptr = wave_incl_scan(n) - n;                              // index inside the tile's piece of AC_exact
    if (S_t + total_t > p.ac_count) underrun = true;                  // the stream promises more than the caller provides
    T x[64];
    if constexpr (sizeof(T) == 8) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const unsigned wgd = w[g];
      const unsigned nv = ~wgd;                                        // a zero byte of nv <=> bin id 255
      unsigned m = ~(((nv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nv) & 0x80808080u;
      if (g == 0) m &= ~0x80u;                                         // j = 0 is the DC slot (:392 / :438)
      const unsigned w1 = ((wgd >> 1) & 0x7F7F7F7Fu) + (wgd & 0x01010101u);   // four magnitudes (b + 1) >> 1
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(m != 0u)) {                      // :400 / :446 somewhere in the wave
        unsigned at[4];
        at[0] = ptr;
        at[1] = at[0] + ((m >> 7) & 1u);
        at[2] = at[1] + ((m >> 15) & 1u);
        at[3] = at[2] + ((m >> 23) & 1u);
        ptr = at[3] + (m >> 31);
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = stage[min(at[i], stage_last)];
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int j = 4 * g + i;
        if (j == 0) { x[0] = (T)dc_t; continue; }                      // :392 / :438
        T v;
        if constexpr (BC_ARITH) v = bin_centre<T>(w1, nv, i, p.bin_width);
        else v = bctab[(wgd >> (8 * i)) & 255u];                       // :416 / :462
        if ((m >> (8 * i + 7)) & 1u) v = (T)e[i];
        x[j] = v;
      }
    }
    if constexpr (MODE == DCTZHIP_QT) {
      // dctz-decomp-lib.c:404-409 in a pass of its own over the flagged positions: inside the loop above every division sat
      // right behind the LDS read of its coefficient, one exposed round trip per flagged group (with one wave per SIMD the
      // QT decoder spent 76 % more cycles waiting than its EC twin for FEWER vector instructions, profiles/r04_pmc_qt.txt);
      // here the coefficients are in registers already and the reads of all sixteen groups overlap as they do in EC mode
#pragma unroll
      for (int g = 0; g < 16; g++) {
        const unsigned nv = ~w[g];
        unsigned m = ~(((nv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nv) & 0x80808080u;
        if (g == 0) m &= ~0x80u;
        if (__builtin_amdgcn_ballot_w64(m != 0u)) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int j = 4 * g + i;
            if (j != 0 && ((m >> (8 * i + 7)) & 1u)) x[j] = qt_restore(x[j], qtl.at(j), p.eb, T(10), p.range_min, p.range_max);
          }
        }
      }
    }
    } else {
    x[0] = (T)dc_t;                                                    // :392 / :438
#pragma unroll
    for (int j = 1; j < 64; j++) {
      const unsigned b = (w[j >> 2] >> (8 * (j & 3))) & 255u;
      T v;
      if constexpr (BC_ARITH) {
        const unsigned wj = w[j >> 2];
        v = bin_centre<T>(((wj >> 1) & 0x7F7F7F7Fu) + (wj & 0x01010101u), ~wj, j & 3, p.bin_width);
      } else v = bctab[b];                                             // :416 / :462
      if (b == 255u) {                                                 // :400 / :446
        const float e = stage[min(ptr, stage_last)];
        ptr++;
        v = (T)e;
        if (MODE == DCTZHIP_QT) v = qt_restore(v, qtl.at(j), p.eb, T(10), p.range_min, p.range_max);
      }
      x[j] = v;
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the staged coefficients are consumed: the strip is free
    if (r + 1 < my_tiles) prefetch(tile + nwg, r + 1);
    block_inv<T, CTab<T>, GEOM_1D, (PH > 1)>(x, tab);
    if (scale) {
#pragma unroll
      for (int j = 0; j < 64; j++) x[j] = x[j] * p.sf;                 // dctz-decomp-lib.c:494-511
    }
    if (r + 1 < my_tiles) landed();
    auto store_phase = [&](auto phase) {
      constexpr int PHASE = decltype(phase)::value;
      write_phase<T, PH, PHASE>(x, outbuf, tm);
#pragma unroll
      for (int jg = 0; jg < 8; jg++) {
        const int vo = jg * 8 * G::BLKB + tm.g_of(jg);
#pragma unroll
        for (int sg = 0; sg < G::SEGP; sg++) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(outbuf + (jg * G::SEGP + sg) * 1024 + lane * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, r_out, vo + (PHASE * G::SEGP + sg) * 128, 0, DCTZ_DEC_STORE_AUX);
        }
      }
    };
    store_phase(std::integral_constant<int, 0>{});
    if (PH == 2) store_phase(std::integral_constant<int, PH - 1>{});
  }
  if (underrun) atomicExch(&p.ctl->error, 2u);
}
template <typename T, int MODE, int PH>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : PH, (sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : PH))) void k_decompress_il(InvParams<T> p, FinArgs fin) {
  decompress_il_body<T, MODE, PH>(p, blockIdx.x, gridDim.x, [&]() {
    if (fin.box != nullptr && blockIdx.x == 0) {
      const int lane = threadIdx.x;
      unsigned all = 0;
      for (unsigned i = (unsigned)lane; i < p.nwg; i += WG) all += p.wg_cnt[i];
      all = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(all), 63);
      FinBody f;
      f.ctl = fin.ctl; f.part = nullptr; f.nparts = 0; f.box = fin.box; f.seq = fin.seq; f.guess = nullptr;
      f.cnt_known = true; f.cnt_total = all;
      f.err_known = true; f.error = all > p.ac_count ? 2u : 0u;
      finish_body<false>(f);
    }
  });
}

// Last, short block on decode (dctz-decomp-lib.c:423-428, dct.c:144-199).
template <typename T, int MODE>
__device__ __forceinline__ void decompress_rem_body(const InvParams<T>& p, const int l, const bool SCALE) {
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
  unsigned before = 0;                                             // everything the full blocks consumed
  for (unsigned i = (unsigned)k; i < p.nwg; i += 64) before += p.wg_cnt[i];
  const unsigned start = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(before), 63);
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
template <typename T, int MODE, bool SCALE>
__global__ __launch_bounds__(64) void k_decompress_rem(InvParams<T> p, int l) { decompress_rem_body<T, MODE>(p, l, SCALE); }

// ================================================================= launchers ==
#if DCTZ_PART == 0
void launch_finish(Ctl* ctl, const double* part, int nparts, HostBox* box, unsigned long long seq, hipStream_t s, const SfGuess* guess,
                   unsigned* zero_words, unsigned nzero) {
  const FinArgs f = {ctl, part, nparts, box, seq, guess, zero_words, nzero};
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(SWG), 0, s, f);
}
#endif

template <typename T>
void launch_compress(const FwdParams<T>& p, int mode, bool stats, int grid, int geom, hipStream_t s) {
  if (geom == GEOM_1D && p.scaled != nullptr) {        // the variant that writes x / sf back as well
    if (mode == DCTZHIP_EC) {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, true, Phases<T>::C, GEOM_1D, true>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, false, Phases<T>::C, GEOM_1D, true>), dim3(grid), dim3(WG), 0, s, p);
    } else {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, true, Phases<T>::C, GEOM_1D, true>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, false, Phases<T>::C, GEOM_1D, true>), dim3(grid), dim3(WG), 0, s, p);
    }
  } else if (geom == GEOM_1D) {
    if (mode == DCTZHIP_EC) {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, true, Phases<T>::C, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, false, Phases<T>::C, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p);
    } else {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, true, Phases<T>::C, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, false, Phases<T>::C, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p);
    }
  } else if (geom == GEOM_2D) {
    if (mode == DCTZHIP_EC) {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, true, Phases<T>::C, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, false, Phases<T>::C, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p);
    } else {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, true, Phases<T>::C, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, false, Phases<T>::C, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p);
    }
  } else {
    if (mode == DCTZHIP_EC) {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, true, Phases<T>::C, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_EC, false, Phases<T>::C, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p);
    } else {
      if (stats) hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, true, Phases<T>::C, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p);
      else hipLaunchKernelGGL((k_compress<T, DCTZHIP_QT, false, Phases<T>::C, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p);
    }
  }
}

// Resident workgroups per CU of the k_compress instantiation a launch would pick (registers AND LDS: the persistent
// grid must not exceed what is resident at once, or its tail runs as a second round)
template <typename T>
int compress_occupancy(int mode, bool stats, int geom, bool scaled) {
  int n = 0;
  hipError_t e;
#define OCC(M, S, G) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress<T, M, S, Phases<T>::C, G>, WG, 0)
#define OCCS(M, S) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress<T, M, S, Phases<T>::C, GEOM_1D, true>, WG, 0)
  if (geom == GEOM_1D && scaled) { if (mode == DCTZHIP_EC) { if (stats) OCCS(DCTZHIP_EC, true); else OCCS(DCTZHIP_EC, false); } else { if (stats) OCCS(DCTZHIP_QT, true); else OCCS(DCTZHIP_QT, false); } }
  else if (geom == GEOM_1D) { if (mode == DCTZHIP_EC) { if (stats) OCC(DCTZHIP_EC, true, GEOM_1D); else OCC(DCTZHIP_EC, false, GEOM_1D); } else { if (stats) OCC(DCTZHIP_QT, true, GEOM_1D); else OCC(DCTZHIP_QT, false, GEOM_1D); } }
  else if (geom == GEOM_2D) { if (mode == DCTZHIP_EC) { if (stats) OCC(DCTZHIP_EC, true, GEOM_2D); else OCC(DCTZHIP_EC, false, GEOM_2D); } else { if (stats) OCC(DCTZHIP_QT, true, GEOM_2D); else OCC(DCTZHIP_QT, false, GEOM_2D); } }
  else { if (mode == DCTZHIP_EC) { if (stats) OCC(DCTZHIP_EC, true, GEOM_3D); else OCC(DCTZHIP_EC, false, GEOM_3D); } else { if (stats) OCC(DCTZHIP_QT, true, GEOM_3D); else OCC(DCTZHIP_QT, false, GEOM_3D); } }
#undef OCC
#undef OCCS
  return e == hipSuccess ? n : 0;
}
template <typename T>
int decompress_occupancy(int mode, int geom) {
  int n = 0;
  hipError_t e;
#define OCC(M, G) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_decompress<T, M, Phases<T>::D, G>, WG, 0)
  if (geom == GEOM_1D) { if (mode == DCTZHIP_EC) OCC(DCTZHIP_EC, GEOM_1D); else OCC(DCTZHIP_QT, GEOM_1D); }
  else if (geom == GEOM_2D) { if (mode == DCTZHIP_EC) OCC(DCTZHIP_EC, GEOM_2D); else OCC(DCTZHIP_QT, GEOM_2D); }
  else { if (mode == DCTZHIP_EC) OCC(DCTZHIP_EC, GEOM_3D); else OCC(DCTZHIP_QT, GEOM_3D); }
#undef OCC
  return e == hipSuccess ? n : 0;
}

template <typename T>
void launch_compress_rem(const FwdParams<T>& p, int mode, int l, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_EC>), dim3(1), dim3(64), 0, s, p, l);
  else hipLaunchKernelGGL((k_compress_rem<T, DCTZHIP_QT>), dim3(1), dim3(64), 0, s, p, l);
}

#if DCTZ_PART == 0
void launch_count_tiles(const uint8_t* bin, unsigned nfull, unsigned ntiles, unsigned nwg, unsigned* tile_cnt, unsigned* wg_cnt, hipStream_t s,
                        const void* qtab_host, size_t qtab_bytes, void* qtab_dev, unsigned* tile_pre) {
  QtabArg q;
  std::memset(&q, 0, sizeof(q));
  if (qtab_host && qtab_bytes <= sizeof(q)) std::memcpy(&q, qtab_host, qtab_bytes); else qtab_bytes = 0;
  hipLaunchKernelGGL(k_count_tiles, dim3(nwg), dim3(SWG), 0, s, bin, nfull, ntiles, nwg, tile_cnt, wg_cnt, q, (unsigned)(qtab_bytes / 8),
                     reinterpret_cast<unsigned long long*>(qtab_dev), tile_pre);
}
#endif

template <typename T>
void launch_compact_ac(const FwdParams<T>& p, int mode, double eb, unsigned nlists, int grid, const FinArgs& fin, hipStream_t s) {
  // (grid = nlists; second dimension: chunks of COMPACT_TPW tiles of the longest list)
  const unsigned per_list = p.nlists_main ? (p.ntiles + p.nlists_main - 1) / p.nlists_main : 0u;
  const unsigned chunks = per_list ? (per_list + COMPACT_TPW - 1) / COMPACT_TPW : 1u;
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_compact_ac<T, DCTZHIP_EC>), dim3(grid, chunks), dim3(SWG), 0, s, p, eb, nlists, fin);
  else hipLaunchKernelGGL((k_compact_ac<T, DCTZHIP_QT>), dim3(grid, chunks), dim3(SWG), 0, s, p, eb, nlists, fin);
}

template <typename T>
void launch_decompress(const InvParams<T>& p, int mode, int grid, const FinArgs& fin, int geom, hipStream_t s) {
  // tile-interleaved workgroups where the shim sets tile_pre: by default fp64 EC only.  Measured on one box, builds
  // alternating (tools/r04_il.sh, r04_il2.sh): fp64 EC 220 -> 200-206 us, fp64 EC at p = 0.69 316-335 -> 313-320; fp64 QT
  // 245 -> 257 and fp32 119 -> 125 the OTHER way (kernels their arithmetic holds, not their store stream: they only pay for
  // the scattered reads and the per-tile descriptors), so those keep a contiguous range per workgroup.
  if (geom == GEOM_1D && p.tile_pre != nullptr) {     // (the shim chooses: fp64 EC by default, everything with DCTZHIP_DEC_IL=2)
    if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress_il<T, DCTZHIP_EC, Phases<T>::D>), dim3(grid), dim3(WG), 0, s, p, fin);
    else hipLaunchKernelGGL((k_decompress_il<T, DCTZHIP_QT, Phases<T>::D>), dim3(grid), dim3(WG), 0, s, p, fin);
  } else if (geom == GEOM_1D) {
    if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress<T, DCTZHIP_EC, Phases<T>::D, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p, fin);
    else hipLaunchKernelGGL((k_decompress<T, DCTZHIP_QT, Phases<T>::D, GEOM_1D>), dim3(grid), dim3(WG), 0, s, p, fin);
  } else if (geom == GEOM_2D) {
    if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress<T, DCTZHIP_EC, Phases<T>::D, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p, fin);
    else hipLaunchKernelGGL((k_decompress<T, DCTZHIP_QT, Phases<T>::D, GEOM_2D>), dim3(grid), dim3(WG), 0, s, p, fin);
  } else {
    if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress<T, DCTZHIP_EC, Phases<T>::D, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p, fin);
    else hipLaunchKernelGGL((k_decompress<T, DCTZHIP_QT, Phases<T>::D, GEOM_3D>), dim3(grid), dim3(WG), 0, s, p, fin);
  }
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

// ================================================================== batches ==
// k arrays of one element type in one launch sequence (dctz_device.h: BatchFwd / BatchInv).  A workgroup finds its array
// from the launch's `first[]` (sorted, first[k] = grid; arrays without a workgroup in this launch have an empty range),
// copies that array's parameter block out of the item table with scalar loads and runs the single-array body on it.
// Largest i with first[i] <= b: 64 entries per step, one ballot each (a wave is uniform in b).
__device__ __forceinline__ unsigned batch_item_of(const unsigned* __restrict__ first, const unsigned k, const unsigned b) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned cnt = 0;
  for (unsigned base = 0; base < k; base += 64u) {
    const unsigned i = base + lane;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(i < k && first[i] <= b);
    cnt += (unsigned)__popcll(m);
    if (m != ~0ull) break;
  }
  return cnt - 1u;                                   // first[0] = 0 <= b
}
// a parameter block out of the item table: dword by dword through the constant address space (scalar loads: the index is
// wave-uniform, and nothing in the sequence writes the table after its first kernel)
template <typename S>
__device__ __forceinline__ S load_params(const S* src) {
  static_assert(sizeof(S) % 4 == 0, "parameter blocks are whole dwords");
  constexpr int W = (int)(sizeof(S) / 4);
  const __attribute__((address_space(4))) unsigned* w = (const __attribute__((address_space(4))) unsigned*)(src);
  unsigned buf[W];
#pragma unroll
  for (int i = 0; i < W; i++) buf[i] = w[i];
  S v;
  __builtin_memcpy(&v, buf, sizeof(S));
  return v;
}

// STATS: the sequence has speculative items (BatchFwd::sample) -- every workgroup also reduces max|x|, min|x| and the sum of
// what it reads into p.stat_part (every item of such a sequence has one), as k_compress<..., STATS> does for a single array.
template <typename T, int MODE, bool STATS>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(compress_waves<T, MODE, Phases<T>::C>())))
void k_compress_batch(const BatchFwd<T>* items, const unsigned* __restrict__ first, unsigned k) {
  const unsigned i = batch_item_of(first, k, blockIdx.x);
  const FwdParams<T> p = load_params(&items[i].p);
  compress_body<T, MODE, STATS, Phases<T>::C, GEOM_1D>(p, blockIdx.x - first[i], p.nlists_main);
}

template <typename T, int MODE>
__global__ __launch_bounds__(64) void k_compress_rem_batch(const BatchFwd<T>* items, const unsigned* __restrict__ rem_items) {
  const unsigned i = rem_items[blockIdx.x];
  const FwdParams<T> p = load_params(&items[i].p);
  compress_rem_body<T, MODE>(p, (int)items[i].rem);
}

// Hand-off of a whole batch by ONE workgroup (the first of the sequence's last kernel): per array, what the single-array
// hand-off publishes -- everything was produced by earlier kernels of the sequence.
template <typename T>
__device__ __forceinline__ void batch_finish_compress(const BatchFwd<T>* items, unsigned k, const double* bstats, const BatchFin& fin, bool qt) {
  BatchResC* res = reinterpret_cast<BatchResC*>(fin.res);
  // a speculative item's statistics: what k_compress_batch<STATS> (and the remainder block's kernel) left in its partials,
  // one wave per item
  {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (unsigned i = wave; i < k; i += nwaves) {
      const BatchFwd<T>& it = items[i];
      if (!it.sample) continue;
      const double* part = it.p.stat_part;
      const unsigned np = it.p.nlists_main + (it.rem ? 1u : 0u);
      double dmx = 0.0, dmn = 1.79769313486231570815e308, sum = 0.0;
      for (unsigned j = lane; j < np; j += 64u) { dmx = fmax(dmx, part[3 * j]); dmn = fmin(dmn, part[3 * j + 1]); sum += part[3 * j + 2]; }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        dmx = fmax(dmx, __shfl_down(dmx, d));
        dmn = fmin(dmn, __shfl_down(dmn, d));
        sum += __shfl_down(sum, d);
      }
      if (lane == 0) { double* o = const_cast<double*>(bstats) + 3 * i; o[0] = dmx; o[1] = dmn; o[2] = sum; }
    }
    __syncthreads();                                 // (the records below read bstats)
  }
  auto put = [&](unsigned i, unsigned cnt) {
    const BatchFwd<T>& it = items[i];
    BatchResC r;
    r.sf_used = it.p.guess->sf; r.fast_used = it.p.guess->fast_sf;
    r.stats[0] = bstats[3 * i]; r.stats[1] = bstats[3 * i + 1]; r.stats[2] = bstats[3 * i + 2];
    r.cnt = cnt; r.error = 0; r.pad = 0; r.q0 = it.p.ctl->q0;
    res[i] = r;
  };
  // tot_AC_exact_count (:478-544) of every array = the sum of its list lengths: a wave per array for the long ones (a
  // thread alone would walk some thousand words one round trip after the other), a thread per array while the lists are
  // few (small arrays: one list per tile)
  constexpr unsigned SHORT = 16;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (unsigned i = wave; i < k; i += nwaves) {
    const unsigned nl = items[i].nlists;
    if (nl <= SHORT) continue;
    unsigned c = 0;
    for (unsigned l = lane; l < nl; l += 64u) c += items[i].p.tile_cnt[l] & LIST_LEN;
    c = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(c), 63);
    if (lane == 0) put(i, c);
  }
  for (unsigned i = threadIdx.x; i < k; i += blockDim.x) {
    const unsigned nl = items[i].nlists;
    if (nl > SHORT) continue;
    unsigned c = 0;
    for (unsigned l = 0; l < nl; l++) c += items[i].p.tile_cnt[l] & LIST_LEN;
    put(i, c);
  }
  if (qt)
    for (unsigned e = threadIdx.x; e < k * 64u; e += blockDim.x) fin.resq[e >> 6].qraw[e & 63u] = items[e >> 6].p.ctl->qraw[e & 63u];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && fin.word != nullptr) box_publish(fin.word, fin.seq);
}

template <typename T, int MODE>
__global__ __launch_bounds__(SWG) void k_compact_batch(const BatchFwd<T>* items, const unsigned* __restrict__ first, unsigned k,
                                                       const double* bstats, BatchFin fin) {
  __shared__ unsigned sh[SWG / 64];
  if (blockIdx.x == 0 && blockIdx.y == 0 && fin.res != nullptr) batch_finish_compress<T>(items, k, bstats, fin, MODE == DCTZHIP_QT);
  const unsigned i = batch_item_of(first, k, blockIdx.x);
  const FwdParams<T> p = load_params(&items[i].p);
  compact_ac_body<T, MODE>(p, items[i].eb, items[i].nlists, blockIdx.x - first[i], blockIdx.y, sh);
}

// decode, step 1 for a batch: k_count_tiles per array (+ the flags of its remainder block, so that the hand-off knows
// every array's total before the first block is rebuilt); the first workgroups also bring the item table from the
// host's pinned copy into device memory for the kernels behind this one.
template <typename T>
__global__ __launch_bounds__(SWG) void k_count_batch(const BatchInv<T>* items, const unsigned* __restrict__ first, unsigned k,
                                                     const uint4* __restrict__ blob_src, uint4* __restrict__ blob_dst, unsigned blob_vecs) {
  for (unsigned v = blockIdx.x * SWG + threadIdx.x; v < blob_vecs; v += gridDim.x * SWG) blob_dst[v] = blob_src[v];
  const unsigned i = batch_item_of(first, k, blockIdx.x);
  const BatchInv<T>& it = items[i];
  const unsigned wg = blockIdx.x - first[i];
  const InvParams<T>& p = it.p;
  if (p.nwg) count_tiles_body(p.bin, p.nfull, p.ntiles, p.nwg, const_cast<unsigned*>(p.tile_cnt), const_cast<unsigned*>(p.wg_cnt), wg);
  if (wg == 0 && threadIdx.x < 64) {                 // dctz-decomp-lib.c:400 / :446 over the short last block
    const unsigned t = threadIdx.x;
    const bool flag = t != 0 && t < it.rem && p.bin[(size_t)p.nfull * 64 + t] == 255;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    if (t == 0) *it.rem_cnt = (unsigned)__popcll(m);
  }
}

template <typename T>
__device__ __forceinline__ void batch_finish_decompress(const BatchInv<T>* items, unsigned k, const BatchFin& fin) {
  BatchResD* res = reinterpret_cast<BatchResD*>(fin.res);
  auto put = [&](unsigned i, unsigned all) {
    BatchResD r;
    r.total = all;
    r.error = all > items[i].p.ac_count ? 2u : 0u;   // the stream promises more exact coefficients than the caller provides
    res[i] = r;
  };
  // (a wave per array with many workgroups, a thread per array otherwise: as batch_finish_compress)
  constexpr unsigned SHORT = 16;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (unsigned i = wave; i < k; i += nwaves) {
    const unsigned nw = items[i].p.nwg;
    if (nw <= SHORT) continue;
    unsigned c = 0;
    for (unsigned w = lane; w < nw; w += 64u) c += items[i].p.wg_cnt[w];
    c = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(c), 63);
    if (lane == 0) put(i, c + *items[i].rem_cnt);
  }
  for (unsigned i = threadIdx.x; i < k; i += blockDim.x) {
    const unsigned nw = items[i].p.nwg;
    if (nw > SHORT) continue;
    unsigned all = *items[i].rem_cnt;
    for (unsigned w = 0; w < nw; w++) all += items[i].p.wg_cnt[w];
    put(i, all);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && fin.word != nullptr) box_publish(fin.word, fin.seq);
}

template <typename T, int MODE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : Phases<T>::D, (sizeof(T) == 4 && DCTZ_WPED32) ? DCTZ_WPED32 : Phases<T>::D)))
void k_decompress_batch(const BatchInv<T>* items, const unsigned* __restrict__ first, unsigned k, BatchFin fin) {
  if (blockIdx.x == 0 && fin.res != nullptr) batch_finish_decompress<T>(items, k, fin);
  const unsigned i = batch_item_of(first, k, blockIdx.x);
  const InvParams<T> p = load_params(&items[i].p);
  decompress_body<T, MODE, Phases<T>::D, GEOM_1D>(p, blockIdx.x - first[i], p.nwg, []() {});
}

template <typename T, int MODE>
__global__ __launch_bounds__(64) void k_decompress_rem_batch(const BatchInv<T>* items, const unsigned* __restrict__ rem_items) {
  const unsigned i = rem_items[blockIdx.x];
  const InvParams<T> p = load_params(&items[i].p);
  decompress_rem_body<T, MODE>(p, (int)items[i].rem, items[i].scale != 0u);
}

template <typename T>
void launch_compress_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, int mode, bool stats, hipStream_t s) {
  if (mode == DCTZHIP_EC) {
    if (stats) hipLaunchKernelGGL((k_compress_batch<T, DCTZHIP_EC, true>), dim3(grid), dim3(WG), 0, s, items, first, k);
    else hipLaunchKernelGGL((k_compress_batch<T, DCTZHIP_EC, false>), dim3(grid), dim3(WG), 0, s, items, first, k);
  } else {
    if (stats) hipLaunchKernelGGL((k_compress_batch<T, DCTZHIP_QT, true>), dim3(grid), dim3(WG), 0, s, items, first, k);
    else hipLaunchKernelGGL((k_compress_batch<T, DCTZHIP_QT, false>), dim3(grid), dim3(WG), 0, s, items, first, k);
  }
}
template <typename T>
void launch_compress_rem_batch(const BatchFwd<T>* items, const unsigned* rem_items, unsigned nrem, int mode, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_compress_rem_batch<T, DCTZHIP_EC>), dim3(nrem), dim3(64), 0, s, items, rem_items);
  else hipLaunchKernelGGL((k_compress_rem_batch<T, DCTZHIP_QT>), dim3(nrem), dim3(64), 0, s, items, rem_items);
}
template <typename T>
void launch_compact_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, unsigned chunks, int mode, const double* bstats,
                          const BatchFin& fin, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_compact_batch<T, DCTZHIP_EC>), dim3(grid, chunks), dim3(SWG), 0, s, items, first, k, bstats, fin);
  else hipLaunchKernelGGL((k_compact_batch<T, DCTZHIP_QT>), dim3(grid, chunks), dim3(SWG), 0, s, items, first, k, bstats, fin);
}
template <typename T>
void launch_count_batch(const BatchInv<T>* items_src, const unsigned* first_src, unsigned k, unsigned grid, const void* blob_src, void* blob_dst,
                        size_t blob_bytes, hipStream_t s) {
  hipLaunchKernelGGL(k_count_batch<T>, dim3(grid), dim3(SWG), 0, s, items_src, first_src, k, (const uint4*)blob_src, (uint4*)blob_dst,
                     (unsigned)(blob_bytes / 16));
}
template <typename T>
void launch_decompress_batch(const BatchInv<T>* items, const unsigned* first, unsigned k, unsigned grid, int mode, const BatchFin& fin, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress_batch<T, DCTZHIP_EC>), dim3(grid), dim3(WG), 0, s, items, first, k, fin);
  else hipLaunchKernelGGL((k_decompress_batch<T, DCTZHIP_QT>), dim3(grid), dim3(WG), 0, s, items, first, k, fin);
}
template <typename T>
void launch_decompress_rem_batch(const BatchInv<T>* items, const unsigned* rem_items, unsigned nrem, int mode, hipStream_t s) {
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress_rem_batch<T, DCTZHIP_EC>), dim3(nrem), dim3(64), 0, s, items, rem_items);
  else hipLaunchKernelGGL((k_decompress_rem_batch<T, DCTZHIP_QT>), dim3(nrem), dim3(64), 0, s, items, rem_items);
}

// explicit instantiations used by dctz_shim.hip
// (development: tools/dev_one.sh compiles ONE kernel instantiation alone -- seconds instead of a minute -- for
// register-allocation experiments)
// ---- one hot kernel per translation unit -----------------------------------------------------------------------------
// The register allocation of a k_compress instantiation depends on what ELSE is compiled beside it: alone in its module,
// k_compress<double, QT, STATS> gets 256 registers and no scratch; with a second k_compress instantiation in the same module
// it spills ten (round 3's QT kernel: 20 bytes of scratch inside the tile loop, 5-8 % behind its EC twin), and the headline
// EC kernel two (tools/dev_one.sh shows the 'alone' numbers; the inliner and scheduler see another module).  So the flat
// k_compress instantiations and the batch forms are each built as a translation unit of their own: -DDCTZ_PART=n compiles
// this file down to the n-th kernel of the list below, the main build (DCTZ_PART = 0) declares them `extern template` and
// keeps everything else, launchers included (the host side needs only the kernel's handle, a link-time symbol).
#define DCTZ_PARTS 24
#if DCTZ_PART == 1
template __global__ void k_compress<double, DCTZHIP_EC, true, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_EC, true, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#endif
#if DCTZ_PART == 2
template __global__ void k_compress<double, DCTZHIP_EC, true, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_EC, true, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#endif
#if DCTZ_PART == 3
template __global__ void k_compress<double, DCTZHIP_EC, false, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_EC, false, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#endif
#if DCTZ_PART == 4
template __global__ void k_compress<double, DCTZHIP_EC, false, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_EC, false, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#endif
#if DCTZ_PART == 5
template __global__ void k_compress<double, DCTZHIP_QT, true, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_QT, true, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#endif
#if DCTZ_PART == 6
template __global__ void k_compress<double, DCTZHIP_QT, true, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_QT, true, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#endif
#if DCTZ_PART == 7
template __global__ void k_compress<double, DCTZHIP_QT, false, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_QT, false, Phases<double>::C, GEOM_1D, false>(FwdParams<double>);
#endif
#if DCTZ_PART == 8
template __global__ void k_compress<double, DCTZHIP_QT, false, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<double, DCTZHIP_QT, false, Phases<double>::C, GEOM_1D, true>(FwdParams<double>);
#endif
#if DCTZ_PART == 9
template __global__ void k_compress<float, DCTZHIP_EC, true, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_EC, true, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#endif
#if DCTZ_PART == 10
template __global__ void k_compress<float, DCTZHIP_EC, true, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_EC, true, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#endif
#if DCTZ_PART == 11
template __global__ void k_compress<float, DCTZHIP_EC, false, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_EC, false, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#endif
#if DCTZ_PART == 12
template __global__ void k_compress<float, DCTZHIP_EC, false, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_EC, false, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#endif
#if DCTZ_PART == 13
template __global__ void k_compress<float, DCTZHIP_QT, true, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_QT, true, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#endif
#if DCTZ_PART == 14
template __global__ void k_compress<float, DCTZHIP_QT, true, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_QT, true, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#endif
#if DCTZ_PART == 15
template __global__ void k_compress<float, DCTZHIP_QT, false, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_QT, false, Phases<float>::C, GEOM_1D, false>(FwdParams<float>);
#endif
#if DCTZ_PART == 16
template __global__ void k_compress<float, DCTZHIP_QT, false, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#elif DCTZ_PART == 0
extern template __global__ void k_compress<float, DCTZHIP_QT, false, Phases<float>::C, GEOM_1D, true>(FwdParams<float>);
#endif
#if DCTZ_PART == 17
template __global__ void k_compress_batch<double, DCTZHIP_EC, false>(const BatchFwd<double>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<double, DCTZHIP_EC, false>(const BatchFwd<double>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 18
template __global__ void k_compress_batch<double, DCTZHIP_QT, false>(const BatchFwd<double>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<double, DCTZHIP_QT, false>(const BatchFwd<double>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 19
template __global__ void k_compress_batch<float, DCTZHIP_EC, false>(const BatchFwd<float>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<float, DCTZHIP_EC, false>(const BatchFwd<float>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 20
template __global__ void k_compress_batch<float, DCTZHIP_QT, false>(const BatchFwd<float>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<float, DCTZHIP_QT, false>(const BatchFwd<float>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 21
template __global__ void k_compress_batch<double, DCTZHIP_EC, true>(const BatchFwd<double>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<double, DCTZHIP_EC, true>(const BatchFwd<double>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 22
template __global__ void k_compress_batch<double, DCTZHIP_QT, true>(const BatchFwd<double>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<double, DCTZHIP_QT, true>(const BatchFwd<double>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 23
template __global__ void k_compress_batch<float, DCTZHIP_EC, true>(const BatchFwd<float>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<float, DCTZHIP_EC, true>(const BatchFwd<float>*, const unsigned*, unsigned);
#endif
#if DCTZ_PART == 24
template __global__ void k_compress_batch<float, DCTZHIP_QT, true>(const BatchFwd<float>*, const unsigned*, unsigned);
#elif DCTZ_PART == 0
extern template __global__ void k_compress_batch<float, DCTZHIP_QT, true>(const BatchFwd<float>*, const unsigned*, unsigned);
#endif
#ifdef DCTZ_DEV_ONE
template __global__ void DCTZ_DEV_ONE(DCTZ_DEV_ARGS);
#elif DCTZ_PART > 0
// (this translation unit is one kernel: see above)
#else

#define INST(T)                                                                                         \
  template void launch_compress<T>(const FwdParams<T>&, int, bool, int, int, hipStream_t);              \
  template void launch_compress_rem<T>(const FwdParams<T>&, int, int, hipStream_t);                     \
  template void launch_compact_ac<T>(const FwdParams<T>&, int, double, unsigned, int, const FinArgs&, hipStream_t); \
  template void launch_decompress<T>(const InvParams<T>&, int, int, const FinArgs&, int, hipStream_t);  \
  template void launch_decompress_rem<T>(const InvParams<T>&, int, bool, int, hipStream_t);             \
  template void launch_compress_batch<T>(const BatchFwd<T>*, const unsigned*, unsigned, unsigned, int, bool, hipStream_t);                      \
  template void launch_compress_rem_batch<T>(const BatchFwd<T>*, const unsigned*, unsigned, int, hipStream_t);                              \
  template void launch_compact_batch<T>(const BatchFwd<T>*, const unsigned*, unsigned, unsigned, unsigned, int, const double*, const BatchFin&, hipStream_t); \
  template void launch_count_batch<T>(const BatchInv<T>*, const unsigned*, unsigned, unsigned, const void*, void*, size_t, hipStream_t);    \
  template void launch_decompress_batch<T>(const BatchInv<T>*, const unsigned*, unsigned, unsigned, int, const BatchFin&, hipStream_t);     \
  template void launch_decompress_rem_batch<T>(const BatchInv<T>*, const unsigned*, unsigned, int, hipStream_t);                            \
  template int compress_occupancy<T>(int, bool, int, bool);                                                      \
  template int decompress_occupancy<T>(int, int);                                                          \
  template size_t compress_lds_bytes<T>(int);                                                              \
  template size_t decompress_lds_bytes<T>();
INST(double)
INST(float)
#endif

}  // namespace dctz
