// dctz_kernels_one.hip -- ONE launch per call for arrays whose tiles are all resident at once (gfx950, MI355X).
//
// Why.  The big kernels of dctz_kernels.hip are persistent pipelines: they pay off when a workgroup owns many tiles.  An
// array of a few thousand tiles or fewer (BASELINE config C1: 256 tiles, C2: 1582, every member of C5) gives each
// workgroup ONE tile, and the call is then a chain of four to six tiny kernels -- statistics, their reduction, the
// transform kernel, the remainder block, the list placement; on decode the flag counts in front of the reconstruction --
// each a launch, a ramp and a drain around a few microseconds of work (round 3: C2 step 60 us of which 28 us in the
// two big kernels).  Here the whole call is ONE kernel:
//
//   compress   a WAVE owns a tile (lane b = block b, as everywhere), ONE_TW waves make a workgroup, one more workgroup
//              takes the short last block.  calc_data_stat (util.c:12-44): the tile is in registers, so its max|x| / min|x|
//              are known before anything is scaled; every workgroup posts the DECADE of its maximum (all that enters sf,
//              util.c:29) on the board and its first wave sweeps the board for the largest one -- an all-gather of 8-byte
//              granules instead of a kernel boundary -- then the waves scale (dctz-comp-lib.c:193-216), transform
//              (dct.c:55-103) and bin (:363-414).  The coefficients stored exactly are put into the reference's order
//              (:478-544: block after block, j ascending) inside LDS -- a lane knows its block's count, a wave scan gives
//              its place --, the workgroup's count goes on the board, the sum of the counts in front of it (the running
//              tot_AC_exact_count) comes back from it, and every tile's piece of AC_exact[] leaves in coalesced rows at
//              its final place: no workgroup-local lists, no placement kernel.  QT: the per-position maxima (:371-372)
//              are merged with device atomics before the count is posted, every workgroup then waits for ALL counts,
//              reads the table, and normalises (:488-518) its own coefficients on their way into LDS.
//   decompress a wave counts the flags of its tile (dctz-decomp-lib.c:400 / :446) -- the bin ids are in its registers
//              anyway --, the workgroup posts its count, sweeps the counts in front of it (the running pos of :402-412),
//              and every wave stages its piece of AC_exact[] and rebuilds its tile (:389-483, dct.c:115-205, :494-511).
//
// The board (OneBoard, dctz_device.h): granules {tag = epoch of the launch, value}, one per workgroup and step, written by
// ONE agent-scope 8-byte store and read with agent-scope loads (both go past the non-coherent per-XCD caches; the tag makes
// the data its own flag, so no fence and no counter to reset: MI355X guide, "granule" hand-offs).  Everything else a
// workgroup writes for another one (its statistics record for the hand-off, its table atomics) is issued agent-scope too
// and drained (s_waitcnt vmcnt(0)) before the granule that announces it.  Every sweep is bounded (20 ms): a launch whose
// workgroups are not all resident -- the grid is sized from the occupancy query, which is advisory -- ends with
// ONE_ERR_TIMEOUT in the mailbox and the host runs the call through the chain of kernels instead (dctz_shim.hip).
#include "dctz_kernel_common.h"

namespace dctz {

typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load((gu64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store((gu64*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent_f64(const double* p) { return __longlong_as_double((long long)ld_agent(reinterpret_cast<const unsigned long long*>(p))); }
__device__ __forceinline__ void st_agent_f64(double* p, double v) { st_agent(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ unsigned long long granule(unsigned epoch, unsigned value) { return ((unsigned long long)epoch << 32) | value; }

constexpr unsigned long long ONE_SPIN_TICKS = 2000000ull;   // 20 ms of the 100 MHz constant clock (s_memrealtime)
constexpr unsigned ONE_EXC_MAX = 63u * 64u;                  // "stored exactly" coefficients of a tile at most
constexpr unsigned ONE_GAVE_UP = 0xFFFFFFFFu;                // what a sweep that timed out leaves for the other waves
constexpr int OTW = ONE_TW;
// waves per SIMD the register allocation aims at: fp32 two (two workgroups of 64 KiB of LDS per CU), fp64 one (128 KiB)
template <typename T> constexpr int one_waves() { return sizeof(T) == 4 ? 2 : 1; }
constexpr int ONE_SPEC_MIN_WG = 128 * 4 / ONE_TW;                         // workgroups from which k_compress_one scales on a guess (see there)

// Development aid: time stamps of the first wave of every workgroup at the phases of the kernels (OneBoard::dbg != NULL)
__device__ __forceinline__ void one_stamp(const OneBoard& b, int k) {
  if (b.dbg != nullptr && threadIdx.x == 0) b.dbg[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memrealtime();
}
// a record of a batch launch, dword by dword through the constant address space (scalar loads: the index is uniform)
template <typename S>
__device__ __forceinline__ S one_load_rec(const S* src) {
  static_assert(sizeof(S) % 4 == 0, "records are whole dwords");
  constexpr int W = (int)(sizeof(S) / 4);
  const __attribute__((address_space(4))) unsigned* w = (const __attribute__((address_space(4))) unsigned*)(src);
  unsigned buf[W];
#pragma unroll
  for (int i = 0; i < W; i++) buf[i] = w[i];
  S v;
  __builtin_memcpy(&v, buf, sizeof(S));
  return v;
}

// Sweep granules g[0 .. count) by ONE wave: use(value, index) once per granule, by the lane that read it; false: gave up
// (a tag never came).  Up to eight granules per lane in flight; a pass that finds a foreign tag is repeated after a
// short sleep.
template <typename F>
__device__ __forceinline__ bool sweep_granules(const unsigned long long* g, unsigned count, unsigned epoch, F&& use) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned long long t0 = 0;
  for (unsigned i0 = 0; i0 < count; i0 += 512u) {
    for (;;) {
      unsigned long long v[8];
      bool ok = true;
#pragma unroll
      for (unsigned u = 0; u < 8; u++) {
        v[u] = granule(epoch, 0u);
        if (i0 + 64u * u < count) {                  // (uniform: whole rows of 64 granules)
          const unsigned i = i0 + 64u * u + lane;
          v[u] = ld_agent(g + (i < count ? i : count - 1u));
        }
      }
#pragma unroll
      for (unsigned u = 0; u < 8; u++) ok = ok && (unsigned)(v[u] >> 32) == epoch;
      if (!__builtin_amdgcn_ballot_w64(!ok)) {
#pragma unroll
        for (unsigned u = 0; u < 8; u++)
          if (i0 + 64u * u + lane < count) use((unsigned)v[u], i0 + 64u * u + lane);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > ONE_SPIN_TICKS) return false;
    }
  }
  return true;
}

// max / min of one double per lane over the wave (DPP row shifts and broadcasts; the result is uniform)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_move(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_minmax(double v) {           // MAX: values >= 0 (identity 0); else identity = the lane's own value
#define DCTZ_MM_STEP(CTRL, RM) { const double s = dpp_move<CTRL, RM>(MAX ? 0.0 : v, v); v = MAX ? fmax(v, s) : fmin(v, s); }
  DCTZ_MM_STEP(0x111, 0xf) DCTZ_MM_STEP(0x112, 0xf) DCTZ_MM_STEP(0x114, 0xf) DCTZ_MM_STEP(0x118, 0xf)
  DCTZ_MM_STEP(0x142, 0xa) DCTZ_MM_STEP(0x143, 0xc)
#undef DCTZ_MM_STEP
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// (the same for one non-negative value of the element type per lane)
__device__ __forceinline__ double wave_max_pos(double v) { return wave_minmax<true>(v); }
__device__ __forceinline__ float wave_max_pos(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false)));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63); }
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// The scaling factor of util.c:29 / :43 and the FastDiv level from the board's answer (what k_stats_final_sf does for the
// chain, dctz_kernels_aux.hip): kk = 0: every element is zero (sf = 1), else pw[kk - 1].
struct OneSf { double sf; unsigned fast; };
__device__ __forceinline__ OneSf one_sf(const SfTable& t, unsigned kk, bool window_violated) {
  const __attribute__((address_space(4))) double* pw = (const __attribute__((address_space(4))) double*)(t.pw);
  OneSf r;
  const unsigned k = kk - 1u < (unsigned)t.nk ? kk - 1u : (unsigned)t.nk;
  r.sf = kk == 0u ? 1.0 : pw[k];
  const bool f64 = t.dtype == DCTZHIP_F64;
  r.fast = (t.fastdiv && (f64 ? exp_in(r.sf, -250, 250) : exp_in(r.sf, -30, 30))) ? 1u : 0u;
  if (r.fast && t.fastdiv >= 2 && !window_violated) r.fast = 2u;
  return r;
}
// (what a wave says about its own elements: their largest decade index + 1, 0 for "all zero"; bit 16: some element
// outside the exponent window in which FastDiv needs no per-element test)
__device__ __forceinline__ unsigned one_stat_word(const SfTable& t, double mx, double mn, unsigned below) {
  const bool f64 = t.dtype == DCTZHIP_F64;
  const bool inwin = f64 ? (exp_in(mn, -500, 500) && exp_in(mx, -500, 500)) : (exp_in(mn, -63, 63) && exp_in(mx, -63, 63));
  return (mx == 0.0 ? 0u : below + 1u) | (inwin ? 0u : 0x10000u);
}

// What the waves of a workgroup tell each other (and its first wave the board)
struct OneShared {
  unsigned word[OTW];              // statistics words of the waves' tiles
  unsigned gword[OTW];             // ... and their guesses of the ARRAY's decade (own tile and a sample)
  unsigned tot[OTW];               // "stored exactly" coefficients of the waves' tiles
  double mx[OTW], mn[OTW], sum[OTW];
  unsigned res_stats, res_prefix;  // the sweeps' answers (ONE_GAVE_UP: timed out)
  unsigned long long q0;           // bits of the last block's DC (qtable[0], :355-360)
  unsigned long long qm[OTW][64];  // QT: the waves' per-position maxima (raw bits)
};

// QT: position j's maximum over the table's shards (raw bits of T: positive values order like their bits)
template <typename T>
__device__ __forceinline__ unsigned long long one_qt_read(const OneFwd<T>& a, unsigned j) {
  unsigned long long m = 0ull;
  for (unsigned sd = 0; sd < a.qt_shards; sd++) {
    const unsigned long long v = ld_agent(a.qt + (size_t)(sd * 64u + j) * a.qt_stride);
    m = v > m ? v : m;
  }
  return m;
}

// Hand-off of the call's results by the first wave of the launch's LAST workgroup (it has seen every other workgroup's
// count granule, and every workgroup drained its record and its table atomics in front of that granule).
template <typename T, int MODE>
__device__ __forceinline__ void one_handoff_compress(const OneFwd<T>& a, unsigned cnt_total, unsigned error, double sf, unsigned fast,
                                                     unsigned long long q0bits) {
  const unsigned lane = threadIdx.x & 63u;
  double dmx = 0.0, dmn = 1.79769313486231570815e308, sum = 0.0;
  if (error == 0u)
    for (unsigned i = lane; i < a.b.nwg; i += 64u) {
      dmx = fmax(dmx, ld_agent_f64(a.b.rec + 3 * (size_t)i));
      dmn = fmin(dmn, ld_agent_f64(a.b.rec + 3 * (size_t)i + 1));
      sum += ld_agent_f64(a.b.rec + 3 * (size_t)i + 2);
    }
  dmx = wave_minmax<true>(dmx);
  dmn = wave_minmax<false>(dmn);
  sum = wave_sum_f64(sum);
  // the other table of maxima is the next call's: all-zero when that call starts (kernel boundary)
  unsigned long long qr = 0ull;
  if (MODE == DCTZHIP_QT) {
    for (unsigned sd = 0; sd < a.qt_shards; sd++) a.qt_next[(size_t)(sd * 64u + lane) * a.qt_stride] = 0ull;
    if (error == 0u) qr = one_qt_read(a, lane);
  }
  if (a.bres != nullptr) {                           // one array of a batch: its entry of the result table, the tag last
    if (MODE == DCTZHIP_QT) a.bresq->qraw[lane] = qr;
    if (lane == 0) {
      BatchResC* r = a.bres;
      r->sf_used = sf; r->stats[0] = dmx; r->stats[1] = dmn; r->stats[2] = sum;
      r->cnt = cnt_total; r->error = error; r->fast_used = fast; r->q0 = q0bits;
    }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&a.bres->pad, a.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  HostBox* box = a.box;
  box->qraw[lane] = qr;
  if (lane == 0) {
    box->fstats[0] = dmx; box->fstats[1] = dmn; box->fstats[2] = sum;
    box->cnt_total = cnt_total; box->error = error; box->q0 = q0bits;
    box->sf_used = sf; box->fast_used = fast;
  }
  __threadfence_system();
  if (lane == 0) box_publish(&box->seq_done, a.seq);
}

// ================================================================= compress ==
template <typename T, int MODE, bool SC>
__device__ __forceinline__ void compress_one_body(const OneFwd<T>& a, const unsigned wg) {
  using G = Geo<T, 1>;
  using Bits = typename Traits<T>::Bits;
  constexpr bool F64 = sizeof(T) == 8;
  constexpr int NT = F64 ? 10 : 2;                   // decade thresholds per lane (host: nk <= 64 NT)
  // per wave ONE array: the tile's image (DMA target), then -- SC -- the scaled tile on its way out, then the tile's bin ids
  // on their way out, then the tile's exact coefficients in the reference's order (fp32: two workgroups per CU, 64 KiB each);
  // QT: the clamped table
  __shared__ __attribute__((aligned(1024))) unsigned char tile_all[OTW][G::TILEB];
  __shared__ __attribute__((aligned(16))) T qt_all[MODE == DCTZHIP_QT ? OTW : 1][MODE == DCTZHIP_QT ? 64 : 1];
  __shared__ OneShared sh;
  static_assert(G::TILEB >= (int)(ONE_EXC_MAX + 64u) * 4, "a dense tile's coefficients and the dump slots fit the image");
  const FwdParams<T>& p = a.p;
  const unsigned epoch = a.b.epoch, nwg = a.b.nwg;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned char* const tilebuf = tile_all[wv];
  T* const qt_lds = qt_all[MODE == DCTZHIP_QT ? wv : 0];
  const unsigned ntw = (p.ntiles + (unsigned)OTW - 1u) / (unsigned)OTW;    // workgroups that hold tiles; workgroup ntw: the short last block
  const bool rem_wg = wg == ntw;
  const unsigned tile = wg * (unsigned)OTW + (unsigned)wv;
  const bool tile_wave = !rem_wg && tile < p.ntiles;
  const bool rem_wave = rem_wg && wv == 0;
  const bool last_wg = wg == nwg - 1u;
  const int l = (int)a.rem;
  const T rmin = p.range_min, rmax = p.range_max;
  const size_t first_el = (size_t)tile * TILE_ELEMS;
  const unsigned blks_here = tile_wave ? min((unsigned)TILE_BLKS, p.nfull - tile * (unsigned)TILE_BLKS) : 0u;
  const bool active = (unsigned)lane < blks_here;
  const int range_el = (int)(blks_here * 64u);
  const size_t rbase = (size_t)p.nfull * 64;
  TileMap<T, 1> tm;
  tm.init(lane);
  FastDiv<T> bwd;
  bwd.init(p.bin_width, (p.fast_bw & 1u) != 0);

  // Scaling on a guess of the array's decade (below) pays where the first sweep is long: many workgroups (C2: 397; with the
  // 65 of C1 the sweep ends 1.3 us after the last post, and the sample costs as much).  Never with SC: x / sf goes to the
  // caller's memory before the transform.
  const bool speculate = !SC && (nwg >= (unsigned)ONE_SPEC_MIN_WG || (a.bad_guess & 15u) != 0u);

  // ---- phase 1: the data, and calc_data_stat over it (util.c:18-25) ---------------------------------------------------
  one_stamp(a.b, 0);
  T x[64];
  T raw = T(0);
  double mx = 0.0, mn = 1.79769313486231570815e308, rsum = 0.0;
  unsigned word = 0, gword = 0;
  if (tile_wave) {
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.x + first_el), 0, range_el * (int)sizeof(T), 0x00020000);
    issue_phase_dma<T, 1>(r_in, 0u, 0, tilebuf, tm);
    // (under the DMA: the decade thresholds this lane compares the tile's maximum with)
    double thr[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) thr[i] = (lane + 64 * i < a.sft.nk) ? a.sft.thr[lane + 64 * i] : __builtin_inf();
    // (and this wave's quarter of the sample behind the guess of the array's decade: 64 whole blocks spread evenly over the
    // array, the same for every workgroup -- coalesced rows that the caches serve after the first reader)
    constexpr int NS = 16;
    T smp[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) smp[i] = T(0);
    if (speculate) {
#pragma unroll
      for (int i = 0; i < NS; i++) smp[i] = p.x[(((size_t)(wv * NS + i) * p.nfull) >> 6) * 64 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_phase<T, 1, 0>(x, tilebuf, tm);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    StatAcc<T> acc;
    acc.init();
    if (active) {
#pragma unroll
      for (int j = 0; j < 64; j++) acc.minmax(x[j]);
    }
    // (the sum of util.c:18-28 -- never x[0] -- over the raw values, here and now: everything the hand-off reports about
    // the statistics is on the board before the first sweep, and nothing has to be drained in front of the counts later)
    {
      double s0 = 0.0, s1 = 0.0;
      if (active) {
#pragma unroll
        for (int j = 0; j < 64; j += 2) { s0 += (double)x[j]; s1 += (double)x[j + 1]; }
        if (tile == 0 && lane == 0) s0 -= (double)x[0];
      }
      rsum = wave_sum_f64(s0 + s1);
    }
    mx = wave_minmax<true>((double)acc.mx);
    mn = wave_minmax<false>((double)acc.mn);
    unsigned below = 0;
#pragma unroll
    for (int i = 0; i < NT; i++) below += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(thr[i] < mx));
    word = one_stat_word(a.sft, mx, mn, below);
    {
      StatAcc<T> sacc;
      sacc.init();
#pragma unroll
      for (int i = 0; i < NS; i++) sacc.minmax(smp[i]);
      const double gmx = fmax(mx, wave_minmax<true>((double)sacc.mx));
      unsigned gb = 0;
#pragma unroll
      for (int i = 0; i < NT; i++) gb += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(thr[i] < gmx));
      gword = gmx == 0.0 ? 0u : gb + 1u;
    }
  } else if (rem_wave) {
    raw = lane < l ? p.x[rbase + lane] : T(0);
    StatAcc<T> acc;
    acc.init();
    if (lane < l) acc.add(raw, rbase + lane != 0);   // util.c:22 starts at i = 1
    mx = wave_minmax<true>((double)acc.mx);
    mn = wave_minmax<false>((double)acc.mn);
    rsum = wave_sum_f64(acc.sum);
    unsigned below = 0;
    for (int i0 = 0; i0 < a.sft.nk; i0 += 64) below += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(i0 + lane < a.sft.nk && a.sft.thr[i0 + lane] < mx));
    word = one_stat_word(a.sft, mx, mn, below);
  }
  if (lane == 0) { sh.word[wv] = word; sh.gword[wv] = gword; sh.mx[wv] = mx; sh.mn[wv] = mn; sh.sum[wv] = rsum; }
  one_stamp(a.b, 1);
  __syncthreads();
  one_stamp(a.b, 2);
  if (wv == 0) {
    // the workgroup's statistics -> its record and its granule
    unsigned kw = 0, vw = 0;
    double wmx = 0.0, wmn = 1.79769313486231570815e308, wsum = 0.0;
#pragma unroll
    for (int i = 0; i < OTW; i++) { kw = max(kw, sh.word[i] & 0xFFFFu); vw |= sh.word[i] >> 16; wmx = fmax(wmx, sh.mx[i]); wmn = fmin(wmn, sh.mn[i]); wsum += sh.sum[i]; }
    if (lane == 0) {
      st_agent_f64(a.b.rec + 3 * (size_t)wg, wmx);
      st_agent_f64(a.b.rec + 3 * (size_t)wg + 1, wmn);
      st_agent_f64(a.b.rec + 3 * (size_t)wg + 2, wsum);
      // (bad_guess & 16, tests: workgroup 0 withholds its granule -- what a workgroup that is not resident does to the others)
      if (!((a.bad_guess & 16u) && wg == 0u)) st_agent(a.b.ga + wg, granule(epoch, kw | (vw << 16)));
    }
  }
  // The largest decade of the whole array -- all that the scaling factor depends on (util.c:29) -- is what the FIRST SWEEP
  // of the board brings back; it is complete only when the slowest workgroup has posted, ~4 us after this one is ready to
  // scale.  So the tile is scaled, transformed and binned on a GUESS -- the largest decade among the workgroup's own
  // tiles and a sample of the array (64 blocks, the same for every workgroup: cache hits) -- and the sweep, which by
  // then finds every granule in place, only verifies it; a wrong guess (a spike the sample did not see) runs the tile
  // again from its image in LDS, which nothing has touched yet.  The outputs never depend on the guess.
  // (The short last block is cheap: its workgroup waits.  bad_guess: 1 = every first guess wrong, 3 = guess whatever the
  // grid; tests.)
  auto sweep_stats = [&]() {                         // wave 0; the other waves meet it at the barrier behind
    unsigned kmax = 0, viol = 0;
    const bool ok = sweep_granules(a.b.ga, nwg, epoch, [&](unsigned v, unsigned) { kmax = max(kmax, v & 0xFFFFu); viol |= v >> 16; });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the record's stores are done as well: they went out in front of the sweep's loads)
    kmax = wave_max_u32(kmax);
    const bool vany = __builtin_amdgcn_ballot_w64(viol != 0u) != 0ull;
    if (lane == 0) sh.res_stats = ok ? (kmax | (vany ? 0x10000u : 0u)) : ONE_GAVE_UP;
  };
  auto gave_up = [&]() {                             // (every wave of the workgroup leaves)
    if (threadIdx.x == 0) atomicExch(&p.ctl->error, ONE_ERR_TIMEOUT);
    if (last_wg && wv == 0) one_handoff_compress<T, MODE>(a, 0u, ONE_ERR_TIMEOUT, 1.0, 0u, 0ull);
  };
  unsigned kk_use = 0;
  bool verified = false, replay = false;
  unsigned rs = 0;                                   // the sweep's answer: the array's decade word
  if (!speculate || rem_wg || (a.bad_guess & 16u)) {
    if (wv == 0) sweep_stats();
    one_stamp(a.b, 3);
    __syncthreads();
    rs = sh.res_stats;
    if (rs == ONE_GAVE_UP) { gave_up(); return; }
    kk_use = rs & 0xFFFFu;
    verified = true;
  } else {
#pragma unroll
    for (int i = 0; i < OTW; i++) kk_use = max(kk_use, sh.gword[i]);
    if ((a.bad_guess & 15u) == 1u) kk_use += 1u;
  }
  const bool own_inwin = (word >> 16) == 0u;         // FastDiv needs no per-element test for THIS tile (the same quotients either way)

  // ---- phase 2: scale, transform, bin -----------------------------------------------------------------------------------
  unsigned w[16];
  unsigned mlo = 0, mhi = 0;                         // tile: bit j: coefficient j of this block is stored exactly
  unsigned tot = 0, base = 0;                        // the tile's count, this block's place in the tile's piece of AC_exact
  T rcoef = T(0);                                    // remainder block: this lane's coefficient
  bool rexc = false;
  unsigned rbin = 0, rrank = 0;
  unsigned long long qmbits = 0ull;                  // QT: position `lane`'s maximum |coef| over this wave's out-of-range coefficients
  OneSf osf = {1.0, 0u};
  T sf = T(1);
#pragma clang loop unroll(disable)
  for (;;) {
    osf = one_sf(a.sft, kk_use, !own_inwin);
    sf = (T)osf.sf;
    const bool scale = (sf != T(1));                 // dctz-comp-lib.c:193 / :208
    if (tile_wave) {
      if (replay) {
        read_phase<T, 1, 0>(x, tilebuf, tm);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mlo = 0; mhi = 0;
      }
      if (scale) {
        FastDiv<T> sfd;
        sfd.init(sf, osf.fast != 0u);
        if (osf.fast == 2u) {
          if constexpr (!F64) {
  #pragma unroll
            for (int j = 0; j < 64; j += 2) {
              const f32x2 v = fastdiv_core2(sfd, f32x2{(float)x[j], (float)x[j + 1]});
              x[j] = v.x; x[j + 1] = v.y;
            }
          } else {
  #pragma unroll
            for (int j = 0; j < 64; j++) x[j] = sfd.core(x[j]);
          }
        } else if (osf.fast == 1u) {
  #pragma unroll
          for (int j = 0; j < 64; j++) x[j] = sfd.div(x[j]);
        } else {
  #pragma unroll
          for (int j = 0; j < 64; j++) x[j] = x[j] / sfd.d;
        }
      }
      if (SC && p.scaled != nullptr) {               // (an array of a batch may not ask for it)
        // the reference's in-place x / sf of the caller's array (:193-216), into p.scaled (which may be the input itself: the
        // tile is in registers): registers -> the image -> 1 KiB rows, as k_decompress writes its output
        const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc(p.scaled + first_el, 0, range_el * (int)sizeof(T), 0x00020000);
        write_phase<T, 1, 0>(x, tilebuf, tm);
  #pragma unroll
        for (int jg = 0; jg < 8; jg++) {
          const int vo = jg * 8 * G::BLKB + tm.g_of(jg);
  #pragma unroll
          for (int sg = 0; sg < G::SEGP; sg++) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(tilebuf + (jg * G::SEGP + sg) * 1024 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(r, r_sc, vo + sg * 128, 0, 2 /* nt */);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      block_fwd<T, CTab<T>, GEOM_1D, true>(x, as_ctab<T>(p.tab));
      if (p.coef != nullptr && active) {               // test tap: the coefficients as computed
  #pragma unroll
        for (int j = 0; j < 64; j++) p.coef[((size_t)tile * TILE_BLKS + lane) * 64 + j] = x[j];
      }
      if (p.last_is_full && tile == p.ntiles - 1u) {   // :355-360
        const T dc_last = (T)__shfl(x[0], (int)((p.nfull - 1u) & 63u));
        if (lane == 0) sh.q0 = (unsigned long long)to_bits(dc_last);
      }
      // pass-1 binning (:363-414), four coefficients = one dword of bin ids at a time
      __builtin_amdgcn_sched_barrier(0);
      auto bin_tile = [&](auto fast, auto safe) {
  #pragma unroll
        for (int g = 0; g < 16; g++) {
          float h[4];
          if constexpr (!F64 && decltype(fast)::value) {
  #pragma unroll
            for (int i = 0; i < 4; i += 2) {
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
          asm volatile("" : "+v"(wgd));
          w[g] = wgd;
          const unsigned nw = ~wgd;                    // "stored exactly" = id 255: bit 7 of byte i of mm
          const unsigned mm = ~(((nw & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nw) & 0x80808080u;
          const unsigned m4 = ((mm >> 7) | (mm >> 14) | (mm >> 21) | (mm >> 28)) & 0xFu;
          if (g < 8) mlo |= m4 << (4 * g); else mhi |= m4 << (4 * (g - 8));
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (bwd.ok) { if (p.fast_bw & 2u) bin_tile(std::true_type{}, std::false_type{}); else bin_tile(std::true_type{}, std::true_type{}); }
      else bin_tile(std::false_type{}, std::true_type{});
      w[0] |= 0xFFu;                                   // :361 DC slot
      if (!active) { mlo = 0; mhi = 0; }
      const unsigned n = (unsigned)(__popc(mlo) + __popc(mhi));
      const unsigned incl = wave_incl_scan(n);
      tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      base = incl - n;
    } else if (rem_wave) {
      // the short last block (length l = N % 64): the reference re-plans a length-l (l even) or 2l (l odd) FFT for it
      // (dctz-comp-lib.c:326-340, dct.c:59-72); definition-order DFT with host-built roots, lane k = output k
      T* const v = reinterpret_cast<T*>(tilebuf);
      const T* rt = p.rtab;
      const int N = (l & 1) ? 2 * l : l;
      const int k = lane;
      FastDiv<T> sfd;
      sfd.init(sf, osf.fast != 0u);
      if (k < l) {
        T e = raw;
        if (scale) e = sfd.div(e);
        if (p.scaled != nullptr) p.scaled[rbase + k] = e;
        if (l & 1) { v[k] = e; v[l + (l - 1 - k)] = e; }             // dct.c:61-64
        else if (k & 1) v[l - 1 - (k >> 1)] = e;                     // dct.c:75-83
        else v[k >> 1] = e;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (one wave: its LDS operations are in order)
      if (k < l) {
        T sr = T(0), si = T(0);
        for (int j = 0; j < N; j++) {
          const int tt = (j * k) % N;
          sr = sr + v[j] * rt[RTAB_WR + tt];
          si = si + v[j] * rt[RTAB_WI + tt];
        }
        rcoef = rt[RTAB_AS + k] * sr + rt[RTAB_AX + k] * si;         // dct.c:100-102 (Im V = -si)
      }
      // pass-1 binning, the reference's own form (:363-414)
      const bool out = fabs(rcoef) > rmax;                           // == (item < range_min || item > range_max)
      const T u = rcoef - rmin;
      const T q = bwd.ok ? bwd.core(u) : u / bwd.d;
      const int t = (int)q;                                          // (t_bin_id) cast: truncation
      rbin = out ? 255u : (unsigned)(t <= 127 ? 254 - 2 * t : 2 * t - 255);   // conv_tbl :27-43 (t == 255 -> 255)
      if (k == 0) rbin = 255u; else rexc = (rbin == 255u);
      if (k >= l) rexc = false;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(rexc);
      rrank = (unsigned)__popcll(m & ((1ull << k) - 1ull));
      tot = (unsigned)__popcll(m);
      if (MODE == DCTZHIP_QT) qmbits = (rexc && fabs(rcoef) > rmax) ? (unsigned long long)to_bits(fabs(rcoef)) : 0ull;   // :371-372 / :396-397
      if (k == 0) sh.q0 = (unsigned long long)to_bits(rcoef);        // :355-360
    }
    if (verified) break;
    if (wv == 0) sweep_stats();
    one_stamp(a.b, 3);
    __syncthreads();
    rs = sh.res_stats;
    if (rs == ONE_GAVE_UP) { gave_up(); return; }
    verified = true;
    if ((rs & 0xFFFFu) == kk_use) break;
    kk_use = rs & 0xFFFFu;                           // (uniform over the workgroup: every wave runs its tile again)
    replay = true;
  }
  if (tile_wave) {
    if (MODE == DCTZHIP_QT) {
      // per-position maximum |coef| over the out-of-range coefficients (:371-372 / :396-397): one wave reduction per position
    // that has one somewhere in the tile (LDS atomics of 64 lanes on ONE word ran the dense C1 tile at 23 us for this step),
    // lane j keeps position j's; the workgroup's waves merge theirs behind the barrier below
    T qmv = T(0);
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const unsigned mg = ((g < 8 ? mlo >> (4 * g) : mhi >> (4 * (g - 8)))) & 0xFu;
      if (__builtin_amdgcn_ballot_w64(mg != 0u)) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const T av = fabs(x[4 * g + i]);
          const T mxv = wave_max_pos((((mg >> i) & 1u) && av > rmax) ? av : T(0));
          if (lane == 4 * g + i) qmv = mxv;
        }
      }
    }
      qmbits = (unsigned long long)to_bits(qmv);
    }
  }
  if (MODE == DCTZHIP_QT) sh.qm[wv][lane] = qmbits;
  if (lane == 0) sh.tot[wv] = tot;
  one_stamp(a.b, 4);
  __syncthreads();
  one_stamp(a.b, 5);
  unsigned wg_tot = 0;
#pragma unroll
  for (int i = 0; i < OTW; i++) wg_tot += sh.tot[i];
  if (wv == 0) {
    if (MODE == DCTZHIP_QT) {
      // the workgroup's maxima -> the array's table: one device atomic per position that has one, drained before the count
      // is posted (the table is read behind ALL counts)
      unsigned long long m = 0ull;
#pragma unroll
      for (int i = 0; i < OTW; i++) m = sh.qm[i][lane] > m ? sh.qm[i][lane] : m;
      if (m != 0ull) atomicMax(a.qt + (size_t)((wg % a.qt_shards) * 64u + (unsigned)lane) * a.qt_stride, m);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (the record was drained behind the first sweep)
    if (lane == 0) st_agent(a.b.gb + wg, granule(epoch, wg_tot));
  }

  // ---- phase 3a: what does not need the place in AC_exact ------------------------------------------------------------
  const unsigned exc_at = lds_offset(tilebuf);
  // every coefficient of the lane's block in groups that hold a flagged one somewhere in the wave, in j order: f(j, flagged)
  auto for_flagged = [&](auto&& f) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const unsigned mg = ((g < 8 ? mlo >> (4 * g) : mhi >> (4 * (g - 8)))) & 0xFu;
      if (__builtin_amdgcn_ballot_w64(mg != 0u)) {
#pragma unroll
        for (int i = 0; i < 4; i++) f(4 * g + i, ((mg >> i) & 1u) != 0u);
      }
    }
  };
  // (the lane's number through a register the compiler cannot see through, for everything behind the transform: the
  // per-lane addresses made of it were kept alive across the transform otherwise -- in the fp32 EC kernel, at its 256
  // registers, in scratch memory: 22 spilled registers stored in front of the transform and read back behind it)
  const int lane_out = lane;
  auto lane_again = [&]() { int l = lane_out; asm volatile("" : "+v"(l)); return l; };
  if (tile_wave) {
    // bin ids: 64 bytes per lane -> (the image) -> 1 KiB rows of 16 consecutive blocks; DC (:350-351 USE_TRUNCATE)
    const int lane = lane_again();
    const int f2 = (lane >> 1) & 3;
#pragma unroll
    for (int i = 0; i < 4; i++)
      lds_store_b128(exc_at + (unsigned)((lane * 4 + (i ^ f2)) * 16), u32x4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]});
    const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(p.bin + first_el, 0, range_el, 0x00020000);
    const int bin_goff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) * 16);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(tilebuf + i * 1024 + lane * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, r_bin, bin_goff + i * 1024, 0, 0);
    }
    if (active) p.dc[(size_t)tile * TILE_BLKS + lane] = (float)x[0];
    if (MODE == DCTZHIP_EC) {
      // EC: the coefficients stored exactly (:535-537 USE_TRUNCATE), in the reference's order, into the image (the bin ids
      // have been read out of it: LDS operations of a wave execute in order)
      unsigned pos = base;
      for_flagged([&](int j, bool f) {
        const unsigned at = f ? pos : ONE_EXC_MAX + (unsigned)lane;
        lds_store_b32(exc_at + at * 4u, __builtin_bit_cast(unsigned, (float)x[j]));
        pos += f ? 1u : 0u;
      });
    }
  } else if (rem_wave) {
    if (lane < l) {
      p.bin[rbase + lane] = (uint8_t)rbin;
      if (p.coef != nullptr) p.coef[rbase + lane] = rcoef;
      if (lane == 0) p.dc[p.nfull] = (float)rcoef;
    }
  }
  one_stamp(a.b, 6);
  if (wv == 0) {
    // the running tot_AC_exact_count in front of this workgroup (:478-544); QT: every count (the table is final then)
    unsigned before = 0;
    const bool ok = sweep_granules(a.b.gb, MODE == DCTZHIP_QT ? nwg : wg, epoch, [&](unsigned v, unsigned i) { before += i < wg ? v : 0u; });
    before = wave_sum_u32(before);
    if (lane == 0) sh.res_prefix = ok ? before : ONE_GAVE_UP;
  }
  one_stamp(a.b, 7);
  __syncthreads();
  one_stamp(a.b, 8);
  const unsigned pre = sh.res_prefix;
  if (pre == ONE_GAVE_UP) {
    if (threadIdx.x == 0) atomicExch(&p.ctl->error, ONE_ERR_TIMEOUT);
    if (last_wg && wv == 0) one_handoff_compress<T, MODE>(a, 0u, ONE_ERR_TIMEOUT, 1.0, 0u, 0ull);
    return;
  }
  // ---- phase 3b: the place is known ----------------------------------------------------------------------------------------
  unsigned E = pre;                                  // this wave's first coefficient in AC_exact[]
#pragma unroll
  for (int i = 0; i < OTW; i++) E += i < wv ? sh.tot[i] : 0u;
  // (the hand-off certifies the call: a workgroup whose sweep gave up -- it leaves its tiles unwritten -- has said so in the
  // control block before the granule it missed can have reached this one)
  if (last_wg && wv == 0) one_handoff_compress<T, MODE>(a, pre + wg_tot, __hip_atomic_load(&p.ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), osf.sf,
                                                         one_sf(a.sft, rs & 0xFFFFu, (rs >> 16) != 0u).fast, sh.q0);
  if (tile_wave) {
    const int lane = lane_again();
    if (MODE == DCTZHIP_QT) {
      // the table is final: clamp (:450-461), normalise this tile's coefficients (:488-518) on their way into the image
      T qv = Traits<T>::from_bits((Bits)one_qt_read(a, (unsigned)lane));
      if (qv < T(1)) qv = T(1);
      qt_lds[lane] = qv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned pos = base;
      for_flagged([&](int j, bool f) {
        const unsigned at = f ? pos : ONE_EXC_MAX + (unsigned)lane;
        const T item = qt_normalise(x[j], qt_lds[j], a.eb, T(10), rmin, rmax);
        lds_store_b32(exc_at + at * 4u, __builtin_bit_cast(unsigned, (float)item));               // :496-497
        pos += f ? 1u : 0u;
      });
    }
    // the tile's piece of AC_exact[], whole rows of 64 coefficients at their final place
    const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(p.ac + E, 0, (int)(tot * 4u), 0x00020000);
    const unsigned* staged = reinterpret_cast<const unsigned*>(tilebuf);
    for (unsigned r = 0; r * 64u < tot; r += 4) {
      unsigned v[4];
#pragma unroll
      for (unsigned u = 0; u < 4; u++) v[u] = staged[min((r + u) * 64u + (unsigned)lane, ONE_EXC_MAX + 63u)];
#pragma unroll
      for (unsigned u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b32(v[u], r_ac, (int)(((r + u) * 64u + (unsigned)lane) * 4u), 0, 0);
    }
  } else if (rem_wave) {
    if (rexc) {
      T item = rcoef;
      if (MODE == DCTZHIP_QT) {
        T qv = Traits<T>::from_bits((Bits)one_qt_read(a, (unsigned)lane));
        if (qv < T(1)) qv = T(1);                                  // :450-461
        item = qt_normalise(item, qv, a.eb, T(10), rmin, rmax);    // :488-518
      }
      p.ac[(size_t)E + rrank] = (float)item;                       // :496-497 / :535-537
    }
  }
  one_stamp(a.b, 9);
}

template <typename T, int MODE, bool SC>
__global__ __launch_bounds__(OTW * 64) __attribute__((amdgpu_waves_per_eu(one_waves<T>(), one_waves<T>()))) void k_compress_one(const OneFwd<T> a) {
  compress_one_body<T, MODE, SC>(a, blockIdx.x);
}
// a batch: the workgroup's record says which array it works for and as which of that array's workgroups
template <typename T, int MODE, bool SC>
__global__ __launch_bounds__(OTW * 64) __attribute__((amdgpu_waves_per_eu(one_waves<T>(), one_waves<T>()))) void k_compress_one_batch(const OneBatchC<T> cm) {
  const OneRecC r = one_load_rec(&cm.recs[blockIdx.x]);
  OneFwd<T> a;
  FwdParams<T>& p = a.p;
  p.x = (const T*)r.x; p.bin = (uint8_t*)r.bin; p.dc = r.dc; p.ac = r.ac; p.coef = nullptr; p.scaled = (T*)r.scaled;
  p.tab = cm.tab; p.rtab = (const T*)r.rtab; p.ctl = cm.ctl + r.item;
  p.nfull = r.nfull; p.ntiles = r.ntiles; p.last_is_full = r.rem ? 0u : 1u; p.fast_bw = r.fast_bw;
  p.bin_width = (T)r.bin_width; p.range_min = (T)r.range_min; p.range_max = (T)r.range_max;
  a.b.ga = cm.b.ga + r.board_base; a.b.gb = cm.b.gb + r.board_base; a.b.rec = cm.b.rec + 3 * (size_t)r.board_base;
  a.b.epoch = cm.b.epoch; a.b.nwg = r.nwg; a.b.dbg = cm.b.dbg;
  a.sft = cm.sft;
  a.box = nullptr; a.seq = 0;
  a.qt = cm.qt + (size_t)r.item * 64; a.qt_next = cm.qt_next + (size_t)r.item * 64; a.qt_stride = 1u; a.qt_shards = 1u;
  a.bres = cm.res + r.item; a.bresq = cm.resq ? cm.resq + r.item : nullptr; a.tag = cm.tag;
  a.eb = r.eb; a.rem = r.rem; a.bad_guess = cm.bad_guess;
  compress_one_body<T, MODE, SC>(a, r.wg_local);
}

// =============================================================== decompress ==
template <typename T>
__device__ __forceinline__ void one_handoff_decompress(const OneInv<T>& a, unsigned total, unsigned error) {
  const unsigned lane = threadIdx.x & 63u;
  if (a.bres != nullptr) {                           // one array of a batch
    if (lane == 0) { a.bres->total = total; a.bres->error = error; }
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&a.bres->tag, a.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  HostBox* box = a.box;
  if (lane == 0) { box->cnt_total = total; box->error = error; }
  __threadfence_system();
  if (lane == 0) box_publish(&box->seq_done, a.seq);
}

template <typename T, int MODE>
__device__ __forceinline__ void decompress_one_body(const OneInv<T>& a, const unsigned wg, const T* qtab_src) {
  using G = Geo<T, 1>;
  // per wave ONE array: the tile's exact coefficients (up to 4032 floats, staged by LDS-DMA), then the output image
  __shared__ __attribute__((aligned(1024))) unsigned char io_all[OTW][G::TILEB];
  constexpr bool BC_ARITH = sizeof(T) == 8;          // (bin centres: computed in the fp64 kernel, looked up in the fp32 one: dctz_kernels.hip)
  __shared__ __attribute__((aligned(16))) T bctab[BC_ARITH ? 1 : 256];
  __shared__ T qt[64];
  __shared__ unsigned sh_tot[OTW], sh_prefix;
  const InvParams<T>& p = a.p;
  const unsigned epoch = a.b.epoch, nwg = a.b.nwg;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned char* const io = io_all[wv];
  const unsigned ntw = (p.ntiles + (unsigned)OTW - 1u) / (unsigned)OTW;
  const bool rem_wg = wg == ntw;
  const unsigned tile = wg * (unsigned)OTW + (unsigned)wv;
  const bool tile_wave = !rem_wg && tile < p.ntiles;
  const bool rem_wave = rem_wg && wv == 0;
  const bool last_wg = wg == nwg - 1u;
  const int l = (int)a.rem;
  const unsigned blks_here = tile_wave ? min((unsigned)TILE_BLKS, p.nfull - tile * (unsigned)TILE_BLKS) : 0u;
  const bool active = (unsigned)lane < blks_here;
  const size_t first_el = (size_t)tile * TILE_ELEMS;
  const int range_el = (int)(blks_here * 64u);
  const size_t rbase = (size_t)p.nfull * 64;
  if (!BC_ARITH)
    for (int b = threadIdx.x; b < 256; b += OTW * 64) {   // gen_bins / gen_bins_f (binning.c:17-23 / :37-43)
      const int ti = (b & 1) ? (b >> 1) + 1 : -(b >> 1);
      bctab[b] = (T)ti * p.bin_width;
    }
  if (MODE == DCTZHIP_QT && threadIdx.x < 64) qt[threadIdx.x] = qtab_src[threadIdx.x];
  // ---- the flags of the tile (dctz-decomp-lib.c:400 / :446) -------------------------------------------------------------
  one_stamp(a.b, 0);
  // (four named registers, not an array of vectors: as an array the bin ids -- and the copy below -- stayed in scratch memory
  // in the fp32 EC kernel, 144 bytes per lane stored and read back per tile)
  u32x4 bw0 = u32x4{0u, 0u, 0u, 0u}, bw1 = bw0, bw2 = bw0, bw3 = bw0;
  float dc_t = 0.f;
  unsigned tot = 0, ptr = 0;
  unsigned rbin = 0, rrank = 0;
  bool rexc = false;
  if (tile_wave) {
    const __amdgpu_buffer_rsrc_t r_bin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.bin + first_el), 0, range_el, 0x00020000);
    bw0 = __builtin_amdgcn_raw_buffer_load_b128(r_bin, lane * 64, 0, 0);
    bw1 = __builtin_amdgcn_raw_buffer_load_b128(r_bin, lane * 64 + 16, 0, 0);
    bw2 = __builtin_amdgcn_raw_buffer_load_b128(r_bin, lane * 64 + 32, 0, 0);
    bw3 = __builtin_amdgcn_raw_buffer_load_b128(r_bin, lane * 64 + 48, 0, 0);
    dc_t = active ? p.dc[(size_t)tile * TILE_BLKS + lane] : 0.f;
    const unsigned w[16] = {bw0.x, bw0.y, bw0.z, bw0.w, bw1.x, bw1.y, bw1.z, bw1.w, bw2.x, bw2.y, bw2.z, bw2.w, bw3.x, bw3.y, bw3.z, bw3.w};
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
    const unsigned incl = wave_incl_scan(n);
    tot = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    ptr = incl - n;                                                    // index inside the tile's piece of AC_exact
  } else if (rem_wave) {
    if (lane < l) rbin = p.bin[rbase + lane];
    rexc = (lane < l) && (lane != 0) && (rbin == 255u);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(rexc);
    rrank = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    tot = (unsigned)__popcll(m);
  }
  if (lane == 0) sh_tot[wv] = tot;
  one_stamp(a.b, 1);
  __syncthreads();
  one_stamp(a.b, 2);
  unsigned wg_tot = 0;
#pragma unroll
  for (int i = 0; i < OTW; i++) wg_tot += sh_tot[i];
  if (wv == 0) {
    // the running pos of dctz-decomp-lib.c:402-412 at this workgroup: the counts of the workgroups in front of it
    if (lane == 0 && !(a.withhold && wg == 0u)) st_agent(a.b.gb + wg, granule(epoch, wg_tot));   // (withhold: tests, as in k_compress_one)
    unsigned before = 0;
    const bool ok = sweep_granules(a.b.gb, wg, epoch, [&](unsigned v, unsigned) { before += v; });
    before = wave_sum_u32(before);
    if (lane == 0) sh_prefix = (ok && !(a.withhold && wg == 0u)) ? before : ONE_GAVE_UP;
  }
  one_stamp(a.b, 3);
  __syncthreads();
  one_stamp(a.b, 4);
  const unsigned pre = sh_prefix;
  if (pre == ONE_GAVE_UP) {
    if (threadIdx.x == 0) atomicExch(&p.ctl->error, ONE_ERR_TIMEOUT);
    if (last_wg && wv == 0) one_handoff_decompress<T>(a, 0u, ONE_ERR_TIMEOUT);
    return;
  }
  unsigned S = pre;
#pragma unroll
  for (int i = 0; i < OTW; i++) S += i < wv ? sh_tot[i] : 0u;
  // does the stream promise more exact coefficients than the caller provides?  The one thing the host waits for.
  if (last_wg && wv == 0) {
    const unsigned err = __hip_atomic_load(&p.ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a workgroup that gave up: see k_compress_one)
    one_handoff_decompress<T>(a, pre + wg_tot, err == ONE_ERR_TIMEOUT ? err : (pre + wg_tot > p.ac_count ? 2u : 0u));
  }
  if (tile_wave) {
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out + first_el, 0, range_el * (int)sizeof(T), 0x00020000);
    TileMap<T, 1> tm;
    tm.init(lane);
    // the tile's exact coefficients AC_exact[S, S + tot) -> LDS (reads beyond ac_count return zeros: descriptor range)
    {
      const size_t ac_left = S < p.ac_count ? (size_t)(p.ac_count - S) * 4 : 0;
      const __amdgpu_buffer_rsrc_t r_ac = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ac + (ac_left ? S : 0u)), 0, (int)min(ac_left, (size_t)0x7ffffffc), 0x00020000);
      for (unsigned i = 0; i * 256u < tot; i++) DMA16(r_ac, io + i * 1024u, lane * 16, (int)(i * 1024u), 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float* const stage = reinterpret_cast<const float*>(io);
    constexpr unsigned stage_last = (unsigned)(G::TILEB / 4) - 1u;
    const unsigned w[16] = {bw0.x, bw0.y, bw0.z, bw0.w, bw1.x, bw1.y, bw1.z, bw1.w, bw2.x, bw2.y, bw2.z, bw2.w, bw3.x, bw3.y, bw3.z, bw3.w};
    T x[64];
    if constexpr (sizeof(T) == 8) {
      // four coefficients = one dword of bin ids at a time (dctz_kernels.hip: decompress_body)
#pragma unroll
      for (int g = 0; g < 16; g++) {
        const unsigned wgd = w[g];
        const unsigned nv = ~wgd;                                      // a zero byte of nv <=> bin id 255
        unsigned m = ~(((nv & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | nv) & 0x80808080u;
        if (g == 0) m &= ~0x80u;                                       // j = 0 is the DC slot (:392 / :438)
        const unsigned w1 = ((wgd >> 1) & 0x7F7F7F7Fu) + (wgd & 0x01010101u);   // four magnitudes (b + 1) >> 1
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (__builtin_amdgcn_ballot_w64(m != 0u)) {                    // :400 / :446 somewhere in the wave
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
          if (j == 0) { x[0] = (T)dc_t; continue; }                    // :392 / :438
          T v;
          if constexpr (BC_ARITH) v = bin_centre<T>(w1, nv, i, p.bin_width);
          else v = bctab[(wgd >> (8 * i)) & 255u];                     // :416 / :462
          if ((m >> (8 * i + 7)) & 1u) {
            v = (T)e[i];
            if (MODE == DCTZHIP_QT) v = qt_restore(v, qt[j], p.eb, T(10), p.range_min, p.range_max);
          }
          x[j] = v;
        }
      }
    } else {
      // position by position (two workgroups per CU hide the round trips; the grouped form measured slower for fp32), written
      // as sixteen dwords of four: as ONE loop of 63 trips the compiler leaves it partly rolled, and the bin ids and the
      // block then live in scratch memory -- 144 bytes per lane, stored and read back per tile: 15 MB of the 41 MB that
      // k_decompress_one<float> wrote for a 26 MB array (profiles/r05_c2_raw_traffic.txt)
#pragma unroll
      for (int g = 0; g < 16; g++) {
        const unsigned wgd = w[g];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int j = 4 * g + i;
          if (j == 0) { x[0] = (T)dc_t; continue; }                    // :392 / :438
          const unsigned b = (wgd >> (8 * i)) & 255u;
          T v = bctab[b];                                              // :416 / :462
          if (b == 255u) {                                             // :400 / :446
            const float e = stage[min(ptr, stage_last)];
            ptr++;
            v = (T)e;
            if (MODE == DCTZHIP_QT) v = qt_restore(v, qt[j], p.eb, T(10), p.range_min, p.range_max);
          }
          // (each value through a register of its own: paired into <2 x float> stores by the vectoriser, the first 33
          // elements of the block stayed an array in scratch memory -- 144 bytes per lane, stored and read back per tile:
          // 15 MB of the 41 MB that k_decompress_one<float> wrote for a 26 MB array, profiles/r05_c2_raw_traffic.txt)
          asm volatile("" : "+v"(v));
          x[j] = v;
        }
      }
    }
    one_stamp(a.b, 5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the staged coefficients are consumed: the array becomes the image
    block_inv<T, CTab<T>, GEOM_1D, true>(x, as_ctab<T>(p.tab));
    if (p.sf != T(1)) {                                                // dctz-decomp-lib.c:496 / :505
#pragma unroll
      for (int j = 0; j < 64; j++) x[j] = x[j] * p.sf;                 // :494-511
    }
    one_stamp(a.b, 6);
    write_phase<T, 1, 0>(x, io, tm);
#pragma unroll
    for (int jg = 0; jg < 8; jg++) {
      const int vo = jg * 8 * G::BLKB + tm.g_of(jg);
#pragma unroll
      for (int s = 0; s < G::SEGP; s++) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(io + (jg * G::SEGP + s) * 1024 + lane * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, r_out, vo + s * 128, 0, 2 /* nt */);
      }
    }
  } else if (rem_wave) {
    // last, short block on decode (dctz-decomp-lib.c:423-428, dct.c:144-199): one wave, its LDS operations are in order
    T* const av = reinterpret_cast<T*>(io);
    T* const cr = av + 64;
    T* const ci = cr + 128;
    const T* rt = p.rtab;
    const int N = (l & 1) ? 2 * l : l;
    const int k = lane;
    cr[k] = T(0); ci[k] = T(0); cr[k + 64] = T(0); ci[k + 64] = T(0);
    if (k < l) {
      T val;
      if (k == 0) val = (T)p.dc[p.nfull];
      else if (rexc) {
        T v = T(0);
        if (S + rrank < p.ac_count) v = (T)p.ac[(size_t)S + rrank];
        if (MODE == DCTZHIP_QT) v = qt_restore(v, qt[k], p.eb, T(10), p.range_min, p.range_max);
        val = v;
      } else {
        const int ti = (rbin & 1u) ? (int)(rbin >> 1) + 1 : -(int)(rbin >> 1);
        val = (T)ti * p.bin_width;
      }
      av[k] = val;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (k < l) {
      cr[k] = rt[RTAB_IAS + k] * av[k];                            // dct.c:146-151 / :166-172
      ci[k] = rt[RTAB_IAX + k] * av[k];
      if ((l & 1) && k >= 1) {                                     // dct.c:152-153
        cr[l + k] = rt[RTAB_IAX + k] * av[l - k];
        ci[l + k] = -(rt[RTAB_IAS + k] * av[l - k]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (k < l) {
      const int s = (l & 1) ? k : ((k & 1) ? l - 1 - (k >> 1) : (k >> 1));   // dct.c:189-199
      T accv = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (s * j) % N;
        accv = accv + (cr[j] * rt[RTAB_WR + tt] - ci[j] * rt[RTAB_WI + tt]);
      }
      T val = (l & 1) ? (accv / (T)l) / T(2) : accv / (T)l;        // dct.c:163 / :185
      if (p.sf != T(1)) val = val * p.sf;                          // :496 / :505
      p.out[rbase + k] = val;
    }
  }
  one_stamp(a.b, 9);
}

template <typename T, int MODE>
__global__ __launch_bounds__(OTW * 64) __attribute__((amdgpu_waves_per_eu(one_waves<T>(), one_waves<T>()))) void k_decompress_one(const OneInv<T> a) {
  decompress_one_body<T, MODE>(a, blockIdx.x, a.qtab);
}
template <typename T, int MODE>
__global__ __launch_bounds__(OTW * 64) __attribute__((amdgpu_waves_per_eu(one_waves<T>(), one_waves<T>()))) void k_decompress_one_batch(const OneBatchD<T> cm) {
  const OneRecD r = one_load_rec(&cm.recs[blockIdx.x]);
  OneInv<T> a;
  InvParams<T>& p = a.p;
  p.bin = (const uint8_t*)r.bin; p.dc = r.dc; p.ac = r.ac; p.out = (T*)r.out;
  p.tab = cm.tab; p.rtab = (const T*)r.rtab; p.qtab = nullptr; p.ctl = cm.ctl + r.item;
  p.nfull = r.nfull; p.ntiles = r.ntiles; p.ac_count = r.ac_count;
  p.sf = (T)r.sf; p.bin_width = (T)r.bin_width; p.range_min = (T)r.range_min; p.range_max = (T)r.range_max; p.eb = r.eb;
  a.b.ga = nullptr; a.b.gb = cm.b.gb + r.board_base; a.b.rec = nullptr; a.b.epoch = cm.b.epoch; a.b.nwg = r.nwg; a.b.dbg = cm.b.dbg;
  a.box = nullptr; a.seq = 0; a.rem = r.rem; a.tag = cm.tag; a.withhold = cm.pad;
  a.bres = cm.res + r.item;
  decompress_one_body<T, MODE>(a, r.wg_local, (const T*)r.qtab);
}

// ================================================================= launchers ==
template <typename T>
void launch_compress_one_batch(const OneBatchC<T>& cm, unsigned grid, int mode, bool scaled, hipStream_t s) {
  const dim3 g(grid), blk(OTW * 64);
  if (mode == DCTZHIP_EC) {
    if (scaled) hipLaunchKernelGGL((k_compress_one_batch<T, DCTZHIP_EC, true>), g, blk, 0, s, cm);
    else hipLaunchKernelGGL((k_compress_one_batch<T, DCTZHIP_EC, false>), g, blk, 0, s, cm);
  } else {
    if (scaled) hipLaunchKernelGGL((k_compress_one_batch<T, DCTZHIP_QT, true>), g, blk, 0, s, cm);
    else hipLaunchKernelGGL((k_compress_one_batch<T, DCTZHIP_QT, false>), g, blk, 0, s, cm);
  }
}
template <typename T>
void launch_decompress_one_batch(const OneBatchD<T>& cm, unsigned grid, int mode, hipStream_t s) {
  const dim3 g(grid), blk(OTW * 64);
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress_one_batch<T, DCTZHIP_EC>), g, blk, 0, s, cm);
  else hipLaunchKernelGGL((k_decompress_one_batch<T, DCTZHIP_QT>), g, blk, 0, s, cm);
}
template <typename T>
void launch_compress_one(const OneFwd<T>& a, int mode, bool scaled, hipStream_t s) {
  const dim3 grid(a.b.nwg), blk(OTW * 64);
  if (mode == DCTZHIP_EC) {
    if (scaled) hipLaunchKernelGGL((k_compress_one<T, DCTZHIP_EC, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_compress_one<T, DCTZHIP_EC, false>), grid, blk, 0, s, a);
  } else {
    if (scaled) hipLaunchKernelGGL((k_compress_one<T, DCTZHIP_QT, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_compress_one<T, DCTZHIP_QT, false>), grid, blk, 0, s, a);
  }
}
template <typename T>
void launch_decompress_one(const OneInv<T>& a, int mode, hipStream_t s) {
  const dim3 grid(a.b.nwg), blk(OTW * 64);
  if (mode == DCTZHIP_EC) hipLaunchKernelGGL((k_decompress_one<T, DCTZHIP_EC>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((k_decompress_one<T, DCTZHIP_QT>), grid, blk, 0, s, a);
}
template <typename T>
int compress_one_occupancy(int mode, bool scaled) {
  int n = 0;
  hipError_t e;
  if (mode == DCTZHIP_EC) e = scaled ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_one<T, DCTZHIP_EC, true>, OTW * 64, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_one<T, DCTZHIP_EC, false>, OTW * 64, 0);
  else e = scaled ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_one<T, DCTZHIP_QT, true>, OTW * 64, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_compress_one<T, DCTZHIP_QT, false>, OTW * 64, 0);
  return e == hipSuccess ? n : 0;
}
template <typename T>
int decompress_one_occupancy(int mode) {
  int n = 0;
  const hipError_t e = mode == DCTZHIP_EC ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_decompress_one<T, DCTZHIP_EC>, OTW * 64, 0)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k_decompress_one<T, DCTZHIP_QT>, OTW * 64, 0);
  return e == hipSuccess ? n : 0;
}

#define INST_ONE(T)                                                                   \
  template void launch_compress_one<T>(const OneFwd<T>&, int, bool, hipStream_t);     \
  template void launch_decompress_one<T>(const OneInv<T>&, int, hipStream_t);         \
  template void launch_compress_one_batch<T>(const OneBatchC<T>&, unsigned, int, bool, hipStream_t); \
  template void launch_decompress_one_batch<T>(const OneBatchD<T>&, unsigned, int, hipStream_t);     \
  template int compress_one_occupancy<T>(int, bool);                                  \
  template int decompress_one_occupancy<T>(int);
INST_ONE(double)
INST_ONE(float)

}  // namespace dctz
