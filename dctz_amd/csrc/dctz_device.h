// dctz_device.h -- types shared by the kernels (dctz_kernels.hip) and the C-ABI
// shim (dctz_shim.hip): per-dtype traits, the per-call control block, kernel
// parameter blocks and the launcher prototypes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/dctz_hip.h"
#include "dct64_block.h"
#include "dct_nd_block.h"

namespace dctz {

// A TILE = 64 consecutive 64-element blocks = the work of ONE wavefront per loop trip: lane b owns block b
// of the tile and runs its whole transform in registers (dct64_block.h).  Workgroups of the two big kernels
// are single wavefronts.
#ifndef DCTZ_DEC_EXC_CAP
#define DCTZ_DEC_EXC_CAP 1024
#endif
constexpr int TILE_BLKS = 64;
constexpr int TILE_ELEMS = TILE_BLKS * 64;   // 4096
constexpr int WG = 64;                       // threads per workgroup of k_compress / k_decompress
constexpr int SWG = 256;                     // threads per workgroup of the streaming helpers (stats, count, compact, ...)
constexpr int EXC_BYTES = 4096;              // k_compress: the lanes' exception strips, then the 64 x 64 bin ids of the tile on their way out
constexpr int DEC_EXC_CAP = DCTZ_DEC_EXC_CAP;            // decode: exact coefficients of one tile staged in LDS one tile ahead (floats), fp64
// fp32 decodes through a half-tile image (8 KiB): with 2048 staged floats behind it the two hold a dense tile's 4032
template <typename T> struct DecStage { static constexpr int CAP = sizeof(T) == 8 ? DEC_EXC_CAP : 2 * DEC_EXC_CAP; };

template <typename T> struct Traits;
template <> struct Traits<double> {
  using Vec = double2;
  using Bits = unsigned long long;
  static constexpr int EPV = 2;           // elements per 16-byte vector
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
  __host__ __device__ static Vec zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ static void div(Vec& v, float s) { v.x = v.x / s; v.y = v.y / s; v.z = v.z / s; v.w = v.w / s; }
  __device__ static void mul(Vec& v, float s) { v.x = v.x * s; v.y = v.y * s; v.z = v.z * s; v.w = v.w * s; }
  __device__ static void unpack(const Vec& v, float* e) { e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
  __device__ static Vec pack(const float* e) { return make_float4(e[0], e[1], e[2], e[3]); }
  __device__ static float huge() { return 3.40282346638528859812e38f; }
  __device__ static float from_bits(Bits b) { return __uint_as_float(b); }
};

// Geometry of a tile in bytes for element type T, moved through LDS in PH phases (elements [64 p / PH, 64 (p + 1) / PH)
// of every block per phase).  PH = 2 halves a wave's LDS footprint (fp64: 16 KiB), which leaves room for two waves per
// SIMD: one issues instructions while the other waits.
template <typename T, int PHASES = 1> struct Geo {
  static constexpr int BLKB = 64 * (int)sizeof(T);      // bytes per block (512 / 256)
  static constexpr int NSEG = BLKB / 128;               // 128-byte segments per block (4 / 2)
  static constexpr int NCH = BLKB / 16;                 // 16-byte chunks per block (32 / 16)
  static constexpr int TILEB = TILE_BLKS * BLKB;        // bytes per tile (32 KiB / 16 KiB)
  static constexpr int NROW = TILEB / 1024;             // 1 KiB LDS rows per tile (32 / 16)
  static constexpr int PH = PHASES;
  static constexpr int SEGP = NSEG / PH;                 // 128-byte segments of a block per phase
  static constexpr int CHP = NCH / PH;                   // 16-byte chunks of a block per phase
  static constexpr int PHB = TILEB / PH;                 // bytes of a phase image
  static_assert(NSEG % PH == 0, "a phase is a whole number of 128-byte segments");
};
// The "stored exactly" coefficients of a tile leave k_compress as NQ SUB-LISTS of QW positions each: the coefficients j in
// [QW q, QW q + QW) of all 64 blocks, block after block, compacted in LDS by the whole wave (one prefix sum over the
// lanes' counts per sub-list) and written out in whole rows of 64 items.  The staging buffer holds SLOTS items, of which
// the last 64 are the lanes' dump slots (a coefficient that is not stored exactly is written there); a sub-list with more
// than CAP items goes out in several rounds.  k_compact_ac puts the sub-lists of a tile back into the reference's order
// (dctz-comp-lib.c:478-544: block-major, j ascending) from the per-block counts k_compress leaves: one word per block,
// CBITS bits per sub-list.  An item is a float (EC: what AC_exact stores) or the coefficient in full precision plus its
// position (QT: the normalisation needs the table of the whole array first).
// QW = 16 (four sub-lists), except QT on fp64 (8).  Wider sub-lists mean fewer prefix sums, lists that are more often in
// the reference's order as they stand (nothing beyond position QW stored exactly: a flat copy for k_compact_ac) and fewer
// runs to merge otherwise -- and more coefficient registers pinned while a sub-list is staged.  Measured on 512^3,
// k_compress + k_compact_ac in ms at p = 5 % / 17 % / 69 %: fp64 EC  QW 8: 0.289 / 0.366 / 0.577, 16: 0.274 / 0.363 / 0.550,
// 32 (spills): 0.335 / 0.392 / 0.625; fp64 QT  8: 0.332 / 0.456 / 0.910, 16 (20 spilled registers): 0.327 / 0.470 / 0.887.
template <typename T, int MODE> struct Sub {
  using Item = typename std::conditional<MODE == DCTZHIP_EC, float, T>::type;
#ifndef DCTZ_QW64
#define DCTZ_QW64 16
#endif
#ifndef DCTZ_QW64_QT
#define DCTZ_QW64_QT 8
#endif
#ifndef DCTZ_QW32
#define DCTZ_QW32 16
#endif
  static constexpr int QW = sizeof(T) == 8 ? (MODE == DCTZHIP_QT ? DCTZ_QW64_QT : DCTZ_QW64) : DCTZ_QW32;   // coefficients per sub-list
  static constexpr int NQ = 64 / QW;                                                // sub-lists per tile
  // QT, fp64: items (8 B), positions (1 B) AND the wave's per-position maxima of the tile (64 x 8 B, QMAX_AT) share the
  // 4 KiB the tile's bin ids need on their way out anyway -- with anything more the kernel loses its eighth workgroup
  // per CU (20 KiB each): 384 slots.  Otherwise the buffer holds 4 KiB of items, the positions (QT, fp32) behind them.
  static constexpr bool PACKED = (MODE == DCTZHIP_QT) && sizeof(T) == 8;
  static constexpr int SLOTS = PACKED ? 384 : EXC_BYTES / (int)sizeof(Item);        // 1024 floats | 384 doubles
  static constexpr int CAP = SLOTS - 64;
  static constexpr int ITEM_BYTES = SLOTS * (int)sizeof(Item);
  static constexpr int POS_BYTES = (MODE == DCTZHIP_QT) ? SLOTS : 0;
  static constexpr int QMAX_AT = EXC_BYTES - 512;                                   // (PACKED only)
  static constexpr int BYTES = PACKED ? EXC_BYTES : ITEM_BYTES + POS_BYTES;
  static_assert(!PACKED || ITEM_BYTES + POS_BYTES <= QMAX_AT, "items and positions end in front of the maxima");
  static constexpr int CBITS = 32 / NQ;                                             // bits per count in a block's word (counts <= QW)
  // k_compact_ac: the counts of two sub-lists share a dword for the prefix sums over the blocks of a tile (sums <= 1024)
  static constexpr int FB = 16;
  static constexpr int FPD = 32 / FB;
  static constexpr int NPK = (NQ + FPD - 1) / FPD;
};
// phases of k_compress / k_decompress per element type (build knobs for A/B runs)
#ifndef DCTZ_PHC64
#define DCTZ_PHC64 2
#endif
#ifndef DCTZ_PHD64
#define DCTZ_PHD64 1
#endif
#ifndef DCTZ_PHC32
#define DCTZ_PHC32 2      /* fp32 compress: 8 KiB half-tile images; with the packed transform the kernel needs 160 VGPRs, so three
                             waves fit a SIMD (12 workgroups per CU): 0.168 -> 0.148 ms on 512^3 (one phase, 8 per CU: =1) */
#endif
#ifndef DCTZ_PHD32
#define DCTZ_PHD32 2      /* fp32 decompress: two phases, 13 KiB of LDS -> 8 workgroups per CU instead of 7 (-4 %) */
#endif
template <typename T> struct Phases { static constexpr int C = DCTZ_PHC32, D = DCTZ_PHD32; };
template <> struct Phases<double> { static constexpr int C = DCTZ_PHC64, D = DCTZ_PHD64; };

// Per-call control block in device memory; zeroed by the first kernel of every compress call (k_stats_final) --
// a decode call only ever sets `error`, and the host clears the block after a failed call.
struct Ctl {
  unsigned pad_ticket;
  unsigned cnt_total;              // exceptions emitted / consumed so far
  unsigned error;                  // 2: AC_exact underrun on decode
  unsigned pad0;
  unsigned long long qraw[64];     // max |coef| per position, raw bits of T (dctz-comp-lib.c:371-372)
  unsigned long long q0;           // bits of the last block's DC (qtable[0], :355-360)
  unsigned long long pad1;
};
static_assert(sizeof(Ctl) % 16 == 0, "control block must be a multiple of 16 bytes");

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
  double sf_used;                          // scaling factor / FastDiv level the device chose for this call (SfGuess)
  unsigned fast_used, pad_used;
  double fstats[3];                        // statistics fused into k_compress
  double psnr[6];                          // calc_psnr reduction: min, max, sum e^2, max |e|, max |e/x| (k_psnr_final)
};

// The scaling factor of a speculative compress call, chosen ON THE DEVICE from the sampled statistics (k_stats_final)
// so that the host need not be asked between the sample and k_compress; the host checks it afterwards against the true
// statistics with its own libm (util.c:29 / :43), exactly as it checks its own guesses.
struct SfGuess {
  double sf;                       // value of the data type (exactly representable in a double)
  unsigned fast_sf;                // FastDiv level the kernel may assume for x / sf (FwdParams::fast_sf)
  unsigned pad;
};
// Decade tables built by the host with ITS log10 / pow (so that the device's choice is the host's, by construction):
// thr[i] = largest value m of the data type with ceil(log10(m)) <= kmin + i,  pw[i] = 10^(kmin - 1 + i) in the data
// type; both widened to double.  sf(max) = pw[#{i : thr[i] < max}].
struct SfTable {
  const double* thr;
  const double* pw;
  int nk;
  int fastdiv;                     // the context's DCTZHIP_FASTDIV level
  int dtype;
};

// What the hand-off of a call's results to the host needs (k_finish, or the first workgroup of k_compact_ac /
// k_decompress; box == NULL there: no hand-off in that kernel).
struct FinArgs {
  Ctl* ctl;
  const double* part;              // fused statistics partials ({max|x|, min|x|, sum} per slot), nparts of them (0: none)
  int nparts;
  HostBox* box;
  unsigned long long seq;
  const SfGuess* guess;            // device-chosen scaling factor of this call, reported with the results (NULL: the host chose it)
  unsigned* zero_words;            // k_finish only: words to clear for the next call (the ticket counters of k_compress_eo), or NULL
  unsigned nzero;
};

// A multi-dimensional array and its tile grid (dct_nd_block.h): nd = 2 -> 8 x 8 tiles, nd = 3 -> 4 x 4 x 4 tiles.
struct NdShape {
  int nd;
  size_t d[3];                     // extents, last axis fastest
  size_t nb[3];                    // tiles per axis = ceil(d / edge)
  size_t nblk;                     // number of blocks = product of nb
};

// Direct addressing of a multi-dimensional array by the big kernels (no gather / scatter pass): possible when every
// extent is a multiple of the tile edge (no padding) and the array is below 4 GiB (one buffer descriptor covers it).
// Block B of the tile grid -> its origin in the array needs one or two divisions by launch constants (multiply-high +
// one correction step).
struct NdDirect {
  unsigned on;                     // 0: the kernels read / write the block-after-block layout
  unsigned nd;                     // 2 | 3
  unsigned dx, dy;                 // extents (elements): dx = last (fastest) axis; dy = middle axis (3-D only)
  unsigned nbx, nby;               // tiles along x / y
  unsigned mx, my;                 // floor(2^32 / nbx), floor(2^32 / nby)
  unsigned nblk;
  unsigned bytes;                  // size of the array (buffer descriptor range)
};

template <typename T>
struct FwdParams {
  const T* x;                      // input
  uint8_t* bin;                    // bin_index out
  float* dc;                       // DC out
  float* ac;                       // AC_exact out
  T* coef;                         // optional coefficient tap (tests)
  T* scaled;                       // optional: x / sf of every full block written here by k_compress itself (dctz-comp-lib.c:193-216); never x
  float* ac_tmp;                   // workgroup-local AC_exact lists, list of workgroup b starts at the slot of its first tile
  unsigned* tile_cnt;              // list lengths, one per workgroup (+1 for the remainder block)
  T* qt_item;                      // QT scratch: flagged coefficients, full precision (same list layout as ac_tmp)
  uint8_t* qt_j;                   // QT scratch: their position j
  unsigned* qcnt;                  // per block: how many of its coefficients every sub-list of its tile holds (Sub::CBITS bits each)
  unsigned* ttot;                  // per tile: its "stored exactly" coefficients
  const T* tab;                    // TB_* block (device)
  const T* rtab;                   // RTAB_* block (device), remainder block only
  Ctl* ctl;
  const SfGuess* guess;            // speculative call: sf / fast_sf chosen on the device (k_stats_final); NULL: the two fields below
  double* stat_part;               // fused statistics: {max|x|, min|x|, sum} per workgroup (+1 slot for the remainder block), else NULL
  unsigned nfull;                  // number of full 64-element blocks
  unsigned ntiles;
  unsigned last_is_full;           // N % 64 == 0
  unsigned fast_sf, fast_bw;       // divisor inside FastDiv's exponent window (host check); fast_sf == 2: every element too
  unsigned nlists_main;            // number of workgroup lists = grid of k_compress
  T sf, bin_width, range_min, range_max;
  NdDirect nd;                     // multi-dimensional blocks straight from the array (GEOM != GEOM_1D only)
  // k_compress_eo, single-pass placement of AC_exact (direct != 0): per-tile descriptors of the look-back (dctz_kernels_eo.hip),
  // the tag of this call's descriptors; k_compress_rem then appends behind Ctl::cnt_total instead of writing a list
  unsigned long long* lb_desc;
  unsigned* lb_ticket;             // eight ticket counters, 16 words apart (zero when the kernel starts: k_finish leaves them so)
  unsigned lb_epoch;
  unsigned direct;
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
  const unsigned* tile_cnt;        // per-TILE counts of "stored exactly" flags (k_count_tiles)
  const unsigned* wg_cnt;          // the same summed over the tile range of every workgroup of k_decompress (nwg entries)
  const unsigned* tile_pre;        // tile-interleaved k_decompress: counts of the tiles of the same RANGE in front of a tile (k_count_tiles)
  Ctl* ctl;
  unsigned nfull, ntiles, ac_count;
  unsigned nwg;                    // grid of k_decompress
  NdDirect nd;                     // multi-dimensional blocks straight into the array (GEOM != GEOM_1D only)
  T sf, bin_width, range_min, range_max;
  double eb;
};

// ---- batches of arrays (dctzhip_compress_batch / dctzhip_decompress_batch) -------------------------------------
// k arrays of ONE element type go through ONE launch sequence: every kernel of the single-array path has a batch
// form whose workgroups look their array up (`first[]`: first workgroup of array i in that launch, first[k] = grid)
// and then run the single-array body on that array's own parameter block -- exactly the FwdParams / InvParams the
// single-array launch would have been given, with the scratch pointers offset to the array's slices.  Nothing couples
// the arrays (own statistics, sf, bin ranges, tot_AC_exact_count, QT table: dctz-comp-lib.c:186), so every array's
// outputs are those of its own dctzhip_compress call, bit for bit.
template <typename T>
struct BatchFwd {
  FwdParams<T> p;
  double eb;
  T* scaled;                       // receives x / sf (dctz-comp-lib.c:193-216); NULL: not asked for; may alias p.x
  unsigned n;                      // elements
  unsigned rem;                    // n % 64 (the short last block)
  unsigned nlists;                 // workgroup lists of this array (+1 for the remainder block)
  unsigned nparts, part_base;      // statistics partials of this array: part[3 * (part_base + j)], j < nparts
  unsigned sample;                 // != 0: a speculative item -- its statistics pass reads one 4 KiB chunk out of every `sample`, k_compress_batch<STATS>
                                   // takes the true statistics into p.stat_part, the hand-off reports those (0: the pass reads everything)
};
template <typename T>
struct BatchInv {
  InvParams<T> p;
  unsigned n, rem, scale, cnt_wgs; // cnt_wgs: workgroups of k_count_batch for this array (max(p.nwg, 1))
  unsigned* rem_cnt;               // flags of the remainder block (its own word)
  T qtab[64];                      // QT: this array's table (p.qtab points at the device copy of this field)
};
// What a batch hands back per array (fine-grained pinned host memory, written by the hand-off workgroup)
struct BatchResC {
  double sf_used;                  // scaling factor the device chose (k_sf_batch), verified by the host afterwards
  double stats[3];                 // max|x|, min|x|, sum
  unsigned cnt, error, fast_used, pad;   // pad: one-launch batches write their tag here, last
  unsigned long long q0;           // bits of the last block's DC (qtable[0])
};
struct BatchResQ { unsigned long long qraw[64]; };        // QT: per-position maxima, raw bits of T
struct BatchResD { unsigned total, error, tag, pad; };    // decode: flags found / 2 = more than ac_count provides; tag: one-launch batches (written last)
struct BatchFin {
  unsigned long long* word;        // mailbox word (device view) this sequence publishes `seq` into; NULL: it does not publish
                                   // (a later sequence of its chain does)
  unsigned long long seq;
  void* res;                       // BatchResC[k] / BatchResD[k] (device view of the host table)
  BatchResQ* resq;                 // QT only
};
template <typename T> void launch_stats_batch(const BatchFwd<T>* items_src, const unsigned* first_src, unsigned k, unsigned grid,
                                              const void* blob_src, void* blob_dst, size_t blob_bytes, double* part, hipStream_t s);
template <typename T> void launch_sf_batch(const BatchFwd<T>* items, unsigned k, const double* part, double* bstats, SfTable tab, hipStream_t s);
template <typename T> void launch_scale_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, hipStream_t s);
template <typename T> void launch_compress_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, int mode, bool stats, hipStream_t s);
template <typename T> void launch_compress_rem_batch(const BatchFwd<T>* items, const unsigned* rem_items, unsigned nrem, int mode, hipStream_t s);
template <typename T> void launch_compact_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, unsigned chunks, int mode,
                                                const double* bstats, const BatchFin& fin, hipStream_t s);
template <typename T> void launch_count_batch(const BatchInv<T>* items_src, const unsigned* first_src, unsigned k, unsigned grid,
                                              const void* blob_src, void* blob_dst, size_t blob_bytes, hipStream_t s);
template <typename T> void launch_decompress_batch(const BatchInv<T>* items, const unsigned* first, unsigned k, unsigned grid, int mode,
                                                   const BatchFin& fin, hipStream_t s);
template <typename T> void launch_decompress_rem_batch(const BatchInv<T>* items, const unsigned* rem_items, unsigned nrem, int mode, hipStream_t s);

// ---- one launch per call (dctz_kernels_one.hip): arrays whose tiles are all resident at once --------------------
// One WAVE per tile, ONE_TW tiles per workgroup (+ one workgroup for the short last block): calc_data_stat, scaling,
// transform, binning AND the ordered placement of AC_exact in a single kernel; on decode the flag counts, their prefix
// and the reconstruction.  What the chain of kernels hands over at kernel boundaries travels through the BOARD here: one
// 8-byte {tag = epoch, value} granule per workgroup and step, written by one agent-scope store and swept with agent-scope
// loads by the first wave of every workgroup (every workgroup of the launch is resident: the grid is at most what the
// chip holds at once).  Several tiles per workgroup because a sweep reads a granule per WORKGROUP: with single-wave
// workgroups the sweeps of a 2000-tile array would move 2000 x 2000 x 8 bytes past the caches.
#ifndef DCTZ_ONE_TW
#define DCTZ_ONE_TW 4
#endif
constexpr int ONE_TW = DCTZ_ONE_TW;
struct OneBoard {
  unsigned long long* ga;          // per workgroup: decade index of its max|x| (+ "outside FastDiv's window"), compress only
  unsigned long long* gb;          // per workgroup: its "stored exactly" coefficients (tot_AC_exact_count of the tile)
  double* rec;                     // per workgroup: max|x|, min|x|, sum (what the host is told; never on the critical path)
  unsigned epoch;                  // tag of this launch's granules (never 0; older launches left other tags)
  unsigned nwg;                    // workgroups of the launch = ceil(ntiles / ONE_TW) + (rem ? 1 : 0)
  unsigned long long* dbg;         // NULL, or 16 time stamps (100 MHz clock) per workgroup: DCTZHIP_ONE_STAMPS, tools/one_stamps.py
};
template <typename T>
struct OneFwd {
  FwdParams<T> p;                  // x, bin, dc, ac, coef, scaled, tab, rtab, ctl, nfull, ntiles, last_is_full, fast_bw, bin_width, range_*
  OneBoard b;
  SfTable sft;
  HostBox* box;                    // hand-off by the launch's last workgroup
  unsigned long long seq;
  // QT: the per-position maxima of the whole array (dctz-comp-lib.c:371-372), merged with device atomics -- one per
  // workgroup and position -- in a table of qt_shards x 64 words, qt_stride words apart: a word per 256 bytes and four shards
  // for a single array (a few memory channels serialise the atomics on neighbouring words: 1582 waves on two lines of a
  // contiguous table took 30 us), contiguous and unsharded for the small arrays of a batch.  qt_next: the next call's
  // table, zeroed by the hand-off.
  unsigned long long* qt;
  unsigned long long* qt_next;
  unsigned qt_stride, qt_shards;
  struct BatchResC* bres;          // batch: the array's entry of the result table instead of the mailbox (box == NULL then)
  struct BatchResQ* bresq;
  unsigned tag, pad_tag;
  double eb;
  unsigned rem;                    // N % 64
  unsigned bad_guess;              // (tests) the first guess of the array's decade is made wrong on purpose: every tile runs twice
};
template <typename T>
struct OneInv {
  InvParams<T> p;                  // bin, dc, ac, out, tab, rtab, qtab, ctl, nfull, ntiles, ac_count, sf, bin_width, range_*, eb
  OneBoard b;
  HostBox* box;
  unsigned long long seq;
  unsigned rem;
  unsigned tag;
  unsigned withhold, pad_w;        // (tests) workgroup 0 withholds its granule: every sweep that needs it gives up
  struct BatchResD* bres;          // batch: the array's entry of the result table instead of the mailbox
  T qtab[64];                      // QT: the clamped table (dctz-decomp-lib.c:193-199), in the kernel's arguments
};
// A BATCH through the one-launch kernels: the arrays of one element type share a launch, every array with its own
// workgroups, its own stretch of the board and its own entry of the result table.  A workgroup finds everything about its
// array in ONE 128-byte record indexed by its number (pinned host memory the kernel reads once, with scalar loads: one trip
// over PCIe per workgroup instead of a copy in front of the launch); what the arrays share rides in the kernel's arguments.
struct OneRecC {
  const void* x; void* bin; float* dc; float* ac; void* scaled; const void* rtab;
  double bin_width, range_min, range_max, eb;      // values of the element type, widened
  unsigned nfull, ntiles, rem, fast_bw;
  unsigned wg_local, nwg, board_base, item;        // this workgroup among those of its array; first granule of the array; array
  unsigned pad[4];
};
struct OneRecD {
  const void* bin; const float* dc; const float* ac; void* out; const void* rtab; const void* qtab;
  double sf, bin_width, range_min, range_max, eb;
  unsigned nfull, ntiles, rem, ac_count;
  unsigned wg_local, nwg, board_base, item;
  unsigned pad[2];
};
static_assert(sizeof(OneRecC) == 128 && sizeof(OneRecD) == 128, "one record = two 64-byte scalar loads");
template <typename T>
struct OneBatchC {
  const OneRecC* recs;
  const T* tab;
  Ctl* ctl;                        // per array: control blocks (only `error` is used)
  unsigned long long* qt;          // QT, per array: 64 words of this call's table of maxima ...
  unsigned long long* qt_next;     // ... and of the next call's (zeroed by every array's hand-off)
  OneBoard b;                      // base pointers and the epoch (nwg: per record)
  SfTable sft;
  struct BatchResC* res;           // per array (device view of the host table); `pad` receives `tag` last
  struct BatchResQ* resq;          // QT
  unsigned tag;
  unsigned bad_guess;
};
template <typename T>
struct OneBatchD {
  const OneRecD* recs;
  const T* tab;
  Ctl* ctl;
  OneBoard b;
  struct BatchResD* res;
  unsigned tag;
  unsigned pad;                    // (tests) != 0: workgroup 0 of every array withholds its granule
};
// A list's entry in tile_cnt[] (k_compress / k_compress_eo -> k_compact_ac): its length, and LIST_IN_ORDER when the list is in the
// reference's order as it stands (dctz_kernels.hip).
constexpr unsigned LIST_IN_ORDER = 0x80000000u, LIST_LEN = 0x7FFFFFFFu;
// k_compress_eo (dctz_kernels_eo.hip): k_compress for flat fp64 blocks with every block shared by a lane of an "even" and a
// lane of an "odd" wavefront (workgroups of two waves); same parameters, same outputs
void launch_compress_eo(const FwdParams<double>& p, int mode, bool stats, int grid, hipStream_t s);
int compress_eo_occupancy(int mode, bool stats, bool direct);
constexpr unsigned ONE_ERR_TIMEOUT = 3u;   // Ctl::error / HostBox::error: a sweep of the board gave up (a workgroup was not resident)
template <typename T> void launch_compress_one(const OneFwd<T>& a, int mode, bool scaled, hipStream_t s);
template <typename T> void launch_decompress_one(const OneInv<T>& a, int mode, hipStream_t s);
template <typename T> void launch_compress_one_batch(const OneBatchC<T>& cm, unsigned grid, int mode, bool scaled, hipStream_t s);
template <typename T> void launch_decompress_one_batch(const OneBatchD<T>& cm, unsigned grid, int mode, hipStream_t s);
template <typename T> int compress_one_occupancy(int mode, bool scaled);
template <typename T> int decompress_one_occupancy(int mode);

template <typename T> void launch_stats(const T* x, size_t n, double* part, int nparts, double* out, hipStream_t s,
                                        HostBox* box = nullptr, unsigned long long seq = 0, Ctl* zero = nullptr,
                                        const SfTable* tab = nullptr, SfGuess* guess = nullptr);
template <typename T> void launch_stats_sample(const T* x, size_t n, unsigned group, double* part, int nparts, double* out, hipStream_t s,
                                               HostBox* box = nullptr, unsigned long long seq = 0, Ctl* zero = nullptr,
                                               const SfTable* tab = nullptr, SfGuess* guess = nullptr);
void launch_stats_final(const double* part, int nparts, double* out, hipStream_t s, HostBox* box = nullptr,
                        unsigned long long seq = 0, Ctl* zero = nullptr, const SfTable* tab = nullptr, SfGuess* guess = nullptr);
void launch_finish(Ctl* ctl, const double* part, int nparts, HostBox* box, unsigned long long seq, hipStream_t s, const SfGuess* guess = nullptr,
                   unsigned* zero_words = nullptr, unsigned nzero = 0);
template <typename T> void launch_debug_divide(const T* x, size_t n, T d, int ok, T* fast, T* ref, hipStream_t s);
template <typename T> void launch_serial_sum(const T* x, size_t n, double* out, hipStream_t s);
template <typename T> void launch_scale(const T* x, T* out, size_t n, T sf, int grid, hipStream_t s);
template <typename T> void launch_compress(const FwdParams<T>& p, int mode, bool stats, int grid, int geom, hipStream_t s);
template <typename T> void launch_gather_nd(const T* x, T* lin, const NdShape& sh, double* part, int nparts, hipStream_t s);
template <typename T> void launch_scatter_nd(const T* lin, T* out, const NdShape& sh, int grid, hipStream_t s);
template <typename T> void launch_compress_rem(const FwdParams<T>& p, int mode, int l, hipStream_t s);
template <typename T> void launch_compact_ac(const FwdParams<T>& p, int mode, double eb, unsigned nlists, int grid, const FinArgs& fin, hipStream_t s);
struct QtabArg { unsigned long long w[64]; };       // a QT table (64 values of either element type) as a kernel argument
void launch_count_tiles(const uint8_t* bin, unsigned nfull, unsigned ntiles, unsigned nwg, unsigned* tile_cnt, unsigned* wg_cnt, hipStream_t s,
                        const void* qtab_host = nullptr, size_t qtab_bytes = 0, void* qtab_dev = nullptr, unsigned* tile_pre = nullptr);
template <typename T> void launch_decompress(const InvParams<T>& p, int mode, int grid, const FinArgs& fin, int geom, hipStream_t s);
template <typename T> void launch_decompress_rem(const InvParams<T>& p, int mode, bool scale, int l, hipStream_t s);
template <typename T> void launch_dct_blocks(const T* x, T* out, const T* gtab, const T* rtab, size_t n,
                                             bool inverse, int grid, hipStream_t s);
template <typename T> void launch_psnr(const T* x, const T* r, size_t n, double* part, int nparts, double* out, hipStream_t s);

// GPU entropy stage (dctz_deflate.hip): one section -> one zlib stream, everything in device memory
size_t deflate_chunk_bytes();
size_t deflate_scratch_bytes(size_t n);
size_t deflate_bound(size_t n);
hipError_t launch_deflate(const void* src, size_t n, void* dst, void* scratch, unsigned long long* box_len, uint32_t* host_sizes, bool literals_only,
                          hipStream_t st);
hipError_t launch_inflate(const void* sec, const uint32_t* offs, size_t nch, size_t n, void* dst, unsigned long long* adler, uint32_t* status, hipStream_t st);
template <typename T> int compress_occupancy(int mode, bool stats, int geom, bool scaled = false);
template <typename T> int decompress_occupancy(int mode, int geom);
template <typename T> size_t compress_lds_bytes(int mode);
template <typename T> size_t decompress_lds_bytes();

}  // namespace dctz
