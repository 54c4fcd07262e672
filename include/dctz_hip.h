/*
 * dctz_hip.h -- C ABI of the MI355X (gfx950) hot path of DCTZ.
 *
 * This is the seam a DCTZ maintainer binds to: it sits INSIDE the reference's
 * dctz_compress() / dctz_decompress() (dctz.h:126-127) and replaces exactly the
 * serial CPU stages between "input in memory" and "three byte streams ready for
 * zlib" (and the mirror image on the way back).  Plain C types only; device
 * buffers are passed as void* device pointers; the library never frees or
 * reallocates caller memory.  Every entry point returns 0 on success or a
 * negative DCTZHIP_E_* code; dctzhip_last_error() gives the text.  There is NO
 * CPU fallback: without a usable HIP device every call fails.
 *
 * Each function names the reference code it replaces (file:line under the
 * upstream tree).
 */
#ifndef DCTZ_HIP_H
#define DCTZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCTZHIP_BLK 64            /* dctz.h:28  BLK_SZ */
#define DCTZHIP_NBINS 255         /* dctz.h:65-66 (t_bin_id = unsigned char) */

enum { DCTZHIP_F32 = 0, DCTZHIP_F64 = 1 };   /* values of t_datatype, dctz.h:44-47 */
enum { DCTZHIP_EC = 0, DCTZHIP_QT = 1 };     /* -DUSE_QTABLE off/on, Makefile:12-17 */

enum {
  DCTZHIP_OK = 0,
  DCTZHIP_E_ARG = -1,        /* bad argument (null, misaligned, n == 0, n > INT_MAX) */
  DCTZHIP_E_BOUND = -2,      /* error_bound < 1e-6: dctz-comp-lib.c:135-138 */
  DCTZHIP_E_HIP = -3,        /* a HIP runtime call failed (no device, OOM, ...) */
  DCTZHIP_E_INTERNAL = -4    /* the library caught itself out: a launch plan that does not fit its scratch tables, a
                                hand-off that never arrived (10 s), an error flag set by a kernel */
};

typedef struct dctzhip_ctx dctzhip_ctx;

/* What the compress stage hands back to the host besides the three streams.
 * For DCTZHIP_F32 the stats are float values widened exactly to double. */
typedef struct {
  double sf;                 /* scaling factor, util.c:29/43 (host libm, same expr) */
  double mean;               /* sum/N, util.c:28/41 -- DEVICE summation order       */
  double max_abs, min_abs;   /* util.c:18-25 */
  uint32_t cnt;              /* tot_AC_exact_count, dctz-comp-lib.c:323,497,537     */
  uint32_t nblk;             /* CEIL(N, 64), dctz-comp-lib.c:227                    */
  double qtable[64];         /* QT: clamped table as appended to the stream
                                (dctz-comp-lib.c:450-461, 815-820); slot 0 = DC of
                                the last block (:355-360).  EC: zeros.             */
  double qtable_raw[64];     /* QT: table before clamping (= ./qtable.bin, :443-448) */
  uint32_t flags;            /* DCTZHIP_INFO_*: how the statistics were obtained (below) */
  uint32_t reserved;
} dctzhip_cinfo;

/* calc_data_stat (util.c:12-44) needs the whole array before the first division
 * (dctz-comp-lib.c:193-216), i.e. a second read of the input.  Only the decade of
 * max|x| enters sf (util.c:29), so for large inputs the library guesses sf from a
 * sample, lets the compress kernel compute the TRUE max / min / sum while it
 * streams the data anyway, and checks the guess afterwards.  The outputs never
 * depend on the guess: a wrong one is detected and the kernels run again with
 * the true statistics. */
#define DCTZHIP_INFO_STATS_FUSED 1u   /* guess verified: the separate statistics pass was saved */
#define DCTZHIP_INFO_RESPUN 2u        /* guess wrong: compress kernels were run a second time    */
#define DCTZHIP_INFO_SPLIT 8u         /* the compress kernel was k_compress_eo (a block over two lanes: dctzhip_set_split)          */
#define DCTZHIP_INFO_SINGLE_PASS 16u  /* ... and AC_exact was placed by that kernel itself (look-back over the tiles' counts)       */
#define DCTZHIP_INFO_LB_FALLBACK 32u  /* ... whose look-back gave up: the pass was run again through the workgroup-local lists      */
#define DCTZHIP_INFO_ONE_LAUNCH 4u    /* the whole call was one kernel (arrays whose tiles are all resident at once)         */

/* Per-kernel device time of the last compress / decompress call, milliseconds,
 * measured with HIP events on the context's stream (only when profiling is on). */
typedef struct {
  float stats_ms;            /* calc_data_stat kernels          */
  float main_ms;             /* fused block-DCT + binning (or de-quantise + IDCT) */
  float tail_ms;             /* remainder block + QT normalise   */
  float total_ms;            /* first kernel start -> last kernel end */
} dctzhip_timings;

/* ---- context ------------------------------------------------------------- */
/* device < 0: use the current HIP device.  The context owns a stream, constant
 * tables and a scratch arena that grows on demand (dctzhip_reserve pre-sizes it
 * so that later calls do no allocation).  One context per GPU per thread of
 * control; a context is not re-entrant (neither is the reference: file-static
 * FFTW plan, dct.c:18-22). */
int dctzhip_device_count(void);                            /* GPUs visible to this process (0 if none) */
int dctzhip_ctx_create(dctzhip_ctx **out, int device);
void dctzhip_ctx_destroy(dctzhip_ctx *ctx);
const char *dctzhip_last_error(const dctzhip_ctx *ctx);   /* ctx may be NULL: last create error */
int dctzhip_reserve(dctzhip_ctx *ctx, size_t n, int dtype, int mode);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * context's own.  NULL is the legacy default stream, as everywhere in HIP (what a
 * caller that never created a stream runs on -- ordered against its other work);
 * dctzhip_use_own_stream goes back to the context's private non-blocking stream. */
int dctzhip_set_stream(dctzhip_ctx *ctx, void *hip_stream);
int dctzhip_use_own_stream(dctzhip_ctx *ctx);
void *dctzhip_get_stream(dctzhip_ctx *ctx);
int dctzhip_set_profiling(dctzhip_ctx *ctx, int on);
/* Speculative fused statistics (see DCTZHIP_INFO_*): on != 0 enables it for inputs of
 * at least min_elements (0 keeps the current threshold, default 2^22); on == 0 always
 * runs the separate statistics pass first.  Default: on (env DCTZHIP_SPECULATE=0 turns
 * it off).  d_scaled holds x / sf for the verified sf when the call returns: k_compress
 * writes it while it runs (a pass with a wrong guess is run again and writes it again); a
 * call whose d_scaled aliases d_in does not speculate (the input would not survive a wrong
 * guess) and takes the full statistics pass first.  DCTZHIP_FUSE_SCALED=0: a pass of its
 * own behind the kernels, as before round 3. */
int dctzhip_set_speculation(dctzhip_ctx *ctx, int on, size_t min_elements);
int dctzhip_last_timings(dctzhip_ctx *ctx, dctzhip_timings *t);
/* One launch per call.  An array whose tiles are all resident on the chip at once (up to about 8 M fp32 / 4 M fp64 elements:
 * 2048 / 1024 tiles of 64 blocks) is compressed -- and decompressed -- by ONE kernel: calc_data_stat, scaling, transform,
 * binning and the ordered placement of AC_exact (decode: flag counts, their prefix, reconstruction) exchange what the chain
 * of kernels hands over at kernel boundaries through 8-byte tagged words in device memory instead.  The outputs are bit for
 * bit the chain's (DCTZHIP_INFO_ONE_LAUNCH in info->flags tells which ran); a launch that finds its workgroups not all
 * resident (another process on the GPU) gives up after 20 ms and the call is run through the chain.  Default: on (env
 * DCTZHIP_ONE=0 turns it off); on == 0 always takes the chain of kernels. */
int dctzhip_set_one_launch(dctzhip_ctx *ctx, int on);
/* A block over two lanes.  on != 0: the compress kernel of the chain (arrays beyond the one-launch size) for flat fp64
 * blocks is k_compress_eo -- workgroups of two wavefronts that share a tile's blocks, one computing the even-numbered
 * coefficients of every block, the other the odd-numbered ones (half the registers per lane, three waves per SIMD instead
 * of two) -- with the tile's exact coefficients put into the reference's order inside the kernel.  Outputs are bit for bit
 * those of k_compress.  Calls that ask for the scaled copy, fp32 and multi-dimensional blocks keep k_compress.
 * on == 3, EC mode (experimental: measured slower than the lists): the kernel also writes every exact coefficient at its
 * final place in AC_exact[] -- tiles are handed out in order by ticket counters, each posts its count and looks back over
 * the counts in front of it for the running tot_AC_exact_count (dctz-comp-lib.c:478-544) -- so the call has no
 * workgroup-local lists and no k_compact_ac pass (DCTZHIP_INFO_SINGLE_PASS); a look-back that sees no progress for 20 ms
 * gives up and the pass is run again through the lists (DCTZHIP_INFO_LB_FALLBACK).  Env DCTZHIP_EO=0/1 and
 * DCTZHIP_EO_DIRECT=0/1 set the defaults (off, off). */
int dctzhip_set_split(dctzhip_ctx *ctx, int on);
/* (tests and tools)  Counters of the context -- which: 0 one-launch calls, 1 one-launch launches that gave up (run again
 * through the chain), 2 calls left on the chain after such a launch, 3 calls through k_compress_eo, 4 of them with
 * single-pass placement, 5 look-backs that gave up, 6 / 7 verified / wrong guesses of the scaling factor, 8 / 9 speculative
 * items of batches / those whose guess was refused -- and knobs that
 * make a rare path run on purpose -- key 0: workgroup 0 of the one-launch kernels withholds its granule (the launch gives
 * up after 20 ms, the call is run through the chain), 1: one look-back of k_compress_eo gives up, 2: sets counter 2. */
int dctzhip_debug_counter(dctzhip_ctx *ctx, int which, unsigned long long *value);
int dctzhip_debug_knob(dctzhip_ctx *ctx, int key, int value);
/* (tools) the name rocprofv3 lists the big kernel of the last call under, every template argument: which = 0 compress,
 * 1 decompress, 2 / 3 batch compress of the fp64 / fp32 arrays, 4 / 5 batch decompress ("" if none ran yet) */
int dctzhip_debug_last_kernel(dctzhip_ctx *ctx, int which, char *buf, size_t cap);
/* on != 0: every compress / decompress call ends with a synchronisation of the context's stream, i.e. its outputs are
 * complete for ANY observer when it returns (default off: complete in stream order, see the two calls below; env
 * DCTZHIP_BLOCKING=1 does the same).  For callers that read the buffers from another stream or from the host without
 * going through the context's stream. */
int dctzhip_set_blocking(dctzhip_ctx *ctx, int on);

/* ---- device memory helpers (so plain-C hosts need no HIP headers) --------- */
int dctzhip_malloc(dctzhip_ctx *ctx, void **dptr, size_t bytes);
int dctzhip_free(dctzhip_ctx *ctx, void *dptr);
int dctzhip_memcpy_h2d(dctzhip_ctx *ctx, void *dst, const void *src, size_t bytes);
int dctzhip_memcpy_d2h(dctzhip_ctx *ctx, void *dst, const void *src, size_t bytes);
int dctzhip_sync(dctzhip_ctx *ctx);
/* A D2H copy on a stream BESIDE the context's, for a second host thread (the pipelined dctz_compress brings finished pieces
 * back while the calling thread queues the next kernels): the caller has synchronised with whatever produced d_src, and
 * makes such copies from one thread at a time. */
int dctzhip_memcpy_d2h_side(dctzhip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* A D2H copy into pageable host memory that FOLLOWS its producer (dctz_decompress rebuilds a large array group by group
 * and brings finished groups back while the next ones are built): begin() starts the copy of bytes [0, bytes) of d_src to
 * dst through pinned slots, piece by piece, each piece as soon as advance() has announced it -- "bytes [0, upto) are
 * complete in the order of the context's stream" (an event behind the kernels queued so far; no host synchronisation) --,
 * end() waits for the last piece (abandon != 0: stops after the pieces under way).  One pipe at a time per process. */
/* capacities of the two pipes below: dctzhip_d2h_pipe_advance may be called this many times per pipe, a dctzhip_h2d_pipe_begin
 * takes at most this many groups (a caller with more takes its unpipelined path instead) */
#define DCTZHIP_D2H_PIPE_MAX_MARKS 256
#define DCTZHIP_H2D_PIPE_MAX_GROUPS 4096
int dctzhip_d2h_pipe_begin(dctzhip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int dctzhip_d2h_pipe_advance(dctzhip_ctx *ctx, size_t upto);
int dctzhip_d2h_pipe_end(dctzhip_ctx *ctx, int abandon);
/* An H2D copy from pageable host memory whose CONSUMER follows it (dctz_compress of a large array starts the kernels of a
 * group of elements as soon as that group is on the device): begin() starts the copy of src[0, bytes) to d_dst in groups
 * of group_bytes, in order, on a stream of its own; wait(upto) makes the context's stream wait -- on the GPU -- for the
 * group that holds byte upto - 1 (the host waits only until that group's copy has been issued); landed(upto) blocks the
 * host until bytes [0, upto) are on the device, i.e. until the source range is no longer read; end() joins (abandon != 0:
 * no further group is issued).  One pipe at a time per process. */
int dctzhip_h2d_pipe_begin(dctzhip_ctx *ctx, void *d_dst, const void *src, size_t bytes, size_t group_bytes);
int dctzhip_h2d_pipe_wait(dctzhip_ctx *ctx, size_t upto);
int dctzhip_h2d_pipe_landed(dctzhip_ctx *ctx, size_t upto);
int dctzhip_h2d_pipe_end(dctzhip_ctx *ctx, int abandon);
/* Page-lock a caller-owned host buffer for the copies above (hipHostRegister): pageable copies run at about 24 GB/s,
 * pinned ones at PCIe speed.  Pinning itself costs about as much as one pageable copy of the buffer, so it pays for
 * buffers that are reused across calls; the buffer must be unregistered before it is freed. */
int dctzhip_host_register(dctzhip_ctx *ctx, void *ptr, size_t bytes);
int dctzhip_host_unregister(dctzhip_ctx *ctx, void *ptr);

/* ---- compress stage ------------------------------------------------------ */
/* Replaces dctz-comp-lib.c:186-217 (calc_data_stat + scale, util.c:12-44),
 * :271-281 (bin ranges), :318-416 (dct_init + per-block dct_fftw + DC + pass-1
 * binning), :435-476 (QT table) and :478-544 (pass-2 exception compaction).
 *   d_in        n elements (float|double), device, 16-byte aligned; not modified
 *   d_bin_index n bytes out (bin ids; 255 = DC slot / stored exactly)
 *   d_dc        nblk floats out (USE_TRUNCATE)
 *   d_ac_exact  capacity n floats out; info->cnt of them are valid, block-major,
 *               j ascending -- byte-identical to the reference's AC_exact[]
 *   d_scaled    optional: receives x/sf (the reference's in-place scaling of the
 *               caller's buffer, :193-216); may be NULL, may alias d_in (then the
 *               call runs on verified statistics: see dctzhip_set_speculation).
 *               Like every output it is complete in STREAM order, not at return
 *               (single arrays and batches alike), unless dctzhip_set_blocking is on
 *   d_coef      optional debug tap: the DCT coefficients a_x after pass 1
 *               (= dct_result.bin under -DDCT_FILE_DEBUG, :422-428); may be NULL
 * On return *info is filled (the host has waited for it); the last kernels of the
 * call may still be running: the output buffers are complete in STREAM order --
 * for any later call or copy on the context's stream, or after dctzhip_sync(). */
int dctzhip_compress(dctzhip_ctx *ctx, const void *d_in, size_t n, int dtype,
                     double error_bound, int mode, void *d_bin_index, float *d_dc,
                     float *d_ac_exact, void *d_scaled, void *d_coef,
                     dctzhip_cinfo *info);

/* A PART of an array whose statistics the caller already has (EC mode).  The streams of elements [lo, lo + n) of an array
 * are the same whether the array is compressed in one call or part by part -- blocks are independent
 * (dctz-comp-lib.c:323-416), AC_exact is block-major (:478-544) -- provided every part is scaled by the ARRAY's scaling
 * factor (util.c:29), the one thing that couples them: max_abs / min_abs are max|x| and min|x| of the WHOLE array
 * (calc_data_stat, util.c:18-25).  Parts start on block boundaries (lo % 64 == 0); only the array's last part may end in
 * a short block.  d_bin / d_dc: the part's own positions (bin_index + lo, DC + lo / 64); d_ac: where the part's exact
 * coefficients go -- AC_exact + the counts of the parts in front; *cnt: the part's count; part_stats (or NULL): max|x|,
 * min|x| of the part and the sum of its elements from the SECOND on (util.c:22 starts at i = 1: the caller adds a part's
 * first element unless the part is the array's first); *sf (or NULL): the scaling factor used.  A part whose own extremes
 * lie outside [min_abs, max_abs] is refused (DCTZHIP_E_ARG: the statistics are not this array's). */
int dctzhip_compress_part(dctzhip_ctx *ctx, const void *d_in, size_t n, int dtype, double error_bound, double max_abs,
                          double min_abs, void *d_bin, float *d_dc, float *d_ac, uint32_t *cnt, double *part_stats, double *sf);

/* calc_data_stat alone (util.c:12-44): fills sf, mean (device order), max_abs,
 * min_abs and nblk of *info. */
int dctzhip_stats(dctzhip_ctx *ctx, const void *d_in, size_t n, int dtype, dctzhip_cinfo *info);

/* The header's `mean` in the reference's SERIAL summation order (util.c:18-28:
 * sum of x[1..N-1] in the data type, then / N) -- bit-identical to the
 * reference, which a parallel reduction cannot be.  begin() enqueues a
 * single-wavefront kernel on a side stream and returns at once; end() waits
 * for it.  d_in must stay unmodified in between.  (~0.5 s per GiB of fp64: meant
 * to run underneath the host zlib tail, which is 20x longer.) */
int dctzhip_serial_mean_begin(dctzhip_ctx *ctx, const void *d_in, size_t n, int dtype);
int dctzhip_serial_mean_end(dctzhip_ctx *ctx, double *mean);

/* x[i] /= sf in place on the device (dctz-comp-lib.c:193-216), no-op if sf == 1.
 * Synchronous. */
int dctzhip_scale_inplace(dctzhip_ctx *ctx, void *d_x, size_t n, int dtype, double sf);

/* ---- decompress stage ---------------------------------------------------- */
/* Replaces dctz-decomp-lib.c:358-361 (gen_bins, binning.c:12-50), :372-386,
 * :389-483 (de-quantise + ifft_idct per block) and :494-511 (de-scale).
 *   ac_count     number of valid floats in d_ac_exact (header.tot_AC_exact_count);
 *                a bin_index that flags more than that fails with DCTZHIP_E_ARG
 *   qtable_host  QT: the 64 table values in the data type, HOST memory (as read
 *                from the stream tail, dctz-decomp-lib.c:193-199); EC: NULL
 *   sf           header scaling factor (scaling_factor.d, or .f widened)
 *   d_out        n elements out
 * Returns once the stream is known to be consistent with ac_count (the only thing the
 * host has to learn); the reconstruction itself is complete in STREAM order -- for
 * any later call or copy on the context's stream, or after dctzhip_sync(). */
int dctzhip_decompress(dctzhip_ctx *ctx, const void *d_bin_index, const float *d_dc,
                       const float *d_ac_exact, uint32_t ac_count,
                       const void *qtable_host, size_t n, int dtype,
                       double error_bound, double sf, int mode, void *d_out);

/* ---- batches of arrays ------------------------------------------------------ */
/* The reference's own workloads are LISTS of small arrays, one dctz_compress() call and one process each
 * (tests/test-dctz.sh:13-56 over tests/list-msst19.txt:1-6: 12 960 ... 37 024 doubles; tests/list-CESM-ATM-tylor.txt:1-5).
 * On a GPU one such call is four or five launches and a host hand-off around a few microseconds of kernel work.  A batch
 * runs k arrays -- own element type, error bound, buffers and results each; nothing in the reference couples two arrays
 * (own calc_data_stat, sf, bin ranges, tot_AC_exact_count, QT table: dctz-comp-lib.c:186 onwards) -- through ONE launch
 * sequence per element type with ONE hand-off.  Every array's outputs (streams, scaled copy, *info) are bit for bit
 * those of its own dctzhip_compress() / dctzhip_decompress() call.  Arrays of 2^24 elements or more are handed to the
 * single-array path inside the call (their kernels dwarf the launch cost, and that path saves the statistics pass).
 * Fields have the meaning of the same-named arguments of dctzhip_compress / dctzhip_decompress. */
typedef struct {
  const void *d_in;          /* n elements of dtype, device, 16-byte aligned; not modified unless d_scaled aliases it */
  size_t n;
  int dtype;                 /* DCTZHIP_F32 | DCTZHIP_F64, per array */
  double error_bound;        /* per array */
  void *d_bin_index;         /* n bytes out */
  float *d_dc;               /* ceil(n / 64) floats out */
  float *d_ac_exact;         /* capacity n floats out */
  void *d_scaled;            /* optional: x / sf (dctz-comp-lib.c:193-216); may be NULL, may alias d_in */
} dctzhip_batch_citem;
typedef struct {
  const void *d_bin_index;
  const float *d_dc;
  const float *d_ac_exact;
  uint32_t ac_count;         /* header.tot_AC_exact_count of this array */
  const void *qtable_host;   /* QT: 64 values in the data type, host memory; EC: NULL */
  size_t n;
  int dtype;
  double error_bound;
  double sf;                 /* header scaling factor of this array */
  void *d_out;               /* n elements out */
} dctzhip_batch_ditem;
/* infos: k entries (may be NULL).  On return every infos[i] is filled; outputs are complete in STREAM order. */
int dctzhip_compress_batch(dctzhip_ctx *ctx, int k, const dctzhip_batch_citem *items, int mode, dctzhip_cinfo *infos);
/* status: k entries (may be NULL): DCTZHIP_OK, or DCTZHIP_E_ARG for an array whose bin_index flags more exact coefficients
 * than its ac_count provides (the call then returns DCTZHIP_E_ARG; the other arrays are reconstructed all the same). */
int dctzhip_decompress_batch(dctzhip_ctx *ctx, int k, const dctzhip_batch_ditem *items, int mode, int *status);
/* Device time of the last batch call per element-type sequence, t[DCTZHIP_F32] and t[DCTZHIP_F64] (profiling on):
 * stats_ms = statistics + scaling factors (decode: flag counts), main_ms = k_compress_batch / k_decompress_batch,
 * tail_ms = remainder blocks + QT maxima + list placement + scaled copies. */
int dctzhip_last_batch_timings(dctzhip_ctx *ctx, dctzhip_timings t[2]);

/* ---- multi-dimensional blocks (optional mode) ------------------------------ */
/* SURVEY section 8 f4.  NOT a path of the reference's library, which treats every array as flat
 * (dctz-test.c:77-91); the hint is its stand-alone experiment dct-fftw-test.c:74-97 (FFTW_REDFT10 /
 * FFTW_REDFT01 along every axis of a 2-D / 3-D array).  Here the 64 values of a block are an 8 x 8 tile
 * (ndims = 2) or a 4 x 4 x 4 tile (ndims = 3) of the array -- dims[] row-major, last axis fastest, edge tiles
 * padded by repeating the last sample -- transformed with the separable orthonormal DCT-II / DCT-III; tiles
 * are numbered row-major over the tile grid and coefficients row-major inside a tile (position 0 = DC).
 * Everything after the transform is the 1-D pipeline (DC, bins, AC_exact order, QT table per position), so the
 * three streams have the reference's meaning over nblk = prod ceil(dims[i] / edge) blocks:
 *   d_bin_index  nblk * 64 bytes,  d_dc  nblk floats,  d_ac_exact  capacity nblk * 64 floats.
 * Statistics (sf, mean) are those of the original array.  Arrays whose extents are multiples of the tile edge (and
 * below 4 GiB) are read and written in place by the big kernels, at the flat path's speed; ragged ones go through a
 * gather / scatter pass (+2 element sizes of HBM traffic per element and direction).  dctzhip_nd_blocks returns nblk
 * (0 for bad arguments). */
#define DCTZHIP_GEOM_1D 0
#define DCTZHIP_GEOM_2D 1
#define DCTZHIP_GEOM_3D 2
size_t dctzhip_nd_blocks(int ndims, const size_t *dims);
int dctzhip_compress_nd(dctzhip_ctx *ctx, const void *d_in, int ndims, const size_t *dims, int dtype,
                        double error_bound, int mode, void *d_bin_index, float *d_dc,
                        float *d_ac_exact, void *d_scaled, dctzhip_cinfo *info);
int dctzhip_decompress_nd(dctzhip_ctx *ctx, const void *d_bin_index, const float *d_dc,
                          const float *d_ac_exact, uint32_t ac_count, const void *qtable_host,
                          int ndims, const size_t *dims, int dtype, double error_bound, double sf,
                          int mode, void *d_out);

/* ---- transform only ------------------------------------------------------ */
/* Batched drop-in for dct_init + per-block dct_fftw / ifft_idct (+ the
 * remainder-length re-init), dct.h:17-27 as driven by dct-test.c:81-89, 144-152:
 * forward (inverse = 0) or inverse (inverse = 1) orthonormal DCT of every
 * 64-element block of d_in, last block of length n % 64 if non-zero. */
int dctzhip_dct_blocks(dctzhip_ctx *ctx, const void *d_in, void *d_out, size_t n,
                       int dtype, int inverse);

/* ---- multi-GPU: gather of the pre-zlib streams ------------------------------ */
/* Shards are independent dctz_compress calls (own sf, own header; nothing in the reference couples them,
 * dctz-comp-lib.c:186): one process or thread per GPU, one context each, no data-path collective.  The one exchange
 * step is bringing the three streams of every shard to the rank whose host runs the zlib tail
 * (dctz-comp-lib.c:620-760).  It runs over RCCL (loaded with dlopen on first use: single-GPU users need no RCCL):
 * an all-gather of three 64-bit sizes per rank, then grouped point-to-point sends straight to the root, so that
 * all seven inbound xGMI links of the root carry data at once (a ring would be bound by one link).
 *   dctzhip_comm_unique_id   rank 0 makes the 128-byte id and hands it to the other ranks (file, socket, MPI, ...)
 *   dctzhip_comm_create      collective over all ranks; the communicator lives in the context
 *   dctzhip_comm_sizes       collective: sizes[3 r .. 3 r + 2] = {n, nblk, cnt} of rank r, on every rank (host memory)
 *   dctzhip_comm_gather      collective: on `root`, d_*_all receive the streams of all ranks back to back in rank
 *                            order (offsets = prefix sums of `sizes`; root's own streams are copied); the other
 *                            ranks pass NULL.  Synchronous with respect to the host on return. */
#define DCTZHIP_COMM_ID_BYTES 128
int dctzhip_comm_unique_id(void *id_out);
int dctzhip_comm_create(dctzhip_ctx *ctx, int rank, int world, const void *id);
int dctzhip_comm_destroy(dctzhip_ctx *ctx);
int dctzhip_comm_sizes(dctzhip_ctx *ctx, uint64_t n, uint64_t cnt, uint64_t *sizes);
int dctzhip_comm_gather(dctzhip_ctx *ctx, int root, const void *d_bin, const float *d_dc, const float *d_ac,
                        const uint64_t *sizes, void *d_bin_all, float *d_dc_all, float *d_ac_all);

/* ---- harness metric ------------------------------------------------------- */
/* The reductions of calc_psnr (util.c:54-104) over device-resident arrays:
 * out[0] = min x, out[1] = max x (util.c:61-66 / :77-82), out[2] = max |x - r|,
 * out[3] = sum (x - r)^2 with the difference and its square taken in the data type
 * (util.c:67-73 / :83-89).  The sum is a tree reduction: its last digits differ from
 * the reference's serial loop (relative 1e-15), min / max / maxdiff are exact. */
int dctzhip_psnr_terms(dctzhip_ctx *ctx, const void *d_x, const void *d_r, size_t n, int dtype, double out[4]);

/* ---- entropy stage on the GPU -------------------------- */
/* Replaces the reference's zlib tail on the compress side -- deflateInit / deflate of bin_index, DC and AC_exact on
 * three host threads, dctz-comp-lib.c:620-732 -- with kernels: every section becomes ONE standard zlib stream
 * (RFC 1950 frame, RFC 1951 blocks) in device memory, which the reference's reader inflates unchanged
 * (dctz-decomp-lib.c:244-322).  Compressed BYTES differ from zlib's (dynamic-Huffman blocks of 16 KiB of input with
 * matches at a fixed set of distances; sizes within a few per cent of zlib level 6 on DCTZ streams), inflated CONTENT
 * is identical.
 *   nsec      sections of this call (<= 8); they share scratch and run one after the other on the context's stream
 *   d_src[i]  n[i] bytes, device (4-byte alignment gives the fast load path)
 *   d_dst[i]  cap[i] >= dctzhip_deflate_bound(n[i]) bytes, device
 *   out_len   stream lengths; the call returns after the stream has drained (it needs them on the host)
 *   chunk_sizes  NULL, or per section a host array of ceil(n[i] / dctzhip_deflate_chunk_bytes()) entries (or NULL) that
 *             receives the compressed bytes of every chunk.  The stream is 2 bytes of zlib header (78 5E), the chunks
 *             back to back, 03 00 and the adler32; every chunk is a raw deflate block sequence that starts and ends on a
 *             byte boundary and references nothing in front of itself, so a reader that knows these sizes can inflate
 *             the chunks in parallel (the drop-in library stores them as the container's "DZIX" trailer). */
size_t dctzhip_deflate_bound(size_t n);
size_t dctzhip_deflate_chunk_bytes(void);
int dctzhip_deflate(dctzhip_ctx *ctx, int nsec, const void *const *d_src, const size_t *n, void *const *d_dst,
                    const size_t *cap, size_t *out_len, uint32_t *const *chunk_sizes);
/* The same with one flag word per section (NULL: none).  DCTZHIP_DEFLATE_LITERALS: do not search the section for
 * matches -- for bytes of floats (DC, AC_exact), where zlib's own search finds next to nothing; the section is then
 * coded with its byte statistics alone (same size to 0.1 %, a fifth of the parse time). */
#define DCTZHIP_DEFLATE_LITERALS 1u
int dctzhip_deflate_ex(dctzhip_ctx *ctx, int nsec, const void *const *d_src, const size_t *n, void *const *d_dst,
                       const size_t *cap, size_t *out_len, uint32_t *const *chunk_sizes, const unsigned *flags);

/* The reader's side of the same stage: sections written by dctzhip_deflate inflated on the device, one lane per chunk
 * (replaces, for such sections, inflateInit / inflate of dctz-decomp-lib.c:244-322 and the H2D copy of the raw streams).
 *   d_z[i], zlen[i]   the section's zlib stream in device memory
 *   chunk_sizes[i]    host array: compressed bytes of every chunk (the container's "DZIX" index)
 *   raw[i], d_dst[i]  bytes the section inflates to, and where (device)
 *   *ok               1: every chunk decoded, lengths and the adler32 of the content agree with the stream;
 *                     0: something is inconsistent (or the stream is not of this kind) -- the outputs are undefined and
 *                        the caller takes the zlib path, which reports damage the way the reference does.
 * A damaged stream cannot make the kernel read or write outside the buffers described here. */
int dctzhip_inflate(dctzhip_ctx *ctx, int nsec, const void *const *d_z, const size_t *zlen, const uint32_t *const *chunk_sizes,
                    const size_t *raw, void *const *d_dst, int *ok);

/* Diagnostics: element-wise x / divisor computed (a) by the kernels' hoisted-
 * reciprocal division and (b) by the compiler's IEEE division; the two outputs
 * must be bit-identical (tests/test_gpu_parity.py::test_fast_division_is_exact). */
int dctzhip_debug_divide(dctzhip_ctx *ctx, const void *d_x, size_t n, int dtype,
                         double divisor, void *d_fast, void *d_ref);

/* Library/ABI version, "major.minor.patch". */
const char *dctzhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DCTZ_HIP_H */
