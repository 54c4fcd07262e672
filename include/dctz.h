/*
 * dctz.h -- public API of the drop-in host library libdctz-{ec,qt}.so.
 *
 * Binary- and source-compatible with the reference's dctz.h (types, constants
 * and prototypes at dctz.h:28-128) and dct.h (dct.h:17-27), so a caller such as
 * dctz-test.c or the Z-checker glue compiles and links against this library
 * unchanged.  The block-DCT + binning stages behind dctz_compress() /
 * dctz_decompress() run on an MI355X through include/dctz_hip.h; the zlib tail
 * and the container stay on the host.
 *
 * Build-time variants, as in the reference Makefile:12-17: USE_TRUNCATE is
 * always defined (DC / AC_exact stored as float); USE_QTABLE selects the QT
 * library and adds header.bindex_count.
 *
 * Differences from the reference header: fftw3.h is not included (nothing here
 * needs FFTW), and the never-defined ceili() prototype (dctz.h:122) is omitted.
 */
#ifndef _DCTZ_H_
#define _DCTZ_H_

#include <math.h>
#include <memory.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "zlib.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DCTZ_VERSION "0.2.2"          /* wire-format version we interoperate with */
#define DCTZ_VERSION_MAJOR 0
#define DCTZ_VERSION_MINOR 2
#define DCTZ_VERSION_PATCH 2

#ifndef M_PI
#define M_PI 3.14159265358979323846   /* dct.h:13-15 */
#endif

#define BLK_SZ 64                     /* elements per block              (dctz.h:28) */
#define BRSF 1.0                      /* bin range scaling factor        (dctz.h:29) */
#define SF_ADJ_AMT 1                  /* scaling factor exponent offset  (dctz.h:30) */

#ifndef MAX
#define MAX(a, b) ({ __typeof__(a) a_ = (a); __typeof__(b) b_ = (b); a_ > b_ ? a_ : b_; })
#endif
#ifndef MIN
#define MIN(a, b) ({ __typeof__(a) a_ = (a); __typeof__(b) b_ = (b); a_ < b_ ? a_ : b_; })
#endif
#define CEIL(a, b) (a + b - 1) / b    /* as in dctz.h:42 (unparenthesised) */

/* element type tag (dctz.h:44-47) */
typedef enum { FLOAT = 0, DOUBLE } t_datatype;

/* a typed view of a caller-owned buffer (dctz.h:49-59); 32 bytes on LP64 */
typedef struct {
  t_datatype datatype;
  double err_bound;
  char *var_name;
  union {
    float *f;
    double *d;
  } buf;
} t_var;

typedef unsigned char t_bin_id;                 /* dctz.h:63 */
#define NBITS (sizeof(t_bin_id) << 3)           /* dctz.h:65 */
#define NBINS ((1 << (NBITS)) - 1)              /* 255; id 255 = "stored exactly" */

/* statistics of one array (dctz.h:68-94) */
typedef struct {
  union { double d; float f; } mean;
  union { double d; float f; } min;
  union { double d; float f; } max;
  union { double d; float f; } range;
  union { double d; float f; } sf;
} t_bstat;

/* container header, 56 bytes, native endianness (dctz.h:96-119) */
struct header {
  t_datatype datatype;
  unsigned int num_elements;
  double error_bound;
  unsigned int tot_AC_exact_count;
  union { double d; float f; } scaling_factor;
  union { double d; float f; } mean;
  unsigned int bindex_sz_compressed;
  unsigned int DC_sz_compressed;
  unsigned int AC_exact_sz_compressed;
#ifdef USE_QTABLE
  unsigned int bindex_count;
#endif
};

/* dctz.h:121-128 */
void calc_data_stat(t_var *in, t_bstat *bs, int N);
void gen_bins(double min, double max, double *bin_center, int nbins, double error_bound);
void gen_bins_f(float min, float max, float *bin_center, int nbins, float error_bound);
void *compress_thread(void *arg);
int dctz_compress(t_var *var, int N, size_t *outSize, t_var *var_z, double error_bound);
int dctz_decompress(t_var *var_z, t_var *var_r);
double calc_psnr(t_var *var, t_var *var_r, int N, double error_bound);

/* dct.h:17-27 -- per-block transform entry points (driven by dct-test.c) */
void dct_init(int dn);
void dct_init_f(int dn);
void dct_fftw(double *a, double *b, int dn, int nblk);
void dct_fftw_f(float *a, float *b, int dn, int nblk);
void ifft_idct(int dn, double *a, double *data);
void ifft_idct_f(int dn, float *a, float *data);
void dct_finish(void);
void dct_finish_f(void);
void idct_finish(void);
void idct_finish_f(void);

/* ---- additions (not in the reference) ------------------------------------ */
/* Whole-array forms of dct_fftw / ifft_idct: every 64-element block of a[0..n)
 * (last one of length n % 64) in one GPU pass; what dct-test.c:81-89 / 144-152
 * compute with a loop. */
void dctz_dct_blocks(double *a, double *b, size_t n, int inverse);
void dctz_dct_blocks_f(float *a, float *b, size_t n, int inverse);
/* Lists of arrays (ADDITIONS, round 4).  The reference's own workloads are lists of small arrays, one dctz_compress()
 * call -- one process -- per array (tests/test-dctz.sh:13-56 over tests/list-msst19.txt:1-6).  These take k arrays at once:
 * one copy in, ONE batch launch (every array with its own calc_data_stat, scaling factor, bin ranges, tot_AC_exact_count
 * and QT table: dctz-comp-lib.c:186 onwards couples nothing across arrays), one copy back, and the reference's tail -- one
 * single-shot deflate per section, dctz-comp-lib.c:620-732 -- for all 3 k sections on a pool of host threads.  Arguments
 * as dctz_compress()'s, per array: vars[i] (N[i] elements, scaled in place by its sf on return), vars_z[i] (>= N[i] *
 * type_size bytes), outSizes[i], error_bounds[i].  Container i is byte for byte what dctz_compress(vars[i], ...) writes
 * with the default tail.  Flat blocks only.  dctz_decompress_batch() is the mirror image (any flat containers). */
int dctz_compress_batch(int k, t_var *const *vars, const int *N, size_t *outSizes, t_var *const *vars_z, const double *error_bounds);
int dctz_decompress_batch(int k, t_var *const *vars_z, t_var *const *vars_r);
/* Multi-dimensional blocks (optional; SURVEY section 8 f4 -- NOT in the reference, whose library flattens every
 * array, dctz-test.c:77-91; the hint is its FFTW r2r experiment dct-fftw-test.c:74-97).  The NEXT dctz_compress call
 * treats var->buf as a row-major ndims-dimensional array (ndims = 2: 8 x 8 tiles, ndims = 3: 4 x 4 x 4 tiles, last
 * extent fastest, product = N) and transforms tiles with the separable orthonormal DCT instead of runs of 64
 * consecutive elements; the call after that is flat again.  Environment DCTZ_BLOCK_DIMS="1800x3600" does the same
 * for every call whose N matches (callers that cannot be changed).  ndims <= 1 cancels.  Returns 0, or -1 for
 * bad arguments.
 * Container of such a call: header.datatype carries the geometry in bits 8..15 (DCTZ_GEOM_OF), the three
 * sections cover nblk * 64 positions (edge tiles are padded), and 16 bytes follow the last section:
 * "DZND" + three 32-bit extents.  The reference's decoder cannot read it; dctz_decompress here reads both. */
int dctz_set_block_dims(int ndims, const size_t *dims);
#define DCTZ_GEOM_SHIFT 8
#define DCTZ_GEOM_OF(datatype) (((unsigned)(datatype) >> DCTZ_GEOM_SHIFT) & 0xffu)   /* 0: flat; 2, 3: number of axes */
#define DCTZ_TYPE_OF(datatype) ((t_datatype)((unsigned)(datatype) & 0xffu))
#define DCTZ_ND_MAGIC 0x444E5A44u   /* "DZND" little-endian */
/* Chunk index (optional trailer, written when the entropy stage ran on the GPU: DCTZ_ZLIB_GPU=1).  The three sections
 * are then standard zlib streams whose deflate blocks are independent chunks (header bytes 78 5E instead of zlib's
 * 78 9C; include/dctz_hip.h: dctzhip_deflate), and behind everything else of the container follow
 *   "DZIX" | u32 chunk bytes | u32 chunks of section 0, 1, 2 | u16 compressed bytes of every chunk ... | pad to 4.
 * The reference's reader never looks there (it inflates each section as one stream, dctz-decomp-lib.c:244-322);
 * dctz_decompress here looks for the trailer only when all three sections start 78 5E (zlib itself writes 78 9C at the
 * reference's settings), validates it against the header (chunk counts, sizes that tile each stream exactly up to its
 * 03 00 + adler32) and then inflates the chunks side by side on DCTZ_ZLIB_THREADS host threads (default: the cores) --
 * or on the device with DCTZ_INFLATE_GPU=1; anything inconsistent -- an index that does not describe the sections, a chunk
 * that does not inflate, content whose adler32 is not the stream's -- falls back to the ordinary one-stream inflate, which
 * treats damage as the reference's reader does.
 * dctz_decompress() is not told the size of var_z->buf (dctz.h:127), so it can only TRUST that a container whose three
 * sections start 78 5E is followed by its index (20 bytes + 2 per chunk): for containers of unknown origin -- zlib at levels
 * 2 .. 5 writes 78 5E too -- call dctz_check_container(z, zbytes, ...) first, which has the size and refuses a marked
 * container whose index is missing, cut short or does not tile the sections. */
#define DCTZ_IX_MAGIC 0x58495A44u   /* "DZIX" little-endian */
/* Stage timers of the last dctz_compress / dctz_decompress call, seconds
 * (the reference's -DTIME_DEBUG split, dctz-comp-lib.c:762-773). */
typedef struct {
  double h2d_s, gpu_s, d2h_s, zlib_s, total_s;
} dctz_stage_times;
void dctz_last_stage_times(dctz_stage_times *t);
/* The zlib tail (dctz-comp-lib.c:620-732) as a chunked multi-threaded deflate that still
 * yields ONE standard zlib stream, so dctz-decomp-lib.c:244-322 inflates it unchanged
 * (csrc/pdeflate.c).  dctz_compress uses it when the environment has
 * DCTZ_ZLIB_THREADS > 3; exported for tools that write DCTZ containers themselves.
 * cap >= dctz_pdeflate_bound(n, chunk); returns 0 on success. */
/* Bounds / plausibility check of a DCTZ container held in `zbytes` bytes, to be run before
 * dctz_decompress(), which -- like the reference (dctz-decomp-lib.c:84-100) -- trusts the header.
 * max_elements > 0 additionally bounds N (the size of the caller's output buffer); deep != 0 also
 * inflates the three sections and checks their sizes.  No GPU involved.  The library variant must
 * match the file's (EC vs QT): the QT check expects the trailing table and bindex_count. */
#define DCTZ_CHECK_OK 0
#define DCTZ_CHECK_TRUNCATED (-1)     /* fewer bytes than the header describes                */
#define DCTZ_CHECK_BAD_HEADER (-2)    /* a field is impossible (datatype, N, error bound, cnt) */
#define DCTZ_CHECK_TOO_LARGE (-3)     /* N exceeds max_elements                                */
#define DCTZ_CHECK_BAD_STREAM (-4)    /* a section does not inflate to its expected size       */
int dctz_check_container(const void *z, size_t zbytes, int max_elements, int deep);
size_t dctz_pdeflate_bound(size_t n, size_t chunk);
int dctz_pdeflate(const void *src, size_t n, void *dst, size_t cap, size_t *out_len, int threads, size_t chunk);
void dctz_pdeflate_set_level(int level);   /* 1..9; anything else = zlib's default level (the reference's) */

#ifdef __cplusplus
}
#endif
#endif /* _DCTZ_H_ */
