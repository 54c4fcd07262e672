/*
 * dctz_oracle.h -- CPU restatement of the DCTZ block-DCT + binning hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under dctz_amd/ (the product) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: "parity unpinned" by reference-held fixtures -- the reference
 * (swson/DCTZ) ships no golden vectors or asserting tests, and it cannot be
 * built in this image (it needs fftw3.h / libfftw3{,f}, both absent; building
 * against a stand-in header is not allowed).  The oracle is instead anchored on
 *   (1) the known answers recorded in SURVEY.md section 8c (produced by the survey
 *       from the reference itself) -- tests/test_oracle_known_answers.py,
 *   (2) the mathematical identity "orthonormal DCT-II / DCT-III" checked against
 *       scipy.fft (an independent implementation),
 *   (3) an independent numpy restatement (oracle/np_restatement.py).
 *
 * Third-party arithmetic that is not under /root/reference: FFTW3 (Makefile:4,
 * README.md:28 suggests 3.3.10, unpinned).  The reference calls it only as
 * fftw_plan_dft_1d(n, in, out, FFTW_FORWARD|FFTW_BACKWARD) + fftw_execute
 * (dct.c:48,51,72,91,157,160,179,182).  Its published contract is the
 * un-normalised DFT  out[k] = sum_j in[j] exp(-/+ 2 pi i jk/n); that contract is
 * what `orc_dft_naive_*` restates (definition-order summation).  For n = 64 the
 * oracle additionally carries the *pinned fast flow* (`ORC_DCT_FAST`): the same
 * DFT evaluated as real-FFT-via-32-point-complex-FFT, radix 8x4, with a fixed
 * operation order and no fused multiply-add.  The HIP kernels evaluate exactly
 * that expression tree, so kernel-vs-oracle comparisons are bit-exact; the
 * fast flow is checked against the naive definition to rounding error.
 *
 * All functions follow the reference file:line cited at their definition in
 * dctz_oracle_impl.inc.
 */
#ifndef DCTZ_ORACLE_H
#define DCTZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLK 64          /* dctz.h:28  BLK_SZ  */
#define ORC_NBINS 255       /* dctz.h:65-66 NBINS for 8-bit bin ids */

enum { ORC_F32 = 0, ORC_F64 = 1 };          /* dctz.h:44-47 t_datatype */
enum { ORC_EC = 0, ORC_QT = 1 };            /* Makefile:12-17 build variants */
enum { ORC_DCT_NAIVE = 0, ORC_DCT_FAST = 1 };
/* `impl` arguments: engine | (geometry << 4).  Geometry 0 = the reference's 1-D blocks of 64 consecutive
 * elements; 1 = 8 x 8 and 2 = 4 x 4 x 4 tiles with the separable orthonormal DCT (not a path of the
 * reference's library -- pinned on the definition, scipy.fft.dctn(norm="ortho"), only). */
#define ORC_GEOM(g) ((g) << 4)

/* Statistics + scaling factor (util.c:12-44).  `sum` accumulates in the data
 * type (float for f32, util.c:31).  Outputs are written as doubles; for f32
 * they hold exactly-representable float values. */
typedef struct {
  double max, min, sum, mean, sf;
} orc_stats;

void orc_stats_f64(const double *x, size_t n, orc_stats *st);
void orc_stats_f32(const float *x, size_t n, orc_stats *st);

/* In-place scaling x[i] /= sf when sf != 1 (dctz-comp-lib.c:188-217). */
void orc_scale_f64(double *x, size_t n, double sf);
void orc_scale_f32(float *x, size_t n, float sf);

/* Orthonormal DCT-II / DCT-III of one block of length n (1..64)
 * (dct.c:24-103, 115-205; dct-float.c likewise).  impl selects the DFT engine
 * for n == 64; other lengths always use the naive DFT. */
void orc_dct_fwd_f64(const double *a, double *b, int n, int impl);
void orc_dct_inv_f64(const double *a, double *data, int n, int impl);
void orc_dct_fwd_f32(const float *a, float *b, int n, int impl);
void orc_dct_inv_f32(const float *a, float *data, int n, int impl);

/* Twiddle tables exactly as the reference builds them (dct.c:37-47, 130-134). */
void orc_dct_tables_f64(int n, double *as, double *ax, double *ias, double *iax);
void orc_dct_tables_f32(int n, float *as, float *ax, float *ias, float *iax);

/* Bin centres (binning.c:12-50). */
void orc_gen_bins_f64(double *bin_center, int nbins, double error_bound);
void orc_gen_bins_f32(float *bin_center, int nbins, float error_bound);

/* Whole compress hot path a2..a9 of SURVEY section 8 (dctz-comp-lib.c:186-544).
 *   x          : n elements, SCALED IN PLACE like the reference does
 *   bin_index  : n bytes out
 *   dc         : ceil(n/64) floats out (USE_TRUNCATE)
 *   ac_exact   : capacity n floats out, *cnt used
 *   qtable     : 64 elements of the data type out (QT: clamped table incl.
 *                slot 0 = last block's DC; EC: untouched)
 *   qtable_raw : optional, 64 elements, the table before clamping (qtable.bin)
 *   coef       : optional, n elements: DCT coefficients a_x after pass 1
 * returns 0, or -1 if error_bound < 1e-6 (dctz-comp-lib.c:135-138). */
int orc_compress_f64(double *x, size_t n, double error_bound, int mode, int impl,
                     orc_stats *st, uint8_t *bin_index, float *dc,
                     float *ac_exact, uint32_t *cnt, double *qtable,
                     double *qtable_raw, double *coef);
int orc_compress_f32(float *x, size_t n, double error_bound, int mode, int impl,
                     orc_stats *st, uint8_t *bin_index, float *dc,
                     float *ac_exact, uint32_t *cnt, float *qtable,
                     float *qtable_raw, float *coef);

/* Decompress hot path a11..a15 (dctz-decomp-lib.c:358-511).  sf is the header's
 * scaling factor (as double; f32 passes an exactly representable float). */
int orc_decompress_f64(const uint8_t *bin_index, const float *dc,
                       const float *ac_exact, const double *qtable, size_t n,
                       double error_bound, double sf, int mode, int impl,
                       double *out);
int orc_decompress_f32(const uint8_t *bin_index, const float *dc,
                       const float *ac_exact, const float *qtable, size_t n,
                       double error_bound, double sf, int mode, int impl,
                       float *out);

/* PSNR / max error (util.c:54-104).  Returns psnr; optional outputs. */
double orc_psnr_f64(const double *x, const double *r, size_t n, double *maxdiff,
                    double *rmse, double *range);
double orc_psnr_f32(const float *x, const float *r, size_t n, double *maxdiff,
                    double *rmse, double *range);

#ifdef __cplusplus
}
#endif
#endif
