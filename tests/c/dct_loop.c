/* dct-test.c:81-89 / :145-152 in miniature: dct_init, one dct_fftw per 64-element block over an array in host memory,
 * then one ifft_idct per block -- timed per block.  Links against the drop-in library exactly as the reference's
 * dct-test.c would (dct.h:17-27).  Prints "<ns per forward block> <ns per inverse block> <max |x - back|>". */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

void dct_init(int dn);
void dct_fftw(double *a, double *b, int dn, int nblk);
void ifft_idct(int dn, double *a, double *data);
void dct_finish(void);
void idct_finish(void);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 16384;
  double *x = malloc(sizeof(double) * 64 * nblk), *c = malloc(sizeof(double) * 64 * nblk), *r = malloc(sizeof(double) * 64 * nblk);
  for (int i = 0; i < 64 * nblk; i++) x[i] = sin(0.01 * i) + 0.3 * cos(0.37 * i);
  dct_init(64);
  double t0 = now();
  for (int b = 0; b < nblk; b++) dct_fftw(x + 64 * b, c + 64 * b, 64, nblk);
  double t1 = now();
  dct_finish();
  dct_init(64);
  double t2 = now();
  for (int b = 0; b < nblk; b++) ifft_idct(64, c + 64 * b, r + 64 * b);
  double t3 = now();
  idct_finish();
  double worst = 0;
  for (int i = 0; i < 64 * nblk; i++) if (fabs(r[i] - x[i]) > worst) worst = fabs(r[i] - x[i]);
  printf("%.1f %.1f %.3g\n", (t1 - t0) / nblk * 1e9, (t3 - t2) / nblk * 1e9, worst);
  return 0;
}
