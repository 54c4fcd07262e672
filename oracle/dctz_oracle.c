/*
 * dctz_oracle.c -- CPU restatement of the DCTZ hot path (TEST INFRASTRUCTURE;
 * see dctz_oracle.h for scope, parity status and the FFTW note).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -mfma, no -ffast-math: -mfma only turns the explicit fma()
 * calls of the pinned transform flow into one instruction; nothing is contracted implicitly)
 */
#define _GNU_SOURCE /* sincos(), sincosf() */
#include "dctz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846 /* dct.h:13-15 */
#endif

#define T double
#define SUF(x) x##_f64
#define IS_F64 1
#define M_SQRT sqrt
#define M_FABS fabs
#define M_FMA __builtin_fma
#include "dctz_oracle_impl.inc"
#undef T
#undef SUF
#undef IS_F64
#undef M_SQRT
#undef M_FABS
#undef M_FMA

#define T float
#define SUF(x) x##_f32
#define IS_F64 0
#define M_SQRT sqrtf
#define M_FABS fabsf
#define M_FMA __builtin_fmaf
#include "dctz_oracle_impl.inc"
