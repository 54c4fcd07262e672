// dctz_tables.h -- host-side construction of the constant tables the kernels use.
//
// The reference builds its twiddles with the host libm (dct.c:37-47 forward,
// dct.c:130-134 inverse; dct-float.c:39-47, 134-136 in single precision with
// cosf/sinf/sqrtf).  We evaluate the very same expressions on the host, so a
// given machine gets the very same as[]/ax[]/ias[]/iax[] values the reference
// would, and upload them; the device never calls a transcendental.
#pragma once

#include <cmath>
#include <cstring>
#include <type_traits>

#include "dct64_block.h"
#include "dct_nd_block.h"
#include "dct64_block_pk.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846 /* dct.h:13-15 */
#endif

namespace dctz {

// as/ax/ias/iax for a block of length n (1..64).
template <typename T>
inline void reference_twiddles(int n, T* as, T* ax, T* ias, T* iax) {
  for (int i = 0; i < n; i++) {
    // The reference takes cos(y) and sin(y) of one argument; gcc -O3 (its
    // Makefile:1-2) lowers the pair to one sincos()/sincosf() call, whose last
    // bit can differ from separate calls.  We call it explicitly so the table
    // does not depend on which compiler built this file.
    if constexpr (std::is_same<T, double>::value) {
      double y = -i * M_PI / (2 * n), c, s, ci, si;
      ::sincos(y, &s, &c);
      as[i] = (T)(c / sqrt(2.0 * n));
      ax[i] = (T)(s / sqrt(2.0 * n));
      double yi = i * M_PI / (2 * n);
      ::sincos(yi, &si, &ci);
      ias[i] = (T)(ci * sqrt(2.0 * n));
      iax[i] = (T)(si * sqrt(2.0 * n));
    } else {
      float y = -i * (float)M_PI / (2 * n), c, s, ci, si;
      ::sincosf(y, &s, &c);
      as[i] = c / sqrtf(2.0 * n);
      ax[i] = s / sqrtf(2.0 * n);
      float yi = i * (float)M_PI / (2 * n);
      ::sincosf(yi, &si, &ci);
      ias[i] = ci * sqrtf(2.0 * n);
      iax[i] = si * sqrtf(2.0 * n);
    }
  }
  as[0] = as[0] / (std::is_same<T, double>::value ? (T)sqrt(2.0) : (T)sqrtf(2.0));
  if (n % 2 == 0)
    for (int i = 0; i < n; i++) { as[i] = as[i] * 2; ax[i] = ax[i] * 2; }
}

template <typename T>
inline T sqrt2() { return std::is_same<T, double>::value ? (T)sqrt(2.0) : (T)sqrtf(2.0); }

// The TB_* block of dct64_block.h for the 64-point fast path (TB_TOTAL elements: the constants of dct_nd_block.h follow it).  Exact definitions (theta_j = j pi / 128,
// cw = cos(2 pi k / 64), sw = sin(2 pi k / 64)):
//   forward:  alpha_j = cos(theta_j) / sqrt(128) = as[j] / 2,  beta_j = sin(theta_j) / sqrt(128) = -ax[j] / 2   (dct.c:37-47)
//   inverse:  C_j = sqrt(128) cos(theta_j) = ias[j],           S_j = sqrt(128) sin(theta_j) = iax[j]            (dct.c:130-134)
// and the merged split constants are products / sums of those, evaluated in long double and rounded to T
// ONCE (the reference rounds every factor to T first; both are rounding-level restatements of the same
// real numbers, and neither can reproduce FFTW's own internal twiddles).
template <typename T>
inline void fill_tab_block(T* tab) {
  typedef long double L;
  std::memset(tab, 0, sizeof(T) * TBP_TOTAL);
  const L pi = 3.141592653589793238462643383279502884L;
  // multi-dimensional blocks (dct_nd_block.h): the 8- and 4-point orthonormal DCT constants, rounded once
  tab[TB_ND_R8] = (T)sqrtl((L)0.125);
  tab[TB_ND_G1] = (T)(cosl(pi / 8) / 2);      tab[TB_ND_G3] = (T)(cosl(3 * pi / 8) / 2);
  tab[TB_ND_C1] = (T)(cosl(pi / 16) / 2);     tab[TB_ND_C3] = (T)(cosl(3 * pi / 16) / 2);
  tab[TB_ND_C5] = (T)(cosl(5 * pi / 16) / 2); tab[TB_ND_C7] = (T)(cosl(7 * pi / 16) / 2);
  tab[TB_ND_H1] = (T)(sqrtl((L)0.5) * cosl(pi / 8)); tab[TB_ND_H3] = (T)(sqrtl((L)0.5) * cosl(3 * pi / 8));
  const L rt128 = sqrtl((L)128);
  auto al = [&](int j) { return cosl(j * pi / 128) / rt128; };
  auto be = [&](int j) { return sinl(j * pi / 128) / rt128; };
  auto Cc = [&](int j) { return cosl(j * pi / 128) * rt128 / 128; };      // inverse: the final 1/128 is folded in
  auto Ss = [&](int j) { return sinl(j * pi / 128) * rt128 / 128; };
  tab[TB_R] = (T)sqrtl((L)0.5);
  for (int n2 = 1; n2 < 4; n2++)
    for (int k1 = 1; k1 < 8; k1++) {
      const int t = n2 * k1;                        // exp(-+ 2 pi i t / 32)
      L c = cosl(2 * pi * t / 32), s = sinl(2 * pi * t / 32);
      if (t % 8 == 0) {                             // multiples of pi/2: exact
        const int quad = (t / 8) % 4;
        c = (quad == 0) ? 1 : (quad == 2) ? -1 : 0;
        s = (quad == 1) ? 1 : (quad == 3) ? -1 : 0;
      }
      tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2] = (T)c;
      tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1] = (T)s;
    }
  for (int k = 1; k < 16; k++) {
    const L cw = cosl(2 * pi * k / 64), sw = sinl(2 * pi * k / 64), m = 1 - sw, pl = 1 + sw;
    T* f = tab + TB_FS + 16 * (k - 1);
    // rows over (Re Z[k], Im Z[k], Re Z[32-k], Im Z[32-k])
    auto rowA = [&](T* o, L a, L b) {              // b[k]-type row: own pair first
      o[0] = (T)(a * m - b * cw); o[1] = (T)(a * cw + b * m); o[2] = (T)(a * pl + b * cw); o[3] = (T)(a * cw - b * pl);
    };
    auto rowB = [&](T* o, L a, L b) {              // b[32-k]-type row: roles of the pair swapped, cw -> -cw
      o[0] = (T)(a * pl - b * cw); o[1] = (T)(-a * cw - b * pl); o[2] = (T)(a * m + b * cw); o[3] = (T)(-a * cw + b * m);
    };
    rowA(f + 0, al(k), be(k));                     // b[k]    = alpha_k Re V[k] + beta_k Im V[k]          (dct.c:100-102)
    rowA(f + 4, al(64 - k), -be(64 - k));          // b[64-k]: V[64-k] = conj V[k]
    rowB(f + 8, al(32 - k), be(32 - k));           // b[32-k]
    rowB(f + 12, al(32 + k), -be(32 + k));         // b[32+k]: V[32+k] = conj V[32-k]
    T* g = tab + TB_IS + 16 * (k - 1);
    // rows over (a[k], a[64-k], a[32-k], a[32+k])
    g[0] = (T)(Cc(k) * m - cw * Ss(k));            g[1] = (T)(Cc(64 - k) * m + cw * Ss(64 - k));
    g[2] = (T)(Cc(32 - k) * pl - cw * Ss(32 - k)); g[3] = (T)(Cc(32 + k) * pl + cw * Ss(32 + k));          // Re Zb[k]
    g[4] = (T)(Ss(k) * m + cw * Cc(k));            g[5] = (T)(-Ss(64 - k) * m + cw * Cc(64 - k));
    g[6] = (T)(-Ss(32 - k) * pl - cw * Cc(32 - k)); g[7] = (T)(Ss(32 + k) * pl - cw * Cc(32 + k));         // Im Zb[k]
    g[8] = (T)(Cc(k) * pl + cw * Ss(k));           g[9] = (T)(Cc(64 - k) * pl - cw * Ss(64 - k));
    g[10] = (T)(Cc(32 - k) * m + cw * Ss(32 - k)); g[11] = (T)(Cc(32 + k) * m - cw * Ss(32 + k));          // Re Zb[32-k]
    g[12] = (T)(-Ss(k) * pl + cw * Cc(k));         g[13] = (T)(Ss(64 - k) * pl + cw * Cc(64 - k));
    g[14] = (T)(Ss(32 - k) * m - cw * Cc(32 - k)); g[15] = (T)(-Ss(32 + k) * m - cw * Cc(32 + k));         // Im Zb[32-k]
  }
  tab[TB_FS16 + 0] = (T)(2 * al(16)); tab[TB_FS16 + 1] = (T)(-2 * be(16));     // b[16]
  tab[TB_FS16 + 2] = (T)(2 * al(48)); tab[TB_FS16 + 3] = (T)(2 * be(48));      // b[48]
  tab[TB_IS16 + 0] = (T)(2 * Cc(16)); tab[TB_IS16 + 1] = (T)(2 * Cc(48));      // Re Zb[16]
  tab[TB_IS16 + 2] = (T)(-2 * Ss(16)); tab[TB_IS16 + 3] = (T)(2 * Ss(48));     // Im Zb[16]
  // the same numbers once more, two rows side by side, for the packed fp32 transform (dct64_block_pk.h): pair j of a
  // group = (row 2g, row 2g + 1) x constant j
  for (int k = 1; k < 16; k++)
    for (int g = 0; g < 2; g++)
      for (int j = 0; j < 4; j++) {
        tab[TBP_FS + 16 * (k - 1) + 8 * g + 2 * j] = tab[TB_FS + 16 * (k - 1) + 8 * g + j];
        tab[TBP_FS + 16 * (k - 1) + 8 * g + 2 * j + 1] = tab[TB_FS + 16 * (k - 1) + 8 * g + 4 + j];
        tab[TBP_IS + 16 * (k - 1) + 8 * g + 2 * j] = tab[TB_IS + 16 * (k - 1) + 8 * g + j];
        tab[TBP_IS + 16 * (k - 1) + 8 * g + 2 * j + 1] = tab[TB_IS + 16 * (k - 1) + 8 * g + 4 + j];
      }
  tab[TBP_FS16 + 0] = tab[TB_FS16 + 0]; tab[TBP_FS16 + 1] = tab[TB_FS16 + 2]; tab[TBP_FS16 + 2] = tab[TB_FS16 + 1]; tab[TBP_FS16 + 3] = tab[TB_FS16 + 3];
  tab[TBP_IS16 + 0] = tab[TB_IS16 + 0]; tab[TBP_IS16 + 1] = tab[TB_IS16 + 2]; tab[TBP_IS16 + 2] = tab[TB_IS16 + 1]; tab[TBP_IS16 + 3] = tab[TB_IS16 + 3];
}

// util.c:29 / util.c:43, with the host libm exactly like the reference
inline double scaling_factor(int dtype, double max_abs) {
  if (max_abs == 0.0) return 1.0;               // documented deviation: reference divides by 0
  if (dtype == 1 /* DCTZHIP_F64 */) return pow(10, ceil(log10(max_abs)) - 1);
  return (double)powf(10, ceil(log10f((float)max_abs)) - 1);
}


// Decade tables for the device-side choice of sf (SfTable): for every decade k, the LARGEST value m of the data type
// with ceil(log10(m)) <= k under THIS host's log10 / log10f, and 10^(k-1) under its pow / powf -- so that
// pw[#{k : thr[k] < max}] is what scaling_factor(max) returns, by construction, rounding quirks of libm included.
template <typename T>
inline void decade_tables(int kmin, int kmax, double* thr, double* pw) {
  const bool f64 = sizeof(T) == 8;
  typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type Bits;
  auto cl = [&](T m) -> double { return f64 ? ceil(log10((double)m)) : ceil((double)log10f((float)m)); };
  auto p10 = [&](int k) -> T { return f64 ? (T)pow(10, (double)k) : (T)powf(10, (float)k); };
  auto bits = [](T v) { Bits b; std::memcpy(&b, &v, sizeof(T)); return b; };
  auto val = [](Bits b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; };
  const T big = f64 ? (T)1.79769313486231570815e308 : (T)3.40282346638528859812e38f;
  const Bits bmin = 1, bmax = bits(big);               // smallest subnormal .. largest finite (positive values order like their bits)
  for (int k = kmin; k <= kmax; k++) {
    pw[k - kmin] = (double)p10(k - 1);
    // largest m with ceil(log10(m)) <= k: bisection over the bit patterns (log10 is monotonic)
    if (cl(val(bmin)) > (double)k) { thr[k - kmin] = 0.0; continue; }               // the decade lies below the smallest subnormal
    if (cl(val(bmax)) <= (double)k) { thr[k - kmin] = (double)INFINITY; continue; } // ... reaches past the largest finite value
    Bits lo = bmin, hi = bmax;                         // invariant: cl(lo) <= k < cl(hi)
    while (hi - lo > 1) {
      const Bits mid = lo + (hi - lo) / 2;
      if (cl(val(mid)) <= (double)k) lo = mid; else hi = mid;
    }
    thr[k - kmin] = (double)val(lo);
  }
  pw[kmax - kmin + 1] = (double)p10(kmax);
}

// Tables for the remainder block (length l = 1..63, transformed with an l- or
// 2l-point DFT exactly like dct.c:59-72 / 144-164 do through FFTW).
// Layout (elements of T): as[64] ax[64] ias[64] iax[64] wr[128] wi[128];
// ias[0] is stored already adjusted (/sqrt2 for even l, *sqrt2 for odd l).
// (offsets RTAB_* live in dct64_block.h)
template <typename T>
inline void fill_rem_tab(int l, T* tab) {
  std::memset(tab, 0, sizeof(T) * RTAB_SIZE);
  if (l <= 0) return;
  reference_twiddles<T>(l, tab + RTAB_AS, tab + RTAB_AX, tab + RTAB_IAS, tab + RTAB_IAX);
  if (l % 2) tab[RTAB_IAS] = tab[RTAB_IAS] * sqrt2<T>();   // dct.c:145
  else       tab[RTAB_IAS] = tab[RTAB_IAS] / sqrt2<T>();   // dct.c:166
  const int N = (l % 2) ? 2 * l : l;
  for (int t = 0; t < N; t++) {
    double cd, sd;
    ::sincos(2.0 * M_PI * (double)t / (double)N, &sd, &cd);
    tab[RTAB_WR + t] = (T)cd;
    tab[RTAB_WI + t] = (T)sd;
  }
}

}  // namespace dctz
