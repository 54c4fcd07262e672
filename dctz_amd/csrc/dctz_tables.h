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

#include "dct64_lane.h"

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

// The TAB_* block of dct64_lane.h for the 64-point fast path.
template <typename T>
inline void fill_tab64(T* tab) {
  std::memset(tab, 0, sizeof(T) * TAB_SIZE);
  T as[64], ax[64], ias[64], iax[64];
  reference_twiddles<T>(64, as, ax, ias, iax);
  for (int k = 0; k < 64; k++) {
    tab[TAB_HS + k] = (T)0.5 * as[k];
    tab[TAB_HX + k] = (T)0.5 * ax[k];
    tab[TAB_IAS + k] = ias[k];
    tab[TAB_IAX + k] = iax[k];
  }
  tab[TAB_IAS + 0] = ias[0] / sqrt2<T>();          // dct.c:166  ias_0 = ias[0]/sqrt(2)
  // The inverse transform ends with a division by 2n = 128 (dct.c:185-186 "/dn" and the factor 2
  // carried by G).  A power of two commutes exactly with every rounding on the way (no underflow:
  // |coef| >= FLT_MIN or 0, the table entries are O(10)), so it is folded into these two tables:
  // same bits out, 16 multiplications per block and lane less.
  for (int k = 0; k < 64; k++) {
    tab[TAB_IAS + k] = tab[TAB_IAS + k] * (T)(1.0 / 128.0);
    tab[TAB_IAX + k] = tab[TAB_IAX + k] * (T)(1.0 / 128.0);
  }
  const T r = (T)sqrt(0.5);
  tab[TAB_R] = r;
  for (int n2 = 0; n2 < 4; n2++)
    for (int k1 = 0; k1 < 8; k1++) {
      const int t = n2 * k1;
      T c, s;
      if (t % 8 == 0) {                            // multiples of pi/2: exact
        const int quad = (t / 8) % 4;
        c = (T)(quad == 0 ? 1 : quad == 2 ? -1 : 0);
        s = (T)(quad == 1 ? 1 : quad == 3 ? -1 : 0);
      } else if (t % 4 == 0) {                     // odd multiples of pi/4: +-r
        const int o = (t / 4) % 8;
        c = (o == 1 || o == 7) ? r : -r;
        s = (o == 1 || o == 3) ? r : -r;
      } else {
        double cd, sd;
        ::sincos(2.0 * M_PI * t / 32.0, &sd, &cd);
        c = (T)cd;
        s = (T)sd;
      }
      tab[TAB_W32R + n2 * 8 + k1] = c;
      tab[TAB_W32I + n2 * 8 + k1] = s;
    }
  for (int k = 0; k <= 16; k++) {
    double cd, sd;
    ::sincos(2.0 * M_PI * k / 64.0, &sd, &cd);
    T c = (T)cd, s = (T)sd;
    if (k == 0) { c = (T)1; s = (T)0; }
    if (k == 8) { c = r; s = r; }
    if (k == 16) { c = (T)0; s = (T)1; }
    tab[TAB_CW + k] = c;
    tab[TAB_SW + k] = s;
  }
  for (int k = 17; k < 32; k++) {                  // exact mirror symmetry
    tab[TAB_CW + k] = -tab[TAB_CW + 32 - k];
    tab[TAB_SW + k] = tab[TAB_SW + 32 - k];
  }
}

// Tables for the remainder block (length l = 1..63, transformed with an l- or
// 2l-point DFT exactly like dct.c:59-72 / 144-164 do through FFTW).
// Layout (elements of T): as[64] ax[64] ias[64] iax[64] wr[128] wi[128];
// ias[0] is stored already adjusted (/sqrt2 for even l, *sqrt2 for odd l).
// (offsets RTAB_* live in dct64_lane.h)
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
