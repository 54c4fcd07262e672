// dct64_block_eo.h -- the forward 64-point transform of dct64_block.h cut into TWO HALVES that share no arithmetic:
// the half that yields the 32 even-numbered coefficients and the half that yields the 32 odd-numbered ones.
//
// Why it can be cut there.  dct64_fwd() runs radix-8 over n1 (four times, n2 = 0..3), twiddles, radix-4 over n2 (for
// every k1 = 0..7), and the merged split step, which pairs Z[k] with Z[32 - k].  With k = k1 + 8 k2:
//   * the radix-8 butterfly is decimation in frequency: its outputs k1 in {0, 2, 4, 6} are functions of the four SUMS
//     z[m] + z[m + 16] alone, its outputs k1 in {1, 3, 5, 7} of the four DIFFERENCES z[m] - z[m + 16] alone;
//   * the radix-4 stage never mixes two k1;
//   * 32 - k has the parity of k, and the four coefficients the split step makes of Z[k], Z[32 - k] -- b[k], b[64 - k],
//     b[32 - k], b[32 + k] -- have that parity too.
// So the operations of dct64_fwd() fall into two disjoint sets, one per parity of k1, and each set is a function of 32
// inputs (16 complex sums, resp. differences).  Every operation below IS an operation of dct64_fwd() -- same operands,
// same instruction, same rounding -- so the two halves together give dct64_fwd()'s coefficients bit for bit
// (tests/test_lane_emulation.py::test_even_odd_halves_are_the_whole_transform, on the CPU, both element types).
//
// Who uses it: k_compress_eo (dctz_kernels_eo.hip), where a tile's blocks belong to TWO wavefronts -- lane b of the
// "even" wave and lane b of the "odd" wave share block b -- so that a lane carries 32 values instead of 64 and three
// waves fit a SIMD where dct64_fwd()'s 64 values per lane allow two.  The reference code replaced is the same:
// dct_fftw(), dct.c:55-103 (dct-float.c likewise).
#pragma once
#include "dct64_block.h"

namespace dctz {

enum : int { EO_EVEN = 0, EO_ODD = 1 };

// The raw elements behind packed point m's first butterfly (m = 0..15), with z[m] = a[pack_pos(m, 0)] + i a[pack_pos(m, 1)]
// (dct.c:75-83 composed with the pairing): z[m] +- z[m + 16] = (a[4m] +- a[63 - 4m]) + i (a[4m + 2] +- a[61 - 4m]).
DCTZ_HD constexpr int eo_lhs(int m, int c) { return pack_pos(m, c); }          // 4m, 4m + 2
DCTZ_HD constexpr int eo_rhs(int m, int c) { return pack_pos(m + 16, c); }     // 63 - 4m, 61 - 4m

// ROLE = EO_EVEN: pr/pi[m] = z[m] + z[m + 16] (of the scaled block), out[i] = coefficient 2 i.
// ROLE = EO_ODD:  pr/pi[m] = z[m] - z[m + 16],                        out[i] = coefficient 2 i + 1.
template <typename T, int ROLE, typename TabPtr, bool FENCED = false>
DCTZ_HD void dct64_fwd_half(const T (&pr)[16], const T (&pi)[16], T (&out)[32], TabPtr tab) {
  T Yr[4][4], Yi[4][4];                              // [n2][h]: k1 = 2 h + ROLE
  const T r = tab[TB_R];
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    T yr[4], yi[4];
    if (ROLE == EO_EVEN) {
      // fft8<FWD>'s a0, a2, a4, a6 are the sums: a0 = p[n2], a4 = p[4 + n2], a2 = p[8 + n2], a6 = p[12 + n2]
      const T a0r = pr[n2], a0i = pi[n2], a4r = pr[4 + n2], a4i = pi[4 + n2];
      const T a2r = pr[8 + n2], a2i = pi[8 + n2], a6r = pr[12 + n2], a6i = pi[12 + n2];
      const T E0r = a0r + a2r, E0i = a0i + a2i, E2r = a0r - a2r, E2i = a0i - a2i;
      const T O0r = a4r + a6r, O0i = a4i + a6i, O2r = a4r - a6r, O2i = a4i - a6i;
      yr[0] = E0r + O0r; yi[0] = E0i + O0i;          // k1 = 0
      yr[1] = E2r + O2i; yi[1] = E2i - O2r;          // k1 = 2
      yr[2] = E0r - O0r; yi[2] = E0i - O0i;          // k1 = 4
      yr[3] = E2r - O2i; yi[3] = E2i + O2r;          // k1 = 6
    } else {
      // ... a1, a3, a5, a7 the differences: a1 = p[n2], a5 = p[4 + n2], a3 = p[8 + n2], a7 = p[12 + n2]
      const T a1r = pr[n2], a1i = pi[n2], a5r = pr[4 + n2], a5i = pi[4 + n2];
      const T a3r = pr[8 + n2], a3i = pi[8 + n2], a7r = pr[12 + n2], a7i = pi[12 + n2];
      const T E1r = a1r + a3i, E1i = a1i - a3r, E3r = a1r - a3i, E3i = a1i + a3r;
      const T O1r = a5r + a7i, O1i = a5i - a7r, O3r = a5r - a7i, O3i = a5i + a7r;
      const T t1r = (O1r + O1i) * r, t1i = (O1i - O1r) * r;
      const T t3r = (O3i - O3r) * r, t3i = -((O3r + O3i) * r);
      yr[0] = E1r + t1r; yi[0] = E1i + t1i;          // k1 = 1
      yr[1] = E3r + t3r; yi[1] = E3i + t3i;          // k1 = 3
      yr[2] = E1r - t1r; yi[2] = E1i - t1i;          // k1 = 5
      yr[3] = E3r - t3r; yi[3] = E3i - t3i;          // k1 = 7
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int k1 = 2 * h + ROLE;
      if (n2 == 0 || k1 == 0) { Yr[n2][h] = yr[h]; Yi[n2][h] = yi[h]; continue; }
      const T wr = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2], wi = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1];
      Yr[n2][h] = fma_(yi[h], wi, yr[h] * wr);             // times exp(-i 2 pi n2 k1 / 32)
      Yi[n2][h] = fma_(-yr[h], wi, yi[h] * wr);
    }
    DCT64_FENCE();
  }
  T Zr[32], Zi[32];                                  // (only the entries of this half's parity exist)
#pragma unroll
  for (int h = 0; h < 4; h++) {                      // radix-4 over n2: Z[k1 + 8 k2]
    const int k1 = 2 * h + ROLE;
    const T ar = Yr[0][h] + Yr[2][h], ai = Yi[0][h] + Yi[2][h], br = Yr[0][h] - Yr[2][h], bi = Yi[0][h] - Yi[2][h];
    const T cr = Yr[1][h] + Yr[3][h], ci = Yi[1][h] + Yi[3][h], dr = Yr[1][h] - Yr[3][h], di = Yi[1][h] - Yi[3][h];
    Zr[k1] = ar + cr;      Zi[k1] = ai + ci;
    Zr[k1 + 16] = ar - cr; Zi[k1 + 16] = ai - ci;
    Zr[k1 + 8] = br + di;  Zi[k1 + 8] = bi - dr;     // b - i d
    Zr[k1 + 24] = br - di; Zi[k1 + 24] = bi + dr;    // b + i d
  }
  DCT64_FENCE();
  if (ROLE == EO_EVEN) {
    out[0] = (Zr[0] + Zi[0]) * T(0.125);
    out[16] = (Zr[0] - Zi[0]) * T(0.125);
    out[8] = fma_(tab[TB_FS16 + 1], Zi[16], tab[TB_FS16 + 0] * Zr[16]);
    out[24] = fma_(tab[TB_FS16 + 3], Zi[16], tab[TB_FS16 + 2] * Zr[16]);
  }
#pragma unroll
  for (int k = 1; k < 16; k++) {
    if ((k & 1) != ROLE) continue;
    const TabPtr c = tab + TB_FS + 16 * (k - 1);
    out[k >> 1] = lin4<T, TabPtr>(c, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    out[(64 - k) >> 1] = lin4<T, TabPtr>(c + 4, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    out[(32 - k) >> 1] = lin4<T, TabPtr>(c + 8, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    out[(32 + k) >> 1] = lin4<T, TabPtr>(c + 12, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    if ((k >> 1) % 2 == 1) DCT64_FENCE();
  }
}

}  // namespace dctz
