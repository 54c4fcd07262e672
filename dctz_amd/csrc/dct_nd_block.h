// dct_nd_block.h -- multi-dimensional blocks (behind a flag): the 64 values of one block are an
// 8 x 8 tile of a 2-D array or a 4 x 4 x 4 tile of a 3-D array (row-major inside the tile, last axis fastest), and
// the block transform is the SEPARABLE orthonormal DCT-II / DCT-III along every axis -- the normalised form of what
// the reference's stand-alone experiment does to a whole array with FFTW's r2r plans (dct-fftw-test.c:74-97:
// fftw_plan_r2r_2d / _3d with FFTW_REDFT10 per axis forward, FFTW_REDFT01 backward, then a division by (2 n) per
// axis; here the scaling is folded into the transform exactly as the library's 1-D path is orthonormal,
// dct.c:37-47).  Everything downstream of the transform -- DC, binning, the exception stream, the QT table with one
// entry per position in the block -- is the 1-D path's, unchanged (dctz-comp-lib.c:350-544).
//
// One lane evaluates one block in registers, every index a compile-time constant (like dct64_block.h).  The 8- and
// 4-point transforms are the even / odd decomposition written out, one rounding per operation plus the explicit
// fused multiply-adds; the CPU checker under tests/ restates the same sequence of operations.
//
//   8-point forward:  a_n = x_n + x_{7-n},  b_n = x_n - x_{7-n}   (n = 0..3)
//                     c0 = a0 + a3, c1 = a1 + a2, d0 = a0 - a3, d1 = a1 - a2
//                     X0 = (c0 + c1) r8,  X4 = (c0 - c1) r8,  X2 = g1 d0 + g3 d1,  X6 = g3 d0 - g1 d1
//                     X1 = C1 b0 + C3 b1 + C5 b2 + C7 b3      X3 = C3 b0 - C7 b1 - C1 b2 - C5 b3
//                     X5 = C5 b0 - C1 b1 + C7 b2 + C3 b3      X7 = C7 b0 - C5 b1 + C3 b2 - C1 b3
//   with r8 = sqrt(1/8), g_j = cos(j pi / 8) / 2, C_j = cos(j pi / 16) / 2.  The inverse is the transpose.
//   4-point forward:  c0 = x0 + x3, c1 = x1 + x2, d0 = x0 - x3, d1 = x1 - x2
//                     X0 = (c0 + c1) / 2,  X2 = (c0 - c1) / 2,  X1 = h1 d0 + h3 d1,  X3 = h3 d0 - h1 d1
//   with h_j = sqrt(1/2) cos(j pi / 8).
#pragma once

#include "dct64_block.h"

namespace dctz {

// constants behind the 64-point block of dct64_block.h (filled by dctz_tables.h: fill_tab_block), elements of T
enum : int {
  TB_ND = TB_SIZE,        // r8, g1, g3, C1, C3, C5, C7, h1, h3
  TB_ND_R8 = TB_ND + 0, TB_ND_G1 = TB_ND + 1, TB_ND_G3 = TB_ND + 2,
  TB_ND_C1 = TB_ND + 3, TB_ND_C3 = TB_ND + 4, TB_ND_C5 = TB_ND + 5, TB_ND_C7 = TB_ND + 6,
  TB_ND_H1 = TB_ND + 7, TB_ND_H3 = TB_ND + 8,
  TB_TOTAL = TB_SIZE + 12
};

// block geometries (include/dctz_hip.h: DCTZHIP_GEOM_*)
enum : int { GEOM_1D = 0, GEOM_2D = 1, GEOM_3D = 2 };

template <typename T> struct NdConst { T r8, g1, g3, c1, c3, c5, c7, h1, h3; };
template <typename T, typename TabPtr>
DCTZ_HD NdConst<T> nd_load(TabPtr tab) {
  NdConst<T> k;
  k.r8 = tab[TB_ND_R8]; k.g1 = tab[TB_ND_G1]; k.g3 = tab[TB_ND_G3];
  k.c1 = tab[TB_ND_C1]; k.c3 = tab[TB_ND_C3]; k.c5 = tab[TB_ND_C5]; k.c7 = tab[TB_ND_C7];
  k.h1 = tab[TB_ND_H1]; k.h3 = tab[TB_ND_H3];
  return k;
}

// p0 b0 + p1 b1 + p2 b2 + p3 b3, left to right: one product, three fused multiply-adds
template <typename T>
DCTZ_HD T dot4(T p0, T p1, T p2, T p3, T b0, T b1, T b2, T b3) {
  return fma_(p3, b3, fma_(p2, b2, fma_(p1, b1, p0 * b0)));
}

// 8-point orthonormal DCT-II in place on x[O], x[O + S], ..., x[O + 7 S]
template <typename T, int O, int S>
DCTZ_HD void dct8_fwd(T (&x)[64], const NdConst<T>& k) {
  const T a0 = x[O] + x[O + 7 * S], a1 = x[O + S] + x[O + 6 * S], a2 = x[O + 2 * S] + x[O + 5 * S], a3 = x[O + 3 * S] + x[O + 4 * S];
  const T b0 = x[O] - x[O + 7 * S], b1 = x[O + S] - x[O + 6 * S], b2 = x[O + 2 * S] - x[O + 5 * S], b3 = x[O + 3 * S] - x[O + 4 * S];
  const T c0 = a0 + a3, c1 = a1 + a2, d0 = a0 - a3, d1 = a1 - a2;
  x[O] = (c0 + c1) * k.r8;
  x[O + 4 * S] = (c0 - c1) * k.r8;
  x[O + 2 * S] = fma_(k.g3, d1, k.g1 * d0);
  x[O + 6 * S] = fma_(-k.g1, d1, k.g3 * d0);
  x[O + S] = dot4<T>(k.c1, k.c3, k.c5, k.c7, b0, b1, b2, b3);
  x[O + 3 * S] = dot4<T>(k.c3, -k.c7, -k.c1, -k.c5, b0, b1, b2, b3);
  x[O + 5 * S] = dot4<T>(k.c5, -k.c1, k.c7, k.c3, b0, b1, b2, b3);
  x[O + 7 * S] = dot4<T>(k.c7, -k.c5, k.c3, -k.c1, b0, b1, b2, b3);
}

// 8-point orthonormal DCT-III (the transpose), in place
template <typename T, int O, int S>
DCTZ_HD void dct8_inv(T (&x)[64], const NdConst<T>& k) {
  const T X0 = x[O], X1 = x[O + S], X2 = x[O + 2 * S], X3 = x[O + 3 * S], X4 = x[O + 4 * S], X5 = x[O + 5 * S], X6 = x[O + 6 * S], X7 = x[O + 7 * S];
  const T p0 = (X0 + X4) * k.r8, p1 = (X0 - X4) * k.r8;
  const T q0 = fma_(k.g3, X6, k.g1 * X2), q1 = fma_(-k.g1, X6, k.g3 * X2);
  const T e0 = p0 + q0, e3 = p0 - q0, e1 = p1 + q1, e2 = p1 - q1;
  const T o0 = dot4<T>(k.c1, k.c3, k.c5, k.c7, X1, X3, X5, X7);
  const T o1 = dot4<T>(k.c3, -k.c7, -k.c1, -k.c5, X1, X3, X5, X7);
  const T o2 = dot4<T>(k.c5, -k.c1, k.c7, k.c3, X1, X3, X5, X7);
  const T o3 = dot4<T>(k.c7, -k.c5, k.c3, -k.c1, X1, X3, X5, X7);
  x[O] = e0 + o0; x[O + 7 * S] = e0 - o0;
  x[O + S] = e1 + o1; x[O + 6 * S] = e1 - o1;
  x[O + 2 * S] = e2 + o2; x[O + 5 * S] = e2 - o2;
  x[O + 3 * S] = e3 + o3; x[O + 4 * S] = e3 - o3;
}

// 4-point orthonormal DCT-II / DCT-III in place on x[O], x[O + S], x[O + 2 S], x[O + 3 S]
template <typename T, int O, int S>
DCTZ_HD void dct4_fwd(T (&x)[64], const NdConst<T>& k) {
  const T c0 = x[O] + x[O + 3 * S], c1 = x[O + S] + x[O + 2 * S], d0 = x[O] - x[O + 3 * S], d1 = x[O + S] - x[O + 2 * S];
  x[O] = (c0 + c1) * T(0.5);
  x[O + 2 * S] = (c0 - c1) * T(0.5);
  x[O + S] = fma_(k.h3, d1, k.h1 * d0);
  x[O + 3 * S] = fma_(-k.h1, d1, k.h3 * d0);
}
template <typename T, int O, int S>
DCTZ_HD void dct4_inv(T (&x)[64], const NdConst<T>& k) {
  const T X0 = x[O], X1 = x[O + S], X2 = x[O + 2 * S], X3 = x[O + 3 * S];
  const T p0 = (X0 + X2) * T(0.5), p1 = (X0 - X2) * T(0.5);
  const T q0 = fma_(k.h3, X3, k.h1 * X1), q1 = fma_(-k.h1, X3, k.h3 * X1);
  x[O] = p0 + q0; x[O + 3 * S] = p0 - q0;
  x[O + S] = p1 + q1; x[O + 2 * S] = p1 - q1;
}

// compile-time loops over the lines of a block
template <typename T, int I, int N, int OSTEP, int S, bool FWD>
struct Lines8 {
  static DCTZ_HD void run(T (&x)[64], const NdConst<T>& k) {
    if (FWD) dct8_fwd<T, I * OSTEP, S>(x, k); else dct8_inv<T, I * OSTEP, S>(x, k);
    Lines8<T, I + 1, N, OSTEP, S, FWD>::run(x, k);
  }
};
template <typename T, int N, int OSTEP, int S, bool FWD>
struct Lines8<T, N, N, OSTEP, S, FWD> { static DCTZ_HD void run(T (&)[64], const NdConst<T>&) {} };

// 4-point lines: offset of line i = (i / INNER) * OUTER_STEP + (i % INNER) * INNER_STEP
template <typename T, int I, int N, int INNER, int OUTER_STEP, int INNER_STEP, int S, bool FWD>
struct Lines4 {
  static DCTZ_HD void run(T (&x)[64], const NdConst<T>& k) {
    constexpr int O = (I / INNER) * OUTER_STEP + (I % INNER) * INNER_STEP;
    if (FWD) dct4_fwd<T, O, S>(x, k); else dct4_inv<T, O, S>(x, k);
    Lines4<T, I + 1, N, INNER, OUTER_STEP, INNER_STEP, S, FWD>::run(x, k);
  }
};
template <typename T, int N, int INNER, int OUTER_STEP, int INNER_STEP, int S, bool FWD>
struct Lines4<T, N, N, INNER, OUTER_STEP, INNER_STEP, S, FWD> { static DCTZ_HD void run(T (&)[64], const NdConst<T>&) {} };

// ---- 8 x 8: x[8 u + v]; forward: the 8 rows (along v), then the 8 columns (along u); inverse: columns, then rows
template <typename T, typename TabPtr>
DCTZ_HD void dct8x8_fwd(T (&x)[64], TabPtr tab) {
  const NdConst<T> k = nd_load<T, TabPtr>(tab);
  Lines8<T, 0, 8, 8, 1, true>::run(x, k);
  Lines8<T, 0, 8, 1, 8, true>::run(x, k);
}
template <typename T, typename TabPtr>
DCTZ_HD void dct8x8_inv(T (&x)[64], TabPtr tab) {
  const NdConst<T> k = nd_load<T, TabPtr>(tab);
  Lines8<T, 0, 8, 1, 8, false>::run(x, k);
  Lines8<T, 0, 8, 8, 1, false>::run(x, k);
}

// ---- 4 x 4 x 4: x[16 u + 4 v + w]; forward: along w, then v, then u; inverse: u, v, w
template <typename T, typename TabPtr>
DCTZ_HD void dct4x4x4_fwd(T (&x)[64], TabPtr tab) {
  const NdConst<T> k = nd_load<T, TabPtr>(tab);
  Lines4<T, 0, 16, 16, 0, 4, 1, true>::run(x, k);     // lines (u, v): offset 4 i, stride 1
  Lines4<T, 0, 16, 4, 16, 1, 4, true>::run(x, k);     // lines (u, w): offset 16 u + w, stride 4
  Lines4<T, 0, 16, 16, 0, 1, 16, true>::run(x, k);    // lines (v, w): offset i, stride 16
}
template <typename T, typename TabPtr>
DCTZ_HD void dct4x4x4_inv(T (&x)[64], TabPtr tab) {
  const NdConst<T> k = nd_load<T, TabPtr>(tab);
  Lines4<T, 0, 16, 16, 0, 1, 16, false>::run(x, k);
  Lines4<T, 0, 16, 4, 16, 1, 4, false>::run(x, k);
  Lines4<T, 0, 16, 16, 0, 4, 1, false>::run(x, k);
}

}  // namespace dctz
