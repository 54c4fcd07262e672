// dct64_block_pk.h -- the 64-point block transform of dct64_block.h for fp32 with TWO values per instruction
// (gfx950: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).  A complex point is one 2-vector (re, im): the butterflies of
// the radix-8 x radix-4 FFT become packed adds, a multiplication by -i a swizzle with a sign (J below), a twiddle one
// packed product + one packed fused multiply-add, and the merged split step computes two coefficients per chain.
// Component for component these are the SAME IEEE operations in the SAME order as the scalar flow (and as the CPU
// checker's pinned flow): tests/test_lane_emulation.py runs this header on the CPU and requires bit-identity.
// The constants are those of the TB_* block, re-ordered so that the two values of a packed operand sit side by side
// (TBP_* block behind it, filled by dctz_tables.h from the very same numbers).
#pragma once

#include "dct64_block.h"
#include "dct_nd_block.h"

namespace dctz {

#if defined(__clang__)
typedef float pk2 __attribute__((ext_vector_type(2)));
#else
typedef float pk2 __attribute__((vector_size(8)));
#endif

enum : int {
  TBP_FS = TB_TOTAL,          // [15][16]  forward: k = 1..15: pairs (b[k], b[64-k]) x (c0..c3), then (b[32-k], b[32+k]) x (c0..c3)
  TBP_FS16 = TBP_FS + 240,    // [4]       (f0, f2), (f1, f3) of TB_FS16
  TBP_IS = TBP_FS16 + 4,      // [15][16]  inverse: pairs (Re Zb[k], Im Zb[k]) x (c0..c3), then (Re Zb[32-k], Im Zb[32-k]) x (c0..c3)
  TBP_IS16 = TBP_IS + 240,    // [4]       (g0, g2), (g1, g3) of TB_IS16
  TBP_TOTAL = TBP_IS16 + 4
};

DCTZ_HD pk2 pk_mk(float a, float b) { pk2 v = {a, b}; return v; }
DCTZ_HD pk2 pk_bc(float a) { pk2 v = {a, a}; return v; }
DCTZ_HD pk2 pk_J(pk2 v) { pk2 r = {v[1], -v[0]}; return r; }            // times -i: (re, im) -> (im, -re)
DCTZ_HD pk2 pk_swap(pk2 v) { pk2 r = {v[1], v[0]}; return r; }
DCTZ_HD pk2 pk_fma(pk2 a, pk2 b, pk2 c) {
#if defined(__clang__)
  return __builtin_elementwise_fma(a, b, c);
#else
  pk2 r = {__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
  return r;
#endif
}
template <typename TabPtr>
DCTZ_HD pk2 pk_ld(TabPtr tab, int i) { pk2 v = {tab[i], tab[i + 1]}; return v; }

// radix-8 DFT on eight complex points (fft8 of dct64_block.h, component for component)
template <bool FWD>
DCTZ_HD void pk_fft8(pk2 (&z)[8], float r) {
  const pk2 a0 = z[0] + z[4], a1 = z[0] - z[4], a2 = z[2] + z[6], a3 = z[2] - z[6];
  const pk2 a4 = z[1] + z[5], a5 = z[1] - z[5], a6 = z[3] + z[7], a7 = z[3] - z[7];
  const pk2 E0 = a0 + a2, E2 = a0 - a2, O0 = a4 + a6, O2 = a4 - a6;
  const pk2 rr = pk_bc(r);
  if (FWD) {
    const pk2 E1 = a1 + pk_J(a3), E3 = a1 - pk_J(a3), O1 = a5 + pk_J(a7), O3 = a5 - pk_J(a7);
    const pk2 t1 = (O1 + pk_J(O1)) * rr;                  // ((O1r + O1i) r, (O1i - O1r) r)
    const pk2 t3 = (pk_J(O3) - O3) * rr;                  // ((O3i - O3r) r, -((O3r + O3i) r))
    z[0] = E0 + O0; z[4] = E0 - O0;
    z[1] = E1 + t1; z[5] = E1 - t1;
    z[2] = E2 + pk_J(O2); z[6] = E2 - pk_J(O2);
    z[3] = E3 + t3; z[7] = E3 - t3;
  } else {
    const pk2 E1 = a1 - pk_J(a3), E3 = a1 + pk_J(a3), O1 = a5 - pk_J(a7), O3 = a5 + pk_J(a7);
    const pk2 t1 = (O1 - pk_J(O1)) * rr;                  // ((O1r - O1i) r, (O1r + O1i) r)
    const pk2 t3 = (-(O3 + pk_J(O3))) * rr;               // (-((O3r + O3i) r), (O3r - O3i) r)
    z[0] = E0 + O0; z[4] = E0 - O0;
    z[1] = E1 + t1; z[5] = E1 - t1;
    z[2] = E2 - pk_J(O2); z[6] = E2 + pk_J(O2);
    z[3] = E3 + t3; z[7] = E3 - t3;
  }
}

// c0 P + c1 Q + c2 R + c3 S for TWO rows of constants at once (lin4 of dct64_block.h per component)
template <typename TabPtr>
DCTZ_HD pk2 pk_lin4(TabPtr c, float p, float q, float r, float s) {
  return pk_fma(pk_ld(c, 6), pk_bc(s), pk_fma(pk_ld(c, 4), pk_bc(r), pk_fma(pk_ld(c, 2), pk_bc(q), pk_ld(c, 0) * pk_bc(p))));
}

// forward: dct64_fwd<float> of dct64_block.h
template <typename TabPtr, bool FENCED = false, typename Hook = NoHook>
DCTZ_HD void dct64_fwd_pk(float (&x)[64], TabPtr tab, Hook hook = Hook{}) {
  pk2 Y[4][8];
  const float r = tab[TB_R];
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    pk2 y[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) y[n1] = pk_mk(x[pack_pos(4 * n1 + n2, 0)], x[pack_pos(4 * n1 + n2, 1)]);
    pk_fft8<true>(y, r);
    Y[n2][0] = y[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) {
      if (n2 == 0) { Y[0][k1] = y[k1]; continue; }
      const float wr = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2], wi = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1];
      Y[n2][k1] = pk_fma(pk_swap(y[k1]), pk_mk(wi, -wi), y[k1] * pk_bc(wr));      // times exp(-i 2 pi n2 k1 / 32)
    }
    DCT64_FENCE();
    hook_at(hook, n2);
  }
  pk2 Z[32];
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    const pk2 a = Y[0][k1] + Y[2][k1], b = Y[0][k1] - Y[2][k1], c = Y[1][k1] + Y[3][k1], d = Y[1][k1] - Y[3][k1];
    Z[k1] = a + c; Z[k1 + 16] = a - c;
    Z[k1 + 8] = b + pk_J(d); Z[k1 + 24] = b - pk_J(d);
  }
  DCT64_FENCE();
  hook_const<4>(hook);
  {
    const pk2 e = (pk_bc(Z[0][0]) + pk_mk(Z[0][1], -Z[0][1])) * pk_bc(0.125f);      // b[0], b[32]
    x[0] = e[0]; x[32] = e[1];
    const pk2 f = pk_fma(pk_ld(tab, TBP_FS16 + 2), pk_bc(Z[16][1]), pk_ld(tab, TBP_FS16) * pk_bc(Z[16][0]));   // b[16], b[48]
    x[16] = f[0]; x[48] = f[1];
  }
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const TabPtr c = tab + TBP_FS + 16 * (k - 1);
    const pk2 u = pk_lin4<TabPtr>(c, Z[k][0], Z[k][1], Z[32 - k][0], Z[32 - k][1]);
    const pk2 v = pk_lin4<TabPtr>(c + 8, Z[k][0], Z[k][1], Z[32 - k][0], Z[32 - k][1]);
    x[k] = u[0]; x[64 - k] = u[1]; x[32 - k] = v[0]; x[32 + k] = v[1];
    if (k % 2 == 0) { DCT64_FENCE(); hook_at(hook, 4 + k / 2); }
  }
}

// inverse: dct64_inv<float> of dct64_block.h
template <typename TabPtr, bool FENCED = false>
DCTZ_HD void dct64_inv_pk(float (&x)[64], TabPtr tab) {
  pk2 Z[32];
  Z[0] = (pk_bc(x[0]) + pk_mk(x[32], -x[32])) * pk_bc(0.125f);
  Z[16] = pk_fma(pk_ld(tab, TBP_IS16 + 2), pk_bc(x[48]), pk_ld(tab, TBP_IS16) * pk_bc(x[16]));
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const TabPtr c = tab + TBP_IS + 16 * (k - 1);
    Z[k] = pk_lin4<TabPtr>(c, x[k], x[64 - k], x[32 - k], x[32 + k]);
    Z[32 - k] = pk_lin4<TabPtr>(c + 8, x[k], x[64 - k], x[32 - k], x[32 + k]);
    if (k % 2 == 0) DCT64_FENCE();
  }
  DCT64_FENCE();
  const float r = tab[TB_R];
  pk2 Y[4][8];
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    const pk2 a = Z[k1] + Z[k1 + 16], b = Z[k1] - Z[k1 + 16], c = Z[k1 + 8] + Z[k1 + 24], d = Z[k1 + 8] - Z[k1 + 24];
    pk2 t[4];
    t[0] = a + c; t[2] = a - c;
    t[1] = b - pk_J(d); t[3] = b + pk_J(d);                 // b + i d, b - i d
    Y[0][k1] = t[0];
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++) {
      if (k1 == 0) { Y[n2][0] = t[n2]; continue; }
      const float wr = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2], wi = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1];
      Y[n2][k1] = pk_fma(pk_swap(t[n2]), pk_mk(-wi, wi), t[n2] * pk_bc(wr));      // times exp(+i 2 pi n2 k1 / 32)
    }
    if (k1 % 2 == 1) DCT64_FENCE();
  }
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {
    pk2 y[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) y[k1] = Y[n2][k1];
    pk_fft8<false>(y, r);
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) { x[pack_pos(4 * n1 + n2, 0)] = y[n1][0]; x[pack_pos(4 * n1 + n2, 1)] = y[n1][1]; }
    DCT64_FENCE();
  }
}

// the block transform of a geometry; fp32 blocks of the flat geometry take the packed form (DCTZ_PK32=0: the scalar one)
#ifndef DCTZ_PK32
#define DCTZ_PK32 1
#endif
template <typename T, typename TabPtr, int GEOM, bool FENCED, typename Hook = NoHook>
DCTZ_HD void block_fwd(T (&x)[64], TabPtr tab, Hook hook = Hook{}) {
  if constexpr (GEOM == GEOM_2D) dct8x8_fwd<T, TabPtr>(x, tab);                 // (no hook points: the caller does not ask for any there)
  else if constexpr (GEOM == GEOM_3D) dct4x4x4_fwd<T, TabPtr>(x, tab);
  else if constexpr (sizeof(T) == 4 && DCTZ_PK32) dct64_fwd_pk<TabPtr, FENCED, Hook>(x, tab, hook);
  else dct64_fwd<T, TabPtr, FENCED, Hook>(x, tab, hook);
}
template <typename T, typename TabPtr, int GEOM, bool FENCED>
DCTZ_HD void block_inv(T (&x)[64], TabPtr tab) {
  if constexpr (GEOM == GEOM_2D) dct8x8_inv<T, TabPtr>(x, tab);
  else if constexpr (GEOM == GEOM_3D) dct4x4x4_inv<T, TabPtr>(x, tab);
  else if constexpr (sizeof(T) == 4 && DCTZ_PK32) dct64_inv_pk<TabPtr, FENCED>(x, tab);
  else dct64_inv<T, TabPtr, FENCED>(x, tab);
}

}  // namespace dctz
