// dct64_block.h -- the 64-point orthonormal DCT-II / DCT-III of ONE block, evaluated by ONE
// lane entirely in registers (64 values in, 64 values out, every index a compile-time constant).
//
// Replaces the per-block transform of the reference, dct_fftw() / ifft_idct() (dct.c:55-103,
// 115-205; dct-float.c likewise): Makhoul permutation (dct.c:75-83), an n-point complex FFT done
// by FFTW (dct.c:91), and a twiddle by as[]/ax[] (dct.c:100-102) -- resp. ias[]/iax[]
// (dct.c:166-172), backward FFT (dct.c:182), 1/n and de-interleave (dct.c:184-199).
//
// Flow (forward): the permuted real block v[] is packed into 32 complex points z[m] = v[2m] + i v[2m+1],
// a 32-point FFT runs as radix-8 x radix-4, and the split step (32-point spectrum -> 64-point
// spectrum) is MERGED with the reference's twiddle: every coefficient b[k] is a fixed linear
// combination of the four reals of Z[k], Z[32-k],
//     b[k] = c0 Re Z[k] + c1 Im Z[k] + c2 Re Z[32-k] + c3 Im Z[32-k]        (1 mul + 3 fma),
// with the 4 x 63 constants built on the host in extended precision (dctz_tables.h).  The inverse
// is the mirror image: Zb[k] = c0 a[k] + c1 a[64-k] + c2 a[32-k] + c3 a[32+k], then the backward
// radix-4 x radix-8 FFT.  No cross-lane traffic, no LDS, no table lookups by a per-lane index:
// on the GPU the constants are wave-uniform (scalar loads), the lane index only selects the BLOCK.
//
// The fused multiply-adds are written out (fma_); everything else is one rounding per operation
// (-ffp-contract=off).  The file compiles under hipcc (device + host) and under g++ (tests/emu),
// and the CPU checker under tests/ restates the same sequence of operations as its pinned "fast" flow.
#pragma once
#include <type_traits>

#if defined(__HIPCC__)
#define DCTZ_HD __host__ __device__ __forceinline__
#else
#define DCTZ_HD inline
#endif

// Scheduling fence between the stages of the transform (GPU builds only): without it the compiler hoists the scalar
// loads of ALL later constants and the first operations of later stages to the top, which costs a hundred SGPR spills
// and pushes the kernel past the 256 registers that two waves per SIMD allow.
// (Template flag FENCED: a kernel that runs one wave per SIMD wants the hoisting -- it is its only latency cover.)
#if defined(__HIP_DEVICE_COMPILE__)
#define DCT64_FENCE() do { if (FENCED) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DCT64_FENCE() ((void)0)
#endif

namespace dctz {

// Constant block shared by host and device (filled by dctz_tables.h: fill_tab_block).  Offsets in elements of T.
enum : int {
  TB_TW = 0,        // [3][7][2]  (cos, sin)(2 pi n2 k1 / 32), n2 = 1..3, k1 = 1..7
  TB_R = 42,        // [1]        sqrt(1/2)
  TB_FS = 44,       // [15][16]   forward split+twiddle, k = 1..15: rows b[k], b[64-k], b[32-k], b[32+k] x (ar, ai, cr, ci)
  TB_FS16 = 284,    // [4]        b[16] = f0 Re Z[16] + f1 Im Z[16];  b[48] = f2 Re Z[16] + f3 Im Z[16]
  TB_IS = 288,      // [15][16]   inverse merge, k = 1..15: rows Re Zb[k], Im Zb[k], Re Zb[32-k], Im Zb[32-k] x (a[k], a[64-k], a[32-k], a[32+k])
  TB_IS16 = 528,    // [4]        Re Zb[16] = g0 a[16] + g1 a[48];  Im Zb[16] = g2 a[16] + g3 a[48]
  TB_SIZE = 532
};

// Remainder-block tables (length l = N % 64), elements of T:
// as[64] ax[64] ias[64] iax[64] wr[128] wi[128]  (dctz_tables.h: fill_rem_tab)
enum : int { RTAB_AS = 0, RTAB_AX = 64, RTAB_IAS = 128, RTAB_IAX = 192, RTAB_WR = 256,
             RTAB_WI = 384, RTAB_SIZE = 512 };

DCTZ_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
DCTZ_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Radix-8 DFT, natural order in and out.  FWD: exp(-i..), else exp(+i..).
template <typename T, bool FWD>
DCTZ_HD void fft8(T (&xr)[8], T (&xi)[8], T r) {
  T a0r = xr[0] + xr[4], a0i = xi[0] + xi[4], a1r = xr[0] - xr[4], a1i = xi[0] - xi[4];
  T a2r = xr[2] + xr[6], a2i = xi[2] + xi[6], a3r = xr[2] - xr[6], a3i = xi[2] - xi[6];
  T a4r = xr[1] + xr[5], a4i = xi[1] + xi[5], a5r = xr[1] - xr[5], a5i = xi[1] - xi[5];
  T a6r = xr[3] + xr[7], a6i = xi[3] + xi[7], a7r = xr[3] - xr[7], a7i = xi[3] - xi[7];
  T E0r = a0r + a2r, E0i = a0i + a2i, E2r = a0r - a2r, E2i = a0i - a2i;
  T O0r = a4r + a6r, O0i = a4i + a6i, O2r = a4r - a6r, O2i = a4i - a6i;
  if (FWD) {
    T E1r = a1r + a3i, E1i = a1i - a3r, E3r = a1r - a3i, E3i = a1i + a3r;
    T O1r = a5r + a7i, O1i = a5i - a7r, O3r = a5r - a7i, O3i = a5i + a7r;
    T t1r = (O1r + O1i) * r, t1i = (O1i - O1r) * r;
    T t3r = (O3i - O3r) * r, t3i = -((O3r + O3i) * r);
    xr[0] = E0r + O0r; xi[0] = E0i + O0i; xr[4] = E0r - O0r; xi[4] = E0i - O0i;
    xr[1] = E1r + t1r; xi[1] = E1i + t1i; xr[5] = E1r - t1r; xi[5] = E1i - t1i;
    xr[2] = E2r + O2i; xi[2] = E2i - O2r; xr[6] = E2r - O2i; xi[6] = E2i + O2r;
    xr[3] = E3r + t3r; xi[3] = E3i + t3i; xr[7] = E3r - t3r; xi[7] = E3i - t3i;
  } else {
    T E1r = a1r - a3i, E1i = a1i + a3r, E3r = a1r + a3i, E3i = a1i - a3r;
    T O1r = a5r - a7i, O1i = a5i + a7r, O3r = a5r + a7i, O3i = a5i - a7r;
    T t1r = (O1r - O1i) * r, t1i = (O1r + O1i) * r;
    T t3r = -((O3r + O3i) * r), t3i = (O3r - O3i) * r;
    xr[0] = E0r + O0r; xi[0] = E0i + O0i; xr[4] = E0r - O0r; xi[4] = E0i - O0i;
    xr[1] = E1r + t1r; xi[1] = E1i + t1i; xr[5] = E1r - t1r; xi[5] = E1i - t1i;
    xr[2] = E2r - O2i; xi[2] = E2i + O2r; xr[6] = E2r + O2i; xi[6] = E2i - O2r;
    xr[3] = E3r + t3r; xi[3] = E3i + t3i; xr[7] = E3r - t3r; xi[7] = E3i - t3i;
  }
}

// A caller's hook inside the forward transform: called with a compile-time index at the points where the schedule is
// fenced anyway (k_compress issues pieces of the next tile's LDS-DMA there instead of one burst in front of the
// transform).  The default does nothing.
struct NoHook { template <typename I> DCTZ_HD void operator()(I) const {} };
template <int I, typename Hook> DCTZ_HD void hook_const(Hook& h) { h(std::integral_constant<int, I>{}); }
template <typename Hook> DCTZ_HD void hook_at(Hook& h, int i) {
  switch (i) {
    case 0: hook_const<0>(h); break; case 1: hook_const<1>(h); break; case 2: hook_const<2>(h); break; case 3: hook_const<3>(h); break;
    case 4: hook_const<4>(h); break; case 5: hook_const<5>(h); break; case 6: hook_const<6>(h); break; case 7: hook_const<7>(h); break;
    case 8: hook_const<8>(h); break; case 9: hook_const<9>(h); break; case 10: hook_const<10>(h); break; default: hook_const<11>(h); break;
  }
}

// c0 p + c1 q + c2 r + c3 s, accumulated left to right: one product, three fused multiply-adds
template <typename T, typename TabPtr>
DCTZ_HD T lin4(TabPtr c, T p, T q, T r, T s) {
  return fma_(c[3], s, fma_(c[2], r, fma_(c[1], q, c[0] * p)));
}

// Position inside the block of packed point m: z[m] = a[pz(m,0)] + i a[pz(m,1)]  -- the even/odd
// permutation of dct.c:75-83 composed with the pairing (v[2m], v[2m+1]).
DCTZ_HD constexpr int pack_pos(int m, int c) { return (m < 16) ? (4 * m + 2 * c) : (127 - 4 * m - 2 * c); }

// ------------------------------------------------------------------ forward --
// x[0..63]: one block (already scaled) in, its 64 DCT-II coefficients out (dct.c:55-103).
// TabPtr: const T* on the host; on the GPU a pointer into the CONSTANT address space, so that the (wave-uniform)
// table reads become scalar loads.
template <typename T, typename TabPtr, bool FENCED = false, typename Hook = NoHook>
DCTZ_HD void dct64_fwd(T (&x)[64], TabPtr tab, Hook hook = Hook{}) {
  T Yr[4][8], Yi[4][8];
  const T r = tab[TB_R];
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {                 // radix-8 over n1 of z[4 n1 + n2], then the 32-point twiddle
    T yr[8], yi[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) { yr[n1] = x[pack_pos(4 * n1 + n2, 0)]; yi[n1] = x[pack_pos(4 * n1 + n2, 1)]; }
    fft8<T, true>(yr, yi, r);
    Yr[n2][0] = yr[0]; Yi[n2][0] = yi[0];
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) {
      if (n2 == 0) { Yr[0][k1] = yr[k1]; Yi[0][k1] = yi[k1]; continue; }
      const T wr = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2], wi = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1];
      Yr[n2][k1] = fma_(yi[k1], wi, yr[k1] * wr);        // times exp(-i 2 pi n2 k1 / 32)
      Yi[n2][k1] = fma_(-yr[k1], wi, yi[k1] * wr);
    }
    DCT64_FENCE();
    hook_at(hook, n2);
  }
  T Zr[32], Zi[32];
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {                 // radix-4 over n2: Z[k1 + 8 k2]
    const T ar = Yr[0][k1] + Yr[2][k1], ai = Yi[0][k1] + Yi[2][k1], br = Yr[0][k1] - Yr[2][k1], bi = Yi[0][k1] - Yi[2][k1];
    const T cr = Yr[1][k1] + Yr[3][k1], ci = Yi[1][k1] + Yi[3][k1], dr = Yr[1][k1] - Yr[3][k1], di = Yi[1][k1] - Yi[3][k1];
    Zr[k1] = ar + cr;      Zi[k1] = ai + ci;
    Zr[k1 + 16] = ar - cr; Zi[k1 + 16] = ai - ci;
    Zr[k1 + 8] = br + di;  Zi[k1 + 8] = bi - dr;     // b - i d
    Zr[k1 + 24] = br - di; Zi[k1 + 24] = bi + dr;    // b + i d
  }
  DCT64_FENCE();
  hook_const<4>(hook);
  // split + twiddle, merged (header comment); the two self-paired bins are exact scalings:
  // b[0] = (Re Z[0] + Im Z[0]) / 8  (= sum of the block / 8), b[32] = (Re Z[0] - Im Z[0]) / 8
  x[0] = (Zr[0] + Zi[0]) * T(0.125);
  x[32] = (Zr[0] - Zi[0]) * T(0.125);
  x[16] = fma_(tab[TB_FS16 + 1], Zi[16], tab[TB_FS16 + 0] * Zr[16]);
  x[48] = fma_(tab[TB_FS16 + 3], Zi[16], tab[TB_FS16 + 2] * Zr[16]);
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const TabPtr c = tab + TB_FS + 16 * (k - 1);
    x[k] = lin4<T, TabPtr>(c, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    x[64 - k] = lin4<T, TabPtr>(c + 4, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    x[32 - k] = lin4<T, TabPtr>(c + 8, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    x[32 + k] = lin4<T, TabPtr>(c + 12, Zr[k], Zi[k], Zr[32 - k], Zi[32 - k]);
    if (k % 2 == 0) { DCT64_FENCE(); hook_at(hook, 4 + k / 2); }
  }
}

// ------------------------------------------------------------------ inverse --
// x[0..63]: 64 coefficients in, the reconstructed block out (dct.c:115-205, even n).
template <typename T, typename TabPtr, bool FENCED = false>
DCTZ_HD void dct64_inv(T (&x)[64], TabPtr tab) {
  T Zr[32], Zi[32];
  Zr[0] = (x[0] + x[32]) * T(0.125);
  Zi[0] = (x[0] - x[32]) * T(0.125);
  Zr[16] = fma_(tab[TB_IS16 + 1], x[48], tab[TB_IS16 + 0] * x[16]);
  Zi[16] = fma_(tab[TB_IS16 + 3], x[48], tab[TB_IS16 + 2] * x[16]);
#pragma unroll
  for (int k = 1; k < 16; k++) {
    const TabPtr c = tab + TB_IS + 16 * (k - 1);
    Zr[k] = lin4<T, TabPtr>(c, x[k], x[64 - k], x[32 - k], x[32 + k]);
    Zi[k] = lin4<T, TabPtr>(c + 4, x[k], x[64 - k], x[32 - k], x[32 + k]);
    Zr[32 - k] = lin4<T, TabPtr>(c + 8, x[k], x[64 - k], x[32 - k], x[32 + k]);
    Zi[32 - k] = lin4<T, TabPtr>(c + 12, x[k], x[64 - k], x[32 - k], x[32 + k]);
    if (k % 2 == 0) DCT64_FENCE();
  }
  DCT64_FENCE();
  const T r = tab[TB_R];
  T Yr[4][8], Yi[4][8];
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {                 // backward radix-4 over k2 of Zb[k1 + 8 k2] -> n2, then the twiddle
    const T ar = Zr[k1] + Zr[k1 + 16], ai = Zi[k1] + Zi[k1 + 16], br = Zr[k1] - Zr[k1 + 16], bi = Zi[k1] - Zi[k1 + 16];
    const T cr = Zr[k1 + 8] + Zr[k1 + 24], ci = Zi[k1 + 8] + Zi[k1 + 24], dr = Zr[k1 + 8] - Zr[k1 + 24], di = Zi[k1 + 8] - Zi[k1 + 24];
    T tr[4], ti[4];
    tr[0] = ar + cr; ti[0] = ai + ci;
    tr[2] = ar - cr; ti[2] = ai - ci;
    tr[1] = br - di; ti[1] = bi + dr;               // b + i d
    tr[3] = br + di; ti[3] = bi - dr;               // b - i d
    Yr[0][k1] = tr[0]; Yi[0][k1] = ti[0];
#pragma unroll
    for (int n2 = 1; n2 < 4; n2++) {
      if (k1 == 0) { Yr[n2][0] = tr[n2]; Yi[n2][0] = ti[n2]; continue; }
      const T wr = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2], wi = tab[TB_TW + ((n2 - 1) * 7 + (k1 - 1)) * 2 + 1];
      Yr[n2][k1] = fma_(-ti[n2], wi, tr[n2] * wr);       // times exp(+i 2 pi n2 k1 / 32)
      Yi[n2][k1] = fma_(tr[n2], wi, ti[n2] * wr);
    }
    if (k1 % 2 == 1) DCT64_FENCE();
  }
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++) {                 // backward radix-8 over k1 -> z[4 n1 + n2]
    T yr[8], yi[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) { yr[k1] = Yr[n2][k1]; yi[k1] = Yi[n2][k1]; }
    fft8<T, false>(yr, yi, r);
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) { x[pack_pos(4 * n1 + n2, 0)] = yr[n1]; x[pack_pos(4 * n1 + n2, 1)] = yi[n1]; }
    DCT64_FENCE();
  }
}

}  // namespace dctz
