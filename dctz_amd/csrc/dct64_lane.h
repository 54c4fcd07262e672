// dct64_lane.h -- the 64-point orthonormal DCT-II / DCT-III as evaluated by FOUR
// cooperating lanes of a gfx950 wavefront (one quad = one 64-element block).
//
// Replaces the per-block transform of the reference, dct_fftw()/ifft_idct()
// (dct.c:55-103, 115-205; dct-float.c likewise): Makhoul permutation, an
// n-point complex FFT done by FFTW, and a twiddle by as[]/ax[] (ias[]/iax[]).
// Here the 64-point FFT of the real, permuted block is computed as a 32-point
// complex FFT of packed pairs: radix-8 inside a lane, radix-4 across the quad
// through DPP quad_perm moves, then a split step and the reference's own
// as/ax twiddle.  Every lane holds 8 complex points = 16 reals.
//
// Only pure arithmetic lives here (no memory, no cross-lane intrinsics): the
// caller supplies partner values.  The file compiles under hipcc (device) and
// under g++ (tests/emu build that walks the four lanes in a loop).  All code
// is built with -ffp-contract=off: one rounding per operation, which is what
// the parity tests rely on.
#pragma once

#if defined(__HIPCC__)
#define DCTZ_HD __host__ __device__ __forceinline__
#else
#define DCTZ_HD inline
#endif

namespace dctz {

// Table block shared by host and device (filled by dctz_tables.h on the host).
// Offsets are in elements of T.
enum : int {
  TAB_W32R = 0,     // [4][8]  cos(2 pi n2 k1 / 32)
  TAB_W32I = 32,    // [4][8]  sin(2 pi n2 k1 / 32)
  TAB_CW = 64,      // [32]    cos(2 pi k / 64), mirror-symmetric
  TAB_SW = 96,      // [32]    sin(2 pi k / 64)
  TAB_HS = 128,     // [64]    0.5*as[k]   (dct.c:37-47)
  TAB_HX = 192,     // [64]    0.5*ax[k]
  TAB_IAS = 256,    // [64]    ias[k] / 128, ias[0] pre-divided by sqrt(2)  (dct.c:130-134,166,185)
  TAB_IAX = 320,    // [64]    iax[k] / 128
  TAB_R = 384,      // [1]     sqrt(1/2)
  TAB_SIZE = 392
};

// Remainder-block tables (length l = N % 64), elements of T:
// as[64] ax[64] ias[64] iax[64] wr[128] wi[128]  (dctz_tables.h: fill_rem_tab)
enum : int { RTAB_AS = 0, RTAB_AX = 64, RTAB_IAS = 128, RTAB_IAX = 192, RTAB_WR = 256,
             RTAB_WI = 384, RTAB_SIZE = 512 };

// Position inside the block of packed point m, component c (0 = re, 1 = im):
// the even/odd permutation of dct.c:75-83 composed with pairing (v[2m], v[2m+1]).
DCTZ_HD int pack_pos(int m, int c) { return (m < 16) ? (4 * m + 2 * c) : (127 - 4 * m - 2 * c); }

// After the forward cross-lane radix-4, quad lane l holds frequencies 8*q + k1
// with q = bit-reversed lane id.
DCTZ_HD int lane_q(int lane) { return ((lane & 1) << 1) | ((lane >> 1) & 1); }

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

// One cross-lane radix-2 step on one value.  The lower lane of the pair keeps
// mine + theirs, the upper lane theirs - mine: theirs + s*mine with s = +1 / -1.
// The product by +-1 is exact, so the fused form rounds once, exactly like the
// plain add / subtract (one instruction instead of select + add).
DCTZ_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
DCTZ_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename T>
DCTZ_HD T bfly(T mine, T theirs, bool upper) { return fma_(upper ? T(-1) : T(1), mine, theirs); }
template <typename T>
DCTZ_HD T bfly_s(T mine, T theirs, T s) { return fma_(s, mine, theirs); }

// ---------------------------------------------------------------- forward ---
// F1: in-lane radix-8 + 32-point twiddle.  On entry y = packed points
// z[4*n1 + n2], n1 = 0..7, of quad lane n2.
template <typename T>
DCTZ_HD void fwd_stage_lane(T (&yr)[8], T (&yi)[8], int n2, const T* tab) {
  fft8<T, true>(yr, yi, tab[TAB_R]);
#pragma unroll
  for (int k1 = 1; k1 < 8; k1++) {
    T wr = tab[TAB_W32R + n2 * 8 + k1], wi = tab[TAB_W32I + n2 * 8 + k1];
    T a = yr[k1], b = yi[k1];
    yr[k1] = a * wr + b * wi;       // times exp(-i 2 pi n2 k1/32)
    yi[k1] = b * wr - a * wi;
  }
}

// F2a / F2b: the two cross-lane radix-2 steps (partner = lane^2, then lane^1).
template <typename T>
DCTZ_HD void fwd_cross_a(T (&yr)[8], T (&yi)[8], const T (&pr)[8], const T (&pi)[8], int lane) {
  const bool up = (lane & 2) != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    T r = bfly(yr[k], pr[k], up), i = bfly(yi[k], pi[k], up);
    if (lane == 3) { yr[k] = i; yi[k] = -r; }   // times -i
    else { yr[k] = r; yi[k] = i; }
  }
}
template <typename T>
DCTZ_HD void fwd_cross_b(T (&yr)[8], T (&yi)[8], const T (&pr)[8], const T (&pi)[8], int lane) {
  const bool up = (lane & 1) != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { yr[k] = bfly(yr[k], pr[k], up); yi[k] = bfly(yi[k], pi[k], up); }
}

// F3: split the 32-point spectrum into the 64-point one and apply the
// reference's twiddle b[k] = as[k] Re V[k] - ax[k] Im V[k]  (dct.c:100-102).
// z = own Z[8q+k1]; p[k1] = Z[32 - (8q+k1)] (from the mirror lane for k1 >= 1,
// from quad_perm [0,1,3,2] for k1 = 0).  Outputs: lo[k1] = b[8q+k1],
// hi[k1] = b[64-(8q+k1)], except quad lane 0 / k1 = 0 where hi[0] = b[32].
// One frequency pair: (ar, ai) = own Z[k], (cr, ci) = Z[32-k], k = 8q + k1.
template <typename T>
DCTZ_HD void fwd_split_one(int k1, T ar, T ai, T cr, T ci, int lane, const T* tab, T& lo, T& hi) {
  const int k = 8 * lane_q(lane) + k1;
  T Er = ar + cr, Ei = ai - ci, Or = ai + ci, Oi = cr - ar;
  T cw = tab[TAB_CW + k], sw = tab[TAB_SW + k];
  T P = cw * Or + sw * Oi;
  T Q = cw * Oi - sw * Or;
  T Vr = Er + P, Vi = Ei + Q;
  lo = tab[TAB_HS + k] * Vr - tab[TAB_HX + k] * Vi;
  if (k1 == 0 && lane == 0) {
    T Xr = Er - P, Xi = Q - Ei;                       // V[32]
    hi = tab[TAB_HS + 32] * Xr - tab[TAB_HX + 32] * Xi;
  } else {
    hi = tab[TAB_HS + 64 - k] * Vr + tab[TAB_HX + 64 - k] * Vi;   // V[64-k] = conj V[k]
  }
}
template <typename T>
DCTZ_HD void fwd_split(const T (&zr)[8], const T (&zi)[8], const T (&pr)[8], const T (&pi)[8],
                       int lane, const T* tab, T (&lo)[8], T (&hi)[8]) {
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) fwd_split_one<T>(k1, zr[k1], zi[k1], pr[k1], pi[k1], lane, tab, lo[k1], hi[k1]);
}

// ---------------------------------------------------------------- inverse ---
// I1: coefficients -> G[k] = c[k] + conj(c[64-k]), c[k] = (ias[k] + i iax[k]) a[k]
// (dct.c:166-172).  lo/hi as produced by fwd_split.  g32 is meaningful in quad
// lane 0 only (G[32]).
template <typename T>
DCTZ_HD void inv_prepare(const T (&lo)[8], const T (&hi)[8], int lane, const T* tab,
                         T (&gr)[8], T (&gi)[8], T& g32r, T& g32i) {
  const int q = lane_q(lane);
  g32r = T(0); g32i = T(0);
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    const int k = 8 * q + k1;
    T cr = tab[TAB_IAS + k] * lo[k1], ci = tab[TAB_IAX + k] * lo[k1];
    if (k1 == 0 && lane == 0) {
      T dr = tab[TAB_IAS + 32] * hi[0], di = tab[TAB_IAX + 32] * hi[0];
      gr[0] = cr + cr; gi[0] = ci - ci;
      g32r = dr + dr; g32i = di - di;
    } else {
      T dr = tab[TAB_IAS + 64 - k] * hi[k1], di = tab[TAB_IAX + 64 - k] * hi[k1];
      gr[k1] = cr + dr; gi[k1] = ci - di;
    }
  }
}

// I2: merge G[k] with G[32-k] into the 32-point spectrum Zb[8q+k1].
// p[k1] = G[32-(8q+k1)] (same partner pattern as fwd_split; quad lane 0 / k1 = 0
// pairs with its own G[32]).
template <typename T>
DCTZ_HD void inv_merge(const T (&gr)[8], const T (&gi)[8], const T (&pr)[8], const T (&pi)[8],
                       int lane, const T* tab, T (&zr)[8], T (&zi)[8]) {
  const int q = lane_q(lane);
#pragma unroll
  for (int k1 = 0; k1 < 8; k1++) {
    const int k = 8 * q + k1;
    T ar = gr[k1], ai = gi[k1], hr = pr[k1], hi_ = pi[k1];
    T Pr = ar + hr, Pi = ai - hi_, Dr = ar - hr, Di = ai + hi_;
    T cw = tab[TAB_CW + k], sw = tab[TAB_SW + k];
    T Qr = cw * Dr - sw * Di;
    T Qi = cw * Di + sw * Dr;
    zr[k1] = Pr - Qi; zi[k1] = Pi + Qr;
  }
}

// I3a / I3b: cross-lane radix-4, backward (partner = lane^1, then lane^2).
template <typename T>
DCTZ_HD void inv_cross_a(T (&yr)[8], T (&yi)[8], const T (&pr)[8], const T (&pi)[8], int lane) {
  const bool up = (lane & 1) != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    T r = bfly(yr[k], pr[k], up), i = bfly(yi[k], pi[k], up);
    if (lane == 3) { yr[k] = -i; yi[k] = r; }   // times +i
    else { yr[k] = r; yi[k] = i; }
  }
}
template <typename T>
DCTZ_HD void inv_cross_b(T (&yr)[8], T (&yi)[8], const T (&pr)[8], const T (&pi)[8], int lane) {
  const bool up = (lane & 2) != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { yr[k] = bfly(yr[k], pr[k], up); yi[k] = bfly(yi[k], pi[k], up); }
}

// I4: 32-point twiddle + in-lane radix-8 backward (the reference's 1/128 --
// "/dn", dct.c:185-186, and the factor 2 carried by G -- rides in the input tables).  On exit y = packed
// points z[4*n1 + n2] of quad lane n2: re -> pack_pos(m,0), im -> pack_pos(m,1).
template <typename T>
DCTZ_HD void inv_stage_lane(T (&yr)[8], T (&yi)[8], int n2, const T* tab) {
#pragma unroll
  for (int k1 = 1; k1 < 8; k1++) {
    T wr = tab[TAB_W32R + n2 * 8 + k1], wi = tab[TAB_W32I + n2 * 8 + k1];
    T a = yr[k1], b = yi[k1];
    yr[k1] = a * wr - b * wi;       // times exp(+i 2 pi n2 k1/32)
    yi[k1] = a * wi + b * wr;
  }
  fft8<T, false>(yr, yi, tab[TAB_R]);   // (the 1/128 is already inside TAB_IAS / TAB_IAX: dctz_tables.h)
}

}  // namespace dctz
