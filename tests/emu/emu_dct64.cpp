// emu_dct64.cpp -- TEST HARNESS: runs the PRODUCT's per-block transform (dctz_amd/csrc/dct64_block.h,
// the code one GPU lane executes) and its host-built tables (dctz_tables.h) on the CPU, so the kernel's
// arithmetic can be checked bit-for-bit against the oracle without a GPU.
// Build: g++ -O1 -ffp-contract=off -mfma -shared -fPIC (tests/test_lane_emulation.py).
#include "../../dctz_amd/csrc/dct64_block.h"
#include "../../dctz_amd/csrc/dct_nd_block.h"
#include "../../dctz_amd/csrc/dct64_block_eo.h"
#include "../../dctz_amd/csrc/dctz_tables.h"

using namespace dctz;

template <typename T>
static void emu(const T* a, T* b, bool inverse, int geom = 0) {
  static T tab[TBP_TOTAL];
  static bool ready = false;
  if (!ready) { fill_tab_block<T>(tab); ready = true; }
  T x[64];
  for (int i = 0; i < 64; i++) x[i] = a[i];
  if (geom == GEOM_2D) { if (inverse) dct8x8_inv<T, const T*>(x, tab); else dct8x8_fwd<T, const T*>(x, tab); }
  else if (geom == GEOM_3D) { if (inverse) dct4x4x4_inv<T, const T*>(x, tab); else dct4x4x4_fwd<T, const T*>(x, tab); }
  else if (inverse) dct64_inv<T, const T*>(x, tab); else dct64_fwd<T, const T*>(x, tab);
  for (int i = 0; i < 64; i++) b[i] = x[i];
}


// the forward transform as its even-coefficient and odd-coefficient halves (dct64_block_eo.h): what the two waves that
// share a tile in k_compress_eo compute, here one after the other on the CPU
template <typename T>
static void emu_eo(const T* a, T* b) {
  static T tab[TBP_TOTAL];
  static bool ready = false;
  if (!ready) { fill_tab_block<T>(tab); ready = true; }
  T sr[16], si[16], dr[16], di[16], ev[32], od[32];
  for (int m = 0; m < 16; m++) {
    sr[m] = a[eo_lhs(m, 0)] + a[eo_rhs(m, 0)]; si[m] = a[eo_lhs(m, 1)] + a[eo_rhs(m, 1)];
    dr[m] = a[eo_lhs(m, 0)] - a[eo_rhs(m, 0)]; di[m] = a[eo_lhs(m, 1)] - a[eo_rhs(m, 1)];
  }
  dct64_fwd_half<T, EO_EVEN, const T*>(sr, si, ev, tab);
  dct64_fwd_half<T, EO_ODD, const T*>(dr, di, od, tab);
  for (int i = 0; i < 32; i++) { b[2 * i] = ev[i]; b[2 * i + 1] = od[i]; }
}

extern "C" {
void emu_eo_f64(const double* a, double* b) { emu_eo<double>(a, b); }
void emu_eo_f32(const float* a, float* b) { emu_eo<float>(a, b); }
void emu_fwd_f64(const double* a, double* b) { emu<double>(a, b, false); }
void emu_inv_f64(const double* a, double* b) { emu<double>(a, b, true); }
void emu_fwd_f32(const float* a, float* b) { emu<float>(a, b, false); }
void emu_inv_f32(const float* a, float* b) { emu<float>(a, b, true); }
void emu_tab_f64(double* tab) { static double t[TBP_TOTAL]; fill_tab_block<double>(t); for (int i = 0; i < TB_SIZE; i++) tab[i] = t[i]; }
void emu_tab_f32(float* tab) { static float t[TBP_TOTAL]; fill_tab_block<float>(t); for (int i = 0; i < TB_SIZE; i++) tab[i] = t[i]; }
int emu_tab_size(void) { return TB_SIZE; }
// the packed fp32 form of the 64-point transform (dct64_block_pk.h)
void emu_pk_f32(const float* a, float* b, int inverse) {
  static float tab[TBP_TOTAL];
  static bool ready = false;
  if (!ready) { fill_tab_block<float>(tab); ready = true; }
  float x[64];
  for (int i = 0; i < 64; i++) x[i] = a[i];
  if (inverse) dct64_inv_pk<const float*>(x, tab); else dct64_fwd_pk<const float*>(x, tab);
  for (int i = 0; i < 64; i++) b[i] = x[i];
}
// multi-dimensional blocks (dct_nd_block.h): geom 1 = 8 x 8, 2 = 4 x 4 x 4
void emu_nd_f64(const double* a, double* b, int geom, int inverse) { emu<double>(a, b, inverse != 0, geom); }
void emu_nd_f32(const float* a, float* b, int geom, int inverse) { emu<float>(a, b, inverse != 0, geom); }
// scaling factor of util.c:29 / :43 and the decade tables the device chooses it from (dctz_tables.h)
double emu_scaling_factor(int dtype, double max_abs) { return scaling_factor(dtype, max_abs); }
void emu_decades(int dtype, int kmin, int kmax, double* thr, double* pw) {
  if (dtype == 1) decade_tables<double>(kmin, kmax, thr, pw); else decade_tables<float>(kmin, kmax, thr, pw);
}
void emu_rem_tab_f64(int l, double* tab) { fill_rem_tab<double>(l, tab); }
void emu_rem_tab_f32(int l, float* tab) { fill_rem_tab<float>(l, tab); }
}
