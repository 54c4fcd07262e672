// emu_dct64.cpp -- TEST HARNESS: walks the four quad lanes of dct64_lane.h on the
// CPU (cross-lane DPP moves become array indexing) so the kernel's arithmetic
// can be checked bit-for-bit against the oracle without a GPU.
// Build: g++ -O1 -ffp-contract=off -shared -fPIC (tests/test_lane_emulation.py).
#include "../../dctz_amd/csrc/dct64_lane.h"
#include "../../dctz_amd/csrc/dctz_tables.h"

using namespace dctz;

template <typename T>
static void emu_fwd(const T* a, T* b) {
  T tab[TAB_SIZE];
  fill_tab64<T>(tab);
  T yr[4][8], yi[4][8], pr[4][8], pi[4][8];
  for (int l = 0; l < 4; l++) {
    for (int n1 = 0; n1 < 8; n1++) {
      yr[l][n1] = a[pack_pos(4 * n1 + l, 0)];
      yi[l][n1] = a[pack_pos(4 * n1 + l, 1)];
    }
    fwd_stage_lane<T>(yr[l], yi[l], l, tab);
  }
  for (int l = 0; l < 4; l++) for (int k = 0; k < 8; k++) { pr[l][k] = yr[l ^ 2][k]; pi[l][k] = yi[l ^ 2][k]; }
  for (int l = 0; l < 4; l++) fwd_cross_a<T>(yr[l], yi[l], pr[l], pi[l], l);
  for (int l = 0; l < 4; l++) for (int k = 0; k < 8; k++) { pr[l][k] = yr[l ^ 1][k]; pi[l][k] = yi[l ^ 1][k]; }
  for (int l = 0; l < 4; l++) fwd_cross_b<T>(yr[l], yi[l], pr[l], pi[l], l);
  static const int perm0[4] = {0, 1, 3, 2};
  for (int l = 0; l < 4; l++) {
    pr[l][0] = yr[perm0[l]][0]; pi[l][0] = yi[perm0[l]][0];
    for (int k = 1; k < 8; k++) { pr[l][k] = yr[3 - l][8 - k]; pi[l][k] = yi[3 - l][8 - k]; }
  }
  for (int l = 0; l < 4; l++) {
    T lo[8], hi[8];
    fwd_split<T>(yr[l], yi[l], pr[l], pi[l], l, tab, lo, hi);
    const int q = lane_q(l);
    for (int k1 = 0; k1 < 8; k1++) {
      b[8 * q + k1] = lo[k1];
      if (l == 0 && k1 == 0) b[32] = hi[0]; else b[64 - (8 * q + k1)] = hi[k1];
    }
  }
}

template <typename T>
static void emu_inv(const T* a, T* data) {
  T tab[TAB_SIZE];
  fill_tab64<T>(tab);
  T gr[4][8], gi[4][8], pr[4][8], pi[4][8], zr[4][8], zi[4][8], g32r[4], g32i[4];
  for (int l = 0; l < 4; l++) {
    T lo[8], hi[8];
    const int q = lane_q(l);
    for (int k1 = 0; k1 < 8; k1++) {
      lo[k1] = a[8 * q + k1];
      hi[k1] = (l == 0 && k1 == 0) ? a[32] : a[64 - (8 * q + k1)];
    }
    inv_prepare<T>(lo, hi, l, tab, gr[l], gi[l], g32r[l], g32i[l]);
  }
  static const int perm0[4] = {0, 1, 3, 2};
  for (int l = 0; l < 4; l++) {
    pr[l][0] = gr[perm0[l]][0]; pi[l][0] = gi[perm0[l]][0];
    if (l == 0) { pr[0][0] = g32r[0]; pi[0][0] = g32i[0]; }
    for (int k = 1; k < 8; k++) { pr[l][k] = gr[3 - l][8 - k]; pi[l][k] = gi[3 - l][8 - k]; }
  }
  for (int l = 0; l < 4; l++) inv_merge<T>(gr[l], gi[l], pr[l], pi[l], l, tab, zr[l], zi[l]);
  for (int l = 0; l < 4; l++) for (int k = 0; k < 8; k++) { pr[l][k] = zr[l ^ 1][k]; pi[l][k] = zi[l ^ 1][k]; }
  for (int l = 0; l < 4; l++) inv_cross_a<T>(zr[l], zi[l], pr[l], pi[l], l);
  for (int l = 0; l < 4; l++) for (int k = 0; k < 8; k++) { pr[l][k] = zr[l ^ 2][k]; pi[l][k] = zi[l ^ 2][k]; }
  for (int l = 0; l < 4; l++) inv_cross_b<T>(zr[l], zi[l], pr[l], pi[l], l);
  for (int l = 0; l < 4; l++) {
    inv_stage_lane<T>(zr[l], zi[l], l, tab);
    for (int n1 = 0; n1 < 8; n1++) {
      data[pack_pos(4 * n1 + l, 0)] = zr[l][n1];
      data[pack_pos(4 * n1 + l, 1)] = zi[l][n1];
    }
  }
}

extern "C" {
void emu_fwd_f64(const double* a, double* b) { emu_fwd<double>(a, b); }
void emu_inv_f64(const double* a, double* b) { emu_inv<double>(a, b); }
void emu_fwd_f32(const float* a, float* b) { emu_fwd<float>(a, b); }
void emu_inv_f32(const float* a, float* b) { emu_inv<float>(a, b); }
void emu_rem_tab_f64(int l, double* tab) { fill_rem_tab<double>(l, tab); }
void emu_rem_tab_f32(int l, float* tab) { fill_rem_tab<float>(l, tab); }
}
