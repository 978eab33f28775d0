// jamd_device.h -- device-side helpers shared by the scoring kernels.
#pragma once
#include "jamd_internal.h"

typedef float f2 __attribute__((ext_vector_type(2)));

namespace jamd {

// One step of addlog_array() (libsent/src/phmm/addlog.c:108-121): y is the
// running log-sum, sc the next term; the larger stays, the table adds
// log(1+e^d).  Index arithmetic in double exactly as the reference.
__device__ __forceinline__ float addlog_step(float y, float sc, const float *__restrict__ tbl,
                                             float addmin_f) {
  const bool gt = sc > y;
  const float hi = gt ? sc : y;
  const float lo = gt ? y : sc;
  const float d = lo - hi;
  float r = hi;
  if (!(d < addmin_f)) {
    const unsigned idx = (unsigned)((double)(-d) * JAMD_TMAG + 0.5);
    r = hi + tbl[idx];
  }
  return r;
}

// calc_mix.c:73-80 / calc_tied_mix.c:229-236, one stream, stream weight 1.
__device__ __forceinline__ float finish_state(float lse) {
  if (lse <= JAMD_LOG_ZERO || lse == 0.0f) return JAMD_LOG_ZERO;
  return (float)((double)lse * JAMD_INV_LOG_TEN);
}

// compute_g_base() (gprune_none.c:59-82) for a packed pair of frames held in
// registers (DT > 0, compile-time dimension) or in LDS, transposed
// [d][128 frames of the wave] (DT == 0, run-time dimension D).
// r -> record [mean(D) ivar(D) gconst ...]; returns the two tmp*-0.5 scores,
// LOG_ZERO for a NULL density (gconst stored as NaN).
template <int DT>
__device__ __forceinline__ f2 gauss_pair(const f2 *v, const float *vt, int lane, int D,
                                         const float *__restrict__ r) {
  const float gc = r[2 * D];
  f2 acc = {gc, gc};
  if constexpr (DT > 0) {
#pragma unroll
    for (int d = 0; d < DT; d++) {
      const float mu = r[d], iv = r[DT + d];
      f2 x = v[d] - f2{mu, mu};
      x = x * x;
      x = x * f2{iv, iv};
      acc = acc + x;
    }
  } else {
    for (int d = 0; d < D; d++) {
      const float mu = r[d], iv = r[D + d];
      f2 x = f2{vt[d * 128 + lane], vt[d * 128 + 64 + lane]} - f2{mu, mu};
      x = x * x;
      x = x * f2{iv, iv};
      acc = acc + x;
    }
  }
  f2 sc = {acc.x * -0.5f, acc.y * -0.5f};
  if (gc != gc) sc = f2{JAMD_LOG_ZERO, JAMD_LOG_ZERO};
  return sc;
}

// cache_push() (gprune_common.c:88-126): keep the best `cap` (score,id) pairs
// in descending order in a register-resident list of NMAX slots.
//   bottom case (sc[len-1] >= score): append if there is room, else drop;
//   otherwise insert before the first element that is not greater.
template <int NMAX>
__device__ __forceinline__ void topn_push(float (&sc)[NMAX], int (&id)[NMAX], int &len, int cap,
                                          float score, int gid) {
  int p;
  float last = score;             // value of sc[len-1] (register array: no dynamic indexing)
#pragma unroll
  for (int i = 0; i < NMAX; i++) if (i == len - 1) last = sc[i];
  if (len > 0 && last >= score) {
    p = len;                      // bottom
  } else {
    p = 0;
#pragma unroll
    for (int i = 0; i < NMAX; i++) p += (i < len && sc[i] > score) ? 1 : 0;
  }
  if (p >= cap) return;
#pragma unroll
  for (int i = NMAX - 1; i >= 1; i--) {
    if (i > p && i < cap) { sc[i] = sc[i - 1]; id[i] = id[i - 1]; }
  }
#pragma unroll
  for (int i = 0; i < NMAX; i++) {
    if (i == p) { sc[i] = score; id[i] = gid; }
  }
  if (len < cap) len++;
}

// outprob_cd() reductions over one CD_State_Set given a row of state scores
// (libsent/src/phmm/outprob.c:287-400); shared by the cdset kernel and the
// first-pass kernel.  `states[a..b)` are the member state ids.
constexpr int kNbestMax = 16;
// `row` is anything indexable by state id (a global pointer, or the first-pass kernel's RowRef).
template <typename Row>
__device__ __forceinline__ float cd_reduce(const Row &row, const int *__restrict__ states,
                                           int a, int b, int method, int nbest) {
  if (method == JAMD_IWCD_MAX) {                       // outprob_cd_max :332-344
    float m = JAMD_LOG_ZERO;
    for (int k = a; k < b; k++) { const float p = row[states[k]]; if (m < p) m = p; }
    return m;
  }
  if (method == JAMD_IWCD_AVG) {                       // outprob_cd_avg :356-370
    float sum = 0.0f; int j = 0;
    for (int k = a; k < b; k++) { const float p = row[states[k]]; if (p > JAMD_LOG_ZERO) { sum += p; j++; } }
    return sum / (float)j;
  }
  // outprob_cd_nbest :287-321 keeps a descending list of at most nbest values and
  // returns their best-first float sum / n.  The kept multiset is simply the nbest
  // largest values (ties do not change it), so for the usual small nbest a
  // register insertion chain produces the identical sum.
  if (nbest <= 4) {
    float b0 = JAMD_LOG_ZERO, b1 = JAMD_LOG_ZERO, b2 = JAMD_LOG_ZERO, b3 = JAMD_LOG_ZERO;
    int n = 0;
    auto ins = [&](float p) {
      if (p <= JAMD_LOG_ZERO) return;
      n++;
      float t;
      if (p > b0) { t = b0; b0 = p; p = t; }
      if (p > b1) { t = b1; b1 = p; p = t; }
      if (p > b2) { t = b2; b2 = p; p = t; }
      if (p > b3) { b3 = p; }
    };
    int k = a;
    // four members at a time: the id loads, then the score gathers, are independent, so
    // their latencies overlap (the gathers are what this loop waits for)
    for (; k + 4 <= b; k += 4) {
      const int s0 = states[k], s1 = states[k + 1], s2 = states[k + 2], s3 = states[k + 3];
      const float p0 = row[s0], p1 = row[s1], p2 = row[s2], p3 = row[s3];
      ins(p0); ins(p1); ins(p2); ins(p3);
    }
    for (; k < b; k++) ins(row[states[k]]);
    if (n > nbest) n = nbest;
    float sum = 0.0f;
    if (n > 0) sum += b0;
    if (n > 1) sum += b1;
    if (n > 2) sum += b2;
    if (n > 3) sum += b3;
    return sum / (float)n;
  }
  float best[kNbestMax];
  int n = 0;
#pragma unroll
  for (int q = 0; q < kNbestMax; q++) best[q] = JAMD_LOG_ZERO;
  for (int k = a; k < b; k++) {
    const float p = row[states[k]];
    if (p <= JAMD_LOG_ZERO) continue;
    // position = number of kept values >= p when appending at the bottom
    // (outprob.c:297: `prob <= maxprobs[n-1]`), else before the first smaller one
    int pos = 0;
#pragma unroll
    for (int q = 0; q < kNbestMax; q++) pos += (q < n && best[q] >= p) ? 1 : 0;
    if (pos >= nbest) continue;
#pragma unroll
    for (int q = kNbestMax - 1; q >= 1; q--) if (q > pos && q < nbest) best[q] = best[q - 1];
#pragma unroll
    for (int q = 0; q < kNbestMax; q++) if (q == pos) best[q] = p;
    if (n < nbest) n++;
  }
  float sum = 0.0f;
#pragma unroll
  for (int q = 0; q < kNbestMax; q++) if (q < n) sum += best[q];
  return sum / (float)n;
}

}  // namespace jamd
