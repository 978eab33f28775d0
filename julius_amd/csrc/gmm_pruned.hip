// gmm_pruned.hip -- the Gaussian-pruning and tied-mixture forms of GMM scoring.
//
//   gprune_safe() on plain states       libsent/src/phmm/gprune_safe.c:160-202
//                                       + calc_mix.c:63-80
//   tied-mixture codebook cache         libsent/src/phmm/calc_tied_mix.c:189-227
//   per-state re-weighting + log-sum    calc_tied_mix.c:192-198,229-236
//   cache_push() ordering               libsent/src/phmm/gprune_common.c:88-126
//
// gprune_safe is an EXACT top-N: compute_g_safe() abandons a Gaussian only when
// its partial sum already exceeds the running N-th best, so the surviving list
// is the N highest compute_g_base() scores in descending order.  The device
// therefore evaluates every Gaussian fully (same 4-op chain as K1) and keeps a
// register-resident sorted top-N per frame with cache_push()'s insertion rule.
// (With exactly tied scores the reference's survivor depends on its visiting
// order, which for tied-mixture frames t>0 starts from frame t-1's winners;
// the device visits in index order.  See DESIGN.md "ties".)
//
// Same lane = frame-pair mapping as the tile kernel: Gaussian records come
// through scalar loads, the D-loop is packed VALU.
#include "jamd_device.h"

namespace {
using namespace jamd;

constexpr int kWaves = 4;

// Load the wave's 128 frames: registers (DT>0) or LDS transposed (DT==0).
template <int DT>
__device__ __forceinline__ void load_frames(f2 *v, float *vt, const float *__restrict__ frames,
                                            int t0, int T, int D, int lane) {
  int ta = t0 + lane, tb = ta + 64;
  if (ta > T - 1) ta = T - 1;
  if (tb > T - 1) tb = T - 1;
  const float *fa = frames + (size_t)ta * D, *fb = frames + (size_t)tb * D;
  if constexpr (DT > 0) {
#pragma unroll
    for (int d = 0; d < DT; d++) { v[d].x = fa[d]; v[d].y = fb[d]; }
  } else {
    for (int d = 0; d < D; d++) { vt[d * 128 + lane] = fa[d]; vt[d * 128 + 64 + lane] = fb[d]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- plain states, gprune safe (top-`cap` by raw Gaussian score) ------------
template <int DT, int NMAX>
__global__ void __launch_bounds__(64 * kWaves)
gmm_safe_kernel(const float *__restrict__ rec, const int *__restrict__ st_off,
                const float *__restrict__ frames, const float *__restrict__ tbl,
                float *__restrict__ out, int T, int S, int D, int REC, int cap, int nsb,
                float addmin_f) {
  extern __shared__ __align__(16) float dyn[];
  // results of NS states x the wave's 128 frames are staged in a wave-private LDS tile and written as 64-byte
  // row segments (a lane storing its own [t][s] element makes 64 scattered 4-byte stores per instruction)
  constexpr int NS = 16;
  __shared__ float tile[kWaves][128][NS + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * kWaves + wave) * 128;
  if (t0 >= T) return;
  f2 v[DT > 0 ? DT : 1];
  float *vt = dyn + (size_t)wave * D * 128;
  load_frames<DT>(v, vt, frames, t0, T, D, lane);
  const int s_begin = blockIdx.y * nsb, s_end = min(S, s_begin + nsb);
  for (int sg = s_begin; sg < s_end; sg += NS) {
   const int ns = min(NS, s_end - sg);
   for (int si = 0; si < ns; si++) {
    const int s = sg + si;
    const int e0 = st_off[s], e1 = st_off[s + 1];
    float sc0[NMAX], sc1[NMAX];
    int id0[NMAX], id1[NMAX];
    int len0 = 0, len1 = 0;
#pragma unroll
    for (int i = 0; i < NMAX; i++) { sc0[i] = sc1[i] = JAMD_LOG_ZERO; id0[i] = id1[i] = 0; }
    for (int e = e0; e < e1; e++) {
      const f2 g = gauss_pair<DT>(v, vt, lane, D, rec + (size_t)e * REC);
      topn_push<NMAX>(sc0, id0, len0, cap, g.x, e - e0);
      topn_push<NMAX>(sc1, id1, len1, cap, g.y, e - e0);
    }
    // calc_mix.c:66-72: add ln w of the survivors, log-sum from the last slot down
    float y0 = JAMD_LOG_ZERO, y1 = JAMD_LOG_ZERO;
#pragma unroll
    for (int i = NMAX - 1; i >= 0; i--) {
      if (i < len0) y0 = addlog_step(y0, sc0[i] + rec[(size_t)(e0 + id0[i]) * REC + 2 * D + 1], tbl, addmin_f);
      if (i < len1) y1 = addlog_step(y1, sc1[i] + rec[(size_t)(e0 + id1[i]) * REC + 2 * D + 1], tbl, addmin_f);
    }
    tile[wave][lane][si] = finish_state(y0);
    tile[wave][64 + lane][si] = finish_state(y1);
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   constexpr int RPI = 64 / NS;              // rows per store instruction
   const int col = lane % NS, rsub = lane / NS;
#pragma unroll 4
   for (int it = 0; it < 128 / RPI; it++) {
     const int rr = it * RPI + rsub;
     const int t = t0 + rr;
     if (t < T && col < ns) out[(size_t)t * S + sg + col] = tile[wave][rr][col];
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
  }
}

// ---- tied-mixture codebooks: per (frame, book) top-N cache ------------------
// NMAX == 0: gprune none -- every Gaussian of the book in index order.
template <int DT, int NMAX>
__global__ void __launch_bounds__(64 * kWaves)
tmix_book_kernel(const float *__restrict__ brec, const int *__restrict__ book_off,
                 const float *__restrict__ frames, float *__restrict__ c_score,
                 int *__restrict__ c_id, int *__restrict__ c_num, int T, int nbook, int D, int REC,
                 int cap) {
  extern __shared__ __align__(16) float dyn[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * kWaves + wave) * 128;
  if (t0 >= T) return;
  f2 v[DT > 0 ? DT : 1];
  float *vt = dyn + (size_t)wave * D * 128;
  load_frames<DT>(v, vt, frames, t0, T, D, lane);
  const int b = blockIdx.y;
  const int k0 = book_off[b], k1 = book_off[b + 1];
  const int ta = t0 + lane, tb = ta + 64;
  const size_t oa = ((size_t)ta * nbook + b) * cap, ob = ((size_t)tb * nbook + b) * cap;
  if constexpr (NMAX == 0) {
    for (int k = k0; k < k1; k++) {
      const f2 g = gauss_pair<DT>(v, vt, lane, D, brec + (size_t)k * REC);
      if (ta < T) { c_score[oa + (k - k0)] = g.x; c_id[oa + (k - k0)] = k - k0; }
      if (tb < T) { c_score[ob + (k - k0)] = g.y; c_id[ob + (k - k0)] = k - k0; }
    }
    if (ta < T) c_num[(size_t)ta * nbook + b] = k1 - k0;
    if (tb < T) c_num[(size_t)tb * nbook + b] = k1 - k0;
  } else {
    constexpr int N = NMAX > 0 ? NMAX : 1;
    float sc0[N], sc1[N];
    int id0[N], id1[N];
    int len0 = 0, len1 = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { sc0[i] = sc1[i] = JAMD_LOG_ZERO; id0[i] = id1[i] = 0; }
    for (int k = k0; k < k1; k++) {
      const f2 g = gauss_pair<DT>(v, vt, lane, D, brec + (size_t)k * REC);
      topn_push<N>(sc0, id0, len0, cap, g.x, k - k0);
      topn_push<N>(sc1, id1, len1, cap, g.y, k - k0);
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (i < cap) {
        if (ta < T) { c_score[oa + i] = sc0[i]; c_id[oa + i] = id0[i]; }
        if (tb < T) { c_score[ob + i] = sc1[i]; c_id[ob + i] = id1[i]; }
      }
    }
    if (ta < T) c_num[(size_t)ta * nbook + b] = len0;
    if (tb < T) c_num[(size_t)tb * nbook + b] = len1;
  }
}

// ---- tied-mixture codebooks, gprune heu / beam WITH HISTORY --------------------------------
// gprune_heu() (gprune_heu.c:295-352) and gprune_beam() (gprune_beam.c:291-352) called from calc_tied_mix() with
// last_id = the codebook's cached winners of frame t - 1 (calc_tied_mix.c:203-215).  Under eager scoring (every
// state of every frame, outprob.c:230-242) that history is the previous frame's result, so a codebook is a chain over
// the frames of ONE utterance: one wave per (codebook, utterance) walks the frames, the lanes share the codebook's
// Gaussians.  Per frame:
//   1. last frame's winners (lane j = winner j) are computed in full -- the sum starts at 0 and gconst is added at
//      the end (compute_g_beam_updating :153-177, compute_g_heu_updating :138-162: NOT compute_g_base's order) --
//      and their per-dimension partial sums (beam) / terms (heu) go to LDS; the lanes then take one dimension each:
//      beam  th[d] = max(0, partial sums) + TMBEAMWIDTH (added in double, :142);
//      heu   backmax[d] = max(0, terms), summed from the last dimension backwards (make_backmax :107-121);
//   2. every other Gaussian (lane = Gaussian): beam -- dropped if ANY partial sum exceeds th[d] (the reference returns
//      at the first one); heu -- its largest (partial sum + backmax[d + 1]) is kept, because the threshold it is
//      compared with (-2 x the list's last score) moves while the list fills (:334-345);
//   3. the survivors enter the top-N list in index order with cache_push()'s rule; the list lives in registers, one
//      entry per lane.
// Frame 0 of an utterance (and a frame whose predecessor cached nothing) takes the safe-pruning branch (:337-350) =
// compute_g_base() scores pushed in index order.  Same MIXCACHE layout as tmix_book_kernel.
struct TopList {                 // lane p holds entry p of the descending list (cap <= 64)
  float sc; int id; int len, cap;
  __device__ __forceinline__ void push(float score, int gid, int lane) {      // cache_push(), gprune_common.c:88-126
    const float last = __shfl(sc, len > 0 ? len - 1 : 0, 64);
    if (len > 0 && last >= score) {                                            // bottom: append if there is room
      if (len < cap) { if (lane == len) { sc = score; id = gid; } len++; }
      return;
    }
    const int p = __popcll(__ballot(lane < len && sc > score));                // entries strictly greater stay in front
    const float usc = __shfl_up(sc, 1, 64); const int uid = __shfl_up(id, 1, 64);
    if (lane > p && lane <= len && lane < cap) { sc = usc; id = uid; }
    if (lane == p) { sc = score; id = gid; }
    if (len < cap) len++;
  }
};

template <int METHOD>
__global__ void __launch_bounds__(64)
tmix_book_hist_kernel(const float *__restrict__ brec, const int *__restrict__ book_off, const float *__restrict__ frames,
                      const int *__restrict__ utt_off, float *__restrict__ c_score, int *__restrict__ c_id,
                      int *__restrict__ c_num, int nbook, int D, int REC, int cap, int kmax) {
  extern __shared__ __align__(16) float dyn[];
  float *x = dyn;                         // [D]     the frame
  float *th = x + D;                      // [D + 1] beam: thresholds; heu: backmax
  float *part = th + D + 1;               // [cap][D] per-dimension values of last frame's winners
  int *calced = reinterpret_cast<int *>(part + (size_t)cap * D);   // [kmax] mixcalced
  const int lane = threadIdx.x, b = blockIdx.x, u = blockIdx.y;
  const int k0 = book_off[b], K = book_off[b + 1] - k0;
  const int t_begin = utt_off[u], t_end = utt_off[u + 1];
  for (int i = lane; i < K; i += 64) calced[i] = 0;
  TopList L; L.sc = JAMD_LOG_ZERO; L.id = 0; L.len = 0; L.cap = cap;
  int last_id = 0, lnum = 0;              // lane j: winner j of the previous frame
  auto sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int t = t_begin; t < t_end; t++) {
    for (int d = lane; d < D; d += 64) x[d] = frames[(size_t)t * D + d];
    sync();
    L.len = 0;
    if (t == t_begin || lnum == 0) {
      // no history: safe pruning = the N best compute_g_base() scores, pushed in index order
      for (int i0 = 0; i0 < K; i0 += 64) {
        const int i = i0 + lane;
        float sc = JAMD_LOG_ZERO;
        if (i < K) {
          const float *__restrict__ r = brec + (size_t)(k0 + i) * REC;
          const float gc = r[2 * D];
          float acc = gc;
          for (int d = 0; d < D; d++) { float v = x[d] - r[d]; v = v * v; v = v * r[D + d]; acc = acc + v; }
          sc = (gc != gc) ? JAMD_LOG_ZERO : acc * -0.5f;
        }
        // (a NULL density scores LOG_ZERO and is pushed like any other, gprune_safe.c:185-196)
        unsigned long long m = __ballot(i < K);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          L.push(__shfl(sc, l, 64), i0 + l, lane);
        }
      }
    } else {
      // 1. last frame's winners
      float psc = JAMD_LOG_ZERO;
      if (lane < lnum) {
        const float *__restrict__ r = brec + (size_t)(k0 + last_id) * REC;
        const float gc = r[2 * D];
        const bool nulld = gc != gc;
        float tmp = 0.0f;
        for (int d = 0; d < D; d++) {
          float v = x[d] - r[d]; v = v * v; v = v * r[D + d];
          if (METHOD == JAMD_GPRUNE_BEAM) { tmp = tmp + v; part[lane * D + d] = nulld ? 0.0f : tmp; }
          else { tmp = tmp + v; part[lane * D + d] = nulld ? 0.0f : v; }
        }
        psc = nulld ? JAMD_LOG_ZERO : (tmp + gc) * -0.5f;
        calced[last_id] = 1;
      }
      sync();
      for (int d = lane; d < D; d += 64) {
        float m = 0.0f;
        for (int j = 0; j < lnum; j++) { const float v = part[j * D + d]; if (m < v) m = v; }
        th[d] = (METHOD == JAMD_GPRUNE_BEAM) ? (float)((double)m + 5.0) : m;         // TMBEAMWIDTH, hmm_calc.h:54
      }
      for (int j = 0; j < lnum; j++) L.push(__shfl(psc, j, 64), __shfl(last_id, j, 64), lane);
      sync();
      if (METHOD == JAMD_GPRUNE_HEU) {
        if (lane == 0) {
          th[D] = 0.0f;
          for (int d = D - 1; d >= 0; d--) th[d] = th[d] + th[d + 1];
        }
        sync();
      }
      // 2. the rest, 3. pushed in index order
      float thres = __shfl(L.sc, L.len - 1, 64);
      for (int i0 = 0; i0 < K; i0 += 64) {
        const int i = i0 + lane;
        float sc = JAMD_LOG_ZERO, mx = 0.0f;
        bool cand = false;
        if (i < K) {
          if (calced[i]) calced[i] = 0;
          else {
            const float *__restrict__ r = brec + (size_t)(k0 + i) * REC;
            const float gc = r[2 * D];
            float tmp = 0.0f;
            bool pruned = false;
            mx = -3.0e38f;
            for (int d = 0; d < D; d++) {
              float v = x[d] - r[d]; v = v * v; v = v * r[D + d];
              tmp = tmp + v;
              if (METHOD == JAMD_GPRUNE_BEAM) pruned |= tmp > th[d];
              else { const float w = tmp + th[d + 1]; if (mx < w) mx = w; }
            }
            sc = (gc != gc || pruned) ? JAMD_LOG_ZERO : (tmp + gc) * -0.5f;
            cand = sc > JAMD_LOG_ZERO;
          }
        }
        // heu: once the list is full its last score only rises, so a Gaussian over the CURRENT threshold is out for good
        // (while the list still grows an appended entry LOWERS it: no shortcut then)
        if (METHOD == JAMD_GPRUNE_HEU && L.len == L.cap) cand = cand && !(mx > thres * -2.0f);
        unsigned long long m = __ballot(cand);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          const float s_l = __shfl(sc, l, 64);
          if (METHOD == JAMD_GPRUNE_HEU) {
            if (__shfl(mx, l, 64) > thres * -2.0f) continue;
            L.push(s_l, i0 + l, lane);
            thres = __shfl(L.sc, L.len - 1, 64);
          } else {
            L.push(s_l, i0 + l, lane);
          }
        }
      }
    }
    const size_t o = ((size_t)t * nbook + b) * cap;
    if (lane < L.len) { c_score[o + lane] = L.sc; c_id[o + lane] = L.id; }
    if (lane == 0) c_num[(size_t)t * nbook + b] = L.len;
    last_id = L.id; lnum = L.len;
    sync();
  }
}

// ---- tied-mixture states: weights of the cached winners + log-sum -----------
__global__ void __launch_bounds__(256)
tmix_state_kernel(const int *__restrict__ tied_states, int ntied, const int *__restrict__ st_off,
                  const int *__restrict__ st_book, const float *__restrict__ ent_logw,
                  const float *__restrict__ c_score, const int *__restrict__ c_id,
                  const int *__restrict__ c_num, const float *__restrict__ tbl,
                  float *__restrict__ out, int T, int S, int nbook, int cap, float addmin_f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntied) return;
  const int s = tied_states[i];
  const int b = st_book[s];
  const int e0 = st_off[s];
  for (int t = blockIdx.y; t < T; t += gridDim.y) {        // gridDim.y is capped (65535 limit): frames are strided
    const size_t o = ((size_t)t * nbook + b) * cap;
    const int n = c_num[(size_t)t * nbook + b];
    float y = JAMD_LOG_ZERO;
    for (int k = n - 1; k >= 0; k--) {
      const float x = c_score[o + k] + ent_logw[e0 + c_id[o + k]];  // calc_tied_mix.c:194-196
      y = addlog_step(y, x, tbl, addmin_f);
    }
    out[(size_t)t * S + s] = finish_state(y);
  }
}

template <int DT>
int launch_safe(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  const int nfb = (T + 128 * kWaves - 1) / (128 * kWaves);
  int nsb = 16;         // one output tile of states per block: short blocks keep a launch's last round short (see launch_tile())
  while ((g->S + nsb - 1) / nsb > 65535) nsb *= 2;   // grid.y limit
  const dim3 grid(nfb, (g->S + nsb - 1) / nsb);
  const size_t dyn = DT > 0 ? 0 : sizeof(float) * kWaves * g->D * 128;
  const int cap = g->gprune_num < g->maxmix ? g->gprune_num : g->maxmix;
#define JAMD_SAFE(N)                                                                         \
  do {                                                                                       \
    const int rc_ = jamd_reserve_dyn_lds((const void *)gmm_safe_kernel<DT, N>, dyn, "gprune safe"); \
    if (rc_ != JAMD_OK) return rc_;                                                          \
    hipLaunchKernelGGL((gmm_safe_kernel<DT, N>), grid, dim3(64 * kWaves), dyn, st, g->d_rec, \
                       g->d_st_off_plain, frames, g->eng->d_addlog, out, T, g->S, g->D, g->rec, \
                       cap, nsb, g->eng->addmin_f);                                          \
  } while (0)
  if (cap <= 2) JAMD_SAFE(2);
  else if (cap <= 4) JAMD_SAFE(4);
  else if (cap <= 8) JAMD_SAFE(8);
  else if (cap <= 16) JAMD_SAFE(16);
  else if (cap <= 32) JAMD_SAFE(32);
  else JAMD_SAFE(64);
#undef JAMD_SAFE
  snprintf(g->last_kernel, sizeof(g->last_kernel), "gmm_safe<DT=%d> cap=%d", DT, cap);
  return JAMD_OK;
}

int launch_book_hist(jamd_gmm *g, const float *frames, float *c_score, int *c_id, int *c_num, hipStream_t st) {
  const dim3 grid(g->nbook, g->cur_nutt);
  const size_t dyn = sizeof(float) * ((size_t)g->D + (size_t)g->D + 1 + (size_t)g->tm_cap * g->D) + sizeof(int) * (size_t)g->maxbook;
  const void *fn = g->hist_method == JAMD_GPRUNE_BEAM ? (const void *)tmix_book_hist_kernel<JAMD_GPRUNE_BEAM>
                                                       : (const void *)tmix_book_hist_kernel<JAMD_GPRUNE_HEU>;
  const int rc = jamd_reserve_dyn_lds(fn, dyn, "gprune heu/beam over tied-mixture codebooks");
  if (rc != JAMD_OK) return rc;
  if (g->hist_method == JAMD_GPRUNE_BEAM)
    hipLaunchKernelGGL(tmix_book_hist_kernel<JAMD_GPRUNE_BEAM>, grid, dim3(64), dyn, st, g->d_book_rec, g->d_book_off, frames,
                       g->d_cur_utt_off, c_score, c_id, c_num, g->nbook, g->D, g->rec, g->tm_cap, g->maxbook);
  else
    hipLaunchKernelGGL(tmix_book_hist_kernel<JAMD_GPRUNE_HEU>, grid, dim3(64), dyn, st, g->d_book_rec, g->d_book_off, frames,
                       g->d_cur_utt_off, c_score, c_id, c_num, g->nbook, g->D, g->rec, g->tm_cap, g->maxbook);
  return JAMD_OK;
}

template <int DT>
int launch_book(jamd_gmm *g, const float *frames, int T, float *c_score, int *c_id, int *c_num,
                hipStream_t st) {
  const int nfb = (T + 128 * kWaves - 1) / (128 * kWaves);
  const dim3 grid(nfb, g->nbook);
  const size_t dyn = DT > 0 ? 0 : sizeof(float) * kWaves * g->D * 128;
  const int cap = g->tm_cap;
#define JAMD_BOOK(N)                                                                          \
  do {                                                                                        \
    const int rc_ = jamd_reserve_dyn_lds((const void *)tmix_book_kernel<DT, N>, dyn, "tied-mixture codebooks"); \
    if (rc_ != JAMD_OK) return rc_;                                                           \
    hipLaunchKernelGGL((tmix_book_kernel<DT, N>), grid, dim3(64 * kWaves), dyn, st, g->d_book_rec, \
                       g->d_book_off, frames, c_score, c_id, c_num, T, g->nbook, g->D, g->rec, cap); \
  } while (0)
  if (g->gprune == JAMD_GPRUNE_NONE) JAMD_BOOK(0);
  else if (cap <= 2) JAMD_BOOK(2);
  else if (cap <= 4) JAMD_BOOK(4);
  else if (cap <= 8) JAMD_BOOK(8);
  else if (cap <= 16) JAMD_BOOK(16);
  else if (cap <= 32) JAMD_BOOK(32);
  else JAMD_BOOK(64);
#undef JAMD_BOOK
  return JAMD_OK;
}

}  // namespace

// entry points used by gmm_outprob.hip
int jamd_gmm_launch_safe(jamd_gmm *g, const float *frames, int T, float *out, hipStream_t st) {
  switch (g->D) {
    case 39: return launch_safe<39>(g, frames, T, out, st);
    case 38: return launch_safe<38>(g, frames, T, out, st);
    case 26: return launch_safe<26>(g, frames, T, out, st);
    case 25: return launch_safe<25>(g, frames, T, out, st);
    default: return launch_safe<0>(g, frames, T, out, st);
  }
}

int jamd_gmm_launch_tmix(jamd_gmm *g, const float *frames, int T, float *out, float *c_score,
                         int *c_id, int *c_num, hipStream_t st) {
  int rc;
  if (g->hist_method != 0) rc = launch_book_hist(g, frames, c_score, c_id, c_num, st);
  else switch (g->D) {
    case 39: rc = launch_book<39>(g, frames, T, c_score, c_id, c_num, st); break;
    case 38: rc = launch_book<38>(g, frames, T, c_score, c_id, c_num, st); break;
    case 26: rc = launch_book<26>(g, frames, T, c_score, c_id, c_num, st); break;
    case 25: rc = launch_book<25>(g, frames, T, c_score, c_id, c_num, st); break;
    default: rc = launch_book<0>(g, frames, T, c_score, c_id, c_num, st); break;
  }
  if (rc != JAMD_OK || !out) return rc;
  const dim3 grid((g->ntied + 255) / 256, T < 4096 ? T : 4096);
  hipLaunchKernelGGL(tmix_state_kernel, grid, dim3(256), 0, st, g->d_tied_states, g->ntied,
                     g->d_st_off, g->d_st_book, g->d_ent_logw, c_score, c_id, c_num,
                     g->eng->d_addlog, out, T, g->S, g->nbook, g->tm_cap, g->eng->addmin_f);
  return JAMD_OK;
}
