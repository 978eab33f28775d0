/*
 * jamd_oracle_am.c -- CPU restatement of the acoustic-scoring half of the hot
 * path.  TEST INFRASTRUCTURE ONLY -- see
 * jamd_oracle.h.  Written from the behaviour of the reference, not copied:
 * each function names the reference lines it follows.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off (the explicit fmaf() calls below are
 * the only fused operations, mirroring _mm256_fmadd_ps in calc_dnn_fma.c).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "jamd_oracle.h"

/* ---------------------------------------------------------------- addlog -- */
/* libsent/src/phmm/addlog.c:28-30 */
#define TBLSIZE 500000
#define VRANGE 15
#define TMAG 33333.3333

static float g_tbl[TBLSIZE];
static int g_tbl_built = 0;

/* addlog.c:42-57: tbl[i] = log(1 + exp(-(15*i/500000))) with the argument
 * formed in float and log/exp evaluated in double. */
const float *jo_log_tbl(void)
{
  if (!g_tbl_built) {
    for (int i = 0; i < TBLSIZE; i++) {
      float f = -((float)VRANGE * (float)i / (float)TBLSIZE);
      g_tbl[i] = (float)log(1 + exp(f));
    }
    g_tbl_built = 1;
  }
  return g_tbl;
}
int jo_log_tbl_size(void) { return TBLSIZE; }

/* addlog.c:71-92 */
float jo_addlog(float x, float y)
{
  const float *tbl = jo_log_tbl();
  float hi = (x < y) ? y : x;
  float d = (x < y) ? (x - y) : (y - x);
  if (d < JO_LOG_ADDMIN) return hi;
  unsigned int idx = (unsigned int)((-d) * TMAG + 0.5);
  return hi + tbl[idx];
}

/* addlog.c:103-123: right-to-left scan keeping the running maximum y; terms
 * more than 13.8155 below it are dropped; index math in double. */
float jo_addlog_array(const float *a, int n)
{
  const float *tbl = jo_log_tbl();
  float y = JO_LOG_ZERO;
  for (int k = n - 1; k >= 0; k--) {
    float x = a[k];
    if (x > y) { float t = x; x = y; y = t; }
    float d = x - y;
    if (d < JO_LOG_ADDMIN) continue;
    unsigned int idx = (unsigned int)((-d) * TMAG + 0.5);
    y += tbl[idx];
  }
  return y;
}

/* ------------------------------------------------------- Gaussian kernels -- */
/* gprune_none.c:59-82: tmp = gconst; tmp += (x*x)*ivar for d ascending, all in
 * fp32 with separate roundings; result tmp * -0.5. */
float jo_compute_g_base(const float *vec, const float *mean, const float *ivar,
                        float gconst, int D)
{
  float tmp = gconst;
  for (int d = 0; d < D; d++) {
    float x = vec[d] - mean[d];
    float xx = x * x;
    float t = xx * ivar[d];
    tmp = tmp + t;
  }
  return (float)(tmp * -0.5);
}

/* gprune_safe.c:76-97: same accumulation, abandon with LOG_ZERO as soon as the
 * partial sum exceeds -2*thres. */
float jo_compute_g_safe(const float *vec, const float *mean, const float *ivar,
                        float gconst, int D, float thres)
{
  float fthres = (float)(thres * (-2.0));
  float tmp = gconst;
  for (int d = 0; d < D; d++) {
    float x = vec[d] - mean[d];
    float xx = x * x;
    float t = xx * ivar[d];
    tmp = tmp + t;
    if (tmp > fthres) return JO_LOG_ZERO;
  }
  return (float)(tmp * -0.5);
}

/* gprune_common.c:41-60: binary search for the first slot whose score is not
 * greater than `score` in a descending list. */
static int find_slot(const float *sc, float score, int len)
{
  int lo = 0, hi = len - 1;
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (sc[mid] > score) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* gprune_common.c:88-126 (cache_push): keep the best `cap` (score,id) pairs
 * sorted descending; an equal score goes after existing equals at the bottom
 * and before them in the middle (that is what the binary search yields). */
static int topn_push(float *sc, int *id, int cap, int gid, float score, int len)
{
  if (len == 0) { sc[0] = score; id[0] = gid; return 1; }
  if (sc[len - 1] >= score) {
    if (len < cap) { sc[len] = score; id[len] = gid; len++; }
    return len;
  }
  int p = (sc[0] < score) ? 0 : find_slot(sc, score, len);
  if (len < cap) {
    memmove(sc + p + 1, sc + p, sizeof(float) * (len - p));
    memmove(id + p + 1, id + p, sizeof(int) * (len - p));
  } else if (p < len - 1) {
    memmove(sc + p + 1, sc + p, sizeof(float) * (len - p - 1));
    memmove(id + p + 1, id + p, sizeof(int) * (len - p - 1));
  }
  sc[p] = score; id[p] = gid;
  if (len < cap) len++;
  return len;
}

typedef struct {
  int D;
  const float *mean, *ivar, *gconst;
  const float *vec;
} gctx;

static float g_base(const gctx *c, int dens)
{
  if (dens < 0) return JO_LOG_ZERO;          /* NULL density: gprune_none.c:67 */
  return jo_compute_g_base(c->vec, c->mean + (size_t)dens * c->D,
                           c->ivar + (size_t)dens * c->D, c->gconst[dens], c->D);
}
static float g_safe(const gctx *c, int dens, float thres)
{
  if (dens < 0) return JO_LOG_ZERO;
  return jo_compute_g_safe(c->vec, c->mean + (size_t)dens * c->D,
                           c->ivar + (size_t)dens * c->D, c->gconst[dens], c->D, thres);
}

/* gprune_none.c:133-147: every density, ids in order. */
static int gset_none(const gctx *c, const int *dens, int n, float *sc, int *id)
{
  for (int i = 0; i < n; i++) { sc[i] = g_base(c, dens[i]); id[i] = i; }
  return n;
}

/* gprune_safe.c:160-202.  `last_id`/`lnum` are last frame's winners (tied
 * mixture only); `mark` is the mixcalced scratch (all zero on entry/exit). */
static int gset_safe(const gctx *c, const int *dens, int n, int cap,
                     const int *last_id, int lnum, unsigned char *mark,
                     float *sc, int *id)
{
  int num = 0;
  float thres;
  if (last_id != NULL) {
    for (int j = 0; j < lnum; j++) {
      int i = last_id[j];
      num = topn_push(sc, id, cap, i, g_base(c, dens[i]), num);
      mark[i] = 1;
    }
    thres = sc[num - 1];
    for (int i = 0; i < n; i++) {
      if (mark[i]) { mark[i] = 0; continue; }
      float s = g_safe(c, dens[i], thres);
      if (s <= thres) continue;
      num = topn_push(sc, id, cap, i, s, num);
      thres = sc[num - 1];
    }
  } else {
    thres = JO_LOG_ZERO;
    for (int i = 0; i < n; i++) {
      float s;
      if (num < cap) s = g_base(c, dens[i]);
      else { s = g_safe(c, dens[i], thres); if (s <= thres) continue; }
      num = topn_push(sc, id, cap, i, s, num);
      thres = sc[num - 1];
    }
  }
  return num;
}

/* gprune_beam.c:291-352 with history (tied-mixture codebooks, frame t >= 1): last frame's winners are computed in
 * full while the largest partial sum of every dimension is recorded (compute_g_beam_updating :153-177: the sum
 * starts at 0 and gconst is added at the END), TMBEAMWIDTH (5.0, hmm_calc.h:54) is added to those maxima, and every
 * other Gaussian is dropped at the first dimension where its partial sum exceeds the threshold
 * (compute_g_beam_pruning :192-215).  Without history: the safe-pruning branch (:337-350). */
#define JO_TMBEAMWIDTH 5.0
static int gset_beam(const gctx *c, const int *dens, int n, int cap, const int *last_id, int lnum,
                     unsigned char *mark, float *th, float *sc, int *id)
{
  int num = 0, D = c->D;
  if (last_id == NULL) return gset_safe(c, dens, n, cap, NULL, 0, mark, sc, id);
  for (int d = 0; d < D; d++) th[d] = 0.0f;                       /* clear_dimthres */
  for (int j = 0; j < lnum; j++) {
    int i = last_id[j];
    float score = JO_LOG_ZERO;
    if (dens[i] >= 0) {
      const float *mean = c->mean + (size_t)dens[i] * D, *var = c->ivar + (size_t)dens[i] * D;
      float tmp = 0.0f;
      for (int d = 0; d < D; d++) {
        float x = c->vec[d] - mean[d];
        float xx = x * x;
        float t = xx * var[d];
        tmp = tmp + t;
        if (th[d] < tmp) th[d] = tmp;
      }
      score = (float)((tmp + c->gconst[dens[i]]) * -0.5);
    }
    num = topn_push(sc, id, cap, i, score, num);
    mark[i] = 1;
  }
  for (int d = 0; d < D; d++) th[d] = (float)(th[d] + JO_TMBEAMWIDTH);   /* set_dimthres: float += double */
  for (int i = 0; i < n; i++) {
    if (mark[i]) { mark[i] = 0; continue; }
    float score = JO_LOG_ZERO;
    if (dens[i] >= 0) {
      const float *mean = c->mean + (size_t)dens[i] * D, *var = c->ivar + (size_t)dens[i] * D;
      float tmp = 0.0f;
      int d;
      for (d = 0; d < D; d++) {
        float x = c->vec[d] - mean[d];
        float xx = x * x;
        float t = xx * var[d];
        tmp = tmp + t;
        if (tmp > th[d]) break;
      }
      if (d == D) score = (float)((tmp + c->gconst[dens[i]]) * -0.5);
    }
    if (score > JO_LOG_ZERO) num = topn_push(sc, id, cap, i, score, num);
  }
  return num;
}

/* gprune_heu.c:295-352 with history: last frame's winners are computed in full while the largest TERM of every
 * dimension is recorded (compute_g_heu_updating :138-162), the maxima are summed from the last dimension backwards
 * (make_backmax :107-121: backmax[D] = 0), and every other Gaussian is dropped as soon as its partial sum plus the
 * recorded maximum of the dimensions still to come exceeds -2 x the current N-th best score
 * (compute_g_heu_pruning :180-206; the threshold follows the list, :334-345). */
static int gset_heu(const gctx *c, const int *dens, int n, int cap, const int *last_id, int lnum,
                    unsigned char *mark, float *bm, float *sc, int *id)
{
  int num = 0, D = c->D;
  float thres;
  if (last_id == NULL) return gset_safe(c, dens, n, cap, NULL, 0, mark, sc, id);
  for (int d = 0; d <= D; d++) bm[d] = 0.0f;                      /* init_backmax */
  for (int j = 0; j < lnum; j++) {
    int i = last_id[j];
    float score = JO_LOG_ZERO;
    if (dens[i] >= 0) {
      const float *mean = c->mean + (size_t)dens[i] * D, *var = c->ivar + (size_t)dens[i] * D;
      float sum = 0.0f;
      for (int d = 0; d < D; d++) {
        float x = c->vec[d] - mean[d];
        float xx = x * x;
        float tmp = xx * var[d];
        sum = sum + tmp;
        if (bm[d] < tmp) bm[d] = tmp;
      }
      score = (float)((sum + c->gconst[dens[i]]) * -0.5);
    }
    num = topn_push(sc, id, cap, i, score, num);
    mark[i] = 1;
  }
  bm[D] = 0.0f;                                                   /* make_backmax */
  for (int d = D - 1; d >= 0; d--) bm[d] = bm[d] + bm[d + 1];
  thres = sc[num - 1];
  for (int i = 0; i < n; i++) {
    if (mark[i]) { mark[i] = 0; continue; }
    float score = JO_LOG_ZERO;
    if (dens[i] >= 0) {
      const float *mean = c->mean + (size_t)dens[i] * D, *var = c->ivar + (size_t)dens[i] * D;
      float fthres = (float)(thres * (-2.0));
      float tmp = 0.0f;
      int d;
      for (d = 0; d < D; d++) {
        float x = c->vec[d] - mean[d];
        float xx = x * x;
        float t = xx * var[d];
        tmp = tmp + t;
        if (tmp + bm[d + 1] > fthres) break;
      }
      if (d == D) score = (float)((tmp + c->gconst[dens[i]]) * -0.5);
    }
    if (score > JO_LOG_ZERO) {
      num = topn_push(sc, id, cap, i, score, num);
      thres = sc[num - 1];
    }
  }
  return num;
}

/* compute_gaussset as outprob_init.c:99-147 selects it */
static int gset_pruned(int gprune, const gctx *c, const int *dens, int n, int cap, const int *last_id, int lnum,
                       unsigned char *mark, float *dimwork, float *sc, int *id)
{
  if (gprune == JO_GPRUNE_BEAM) return gset_beam(c, dens, n, cap, last_id, lnum, mark, dimwork, sc, id);
  if (gprune == JO_GPRUNE_HEU) return gset_heu(c, dens, n, cap, last_id, lnum, mark, dimwork, sc, id);
  return gset_safe(c, dens, n, cap, last_id, lnum, mark, sc, id);
}

/* calc_mix.c:75-80 / calc_tied_mix.c:231-236 for a single stream with stream
 * weight 1: logprobsum = 0 + logprob*1; LOG_ZERO if 0 or <= LOG_ZERO; result is
 * the double product with INV_LOG_TEN rounded to float. */
static float finish_state(float logprob)
{
  float logprobsum = 0.0f;
  if (!(logprob <= JO_LOG_ZERO)) logprobsum += logprob * 1.0f;
  if (logprobsum == 0.0f) return JO_LOG_ZERO;
  if (logprobsum <= JO_LOG_ZERO) return JO_LOG_ZERO;
  return (float)(logprobsum * JO_INV_LOG_TEN);
}

int jo_tmix_topn(int D, const float *mean, const float *ivar, const float *gconst,
                 const int *book_dens, int book_num, int gprune, int gprune_num,
                 const float *frames, int T,
                 float *out_score, int *out_id, int *out_num)
{
  int cap = (gprune == JO_GPRUNE_NONE) ? book_num : gprune_num;
  unsigned char *mark = calloc(book_num > 0 ? book_num : 1, 1);
  float *sc = malloc(sizeof(float) * (book_num + 1));
  int *id = malloc(sizeof(int) * (book_num + 1));
  gctx c = { D, mean, ivar, gconst, NULL };
  float *dimwork = malloc(sizeof(float) * (D + 1));
  int lastn = 0; const int *last = NULL;
  for (int t = 0; t < T; t++) {
    c.vec = frames + (size_t)t * D;
    int num;
    if (gprune == JO_GPRUNE_NONE) num = gset_none(&c, book_dens, book_num, sc, id);
    else num = gset_pruned(gprune, &c, book_dens, book_num, cap, (t >= 1 && lastn > 0) ? last : NULL,
                           lastn, mark, dimwork, sc, id);
    out_num[t] = num;
    memcpy(out_score + (size_t)t * cap, sc, sizeof(float) * num);
    memcpy(out_id + (size_t)t * cap, id, sizeof(int) * num);
    last = out_id + (size_t)t * cap; lastn = num;
  }
  free(mark); free(sc); free(id); free(dimwork);
  return 0;
}

/* outprob.c:230-242 (all states of a frame) with calc_mix.c:41 for plain states
 * and calc_tied_mix.c:162 (per-(frame,codebook) top-N cache, previous frame's
 * ids seeding the pruning) for tied-mixture states.  Frames are visited in
 * order and states in id order, as the reference's batch loop does. */
int jo_gmm_outprob(int S, int D,
                   const float *mean, const float *ivar, const float *gconst,
                   const int *st_off, const int *ent_dens, const float *ent_logw,
                   const int *st_book, int nbook,
                   int gprune, int gprune_num,
                   const float *frames, int T, float *out)
{
  int maxn = 1;
  for (int s = 0; s < S; s++) if (st_off[s + 1] - st_off[s] > maxn) maxn = st_off[s + 1] - st_off[s];
  int cap = (gprune == JO_GPRUNE_NONE) ? maxn : gprune_num;
  if (cap > maxn) cap = maxn;
  /* gprune_none_init forces OP_gprune_num to the maximum (gprune_none.c:110) */
  float *sc = malloc(sizeof(float) * (maxn + 1));
  int *id = malloc(sizeof(int) * (maxn + 1));
  unsigned char *mark = calloc(maxn + 1, 1);
  float *dimwork = malloc(sizeof(float) * (D + 1));
  /* per-codebook cache for the current and the previous frame */
  float *bsc[2] = { NULL, NULL }; int *bid[2] = { NULL, NULL }; int *bnum[2] = { NULL, NULL };
  if (nbook > 0) {
    for (int k = 0; k < 2; k++) {
      bsc[k] = malloc(sizeof(float) * nbook * cap);
      bid[k] = malloc(sizeof(int) * nbook * cap);
      bnum[k] = calloc(nbook, sizeof(int));
    }
  }
  gctx c = { D, mean, ivar, gconst, NULL };
  for (int t = 0; t < T; t++) {
    int cur = t & 1, prv = cur ^ 1;
    c.vec = frames + (size_t)t * D;
    if (nbook > 0) memset(bnum[cur], 0, sizeof(int) * nbook);
    for (int s = 0; s < S; s++) {
      const int *dens = ent_dens + st_off[s];
      const float *logw = ent_logw + st_off[s];
      int n = st_off[s + 1] - st_off[s];
      int num;
      int b = st_book ? st_book[s] : -1;
      if (b >= 0) {
        float *csc = bsc[cur] + (size_t)b * cap; int *cid = bid[cur] + (size_t)b * cap;
        if (bnum[cur][b] > 0) {                      /* calc_tied_mix.c:192-198 */
          num = bnum[cur][b];
          for (int i = 0; i < num; i++) sc[i] = csc[i] + logw[cid[i]];
        } else {                                     /* calc_tied_mix.c:199-227 */
          const int *last = NULL; int lnum = 0;
          if (t >= 1 && bnum[prv][b] > 0) { last = bid[prv] + (size_t)b * cap; lnum = bnum[prv][b]; }
          if (gprune == JO_GPRUNE_NONE) num = gset_none(&c, dens, n, sc, id);
          else num = gset_pruned(gprune, &c, dens, n, cap, last, lnum, mark, dimwork, sc, id);
          bnum[cur][b] = num;
          for (int i = 0; i < num; i++) { cid[i] = id[i]; csc[i] = sc[i]; sc[i] += logw[id[i]]; }
        }
      } else {                                       /* calc_mix.c:63-70 */
        if (gprune == JO_GPRUNE_NONE) num = gset_none(&c, dens, n, sc, id);
        else num = gset_pruned(gprune, &c, dens, n, cap, NULL, 0, mark, dimwork, sc, id);   /* calc_mix(): last_id == NULL */
        for (int i = 0; i < num; i++) sc[i] += logw[id[i]];
      }
      out[(size_t)t * S + s] = finish_state(jo_addlog_array(sc, num));
    }
  }
  free(sc); free(id); free(mark); free(dimwork);
  for (int k = 0; k < 2; k++) { free(bsc[k]); free(bid[k]); free(bnum[k]); }
  return 0;
}

/* ------------------------------------------------------------ outprob_cd -- */
/* outprob.c:287-400.  state_scores is one frame's row of the [T][S] cache. */
float jo_outprob_cd(const float *state_scores, const int *set_states, int set_num,
                    int method, int nbest)
{
  if (method == JO_IWCD_MAX) {                        /* outprob.c:332-344 */
    float maxprob = JO_LOG_ZERO;
    for (int i = 0; i < set_num; i++) {
      float p = state_scores[set_states[i]];
      if (maxprob < p) maxprob = p;
    }
    return maxprob;
  }
  if (method == JO_IWCD_AVG) {                        /* outprob.c:356-370 */
    float sum = 0.0f; int j = 0;
    for (int i = 0; i < set_num; i++) {
      float p = state_scores[set_states[i]];
      if (p > JO_LOG_ZERO) { sum += p; j++; }
    }
    return sum / (float)j;
  }
  /* outprob.c:287-321: insertion into a descending list of at most nbest */
  float *best = malloc(sizeof(float) * (nbest > 0 ? nbest : 1));
  int n = 0;
  for (int i = 0; i < set_num; i++) {
    float p = state_scores[set_states[i]];
    if (p <= JO_LOG_ZERO) continue;
    if (n == 0 || p <= best[n - 1]) {
      if (n == nbest) continue;
      best[n++] = p;
    } else {
      for (int k = 0; k < n; k++) {
        if (p > best[k]) {
          memmove(best + k + 1, best + k, sizeof(float) * (n - k - ((n == nbest) ? 1 : 0)));
          best[k] = p;
          break;
        }
      }
      if (n < nbest) n++;
    }
  }
  float sum = 0.0f;
  for (int i = 0; i < n; i++) sum += best[i];
  free(best);
  return sum / (float)n;
}

/* ------------------------------------------------------------------- DNN -- */
/* calc_dnn.c:342-369 */
#define LOGISTIC_TABLE_FACTOR 20000
#define LOGISTIC_TABLE_MAX (16 * LOGISTIC_TABLE_FACTOR)
#define LOGISTIC_MIN 0.000334
#define LOGISTIC_MAX 0.999666
static float g_sig[LOGISTIC_TABLE_MAX + 1];
static int g_sig_built = 0;

const float *jo_logistic_tbl(void)
{
  if (!g_sig_built) {
    for (int i = 0; i <= LOGISTIC_TABLE_MAX; i++) {
      double x = (double)i / (double)LOGISTIC_TABLE_FACTOR - 8.0;
      g_sig[i] = (float)(1.0 / (1.0 + exp(-x)));
    }
    g_sig_built = 1;
  }
  return g_sig;
}
int jo_logistic_tbl_size(void) { return LOGISTIC_TABLE_MAX + 1; }

/* calc_dnn.c:364-369 / :813-818: the index expression is float arithmetic
 * ((x + 8.0f) * 20000 is float*int -> float) plus a double 0.5. */
float jo_logistic(float x)
{
  const float *tb = jo_logistic_tbl();
  if (x <= -8.0f) return (float)LOGISTIC_MIN;
  if (x >= 8.0f) return (float)LOGISTIC_MAX;
  return tb[(int)((x + 8.0f) * LOGISTIC_TABLE_FACTOR + 0.5)];
}

/* One layer dst = W src + b in the summation order of the selected reference
 * kernel: calc_dnn.c:510 (scalar, ascending k), calc_dnn_fma.c:19 (8 strided
 * fused partial sums, lanes added 0..7, then bias), calc_dnn_avx.c:19 (same
 * without fusion), calc_dnn_sse.c:19 (4 strided partial sums). */
static void dnn_layer(float *dst, const float *src, const float *w, const float *b,
                      int out, int in, int simd)
{
  for (int i = 0; i < out; i++) {
    const float *wr = w + (size_t)i * in;
    float r;
    if (simd == JO_DNN_SCALAR) {
      float x = 0.0f;
      for (int j = 0; j < in; j++) { float p = wr[j] * src[j]; x = x + p; }
      r = x + b[i];
    } else if (simd == JO_DNN_SSE) {
      float acc[4] = { 0, 0, 0, 0 };
      int n = in / 4;
      for (int j = 0; j < n; j++)
        for (int l = 0; l < 4; l++) { float p = wr[4 * j + l] * src[4 * j + l]; acc[l] = acc[l] + p; }
      r = acc[0] + acc[1]; r = r + acc[2]; r = r + acc[3]; r = r + b[i];
    } else {
      float acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
      int n = in / 8;                                   /* calc_dnn_fma.c:25 truncates */
      for (int j = 0; j < n; j++)
        for (int l = 0; l < 8; l++) {
          if (simd == JO_DNN_FMA) acc[l] = fmaf(src[8 * j + l], wr[8 * j + l], acc[l]);
          else { float p = src[8 * j + l] * wr[8 * j + l]; acc[l] = acc[l] + p; }
        }
      r = acc[0] + acc[1];
      for (int l = 2; l < 8; l++) r = r + acc[l];
      r = r + b[i];
    }
    dst[i] = r;
  }
}

/* calc_dnn.c:774-868: hidden layers with table logistic, output layer, then
 * last_cache[i] = INV_LOG_TEN * (x_i - addlog_array(x)) - state_prior[i]
 * (double product, float subtraction of the prior after rounding? -- no: the
 * whole right-hand side is evaluated in double and rounded once on store). */
int jo_dnn_outprob(int nlayer, const int *dims, const float *const *w, const float *const *b,
                   const float *state_prior, int simd,
                   const float *frames, int T, float *out)
{
  int maxd = 0;
  for (int l = 0; l <= nlayer; l++) if (dims[l] > maxd) maxd = dims[l];
  float *buf0 = malloc(sizeof(float) * maxd), *buf1 = malloc(sizeof(float) * maxd);
  int S = dims[nlayer];
  for (int t = 0; t < T; t++) {
    const float *src = frames + (size_t)t * dims[0];
    float *dst = buf0;
    for (int l = 0; l < nlayer - 1; l++) {
      dnn_layer(dst, src, w[l], b[l], dims[l + 1], dims[l], simd);
      for (int i = 0; i < dims[l + 1]; i++) dst[i] = jo_logistic(dst[i]);
      src = dst; dst = (dst == buf0) ? buf1 : buf0;
    }
    float *o = out + (size_t)t * S;
    dnn_layer(o, src, w[nlayer - 1], b[nlayer - 1], S, dims[nlayer - 1], simd);
    float lse = jo_addlog_array(o, S);
    for (int i = 0; i < S; i++)
      o[i] = (float)(JO_INV_LOG_TEN * (o[i] - lse) - state_prior[i]);
  }
  free(buf0); free(buf1);
  return 0;
}

/* ---- Gaussian mixture selection (-gshmm), libsent/src/phmm/gms.c + gms_gprune.c ------------
 * Per frame: every selection-model state gets max over its Gaussians of the UNWEIGHTED score, plus
 * the weight of that Gaussian, times INV_LOG_TEN (compute_g_max(), gms_gprune.c:119-170, GS_MAX_PROB
 * + LAST_BEST: last frame's best Gaussian first, then from the highest index down, strict >, with
 * the safe partial-sum cut-off of calc_contprob_with_safe_pruning() :80-110 -- which never changes
 * the maximum); the nbest highest states are "selected" (sort_gsindex_upward(), gms.c:189-243: a
 * partial heap sort over an index array that persists from frame to frame); a state of the real
 * model keeps its real score when its selection state is selected and gets that state's score
 * otherwise (gms_state(), gms.c:394-412).  scores is [T][S]: real scores in, GMS scores out;
 * states with state2gs < 0 are left alone (the reference reads out of bounds for them). */
int jo_gms_apply(int Sgs, int D, const float *mean, const float *ivar, const float *gconst,
                 const int *st_off, const int *ent_dens, const float *ent_logw,
                 const int *state2gs, int S, int nbest, const float *frames, int T, float *scores)
{
  float *fs = (float *)malloc(sizeof(float) * (size_t)Sgs);
  int *idx = (int *)malloc(sizeof(int) * (size_t)Sgs);
  int *last = (int *)malloc(sizeof(int) * (size_t)Sgs);
  int t, i, s;
  if (!fs || !idx || !last) return -1;
  for (i = 0; i < Sgs; i++) { idx[i] = i; last[i] = -1; }
  for (t = 0; t < T; t++) {
    const float *vec = frames + (size_t)t * D;
    for (i = 0; i < Sgs; i++) {                                     /* compute_gs_scores() */
      const int e0 = st_off[i], n = st_off[i + 1] - st_off[i];
      float maxprob = JO_LOG_ZERO; int maxi, k, pass;
      /* order: last best first (if any), then n-1 .. 0 skipping it */
      maxi = (last[i] != -1) ? last[i] : n - 1;
      for (pass = 0; pass < 2; pass++) {
        const int k0 = pass == 0 ? maxi : n - 1, k1 = pass == 0 ? maxi : 0;
        for (k = k0; k >= k1; k--) {
          float tmp, thr, prob; int d, g, cut = 0;
          if (pass == 1 && k == ((last[i] != -1) ? last[i] : n - 1)) continue;
          g = ent_dens[e0 + k];
          if (g < 0) prob = JO_LOG_ZERO;
          else {
            thr = (pass == 0 ? JO_LOG_ZERO : maxprob) * (-2.0f);
            tmp = gconst[g];
            for (d = 0; d < D; d++) {
              float x = vec[d] - mean[(size_t)g * D + d];
              tmp += x * x * ivar[(size_t)g * D + d];
              if (tmp > thr) { cut = 1; break; }
            }
            prob = cut ? JO_LOG_ZERO : tmp * -0.5f;
          }
          if (pass == 0) { maxprob = prob; }
          else if (prob > maxprob) { maxprob = prob; maxi = k; }
        }
      }
      last[i] = maxi;
      {
        float logprobsum = 0.0f;
        logprobsum += (maxprob + ent_logw[e0 + maxi]) * 1.0f;
        fs[i] = (float)(logprobsum * JO_INV_LOG_TEN);
      }
    }
    {                                                               /* sort_gsindex_upward() */
      const int totalnum = Sgs, neednum = nbest < Sgs ? nbest : Sgs;
      int n, root, child, parent, sd;
#define SD_(A) idx[(A) - 1]
#define SV_(A) (fs[idx[(A) - 1]])
      for (root = totalnum / 2; root >= 1; root--) {
        sd = SD_(root); parent = root;
        while ((child = parent * 2) <= totalnum) {
          if (child < totalnum && SV_(child) < SV_(child + 1)) child++;
          if (fs[sd] >= SV_(child)) break;
          SD_(parent) = SD_(child); parent = child;
        }
        SD_(parent) = sd;
      }
      n = totalnum;
      while (n > totalnum - neednum) {
        sd = SD_(n); SD_(n) = SD_(1); n--; parent = 1;
        while ((child = parent * 2) <= n) {
          if (child < n && SV_(child) < SV_(child + 1)) child++;
          if (fs[sd] >= SV_(child)) break;
          SD_(parent) = SD_(child); parent = child;
        }
        SD_(parent) = sd;
      }
#undef SD_
#undef SV_
      for (i = totalnum - neednum; i < totalnum; i++) fs[idx[i]] = JO_LOG_ZERO;    /* do_gms(): selected */
    }
    for (s = 0; s < S; s++) {                                       /* gms_state() */
      const int gsid = state2gs[s];
      if (gsid >= 0 && fs[gsid] != JO_LOG_ZERO) scores[(size_t)t * S + s] = fs[gsid];
    }
  }
  free(fs); free(idx); free(last);
  return 0;
}


/* ---- GMM-based input verification / rejection (-gmm, -gmmnum, -gmmreject) --------------------
 * libjulius/src/gmm.c keeps a private copy of the safe pruning with one difference from libsent's:
 * the Gaussians visited while the list is not yet full are scored by gmm_compute_g_base()
 * (gmm.c:177-194: the squared distances are summed from 0 and gconst is added LAST), the later ones
 * by gmm_compute_g_safe() (gmm.c:218-240: gconst first, LOG_ZERO as soon as the partial sum passes
 * -2 * the list's last score).  gmm_gprune_safe() gmm.c:296-313, gmm_calc_mix() gmm.c:335-370 (it
 * returns the last stream's value times INV_LOG_TEN; one stream here), gmm_proceed() gmm.c:574-600:
 * out[t][k] = score of model k's single output state at frame t.  model_state[k] indexes the
 * flattened state pool.  Test infrastructure only. */
int jo_rejgmm_frame_scores(int D, const float *mean, const float *ivar, const float *gconst,
                           const int *st_off, const int *ent_dens, const float *ent_logw,
                           const int *model_state, int nmodel, int gprune_num,
                           const float *frames, int T, float *out)
{
  int t, k, i, d, maxmix = 1;
  float *sc; int *id;
  for (k = 0; k < nmodel; k++) {
    const int n = st_off[model_state[k] + 1] - st_off[model_state[k]];
    if (n > maxmix) maxmix = n;
  }
  sc = (float *)malloc(sizeof(float) * (size_t)maxmix);
  id = (int *)malloc(sizeof(int) * (size_t)maxmix);
  if (!sc || !id) return -1;
  for (t = 0; t < T; t++) {
    const float *vec = frames + (size_t)t * D;
    for (k = 0; k < nmodel; k++) {
      const int e0 = st_off[model_state[k]], n = st_off[model_state[k] + 1] - e0;
      int num = 0;
      float thres = JO_LOG_ZERO, logprob, logprobsum = 0.0f;
      for (i = 0; i < n; i++) {                                   /* gmm_gprune_safe() */
        const int g = ent_dens[e0 + i];
        float score;
        if (num < gprune_num) {
          if (g < 0) score = JO_LOG_ZERO;
          else {
            float tmp = 0.0f;
            for (d = 0; d < D; d++) { float x = vec[d] - mean[(size_t)g * D + d]; tmp += x * x * ivar[(size_t)g * D + d]; }
            score = (float)((tmp + gconst[g]) * -0.5);
          }
        } else {
          if (g < 0) score = JO_LOG_ZERO;
          else {
            const float fthres = (float)(thres * (-2.0));
            float tmp = gconst[g]; int cut = 0;
            for (d = 0; d < D; d++) {
              float x = vec[d] - mean[(size_t)g * D + d];
              tmp += x * x * ivar[(size_t)g * D + d];
              if (tmp > fthres) { cut = 1; break; }
            }
            score = cut ? JO_LOG_ZERO : (float)(tmp * -0.5);
          }
          if (score <= thres) continue;
        }
        num = topn_push(sc, id, gprune_num < maxmix ? gprune_num : maxmix, i, score, num);
        thres = sc[num - 1];
      }
      for (i = 0; i < num; i++) sc[i] += ent_logw[e0 + id[i]];   /* gmm_calc_mix() */
      logprob = jo_addlog_array(sc, num);
      if (!(logprob <= JO_LOG_ZERO)) logprobsum += logprob * 1.0f;
      if (logprobsum == 0.0f || logprobsum <= JO_LOG_ZERO) out[(size_t)t * nmodel + k] = JO_LOG_ZERO;
      else out[(size_t)t * nmodel + k] = (float)(logprob * JO_INV_LOG_TEN);
    }
  }
  free(sc); free(id);
  return 0;
}
