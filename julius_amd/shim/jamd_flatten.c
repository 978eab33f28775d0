/*
 * jamd_flatten.c -- HTK_HMM_INFO -> jamd_gmm_desc (see jamd_flatten.h).
 *
 * Structures walked (all reference headers, nothing redefined here):
 *   HTK_HMM_INFO.ststart / totalstatenum      libsent/include/sent/htk_hmm.h:337,398
 *   HTK_HMM_State {nstream, pdf[], id, next}  htk_hmm.h:163-170
 *   HTK_HMM_PDF {tmix, mix_num, b, bweight}   htk_hmm.h:151-159
 *   GCODEBOOK {num, d[], id}                  htk_hmm.h:196-201
 *   HTK_HMM_Dens {mean, var->vec, gconst}     htk_hmm.h:120-131
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jamd_flatten.h"
#include <sent/speech.h>   /* OUTPROB_CACHE_PERIOD */
#include <sent/util.h>     /* mymalloc, mybmalloc2 */

/* open-addressing pointer -> index map for density de-duplication */
typedef struct { const void **key; int *val; size_t cap; } pmap;

static size_t phash(const void *p, size_t cap) {
  size_t x = (size_t)p;
  x ^= x >> 17; x *= (size_t)0x9E3779B97F4A7C15ull; x ^= x >> 29;
  return x & (cap - 1);
}
static int pmap_get(pmap *m, const void *k, int newval) {
  size_t i = phash(k, m->cap);
  while (m->key[i] != NULL && m->key[i] != k) i = (i + 1) & (m->cap - 1);
  if (m->key[i] == NULL) { m->key[i] = k; m->val[i] = newval; return -1; }
  return m->val[i];
}

int jamd_flatten_hmminfo(HTK_HMM_INFO *hmminfo, jamd_flat_gmm *out)
{
  HTK_HMM_State *st;
  int S = hmminfo->totalstatenum;
  int D = hmminfo->opt.vec_size;
  int E = 0, G = 0, s, i;
  pmap m;

  memset(out, 0, sizeof(*out));
  if (hmminfo->opt.stream_info.num != 1) return JAMD_EINVAL;
  if (!hmminfo->variance_inversed) return JAMD_EINVAL;

  /* pass 1: count entries and distinct densities */
  for (st = hmminfo->ststart; st; st = st->next) {
    HTK_HMM_PDF *p = st->pdf[0];
    E += p->tmix ? ((GCODEBOOK *)p->b)->num : p->mix_num;
  }
  m.cap = 64; while (m.cap < (size_t)(2 * E + 16)) m.cap <<= 1;
  m.key = (const void **)calloc(m.cap, sizeof(void *));
  m.val = (int *)malloc(m.cap * sizeof(int));
  out->st_off = (int *)malloc(sizeof(int) * (S + 1));
  out->ent_dens = (int *)malloc(sizeof(int) * (E ? E : 1));
  out->ent_logw = (float *)malloc(sizeof(float) * (E ? E : 1));
  out->st_book = (int *)malloc(sizeof(int) * S);
  out->mean = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1) * D);
  out->ivar = (float *)malloc(sizeof(float) * (size_t)(E ? E : 1) * D);
  out->gconst = (float *)malloc(sizeof(float) * (E ? E : 1));
  for (s = 0; s <= S; s++) out->st_off[s] = -1;

  /* pass 2: states are visited in list order but placed by id; entry ranges
   * are assigned in id order afterwards, so first record sizes */
  {
    int *cnt = (int *)calloc(S, sizeof(int));
    for (st = hmminfo->ststart; st; st = st->next) {
      HTK_HMM_PDF *p = st->pdf[0];
      if (st->id < 0 || st->id >= S) { free(cnt); free(m.key); free(m.val); return JAMD_EINVAL; }
      cnt[st->id] = p->tmix ? ((GCODEBOOK *)p->b)->num : p->mix_num;
    }
    out->st_off[0] = 0;
    for (s = 0; s < S; s++) out->st_off[s + 1] = out->st_off[s] + cnt[s];
    free(cnt);
  }
  for (st = hmminfo->ststart; st; st = st->next) {
    HTK_HMM_PDF *p = st->pdf[0];
    HTK_HMM_Dens **dl;
    int n, e0 = out->st_off[st->id];
    if (p->tmix) {
      GCODEBOOK *bk = (GCODEBOOK *)p->b;
      dl = bk->d; n = bk->num;
      out->st_book[st->id] = bk->id;
      if (n > out->book_size_max) out->book_size_max = n;
    } else {
      dl = p->b; n = p->mix_num;
      out->st_book[st->id] = -1;
    }
    for (i = 0; i < n; i++) {
      HTK_HMM_Dens *d = dl[i];
      int gi = -1;
      if (d != NULL) {
        if (d->meanlen != D || d->var->len != D) { free(m.key); free(m.val); return JAMD_EINVAL; }
        gi = pmap_get(&m, d, G);
        if (gi < 0) {
          gi = G++;
          memcpy(out->mean + (size_t)gi * D, d->mean, sizeof(float) * D);
          memcpy(out->ivar + (size_t)gi * D, d->var->vec, sizeof(float) * D);
          out->gconst[gi] = d->gconst;
        }
      }
      out->ent_dens[e0 + i] = gi;
      out->ent_logw[e0 + i] = p->bweight[i];
    }
  }
  free(m.key); free(m.val);

  out->desc.nstate = S; out->desc.veclen = D; out->desc.ndens = G; out->desc.nentry = E;
  out->desc.nbook = hmminfo->is_tied_mixture ? hmminfo->codebooknum : 0;
  out->desc.nstream = 1;
  out->desc.mean = out->mean; out->desc.ivar = out->ivar; out->desc.gconst = out->gconst;
  out->desc.st_off = out->st_off; out->desc.ent_dens = out->ent_dens;
  out->desc.ent_logw = out->ent_logw; out->desc.st_book = out->st_book;
  return JAMD_OK;
}

void jamd_flat_gmm_free(jamd_flat_gmm *f)
{
  free(f->mean); free(f->ivar); free(f->gconst); free(f->st_off);
  free(f->ent_dens); free(f->ent_logw); free(f->st_book);
  memset(f, 0, sizeof(*f));
}

float *jamd_pack_param(const HTK_Param *param, int t0, int t1)
{
  int D = param->veclen, t;
  float *buf = (float *)malloc(sizeof(float) * (size_t)(t1 - t0 > 0 ? t1 - t0 : 1) * D);
  for (t = t0; t < t1; t++) memcpy(buf + (size_t)(t - t0) * D, param->parvec[t], sizeof(float) * D);
  return buf;
}

/* Make outprob_cache[0..t0+n) exist (same growth rule and allocator as the static
 * outprob_cache_extend(), libsent/src/phmm/outprob.c:109-142) and copy n rows of device
 * scores in, starting at frame t0: every later outprob_state(t, s) for those frames is a
 * cache hit (outprob.c:245-247) on values bit-identical to calc_mix()'s. */
#define JAMD_LOG_UNDEF (LOG_ZERO - 1)      /* outprob.c:68 */
int jamd_fill_outprob_cache(HMMWork *wrk, const float *scores, int t0, int n, int S)
{
  int t, s, T = t0 + n;
  if (S != wrk->statenum || t0 < 0 || n < 0) return JAMD_EINVAL;
  if (T > wrk->outprob_allocframenum) {
    int newnum = T, size;
    LOGPROB *tmpp;
    if (newnum < wrk->outprob_allocframenum + OUTPROB_CACHE_PERIOD) newnum = wrk->outprob_allocframenum + OUTPROB_CACHE_PERIOD;
    size = (newnum - wrk->outprob_allocframenum) * wrk->statenum;
    if (wrk->outprob_cache == NULL) wrk->outprob_cache = (LOGPROB **)mymalloc(sizeof(LOGPROB *) * newnum);
    else wrk->outprob_cache = (LOGPROB **)myrealloc(wrk->outprob_cache, sizeof(LOGPROB *) * newnum);
    tmpp = (LOGPROB *)mybmalloc2(sizeof(LOGPROB) * size, &(wrk->croot));
    for (t = wrk->outprob_allocframenum; t < newnum; t++) {
      wrk->outprob_cache[t] = &(tmpp[(t - wrk->outprob_allocframenum) * wrk->statenum]);
      for (s = 0; s < wrk->statenum; s++) wrk->outprob_cache[t][s] = JAMD_LOG_UNDEF;
    }
    wrk->outprob_allocframenum = newnum;
  }
  for (t = t0; t < T; t++) memcpy(wrk->outprob_cache[t], scores + (size_t)(t - t0) * S, sizeof(float) * S);
  wrk->OP_time = -1;      /* force outprob_state() to re-latch last_cache for its frame */
  return JAMD_OK;
}

int jamd_flatten_dnn(DNNData *dnn, jamd_flat_dnn *out)
{
  int nl = dnn->hnum + 1, l;
  memset(out, 0, sizeof(*out));
  out->dims = (int *)malloc(sizeof(int) * (nl + 1));
  out->w = (const float **)malloc(sizeof(float *) * nl);
  out->b = (const float **)malloc(sizeof(float *) * nl);
  for (l = 0; l < dnn->hnum; l++) {
    out->dims[l] = dnn->h[l].in;
    if (l > 0 && dnn->h[l].in != dnn->h[l - 1].out) return JAMD_EINVAL;
    out->w[l] = dnn->h[l].w; out->b[l] = dnn->h[l].b;
  }
  out->dims[dnn->hnum] = dnn->o.in;
  out->dims[nl] = dnn->o.out;
  if (dnn->hnum > 0 && dnn->o.in != dnn->h[dnn->hnum - 1].out) return JAMD_EINVAL;
  out->w[dnn->hnum] = dnn->o.w; out->b[dnn->hnum] = dnn->o.b;
  out->desc.nlayer = nl; out->desc.dims = out->dims;
  out->desc.w = out->w; out->desc.b = out->b;
  out->desc.state_prior = dnn->state_prior;      /* log10 already applied by dnn_setup() (calc_dnn.c:699-703) */
  return JAMD_OK;
}

void jamd_flat_dnn_free(jamd_flat_dnn *f) { free(f->dims); free((void *)f->w); free((void *)f->b); memset(f, 0, sizeof(*f)); }

/* ---- acoustic-model blob ---------------------------------------------------------------
 * Same container as the lexicon blob (jamd_flatten_lex.c): magic "JAMDGMM1", int32 nrec,
 * then records {char name[24], int32 dtype (0 int32, 1 float32), int32 count, payload}.
 * Lets batch workers load the flattened model without linking Julius. */
static int gmm_put(FILE *f, const char *name, int dtype, int count, const void *data)
{
  char nm[24];
  memset(nm, 0, sizeof(nm)); strncpy(nm, name, sizeof(nm) - 1);
  if (fwrite(nm, 1, 24, f) != 24 || fwrite(&dtype, 4, 1, f) != 1 || fwrite(&count, 4, 1, f) != 1) return -1;
  if (dtype == 2) {                       /* bytes, padded to a multiple of 4 */
    static const char zero[4] = {0, 0, 0, 0};
    if (count > 0 && fwrite(data, 1, (size_t)count, f) != (size_t)count) return -1;
    if ((count & 3) && fwrite(zero, 1, (size_t)(4 - (count & 3)), f) != (size_t)(4 - (count & 3))) return -1;
    return 0;
  }
  if (count > 0 && fwrite(data, 4, (size_t)count, f) != (size_t)count) return -1;
  return 0;
}

static int gmm_save_with(const jamd_gmm_desc *d, const char *path, const int *state2gs, int nstate, int nbest);

int jamd_gmm_save(const jamd_gmm_desc *d, const char *path)
{
  return gmm_save_with(d, path, NULL, 0, 0);
}

/* The selection model of -gshmm: the same blob plus the records "state2gs" (selection state of every
 * state of the real model, -1 = none) and "gms" {nbest}; jamd_gms_load() reads it. */
int jamd_gms_save(const jamd_gmm_desc *gs, const int *state2gs, int nstate, int nbest, const char *path)
{
  if (state2gs == NULL || nstate <= 0 || nbest < 1 || gs->nbook > 0) return JAMD_EINVAL;
  return gmm_save_with(gs, path, state2gs, nstate, nbest);
}

/* The verification GMMs of -gmm (libjulius/src/gmm.c): the same blob plus "model_state" (state id of
 * each model's output state, recog->gmm->start order), "rej" {-gmmnum}, "is_voice" (gc->is_voice[]:
 * 0 for the models named by -gmmreject) and "model_names" (NUL-separated); jamd_rejgmm_load() reads it. */
typedef struct { const int *model_state; int nmodel, gprune_num; const unsigned char *is_voice; const char *names; int names_len; } rej_extra;
static const rej_extra *g_rej_extra = NULL;

int jamd_rejgmm_save(const jamd_gmm_desc *gmm, const int *model_state, int nmodel, int gprune_num,
                     const unsigned char *is_voice, const char *const *names, const char *path)
{
  rej_extra x;
  char *buf; int k, len = 0, rc;
  if (model_state == NULL || nmodel < 1 || gprune_num < 1 || is_voice == NULL || names == NULL || gmm->nbook > 0) return JAMD_EINVAL;
  for (k = 0; k < nmodel; k++) len += (int)strlen(names[k]) + 1;
  buf = (char *)malloc((size_t)len);
  if (buf == NULL) return JAMD_ENOMEM;
  for (k = 0, len = 0; k < nmodel; k++) { strcpy(buf + len, names[k]); len += (int)strlen(names[k]) + 1; }
  x.model_state = model_state; x.nmodel = nmodel; x.gprune_num = gprune_num; x.is_voice = is_voice; x.names = buf; x.names_len = len;
  g_rej_extra = &x;
  rc = gmm_save_with(gmm, path, NULL, 0, 0);
  g_rej_extra = NULL;
  free(buf);
  return rc;
}

static int gmm_save_with(const jamd_gmm_desc *d, const char *path, const int *state2gs, int nstate, int nbest)
{
  FILE *f = fopen(path, "wb");
  int nrec = (d->st_book ? 8 : 7) + (state2gs ? 2 : 0) + (g_rej_extra ? 4 : 0), rc = 0, ints[6];
  if (f == NULL) return JAMD_EINVAL;
  ints[0] = d->nstate; ints[1] = d->veclen; ints[2] = d->ndens; ints[3] = d->nentry; ints[4] = d->nbook; ints[5] = d->nstream;
  fwrite("JAMDGMM1", 1, 8, f); fwrite(&nrec, 4, 1, f);
  rc |= gmm_put(f, "ints", 0, 6, ints);
  rc |= gmm_put(f, "mean", 1, d->ndens * d->veclen, d->mean);
  rc |= gmm_put(f, "ivar", 1, d->ndens * d->veclen, d->ivar);
  rc |= gmm_put(f, "gconst", 1, d->ndens, d->gconst);
  rc |= gmm_put(f, "st_off", 0, d->nstate + 1, d->st_off);
  rc |= gmm_put(f, "ent_dens", 0, d->nentry, d->ent_dens);
  rc |= gmm_put(f, "ent_logw", 1, d->nentry, d->ent_logw);
  if (d->st_book) rc |= gmm_put(f, "st_book", 0, d->nstate, d->st_book);
  if (state2gs) { rc |= gmm_put(f, "state2gs", 0, nstate, state2gs); rc |= gmm_put(f, "gms", 0, 1, &nbest); }
  if (g_rej_extra) {
    rc |= gmm_put(f, "model_state", 0, g_rej_extra->nmodel, g_rej_extra->model_state);
    rc |= gmm_put(f, "rej", 0, 1, &g_rej_extra->gprune_num);
    rc |= gmm_put(f, "is_voice", 2, g_rej_extra->nmodel, g_rej_extra->is_voice);
    rc |= gmm_put(f, "model_names", 2, g_rej_extra->names_len, g_rej_extra->names);
  }
  if (fclose(f) != 0) rc = -1;
  return rc ? JAMD_EINVAL : JAMD_OK;
}
