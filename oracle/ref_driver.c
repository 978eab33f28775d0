/*
 * ref_driver.c -- our own tap driver over the UNMODIFIED reference library.
 * TEST INFRASTRUCTURE ONLY.  Linked with the reference objects into
 * oracle/_ref/libjref.so (oracle/Makefile) and loaded from Python with ctypes.
 * It only *calls* reference functions (through their public headers) and
 * copies values out of the reference's own data structures; no reference code
 * is reproduced here.
 *
 * Taps (all prefixed jref_):
 *   AM   load an HTK hmmdefs with the reference loader, run outprob_init(),
 *        export the flattened model (via julius_amd/shim/jamd_flatten.c, the
 *        product-side flattening, so that code is exercised against the real
 *        structures), score [T][S] through outprob_state()/outprob_cd().
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <sent/stddefs.h>
#include <sent/htk_hmm.h>
#include <sent/htk_param.h>
#include <sent/hmm.h>
#include <sent/hmm_calc.h>
#include <sent/util.h>

#define JAMD_WITH_LIBJULIUS 1
#include "../julius_amd/shim/jamd_flatten.h"

typedef struct {
  HTK_HMM_INFO *hmminfo;
  HMMWork wrk;
  jamd_flat_gmm flat;
  int have_flat;
  HTK_HMM_State **by_id;   /* state pointer by id */
  HTK_HMM_INFO *gshmm;     /* Gaussian mixture selection model (-gshmm), or NULL */
  int view_only;           /* wrapper around another model's gshmm: nothing to free but the flat copy */
} jref_am;

static double now_sec(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void jref_quiet(int quiet) { jlog_set_output(quiet ? NULL : stderr); }

/* gprune: GPRUNE_SEL_* (1 none, 2 safe, 3 heuristic, 4 beam; hmm_calc.h:45) */
void *jref_am_load(const char *hmmdefs, const char *hmmlist, int gprune, int gprune_num,
                   int cdset_method, int cdmax_num)
{
  jref_am *a = (jref_am *)calloc(1, sizeof(jref_am));
  HTK_HMM_State *st;
  a->hmminfo = hmminfo_new();
  if (!init_hmminfo(a->hmminfo, (char *)hmmdefs, (char *)hmmlist, NULL)) { free(a); return NULL; }
  a->hmminfo->cdset_method = cdset_method;
  a->hmminfo->cdmax_num = cdmax_num;
  memset(&a->wrk, 0, sizeof(a->wrk));
  if (!outprob_init(&a->wrk, a->hmminfo, NULL, 0, gprune, gprune_num, NULL)) { free(a); return NULL; }
  a->by_id = (HTK_HMM_State **)calloc(a->hmminfo->totalstatenum, sizeof(HTK_HMM_State *));
  for (st = a->hmminfo->ststart; st; st = st->next) a->by_id[st->id] = st;
  return a;
}

/* The same with Gaussian mixture selection (-gshmm file, -gsnum n): outprob_init() wires
 * gms_state() in front of the state computation (outprob_init.c:150-180). */
void *jref_am_load_gms(const char *hmmdefs, const char *hmmlist, int gprune, int gprune_num,
                       int cdset_method, int cdmax_num, const char *gshmm, int gms_num)
{
  jref_am *a = (jref_am *)calloc(1, sizeof(jref_am));
  HTK_HMM_State *st;
  a->hmminfo = hmminfo_new();
  if (!init_hmminfo(a->hmminfo, (char *)hmmdefs, (char *)hmmlist, NULL)) { free(a); return NULL; }
  a->hmminfo->cdset_method = cdset_method;
  a->hmminfo->cdmax_num = cdmax_num;
  a->gshmm = hmminfo_new();
  if (!init_hmminfo(a->gshmm, (char *)gshmm, NULL, NULL)) { free(a); return NULL; }
  memset(&a->wrk, 0, sizeof(a->wrk));
  if (!outprob_init(&a->wrk, a->hmminfo, a->gshmm, gms_num, gprune, gprune_num, NULL)) { free(a); return NULL; }
  a->by_id = (HTK_HMM_State **)calloc(a->hmminfo->totalstatenum, sizeof(HTK_HMM_State *));
  for (st = a->hmminfo->ststart; st; st = st->next) a->by_id[st->id] = st;
  return a;
}

/* A read-only view of the selection model for jref_am_dims() / jref_am_export(). */
void *jref_am_gms_model(void *h)
{
  jref_am *a = (jref_am *)h, *v;
  if (!a || !a->gshmm) return NULL;
  v = (jref_am *)calloc(1, sizeof(jref_am));
  v->hmminfo = a->gshmm; v->view_only = 1;
  return v;
}

/* state2gs[S] (gms.c:104-160) and the number of selected states; returns the GS state count. */
int jref_am_gms_map(void *h, int *state2gs, int *nbest)
{
  jref_am *a = (jref_am *)h;
  if (!a || !a->gshmm) return -1;
  memcpy(state2gs, a->wrk.state2gs, sizeof(int) * (size_t)a->hmminfo->totalstatenum);
  *nbest = a->wrk.my_nbest;
  return a->wrk.gsset_num;
}

void jref_am_free(void *h)
{
  jref_am *a = (jref_am *)h;
  if (!a) return;
  if (a->have_flat) jamd_flat_gmm_free(&a->flat);
  if (a->view_only) { free(a); return; }
  outprob_free(&a->wrk);
  hmminfo_free(a->hmminfo);
  free(a->by_id);
  free(a);
}

int jref_am_save(void *h, const char *path);
static int ensure_flat(jref_am *a)
{
  if (a->have_flat) return 0;
  if (jamd_flatten_hmminfo(a->hmminfo, &a->flat) != 0) return -1;
  a->have_flat = 1;
  return 0;
}

/* dims: S, D, G, E, nbook, book_size_max, is_tied_mixture, maxmixturenum */
int jref_am_dims(void *h, int *dims)
{
  jref_am *a = (jref_am *)h;
  if (ensure_flat(a)) return -1;
  dims[0] = a->flat.desc.nstate; dims[1] = a->flat.desc.veclen; dims[2] = a->flat.desc.ndens;
  dims[3] = a->flat.desc.nentry; dims[4] = a->flat.desc.nbook; dims[5] = a->flat.book_size_max;
  dims[6] = a->hmminfo->is_tied_mixture; dims[7] = a->hmminfo->maxmixturenum;
  return 0;
}

int jref_am_export(void *h, float *mean, float *ivar, float *gconst, int *st_off, int *ent_dens,
                   float *ent_logw, int *st_book)
{
  jref_am *a = (jref_am *)h;
  const jamd_gmm_desc *d;
  if (ensure_flat(a)) return -1;
  d = &a->flat.desc;
  memcpy(mean, d->mean, sizeof(float) * (size_t)d->ndens * d->veclen);
  memcpy(ivar, d->ivar, sizeof(float) * (size_t)d->ndens * d->veclen);
  memcpy(gconst, d->gconst, sizeof(float) * d->ndens);
  memcpy(st_off, d->st_off, sizeof(int) * (d->nstate + 1));
  memcpy(ent_dens, d->ent_dens, sizeof(int) * d->nentry);
  memcpy(ent_logw, d->ent_logw, sizeof(float) * d->nentry);
  memcpy(st_book, d->st_book, sizeof(int) * d->nstate);
  return 0;
}

static HTK_Param *make_param(const float *frames, int T, int D)
{
  HTK_Param *p = new_param();
  int t;
  param_alloc(p, T, D);
  p->samplenum = T; p->veclen = D;
  p->header.samplenum = T; p->header.sampsize = D * sizeof(float);
  for (t = 0; t < T; t++) memcpy(p->parvec[t], frames + (size_t)t * D, sizeof(float) * D);
  return p;
}

/* Eager scoring: outprob_prepare() then, frame by frame, every state through
 * outprob_state() with batch_computation on (the -outprobout path).  Returns
 * elapsed seconds of the scoring loop (model load and param packing excluded). */
double jref_am_outprob(void *h, const float *frames, int T, float *out)
{
  jref_am *a = (jref_am *)h;
  int S = a->hmminfo->totalstatenum, D = a->hmminfo->opt.vec_size, t, s;
  HTK_Param *p = make_param(frames, T, D);
  double t0;
  outprob_set_batch_computation(&a->wrk, TRUE);
  t0 = now_sec();
  outprob_prepare(&a->wrk, T);
  for (t = 0; t < T; t++) {
    /* one call triggers the all-state batch loop (outprob.c:230-242) */
    (void)outprob_state(&a->wrk, t, a->by_id[0], p);
    if (out) memcpy(out + (size_t)t * S, a->wrk.outprob_cache[t], sizeof(float) * S);
  }
  t0 = now_sec() - t0;
  (void)s;
  free_param(p);
  return t0;
}

/* Lazy scoring of an explicit (t, state) list, the way the beam touches it. */
double jref_am_outprob_list(void *h, const float *frames, int T, const int *tt, const int *ss,
                            int n, float *out)
{
  jref_am *a = (jref_am *)h;
  int D = a->hmminfo->opt.vec_size, i;
  HTK_Param *p = make_param(frames, T, D);
  double t0;
  outprob_set_batch_computation(&a->wrk, FALSE);
  t0 = now_sec();
  outprob_prepare(&a->wrk, T);
  for (i = 0; i < n; i++) out[i] = outprob_state(&a->wrk, tt[i], a->by_id[ss[i]], p);
  t0 = now_sec() - t0;
  free_param(p);
  return t0;
}

/* Tied-mixture codebook cache after scoring frame t with state `sid`'s
 * codebook: copies mixture_cache[t][book] (calc_tied_mix.c:189-227). */
int jref_am_tmix_cache(void *h, const float *frames, int T, int book, float *score, int *id, int *num)
{
  jref_am *a = (jref_am *)h;
  int S = a->hmminfo->totalstatenum, D = a->hmminfo->opt.vec_size, t, s, i;
  int cap = a->wrk.OP_gprune_num;
  HTK_Param *p = make_param(frames, T, D);
  outprob_set_batch_computation(&a->wrk, TRUE);
  outprob_prepare(&a->wrk, T);
  for (t = 0; t < T; t++) {
    (void)outprob_state(&a->wrk, t, a->by_id[0], p);
    num[t] = a->wrk.mixture_cache_num[t][book];
    for (i = 0; i < num[t]; i++) {
      score[(size_t)t * cap + i] = a->wrk.mixture_cache[t][book][i].score;
      id[(size_t)t * cap + i] = a->wrk.mixture_cache[t][book][i].id;
    }
  }
  (void)S; (void)s;
  free_param(p);
  return cap;
}

/* outprob_cd() (outprob.c:383) for explicit state sets on every frame:
 * sets given as CSR (set_off, states); out is [T][nset]. */
int jref_am_outprob_cd(void *h, const float *frames, int T, int nset, const int *set_off,
                       const int *states, float *out)
{
  jref_am *a = (jref_am *)h;
  int D = a->hmminfo->opt.vec_size, t, i, k;
  HTK_Param *p = make_param(frames, T, D);
  outprob_set_batch_computation(&a->wrk, FALSE);
  outprob_prepare(&a->wrk, T);
  for (t = 0; t < T; t++) {
    for (i = 0; i < nset; i++) {
      CD_State_Set set;
      int n = set_off[i + 1] - set_off[i];
      set.s = (HTK_HMM_State **)malloc(sizeof(HTK_HMM_State *) * (n ? n : 1));
      set.num = n; set.maxnum = n;
      for (k = 0; k < n; k++) set.s[k] = a->by_id[states[set_off[i] + k]];
      out[(size_t)t * nset + i] = outprob_cd(&a->wrk, t, &set, p);
      free(set.s);
    }
  }
  free_param(p);
  return 0;
}

/* Build the reference's own in-memory model structures from flat arrays
 * (plain states only) and run outprob_init() on them -- used to time and check
 * the reference's scoring code at sizes where writing/parsing a 50 MB ascii
 * hmmdefs would dominate.  Only struct fields the scoring path reads are set
 * (hmm_calc.h / htk_hmm.h); the model cannot be used for anything else. */
void *jref_am_from_flat(int S, int D, int G, const float *mean, const float *ivar,
                        const float *gconst, const int *st_off, const int *ent_dens,
                        const float *ent_logw, int gprune, int gprune_num)
{
  jref_am *a = (jref_am *)calloc(1, sizeof(jref_am));
  HTK_HMM_INFO *h = hmminfo_new();
  HTK_HMM_Dens *dens = (HTK_HMM_Dens *)calloc(G, sizeof(HTK_HMM_Dens));
  HTK_HMM_Var *var = (HTK_HMM_Var *)calloc(G, sizeof(HTK_HMM_Var));
  HTK_HMM_State *st = (HTK_HMM_State *)calloc(S, sizeof(HTK_HMM_State));
  HTK_HMM_PDF *pdf = (HTK_HMM_PDF *)calloc(S, sizeof(HTK_HMM_PDF));
  int s, g, i, maxmix = 0;
  for (g = 0; g < G; g++) {
    var[g].vec = (VECT *)(ivar + (size_t)g * D); var[g].len = D;
    dens[g].mean = (VECT *)(mean + (size_t)g * D); dens[g].meanlen = D;
    dens[g].var = &var[g]; dens[g].gconst = gconst[g];
  }
  for (s = 0; s < S; s++) {
    int n = st_off[s + 1] - st_off[s];
    if (n > maxmix) maxmix = n;
    pdf[s].tmix = FALSE; pdf[s].stream_id = 0; pdf[s].mix_num = n;
    pdf[s].b = (HTK_HMM_Dens **)calloc(n ? n : 1, sizeof(HTK_HMM_Dens *));
    pdf[s].bweight = (PROB *)(ent_logw + st_off[s]);
    for (i = 0; i < n; i++) pdf[s].b[i] = ent_dens[st_off[s] + i] >= 0 ? &dens[ent_dens[st_off[s] + i]] : NULL;
    st[s].nstream = 1; st[s].w = NULL; st[s].id = s;
    st[s].pdf = (HTK_HMM_PDF **)calloc(1, sizeof(HTK_HMM_PDF *));
    st[s].pdf[0] = &pdf[s];
    st[s].next = (s + 1 < S) ? &st[s + 1] : NULL;
    pdf[s].next = (s + 1 < S) ? &pdf[s + 1] : NULL;
  }
  h->ststart = st; h->pdfstart = pdf;
  h->opt.stream_info.num = 1; h->opt.stream_info.vsize[0] = D; h->opt.vec_size = D;
  h->totalstatenum = S; h->maxmixturenum = maxmix; h->totalmixnum = G;
  h->is_tied_mixture = FALSE; h->variance_inversed = TRUE; h->cdset_method = IWCD_MAX;
  a->hmminfo = h;
  memset(&a->wrk, 0, sizeof(a->wrk));
  if (!outprob_init(&a->wrk, h, NULL, 0, gprune, gprune_num, NULL)) { free(a); return NULL; }
  a->by_id = (HTK_HMM_State **)calloc(S, sizeof(HTK_HMM_State *));
  for (s = 0; s < S; s++) a->by_id[s] = &st[s];
  return a;   /* intentionally never freed piecewise: test/bench lifetime */
}

/* ------------------------------------------------------------------ DNN tap */
#include <sent/dnn.h>

typedef struct { DNNData *dnn; HMMWork wrk; } jref_dnn;

/* dnn_new() + dnn_setup() (calc_dnn.c:457/528) from .npy / prior files. */
void *jref_dnn_load(int veclen, int contextlen, int in, int out, int hid, int nhid,
                    const char **wfiles, const char **bfiles, const char *ow, const char *ob,
                    const char *prior, float prior_factor, int log10nize, int num_threads)
{
  jref_dnn *d = (jref_dnn *)calloc(1, sizeof(jref_dnn));
  d->dnn = dnn_new();
  if (!dnn_setup(d->dnn, veclen, contextlen, in, out, hid, nhid, (char **)wfiles, (char **)bfiles,
                 (char *)ow, (char *)ob, (char *)prior, prior_factor, log10nize ? TRUE : FALSE, 1,
                 num_threads, "disable")) { free(d); return NULL; }
  make_log_tbl();
  memset(&d->wrk, 0, sizeof(d->wrk));
  d->wrk.OP_dnn = d->dnn;
  d->wrk.statenum = out;
  return d;
}

/* dnn_calc_outprob() (calc_dnn.c:774) frame by frame; returns seconds. */
double jref_dnn_outprob(void *h, const float *frames, int T, float *out)
{
  jref_dnn *d = (jref_dnn *)h;
  int D = d->dnn->inputnodenum, S = d->dnn->outputnodenum, t;
  HTK_Param *p = make_param(frames, T, D);
  float *row = (float *)malloc(sizeof(float) * S);
  double t0 = now_sec();
  d->wrk.OP_param = p;
  for (t = 0; t < T; t++) {
    d->wrk.OP_time = t;
    d->wrk.last_cache = out ? out + (size_t)t * S : row;
    dnn_calc_outprob(&d->wrk);
  }
  t0 = now_sec() - t0;
  free(row);
  free_param(p);
  return t0;
}

const char *jref_simd_string(void) { static char buf[256]; get_builtin_simd_string(buf); return buf; }
int jref_simd_avail(void) { return check_avail_simd(); }

/* ------------------------------------------------- full-engine (pass 1) taps */
#include <julius/juliuslib.h>

typedef struct {
  Jconf *jconf; Recog *recog;
  int res_status, res_wnum; float res_score; int res_wseq[MAXSEQNUM];   /* copied in CALLBACK_RESULT */
} jref_eng;

/* The result sentences are released right after CALLBACK_RESULT
 * (libjulius/src/recogmain.c:1360-1367), so sentence 1 is copied here. */
static void on_result(Recog *recog, void *data)
{
  jref_eng *e = (jref_eng *)data;
  RecogProcess *r = recog->process_list;
  int i;
  e->res_status = r->result.status; e->res_wnum = 0; e->res_score = 0.0f;
  if (r->result.status < 0 || r->result.sentnum <= 0 || r->result.sent == NULL) return;
  e->res_wnum = r->result.sent[0].word_num;
  for (i = 0; i < e->res_wnum; i++) e->res_wseq[i] = r->result.sent[0].word[i];
  e->res_score = r->result.sent[0].score;
}

/* j_config_load_args_new() + j_create_instance_from_jconf() (the julius-simple
 * start-up sequence, julius-simple/julius-simple.c:250-270) */
void *jref_engine_create(int argc, char **argv)
{
  jref_eng *e = (jref_eng *)calloc(1, sizeof(jref_eng));
  e->jconf = j_config_load_args_new(argc, argv);
  if (e->jconf == NULL) { free(e); return NULL; }
  e->recog = j_create_instance_from_jconf(e->jconf);
  if (e->recog == NULL) { free(e); return NULL; }
  if (j_adin_init(e->recog) == FALSE) { free(e); return NULL; }
  callback_add(e->recog, CALLBACK_RESULT, on_result, e);
  return e;
}

/* Eager scoring for the engine's acoustic models (outprob_set_batch_computation(), outprob_init.c:196: every state of
 * a frame is scored at the first request, outprob.c:230-242).  Values are those of the default lazy mode except
 * where the scoring carries history from frame to frame: -gprune heu / beam over tied-mixture codebooks. */
void jref_engine_set_eager(void *h, int on)
{
  jref_eng *e = (jref_eng *)h;
  PROCESS_AM *am;
  for (am = e->recog->amlist; am; am = am->next) outprob_set_batch_computation(&am->hmmwrk, on ? TRUE : FALSE);
}

/* Recognise one HTK parameter file (-input htkparam).  Returns the number of
 * trellis atoms, or -1.  With -1pass the word trellis is left exactly as
 * finalize_1st_pass() built it. */
int jref_engine_recognize(void *h, const char *mfcfile)
{
  jref_eng *e = (jref_eng *)h;
  RecogProcess *r = e->recog->process_list;
  int t, n = 0;
  r->pass1_wnum = 0;              /* find_1pass_result() leaves the previous input's sequence behind when it fails (beam.c:427-431) */
  if (j_open_stream(e->recog, (char *)mfcfile) != 0) return -1;
  if (j_recognize_stream(e->recog) == -1) return -1;
  if (r->backtrellis->num == NULL) return 0;
  for (t = 0; t < r->backtrellis->framelen; t++) n += r->backtrellis->num[t];
  return n;
}

/* Batch driver (only in the build that links the first-pass shim, libjref_amd.so): queue the
 * inputs, decode them all in one device launch; the following jref_engine_recognize() calls find
 * their first pass done.  The three entry points live in julius_amd/shim/jamd_pass1_shim.c. */
extern int jamd_pass1_prefetch_add(RecogProcess *r, HTK_Param *param) __attribute__((weak));
extern int jamd_pass1_prefetch_run(RecogProcess *r) __attribute__((weak));
extern void jamd_pass1_prefetch_clear(RecogProcess *r) __attribute__((weak));

extern int jamd_pass1_prefetch_served(RecogProcess *r) __attribute__((weak));
int jref_engine_prefetch_served(void *h)
{
  return jamd_pass1_prefetch_served ? jamd_pass1_prefetch_served(((jref_eng *)h)->recog->process_list) : -2;
}

int jref_engine_prefetch(void *h, const char **mfcfiles, int n)
{
  jref_eng *e = (jref_eng *)h;
  RecogProcess *r = e->recog->process_list;
  int i;
  if (!jamd_pass1_prefetch_add || !jamd_pass1_prefetch_run) return -2;
  jamd_pass1_prefetch_clear(r);
  for (i = 0; i < n; i++) {
    if (j_open_stream(e->recog, (char *)mfcfiles[i]) != 0) return -1;
    if (jamd_pass1_prefetch_add(r, e->recog->mfcclist->param) != 0) return -1;
  }
  return jamd_pass1_prefetch_run(r) == 0 ? n : -1;
}

/* Copy the sorted word trellis (bt->rw[t][i]) out: per atom wid, begintime,
 * endtime, backscore, lscore and the (wid, endtime) of its predecessor
 * (-1,-1 for the sentence start). */
int jref_engine_trellis(void *h, int *wid, int *bt, int *et, float *backscore, float *lscore,
                        int *pwid, int *pet)
{
  jref_eng *e = (jref_eng *)h;
  BACKTRELLIS *b = e->recog->process_list->backtrellis;
  int t, i, n = 0;
  if (b->num == NULL) return 0;
  for (t = 0; t < b->framelen; t++) {
    for (i = 0; i < b->num[t]; i++) {
      TRELLIS_ATOM *a = b->rw[t][i];
      wid[n] = a->wid; bt[n] = a->begintime; et[n] = a->endtime;
      backscore[n] = a->backscore; lscore[n] = a->lscore;
      if (a->last_tre == NULL || a->last_tre->wid == WORD_INVALID) { pwid[n] = -1; pet[n] = -1; }
      else { pwid[n] = a->last_tre->wid; pet[n] = a->last_tre->endtime; }
      n++;
    }
  }
  return n;
}

/* pass1 best word sequence and score (r->pass1_wseq / pass1_score) */
int jref_engine_pass1(void *h, int *wseq, float *score)
{
  jref_eng *e = (jref_eng *)h;
  RecogProcess *r = e->recog->process_list;
  int i;
  for (i = 0; i < r->pass1_wnum; i++) wseq[i] = r->pass1_wseq[i];
  *score = r->pass1_score;
  return r->pass1_wnum;
}

int jref_engine_info(void *h, int *info)
{
  jref_eng *e = (jref_eng *)h;
  RecogProcess *r = e->recog->process_list;
  if (r == NULL || r->wchmm == NULL || r->wchmm->winfo == NULL) return -1;   /* e.g. a grammar the reference purged */
  info[0] = r->wchmm->n; info[1] = r->wchmm->winfo->num; info[2] = r->wchmm->startnum;
  info[3] = r->wchmm->isolatenum; info[4] = r->trellis_beam_width; info[5] = r->wchmm->hmminfo->totalstatenum;
  info[6] = r->lmtype; info[7] = r->wchmm->hmminfo->multipath; info[8] = r->ccd_flag;
  info[9] = r->backtrellis->framelen;
  return 0;
}

/* Flatten the loaded recogniser's first-pass tables with the product's own
 * reference-side shim (julius_amd/shim/jamd_flatten_lex.c) and write them as a
 * lexicon blob.  Returns 0 or a JAMD_E* code. */
int jref_engine_save_lexicon(void *h, const char *path)
{
  jref_eng *e = (jref_eng *)h;
  jamd_flat_lexicon fl;
  int rc = e->recog->process_list->am->hmminfo->multipath ? jamd_flatten_lexicon_multipath(e->recog->process_list, &fl)
                                                          : jamd_flatten_lexicon(e->recog->process_list, &fl);
  if (rc != 0) return rc;
  rc = jamd_lexicon_save(&fl.desc, path);
  if (rc == 0 && e->recog->process_list->lmtype == LM_PROB && e->recog->process_list->lm != NULL)
    rc = jamd_lexicon_append_ngram_names(path, e->recog->process_list->lm->ngram);     /* as jamd_export does */
  if (rc == 0 && e->recog->process_list->lmtype == LM_PROB && e->recog->process_list->lm != NULL)
    rc = jamd_lexicon_append_separation(path, e->recog->process_list->lm->config->separate_wnum);
  jamd_flat_lexicon_free(&fl);
  return rc;
}


/* Final result after the 2nd pass as captured by on_result(): sentence 1 and its
 * score; *status gets r->result.status.  Returns the word count. */
int jref_engine_result(void *h, int *wseq, float *score, int *status)
{
  jref_eng *e = (jref_eng *)h;
  int i;
  *status = e->res_status; *score = e->res_score;
  for (i = 0; i < e->res_wnum; i++) wseq[i] = e->res_wseq[i];
  return e->res_wnum;
}

/* How much of the [T][S] outprob cache holds a computed score after the last
 * recognition (entries != LOG_UNDEF, libsent/src/phmm/outprob.c:68): the plain
 * reference fills only what the search touched, the device shim fills all of it. */
int jref_engine_cache_fill(void *h, int *defined, int *total)
{
  jref_eng *e = (jref_eng *)h;
  HMMWork *wrk = &(e->recog->amlist->hmmwrk);
  int T = e->recog->process_list->backtrellis->framelen, t, s, n = 0;
  if (T > wrk->outprob_allocframenum) T = wrk->outprob_allocframenum;
  for (t = 0; t < T; t++)
    for (s = 0; s < wrk->statenum; s++)
      if (wrk->outprob_cache[t][s] != (LOG_ZERO - 1)) n++;
  *defined = n; *total = T * wrk->statenum;
  return 0;
}

/* ---- GMM-based input verification (-gmm / -gmmnum / -gmmreject, libjulius/src/gmm.c) ----------
 * info: {number of GMMs, -gmmnum, veclen}; -1 without -gmm. */
int jref_engine_gmm_info(void *h, int *info)
{
  jref_eng *e = (jref_eng *)h;
  if (e->recog->gmm == NULL) return -1;
  info[0] = e->recog->gmm->totalhmmnum;
  info[1] = e->recog->jconf->reject.gmm_gprune_num;
  info[2] = e->recog->gmm->opt.vec_size;
  return 0;
}

/* The GMM definitions as a flat state pool (view for jref_am_dims/_export), and the state id of each
 * model's output state in gmm->start order (the order of gc->gmm_score[]). */
void *jref_engine_gmm_model(void *h)
{
  jref_eng *e = (jref_eng *)h;
  jref_am *v;
  if (e->recog->gmm == NULL) return NULL;
  v = (jref_am *)calloc(1, sizeof(jref_am));
  v->hmminfo = e->recog->gmm; v->view_only = 1;
  return v;
}

int jref_engine_gmm_states(void *h, int *state_id)
{
  jref_eng *e = (jref_eng *)h;
  HTK_HMM_Data *d;
  int i = 0;
  if (e->recog->gmm == NULL) return -1;
  for (d = e->recog->gmm->start; d; d = d->next) state_id[i++] = d->s[1]->id;
  return i;
}

/* Per-frame model scores: gmm_prepare() zeroes gc->gmm_score[], one gmm_proceed() then leaves
 * 0.0 + score = score in it (gmm.c:520-543, 574-600) -- the reference's own entry points, frame by
 * frame, on a parameter block of ours. */
int jref_engine_gmm_frame_scores(void *h, const float *frames, int T, int D, float *out)
{
  jref_eng *e = (jref_eng *)h;
  Recog *recog = e->recog;
  MFCCCalc *mfcc = recog->gmmmfcc;
  HTK_Param *keep, *p;
  boolean keep_valid; int keep_f, t, n;
  if (recog->gmm == NULL || mfcc == NULL || D != recog->gmm->opt.vec_size) return -1;
  n = recog->gmm->totalhmmnum;
  p = make_param(frames, T, D);
  keep = mfcc->param; keep_valid = mfcc->valid; keep_f = mfcc->f;
  mfcc->param = p; mfcc->valid = TRUE;
  for (t = 0; t < T; t++) {
    mfcc->f = t;
    gmm_prepare(recog);
    gmm_proceed(recog);
    memcpy(out + (size_t)t * n, recog->gc->gmm_score, sizeof(float) * (size_t)n);
  }
  mfcc->param = keep; mfcc->valid = keep_valid; mfcc->f = keep_f;
  free_param(p);
  return 0;
}

/* After jref_engine_recognize(): accumulated scores, the winner, its confidence (gmm_end(),
 * gmm.c:614-660), whether gmm_valid_input() accepts, and the frame count. */
int jref_engine_gmm_result(void *h, float *scores, int *max_i, float *cm, int *valid, int *framecount)
{
  jref_eng *e = (jref_eng *)h;
  if (e->recog->gmm == NULL || e->recog->gc == NULL) return -1;
  memcpy(scores, e->recog->gc->gmm_score, sizeof(float) * (size_t)e->recog->gmm->totalhmmnum);
  *max_i = e->recog->gc->max_i;
  *cm = e->recog->gc->gmm_max_cm;
  *valid = gmm_valid_input(e->recog) ? 1 : 0;
  *framecount = e->recog->gc->framecount;
  return 0;
}

/* Only in the shimmed builds (julius_amd/shim/jamd_gmm_wrap.c): frames of this recogniser's
 * verification GMMs that were scored on the device; -1 in the plain reference. */
extern long jamd_gmm_wrap_frames(Recog *recog) __attribute__((weak));
long jref_engine_gmm_device_frames(void *h)
{
  jref_eng *e = (jref_eng *)h;
  return jamd_gmm_wrap_frames ? jamd_gmm_wrap_frames(e->recog) : -1;
}

/* The loaded acoustic model through the product shim's blob writer (jamd_gmm_save). */
int jref_am_save(void *h, const char *path)
{
  jref_am *a = (jref_am *)h;
  if (ensure_flat(a) != 0) return -1;
  return jamd_gmm_save(&a->flat.desc, path);
}

/* ---- binary HMM file, written by the reference's own writer (what mkbinhmm does, mkbinhmm/mkbinhmm.c:65-119), so
 * that the product's direct reader (julius_amd/csrc/readers.hip) can be checked against Julius' own on the same file. */
int jref_write_binhmm(const char *hmmdefs, const char *outfile)
{
  HTK_HMM_INFO *h = hmminfo_new();
  FILE *fp;
  int ok;
  if (!init_hmminfo(h, (char *)hmmdefs, NULL, NULL)) return -1;
  if ((fp = fopen_writefile((char *)outfile)) == NULL) return -2;
  ok = write_binhmm(fp, h, NULL);
  fclose_writefile(fp);
  return ok ? 0 : -3;
}

/* mkbingram's work through the reference's own writer (libsent/src/ngram/ngram_write_bin.c): the input of the direct
 * binary N-gram reader's tests.  arpa_rl != NULL: backward N-gram + additional forward 2-gram (the -nlr / -nrl pair). */
int jref_write_bingram(const char *arpa_lr, const char *arpa_rl, const char *outfile)
{
  NGRAM_INFO *ng = ngram_info_new();
  FILE *fp;
  int ok;
  char header[64] = "written by oracle/ref_driver.c\n";
  if (arpa_rl != NULL) {
    if (!init_ngram_arpa(ng, (char *)arpa_rl, DIR_RL)) return -1;
    if (arpa_lr != NULL && !init_ngram_arpa_additional(ng, (char *)arpa_lr)) return -1;
  } else if (!init_ngram_arpa(ng, (char *)arpa_lr, DIR_LR)) return -1;
  if ((fp = fopen_writefile((char *)outfile)) == NULL) return -2;
  ok = ngram_write_bin(fp, ng, header);
  fclose_writefile(fp);
  return ok ? 0 : -3;
}
