/*
 * jamd_outprob_wrap.c -- reference-side binding of the SCORING boundary (O).
 *
 * Linked into Julius with GNU ld's symbol wrapping
 *     -Wl,--wrap=outprob_state,--wrap=outprob_cd,--wrap=outprob,
 *         --wrap=outprob_prepare,--wrap=outprob_free
 * so that every call libjulius makes into libsent's scoring entry points
 *     outprob_state()   libsent/src/phmm/outprob.c:184   (from outprob_style.c:380-485)
 *     outprob_cd()      outprob.c:383
 *     outprob()         outprob.c:414                    (2nd pass, search_bestfirst_v1.c:939-1212)
 *     outprob_prepare() outprob_init.c:210               (recogmain.c:1156, once per utterance)
 *     outprob_free()    outprob_init.c:231
 * passes through here first.  Nothing of Julius is edited or removed: libsent's own objects
 * stay linked and are what finally answers each call -- but for a supported configuration
 * the whole [T][S] score matrix of the utterance has by then been computed on the device
 * (jamd_gmm_outprob_host) and copied into wrk->outprob_cache, so the answer is a cache
 * hit.  Julius' CPU beam, its grammar / N-gram handling, multipath models, the 2nd pass:
 * all unchanged, all consuming device scores that are bit-identical to calc_mix() /
 * calc_tied_mix().
 *
 * Supported: GMM acoustic models (plain or tied-mixture), single stream, -gprune none,
 * safe, heu or beam (the last two over tied-mixture codebooks: eager-scoring values), with or without Gaussian mixture selection (-gshmm: the selection stage of
 * gms.c runs on the device after the scoring, jamd_gms_apply_host); DNN acoustic models
 * (-dnnconf).  Anything else (heu/beam pruning,
 * multi-stream, -input outprob) is left to libsent's CPU code with one log line -- those paths are outside
 * the engine's scope.  A supported configuration without a usable
 * gfx950 device is a hard error (exit), as in the reference's own CUDA path
 * (libsent/src/phmm/calc_dnn_cuda.cu:22-32).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jamd_flatten.h"
#include <sent/util.h>

LOGPROB __real_outprob_state(HMMWork *wrk, int t, HTK_HMM_State *stateinfo, HTK_Param *param);
LOGPROB __real_outprob_cd(HMMWork *wrk, int t, CD_State_Set *lset, HTK_Param *param);
LOGPROB __real_outprob(HMMWork *wrk, int t, HMM_STATE *hmmstate, HTK_Param *param);
boolean __real_outprob_prepare(HMMWork *wrk, int framenum);
void    __real_outprob_free(HMMWork *wrk);

typedef struct {
  HMMWork *wrk;
  int state;                 /* 0 = not examined, 1 = device, 2 = left to libsent */
  jamd_gmm *gmm;
  jamd_dnn *dnn;
  jamd_gms *gms;             /* -gshmm: selection stage applied to the device scores */
  int nstate;
  const HTK_Param *param;    /* utterance the cache was filled for */
  int filled;                /* frames [0, filled) are in the cache */
  int from_zero;             /* the scoring carries history from frame to frame: a grown input is scored from frame 0 again */
} wrap_ctx;

static jamd_engine *g_eng = NULL;
static wrap_ctx *g_ctx = NULL;       /* one per acoustic model work area, grown on demand */
static int g_nctx = 0, g_capctx = 0;

static wrap_ctx *ctx_get(HMMWork *wrk)
{
  int i;
  for (i = 0; i < g_nctx; i++) if (g_ctx[i].wrk == wrk) return &g_ctx[i];
  if (g_nctx == g_capctx) {
    wrap_ctx *n = (wrap_ctx *)realloc(g_ctx, sizeof(wrap_ctx) * (g_capctx ? 2 * g_capctx : 8));
    if (n == NULL) return NULL;
    g_ctx = n; g_capctx = g_capctx ? 2 * g_capctx : 8;
  }
  memset(&g_ctx[g_nctx], 0, sizeof(wrap_ctx));
  g_ctx[g_nctx].wrk = wrk;
  return &g_ctx[g_nctx++];
}

static void die(const char *what)
{
  jlog("Error: jamd: %s: %s\n", what, jamd_last_error());
  fprintf(stderr, "jamd: %s: %s\n", what, jamd_last_error());
  exit(1);
}

static void examine(wrap_ctx *c)
{
  HMMWork *wrk = c->wrk;
  int gprune;
  jamd_flat_gmm fg;
  c->state = 2;
  if ((wrk->OP_gshmm != NULL && wrk->OP_dnn != NULL) || (wrk->OP_dnn == NULL && wrk->OP_nstream != 1)) {
    jlog("Stat: jamd: multi-stream scoring stays on libsent's CPU code\n");
    return;
  }
  if (jamd_abi_version() != JAMD_ABI_VERSION) die("ABI mismatch between shim and libjulius_amd.so");
  if (g_eng == NULL) {
    const char *dev = getenv("JAMD_DEVICE");
    if (jamd_engine_create(dev ? atoi(dev) : 0, &g_eng) != JAMD_OK) die("no usable gfx950 device");
  }
  if (wrk->OP_dnn != NULL) {                /* dnn_calc_outprob(), calc_dnn.c:774, for whole utterances */
    jamd_flat_dnn fd;
    if (jamd_flatten_dnn(wrk->OP_dnn, &fd) != JAMD_OK) die("cannot flatten the DNN");
    if (jamd_dnn_create(g_eng, &fd.desc, &c->dnn) != JAMD_OK) die("jamd_dnn_create");
    c->nstate = fd.desc.dims[fd.desc.nlayer];
    jamd_flat_dnn_free(&fd);
    c->state = 1;
    jlog("Stat: jamd: DNN scoring on HIP device %d (%d outputs)\n", jamd_engine_device(g_eng), c->nstate);
    return;
  }
  if (wrk->compute_gaussset == gprune_none) gprune = JAMD_GPRUNE_NONE;
  else if (wrk->compute_gaussset == gprune_safe) gprune = JAMD_GPRUNE_SAFE;
  else if (wrk->compute_gaussset == gprune_heu) gprune = JAMD_GPRUNE_HEU;
  else if (wrk->compute_gaussset == gprune_beam) gprune = JAMD_GPRUNE_BEAM;
  else {
    jlog("Stat: jamd: unknown Gaussian pruning function (a plugin?); scoring stays on libsent's CPU code\n");
    return;
  }
  if (gprune >= JAMD_GPRUNE_HEU && wrk->OP_hmminfo->is_tied_mixture) {
    c->from_zero = 1;
    jlog("Stat: jamd: -gprune heu/beam over tied-mixture codebooks: the thresholds of frame t come from frame t-1 of the same input "
         "(every frame is scored on the device), i.e. the reference's values under eager scoring\n");
  }
  if (jamd_flatten_hmminfo(wrk->OP_hmminfo, &fg) != 0) die("cannot flatten the acoustic model");
  if (jamd_gmm_create(g_eng, &fg.desc, gprune, wrk->OP_gprune_num, &c->gmm) != JAMD_OK) die("jamd_gmm_create");
  c->nstate = fg.desc.nstate;
  jamd_flat_gmm_free(&fg);
  if (wrk->OP_gshmm != NULL) {              /* gms_init(), gms.c:275-317: gsset[] is indexed by selection state id */
    jamd_flat_gmm fs;
    int *map = (int *)malloc(sizeof(int) * (size_t)c->nstate), s;
    if (map == NULL || jamd_flatten_hmminfo(wrk->OP_gshmm, &fs) != 0) die("cannot flatten the selection model");
    for (s = 0; s < c->nstate; s++)
      map[s] = (wrk->state2gs[s] >= 0 && wrk->state2gs[s] < wrk->gsset_num) ? wrk->gsset[wrk->state2gs[s]].state->id : -1;
    if (jamd_gms_create(g_eng, &fs.desc, map, c->nstate, wrk->my_nbest, &c->gms) != JAMD_OK) die("jamd_gms_create");
    jamd_gms_set_strict_order(c->gms, 1);   /* this boundary promises the reference's numbers, ties included */
    jamd_flat_gmm_free(&fs); free(map);
    jlog("Stat: jamd: Gaussian mixture selection on the device (%d selection states, %d selected per frame)\n",
         wrk->gsset_num, wrk->my_nbest);
  }
  c->state = 1;
  jlog("Stat: jamd: acoustic scoring on HIP device %d (%d states)\n", jamd_engine_device(g_eng), c->nstate);
}

/* Score every frame of `param` that is not in the cache yet. */
static void ensure(HMMWork *wrk, HTK_Param *param)
{
  wrap_ctx *c = ctx_get(wrk);
  int T, n;
  float *fr, *sc;
  if (c == NULL || param == NULL || param->is_outprob) return;
  if (c->state == 0) examine(c);
  if (c->state != 1) return;
  T = param->samplenum;
  if (c->param != param) { c->param = param; c->filled = 0; }
  if (c->filled >= T) return;
  /* State carried from frame to frame -- the Gaussian selection, and gprune heu/beam over tied-mixture codebooks (frame
   * t's thresholds are the codebook's winners of frame t-1, calc_tied_mix.c:203-215) -- is the state of ONE device
   * call: when the input grows (a second call on the same param), score from frame 0 again, else the first new frame
   * would take the no-history branch and differ from the reference's eager-scoring values. */
  if (c->gms != NULL || c->from_zero) c->filled = 0;
  n = T - c->filled;
  fr = jamd_pack_param(param, c->filled, T);
  sc = (float *)malloc(sizeof(float) * (size_t)n * c->nstate);
  if (fr == NULL || sc == NULL) die("out of memory");
  if (c->dnn != NULL) {
    if (param->veclen != jamd_dnn_veclen(c->dnn)) die("feature vectors are not spliced to the DNN input length");
    if (jamd_dnn_outprob_host(c->dnn, fr, n, sc) != JAMD_OK) die("jamd_dnn_outprob_host");
  } else if (jamd_gmm_outprob_host(c->gmm, fr, n, sc) != JAMD_OK) die("jamd_gmm_outprob_host");
  if (c->gms != NULL && jamd_gms_apply_host(c->gms, fr, n, NULL, 0, sc) != JAMD_OK) die("jamd_gms_apply_host");
  if (jamd_fill_outprob_cache(wrk, sc, c->filled, n, c->nstate) != JAMD_OK) die("cache layout mismatch");
  c->filled = T;
  free(fr); free(sc);
}

LOGPROB __wrap_outprob_state(HMMWork *wrk, int t, HTK_HMM_State *stateinfo, HTK_Param *param)
{
  ensure(wrk, param);
  return __real_outprob_state(wrk, t, stateinfo, param);
}

LOGPROB __wrap_outprob_cd(HMMWork *wrk, int t, CD_State_Set *lset, HTK_Param *param)
{
  ensure(wrk, param);
  return __real_outprob_cd(wrk, t, lset, param);
}

LOGPROB __wrap_outprob(HMMWork *wrk, int t, HMM_STATE *hmmstate, HTK_Param *param)
{
  ensure(wrk, param);
  return __real_outprob(wrk, t, hmmstate, param);
}

boolean __wrap_outprob_prepare(HMMWork *wrk, int framenum)
{
  wrap_ctx *c = ctx_get(wrk);
  if (c != NULL) { c->param = NULL; c->filled = 0; }      /* outprob_cache_prepare() wipes the cache */
  return __real_outprob_prepare(wrk, framenum);
}

void __wrap_outprob_free(HMMWork *wrk)
{
  int i;
  for (i = 0; i < g_nctx; i++)
    if (g_ctx[i].wrk == wrk) {
      if (g_ctx[i].gms) jamd_gms_destroy(g_ctx[i].gms);
      if (g_ctx[i].gmm) jamd_gmm_destroy(g_ctx[i].gmm);
      if (g_ctx[i].dnn) jamd_dnn_destroy(g_ctx[i].dnn);
      g_ctx[i] = g_ctx[--g_nctx];
      break;
    }
  __real_outprob_free(wrk);
}
