/*
 * jamd_gmm_wrap.c -- reference-side binding for GMM-based input verification / rejection
 * (-gmm FILE -gmmnum N -gmmreject NAMES), SURVEY 8f N4.
 *
 * Linked into Julius with GNU ld's symbol wrapping
 *     -Wl,--wrap=gmm_prepare,--wrap=gmm_proceed,--wrap=gmm_end,--wrap=gmm_free
 * so that the calls pass1.c makes into libjulius/src/gmm.c
 *     gmm_prepare()  gmm.c:520   (pass1.c:158, first frame of an input)
 *     gmm_proceed()  gmm.c:574   (pass1.c:161, every frame)
 *     gmm_end()      gmm.c:614   (pass1.c:423/471, end of the first pass)
 *     gmm_free()     gmm.c:700
 * pass through here.  gmm.c itself stays linked and unchanged.  gmm_proceed() only does its
 * bookkeeping here (frame count); the frames it would have scored one by one are scored together on
 * the device (jamd_rejgmm_scores_host) when gmm_end() arrives -- or earlier, if the frame index ever
 * jumps -- and added to gc->gmm_score[] in frame order, exactly the float sums gmm_proceed() builds
 * (gmm.c:599).  Then the real gmm_end() picks the winner, computes the confidence and fires
 * CALLBACK_RESULT_GMM, and recogmain.c:1252 rejects through the real gmm_valid_input().
 *
 * Left to gmm.c (with one log line): multi-stream GMM definitions, short-pause segmentation
 * (-spsegment rewinds the frame index), and a build with GMM_VAD (the VAD needs each frame's score
 * at that frame).  A supported configuration without a usable gfx950 device is a hard error.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define JAMD_WITH_LIBJULIUS 1
#include "jamd_flatten.h"

void __real_gmm_prepare(Recog *recog);
void __real_gmm_proceed(Recog *recog);
void __real_gmm_end(Recog *recog);
void __real_gmm_free(Recog *recog);

typedef struct {
  Recog *recog;
  int state;                 /* 0 = not examined, 1 = device, 2 = left to gmm.c */
  jamd_rejgmm *m;
  int nmodel;
  HTK_Param *param;          /* parameter block the pending frames live in */
  int first_f, npend;        /* pending frames [first_f, first_f + npend) */
  long frames_on_device;     /* statistics for the tests */
} gmmw_ctx;

static jamd_engine *g_eng = NULL;
static gmmw_ctx *g_ctx = NULL;
static int g_nctx = 0, g_capctx = 0;

static gmmw_ctx *ctx_get(Recog *recog)
{
  int i;
  for (i = 0; i < g_nctx; i++) if (g_ctx[i].recog == recog) return &g_ctx[i];
  if (g_nctx == g_capctx) {
    gmmw_ctx *n = (gmmw_ctx *)realloc(g_ctx, sizeof(gmmw_ctx) * (g_capctx ? 2 * g_capctx : 4));
    if (n == NULL) return NULL;
    g_ctx = n; g_capctx = g_capctx ? 2 * g_capctx : 4;
  }
  memset(&g_ctx[g_nctx], 0, sizeof(gmmw_ctx));
  g_ctx[g_nctx].recog = recog;
  return &g_ctx[g_nctx++];
}

static void die(const char *what)
{
  jlog("Error: jamd: %s: %s\n", what, jamd_last_error());
  fprintf(stderr, "jamd: %s: %s\n", what, jamd_last_error());
  exit(1);
}

static void examine(gmmw_ctx *c)
{
  Recog *recog = c->recog;
  jamd_flat_gmm fg;
  HTK_HMM_Data *d;
  int *model_state, i = 0;
  c->state = 2;
#ifdef GMM_VAD
  jlog("Stat: jamd: GMM_VAD build: the verification GMMs stay on gmm.c\n");
  return;
#endif
  if (recog->gmm == NULL || recog->gc == NULL) return;
  if (recog->gc->OP_nstream != 1 || recog->jconf->decodeopt.segment) {
    jlog("Stat: jamd: multi-stream verification GMMs / -spsegment stay on gmm.c\n");
    return;
  }
  if (jamd_abi_version() != JAMD_ABI_VERSION) die("ABI mismatch between shim and libjulius_amd.so");
  if (g_eng == NULL) {
    const char *dev = getenv("JAMD_DEVICE");
    if (jamd_engine_create(dev ? atoi(dev) : 0, &g_eng) != JAMD_OK) die("no usable gfx950 device");
  }
  if (jamd_flatten_hmminfo(recog->gmm, &fg) != 0) die("cannot flatten the verification GMMs");
  c->nmodel = recog->gmm->totalhmmnum;
  model_state = (int *)malloc(sizeof(int) * (size_t)c->nmodel);
  if (model_state == NULL) die("out of memory");
  for (d = recog->gmm->start; d && i < c->nmodel; d = d->next) model_state[i++] = d->s[1]->id;   /* gmm.c:593-598 */
  if (jamd_rejgmm_create(g_eng, &fg.desc, model_state, c->nmodel, recog->jconf->reject.gmm_gprune_num, &c->m) != JAMD_OK)
    die("jamd_rejgmm_create");
  jamd_flat_gmm_free(&fg); free(model_state);
  c->state = 1;
  jlog("Stat: jamd: %d verification GMMs scored on HIP device %d (-gmmnum %d)\n", c->nmodel,
       jamd_engine_device(g_eng), recog->jconf->reject.gmm_gprune_num);
}

/* score the pending frames and add them to gc->gmm_score[] in frame order */
static void flush(gmmw_ctx *c)
{
  GMMCalc *gc = c->recog->gc;
  float *fr, *fs;
  int t, k;
  if (c->npend <= 0) return;
  fr = jamd_pack_param(c->param, c->first_f, c->first_f + c->npend);
  fs = (float *)malloc(sizeof(float) * (size_t)c->npend * c->nmodel);
  if (fr == NULL || fs == NULL) die("out of memory");
  if (jamd_rejgmm_scores_host(c->m, fr, c->npend, NULL, 0, fs, NULL) != JAMD_OK) die("jamd_rejgmm_scores_host");
  for (t = 0; t < c->npend; t++)
    for (k = 0; k < c->nmodel; k++) gc->gmm_score[k] += fs[(size_t)t * c->nmodel + k];     /* gmm.c:599 */
  c->frames_on_device += c->npend;
  c->npend = 0;
  free(fr); free(fs);
}

void __wrap_gmm_prepare(Recog *recog)
{
  gmmw_ctx *c = ctx_get(recog);
  if (c != NULL) c->npend = 0;                 /* gmm_prepare() zeroes the sums: nothing is owed */
  __real_gmm_prepare(recog);
}

void __wrap_gmm_proceed(Recog *recog)
{
  gmmw_ctx *c = ctx_get(recog);
  MFCCCalc *mfcc = recog->gmmmfcc;
  if (c != NULL && c->state == 0) examine(c);
  if (c == NULL || c->state != 1) { __real_gmm_proceed(recog); return; }
  if (!mfcc->valid) return;                    /* gmm.c:589 */
  recog->gc->framecount++;                     /* gmm.c:591 */
  if (c->npend > 0 && (mfcc->param != c->param || mfcc->f != c->first_f + c->npend)) flush(c);
  if (c->npend == 0) { c->param = mfcc->param; c->first_f = mfcc->f; }
  c->npend++;
}

void __wrap_gmm_end(Recog *recog)
{
  gmmw_ctx *c = ctx_get(recog);
  if (c != NULL && c->state == 1) flush(c);
  __real_gmm_end(recog);
}

void __wrap_gmm_free(Recog *recog)
{
  int i;
  for (i = 0; i < g_nctx; i++)
    if (g_ctx[i].recog == recog) {
      if (g_ctx[i].m) jamd_rejgmm_destroy(g_ctx[i].m);
      g_ctx[i] = g_ctx[--g_nctx];
      break;
    }
  __real_gmm_free(recog);
}

/* for the tests: frames of this recogniser scored on the device so far (-1: not on the device) */
long jamd_gmm_wrap_frames(Recog *recog)
{
  gmmw_ctx *c = ctx_get(recog);
  return (c != NULL && c->state == 1) ? c->frames_on_device : -1;
}
