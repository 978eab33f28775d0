/*
 * jamd_calcmix_plugin.c -- Julius' OFFICIAL plugin slot for Gaussian computation, served by the
 * gfx950 engine.  Build as a shared object named *.jpi, put it in a directory and run an
 * UNMODIFIED julius with
 *        julius -plugindir <dir> -gprune jamd ...
 * (the plugin directory option must come before -gprune; libjulius/src/m_options.c:1057-1060,
 * :1318-1321).  Nothing is relinked: libjulius dlopen()s the file (plugin.c:164-233) and wires
 *   calcmix_get_optname / calcmix_init / calcmix / calcmix_free
 * into hmmwrk.compute_gaussset{,_init,_free} (m_fusion.c) -- the interface of plugin/calcmix.c.
 *
 * calcmix() is called once per (state, frame) the search touches and must fill
 * OP_calced_score[i] / OP_calced_id[i] for the state's mixture components; calc_mix()
 * (libsent/src/phmm/calc_mix.c:41) then adds the mixture weights and takes the table log-sum.
 * A launch per call would be absurd, so on the first call for an utterance the plugin computes the
 * per-Gaussian scores of ALL mixture components for ALL frames of the input on the device
 * (jamd_gmm_dens_host, the same four fp32 operations per dimension as gprune_none.c:59-82 and the
 * sample plugin) and every later call is a row lookup.  The values are bit-identical to what the
 * sample plugin's loop computes, so Julius' result does not change.
 *
 * Scope: single-stream GMM-HMMs whose states are all plain or all tied-mixture.  For a tied-mixture
 * model calc_tied_mix() (libsent/src/phmm/calc_tied_mix.c:162-227) sends the Gaussians of a
 * CODEBOOK through the slot (book->d, book->num) and caches what comes back per (frame, codebook);
 * the device rows then hold the codebook Gaussians in codebook order.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sent/stddefs.h>
#include <sent/htk_hmm.h>
#include <sent/htk_param.h>
#include <sent/hmm_calc.h>
#include "jamd_flatten.h"

#define PLUGIN_TITLE "Gaussian computation on an AMD MI355X (julius_amd)"
#define GPRUNE_OPT "jamd"

typedef struct {
  HMMWork *wrk;
  jamd_gmm *gmm;
  int nentry, nstate, veclen;
  int *st_off;                 /* [nstate + 1] first mixture entry of each state (plain model) */
  int tied, nbook; int *book_off;   /* tied-mixture model: first column of each codebook */
  float *dens;                 /* [cap][nentry] per-Gaussian scores of the current input */
  int cap, filled;             /* frames allocated / computed */
  unsigned long long key;      /* content hash of the input the rows belong to */
} plug_ctx;

static jamd_engine *g_eng = NULL;
static plug_ctx *g_ctx = NULL;
static int g_nctx = 0;
static long g_calls = 0, g_fills = 0;

/* counters for tests: lookups served / device fills */
long jamd_calcmix_calls(void) { return g_calls; }
long jamd_calcmix_fills(void) { return g_fills; }

static plug_ctx *ctx_of(HMMWork *wrk)
{
  int i;
  for (i = 0; i < g_nctx; i++) if (g_ctx[i].wrk == wrk) return &g_ctx[i];
  return NULL;
}

static void fatal(const char *what)
{
  fprintf(stderr, "jamd calcmix plugin: %s: %s\n", what, jamd_last_error());
  exit(1);                      /* no CPU fallback: the slot was selected explicitly with -gprune jamd */
}

int initialize(void) { return 0; }

int get_plugin_info(int opcode, char *buf, int buflen)
{
  if (opcode == 0) strncpy(buf, PLUGIN_TITLE, buflen);
  return 0;
}

void calcmix_get_optname(char *buf, int buflen) { strncpy(buf, GPRUNE_OPT, buflen); }

boolean calcmix_init(HMMWork *wrk)
{
  HTK_HMM_INFO *hmm = wrk->OP_hmminfo;
  jamd_flat_gmm fg;
  plug_ctx *c;
  const char *dev = getenv("JAMD_DEVICE");
  if (wrk->OP_nstream != 1) {
    jlog("Error: jamd plugin: multi-stream models are not served through the plugin slot\n");
    return FALSE;
  }
  if (jamd_abi_version() != JAMD_ABI_VERSION) { jlog("Error: jamd plugin: ABI mismatch with libjulius_amd.so\n"); return FALSE; }
  if (g_eng == NULL && jamd_engine_create(dev ? atoi(dev) : 0, &g_eng) != JAMD_OK) {
    jlog("Error: jamd plugin: %s\n", jamd_last_error());
    return FALSE;
  }
  /* what the sample plugin allocates (plugin/calcmix.c:311-321) */
  wrk->OP_calced_maxnum = hmm->maxmixturenum * wrk->OP_nstream;
  wrk->OP_calced_score = (LOGPROB *)malloc(sizeof(LOGPROB) * wrk->OP_calced_maxnum);
  wrk->OP_calced_id = (int *)malloc(sizeof(int) * wrk->OP_calced_maxnum);
  wrk->OP_gprune_num = wrk->OP_calced_maxnum;
  /* the model as outprob_init() left it (variances already inverted, outprob_init.c:74-79) */
  if (jamd_flatten_hmminfo(hmm, &fg) != 0) { jlog("Error: jamd plugin: cannot flatten the acoustic model\n"); return FALSE; }
  g_ctx = (plug_ctx *)realloc(g_ctx, sizeof(plug_ctx) * (size_t)(g_nctx + 1));
  c = &g_ctx[g_nctx];
  memset(c, 0, sizeof(*c));
  c->wrk = wrk;
  if (jamd_gmm_create(g_eng, &fg.desc, JAMD_GPRUNE_NONE, 0, &c->gmm) != JAMD_OK) {
    jlog("Error: jamd plugin: %s\n", jamd_last_error());
    jamd_flat_gmm_free(&fg);
    return FALSE;
  }
  c->nentry = jamd_gmm_nentry(c->gmm); c->nstate = fg.desc.nstate; c->veclen = fg.desc.veclen;
  c->tied = hmm->is_tied_mixture ? 1 : 0; c->nbook = jamd_gmm_nbook(c->gmm);
  c->st_off = (int *)malloc(sizeof(int) * (size_t)(c->nstate + 1));
  memcpy(c->st_off, fg.desc.st_off, sizeof(int) * (size_t)(c->nstate + 1));
  jamd_flat_gmm_free(&fg);
  if (c->nentry <= 0) { jlog("Error: jamd plugin: models mixing plain and tied-mixture states are not served through the slot\n"); return FALSE; }
  if (c->tied) {
    c->book_off = (int *)malloc(sizeof(int) * (size_t)(c->nbook + 1));
    if (jamd_gmm_book_offsets(c->gmm, c->book_off, c->nbook + 1) != JAMD_OK) { jlog("Error: jamd plugin: %s\n", jamd_last_error()); return FALSE; }
  }
  g_nctx++;
  jlog("Stat: jamd plugin: Gaussian scores on HIP device %d (%d states, %d mixture components)\n",
       jamd_engine_device(g_eng), c->nstate, c->nentry);
  return TRUE;
}

#define FNV_INIT 1469598103934665603ull
static unsigned long long hash_frames(const HTK_Param *p, int from, int upto, unsigned long long h)
{
  int t; size_t i;
  for (t = from; t < upto; t++) {
    const unsigned char *b = (const unsigned char *)p->parvec[t];
    for (i = 0; i < sizeof(VECT) * (size_t)p->veclen; i++) { h ^= b[i]; h *= 1099511628211ull; }
  }
  return h;
}

/* rows [from, upto) of the current input: pack, score on the device, keep on the host */
static void fill(plug_ctx *c, const HTK_Param *param, int from, int upto)
{
  float *fr = jamd_pack_param((HTK_Param *)param, from, upto);
  if (fr == NULL) fatal("cannot pack the input");
  if (upto > c->cap) {
    c->cap = upto + 512;
    c->dens = (float *)realloc(c->dens, sizeof(float) * (size_t)c->cap * c->nentry);
    if (c->dens == NULL) { fprintf(stderr, "jamd calcmix plugin: out of memory\n"); exit(1); }
  }
  if (jamd_gmm_dens_host(c->gmm, fr, upto - from, c->dens + (size_t)from * c->nentry) != JAMD_OK) fatal("device scoring");
  free(fr);
  c->key = hash_frames(param, from, upto, from > 0 ? c->key : FNV_INIT);
  c->filled = upto;
  g_fills++;
}

void calcmix(HMMWork *wrk, HTK_HMM_Dens **g, int num, int *last_id, int lnum)
{
  plug_ctx *c = ctx_of(wrk);
  const HTK_Param *param = wrk->OP_param;
  const int t = wrk->OP_time;
  const float *row;
  int i, col0;
  (void)last_id; (void)lnum;
  if (c == NULL || param == NULL || t < 0 || t >= param->samplenum || param->veclen != c->veclen) {
    fprintf(stderr, "jamd calcmix plugin: called outside an input it can serve\n");
    exit(1);
  }
  if (c->tied) {                /* calc_tied_mix(): the current state's codebook */
    const GCODEBOOK *book = (const GCODEBOOK *)(wrk->OP_state->pdf[0]->b);
    if (!wrk->OP_state->pdf[0]->tmix || g != book->d || num != book->num || book->id < 0 || book->id >= c->nbook ||
        c->book_off[book->id + 1] - c->book_off[book->id] != num) {
      fprintf(stderr, "jamd calcmix plugin: called for a Gaussian set that is not the current state's codebook\n");
      exit(1);
    }
    col0 = c->book_off[book->id];
  } else {
    if (g != wrk->OP_state->pdf[0]->b || num != wrk->OP_state->pdf[0]->mix_num) {
      fprintf(stderr, "jamd calcmix plugin: called for a Gaussian set that is not the current state's mixture\n");
      exit(1);
    }
    col0 = c->st_off[wrk->OP_state_id];
  }
  /* every input is entered at frame 0 (init_nodescore(), libjulius/src/beam.c:1552): there the
   * rows kept from the previous input are checked against the frames they were computed from */
  if (t == 0 && c->filled > 0 &&
      (param->samplenum < c->filled || hash_frames(param, 0, c->filled, FNV_INIT) != c->key)) c->filled = 0;
  if (t >= c->filled) fill(c, param, c->filled, param->samplenum);      /* a new input, or one that grew (live) */
  row = c->dens + (size_t)t * c->nentry + col0;
  for (i = 0; i < num; i++) { wrk->OP_calced_id[i] = i; wrk->OP_calced_score[i] = row[i]; }
  wrk->OP_calced_num = num;
  g_calls++;
}

void calcmix_free(HMMWork *wrk)
{
  plug_ctx *c = ctx_of(wrk);
  free(wrk->OP_calced_score);
  free(wrk->OP_calced_id);
  if (c != NULL) {
    if (c->gmm) jamd_gmm_destroy(c->gmm);
    free(c->st_off); free(c->book_off); free(c->dens);
    *c = g_ctx[--g_nctx];
  }
}
