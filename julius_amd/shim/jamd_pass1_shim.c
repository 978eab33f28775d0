/*
 * jamd_pass1_shim.c -- reference-side binding of the first-pass boundary (B).
 *
 * Compiled INSIDE the Julius tree against its own headers and linked IN PLACE OF
 * libjulius/src/beam.c: it exports the complete symbol set of beam.o
 *     get_back_trellis_init()     libjulius/src/beam.c:1825
 *     get_back_trellis_proceed()  beam.c:2663
 *     get_back_trellis_end()      beam.c:3052
 *     finalize_1st_pass()         beam.c:3133
 *     fsbeam_free()               beam.c:3179
 * (prototypes: libjulius/include/julius/extern.h:57-61) so that libjulius' callers
 * (pass1.c:234,242,503,567; instance.c:322) link unchanged.  Everything else of
 * Julius -- loaders, front end, 2nd pass, output -- is untouched.
 *
 * How the frame-by-frame call pattern maps onto the batched engine: _init() opens a
 * streaming session on the device (jamd_beam_stream_begin), _proceed(t) may hand the
 * frames that have arrived so far to it (every JAMD_STREAM_CHUNK frames: live input), and
 * _end() pushes the rest with the final flag.  For buffered input (-input htkparam /
 * rawfile, the benchmark path) the default is one push at _end(): scoring
 * and the whole first pass in one go.  _end() then rebuilds the BACKTRELLIS from the returned atoms with
 * the reference's own allocator (bt_new / bt_store), and finalize_1st_pass()
 * indexes it with bt_relocate_rw / bt_sort_rw exactly as the reference does.
 *
 * Supported configuration (anything else is refused loudly at _init()): N-gram LM, grammar or word
 * list; acoustic model -- multipath ones included -- GMM with any -gprune method, or DNN (-dnnconf);
 * no short-pause segmentation, buffered input.  The "no nodes left in beam" condition is
 * reported by failing the utterance (J_RESULT_STATUS_FAIL) instead of segmenting.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#define JAMD_WITH_LIBJULIUS 1
#include "jamd_flatten.h"

typedef struct {
  RecogProcess *r;             /* key */
  WCHMM_INFO *wchmm;           /* lexicon the handles were built from */
  HTK_HMM_INFO *hmminfo;
  int beam_width; float bs_width;
  jamd_gmm *gmm; jamd_dnn *dnn; jamd_lexicon *lex; jamd_beam *beam; jamd_gms *gms;
  int strict;                /* JAMD_STRICT_ORDER=1 / JAMD_ORDER_MODE=strict, or a multipath model the exact-order kernel cannot serve */
  jamd_trellis_atom *iatoms; int iatom_cap;   /* trellis so far, read back for -progout interim results */
  int order_mode;            /* -1 = the work area's default (exact order where the beam fits), else JAMD_ORDER_* (JAMD_ORDER_MODE) */
  int nstate;
  int nnode, nword; void *dfa; /* further identity of the lexicon tree (grammar updates) */
  float cfg_key[10];           /* scalar configuration baked into the device lexicon / scorer (cfg_key_of()) */
  /* streaming state of the current utterance */
  int chunk;                   /* JAMD_STREAM_CHUNK: push every this many frames from _proceed(); 0 = all at _end() */
  int whole_input;             /* the scoring carries state from frame to frame (-gshmm selection, -gprune heu/beam over
                                * tied-mixture codebooks): the input is scored in one piece, no interim results */
  int pushed;                  /* frames already handed to the device */
  int failed;
  float *host_scores; int host_cap;   /* [pushed][nstate] rows kept for the 2nd pass's cache */
  /* batch-of-utterances first pass: inputs queued by jamd_pass1_prefetch_add(), decoded in
   * one launch by jamd_pass1_prefetch_run(), served to get_back_trellis_*() by content */
  struct pre_entry *pre; int npre, pre_cap;
  int hit;                     /* entry serving the current utterance, -1 = none */
} pass1_ctx;

typedef struct pre_entry {
  unsigned long long key; int T, veclen;            /* identity of the input: length + content hash */
  float *frames;                                   /* [T][veclen], until the batch ran */
  int done, used;
  jamd_pass1_result res; jamd_trellis_atom *atoms; int natom;
  float *scores;                                   /* [T][nstate] for the 2nd pass, NULL with -1pass */
} pre_entry;

static jamd_engine *g_eng = NULL;
static pass1_ctx *g_ctx = NULL;      /* one per recognition process instance, grown on demand */
static int g_nctx = 0, g_capctx = 0;

static void pre_clear(pass1_ctx *c)
{
  int i;
  for (i = 0; i < c->npre; i++) { free(c->pre[i].frames); free(c->pre[i].atoms); free(c->pre[i].scores); }
  free(c->pre); c->pre = NULL; c->npre = c->pre_cap = 0; c->hit = -1;
}

static void ctx_release(pass1_ctx *c)
{
  pre_clear(c);
  if (c->beam) jamd_beam_destroy(c->beam);
  if (c->lex) jamd_lexicon_destroy(c->lex);
  if (c->gms) jamd_gms_destroy(c->gms);
  if (c->gmm) jamd_gmm_destroy(c->gmm);
  if (c->dnn) jamd_dnn_destroy(c->dnn);
  if (c->host_scores != NULL && g_eng != NULL) jamd_host_free(g_eng, c->host_scores);
  c->host_scores = NULL; c->host_cap = 0;
  free(c->iatoms); c->iatoms = NULL; c->iatom_cap = 0;
  c->beam = NULL; c->lex = NULL; c->gmm = NULL; c->dnn = NULL; c->gms = NULL;
}

static pass1_ctx *ctx_get(RecogProcess *r)
{
  int i;
  for (i = 0; i < g_nctx; i++) if (g_ctx[i].r == r) return &g_ctx[i];
  if (g_nctx == g_capctx) {
    pass1_ctx *n = (pass1_ctx *)realloc(g_ctx, sizeof(pass1_ctx) * (g_capctx ? 2 * g_capctx : 8));
    if (n == NULL) return NULL;
    g_ctx = n; g_capctx = g_capctx ? 2 * g_capctx : 8;
  }
  memset(&g_ctx[g_nctx], 0, sizeof(pass1_ctx));
  g_ctx[g_nctx].r = r;
  return &g_ctx[g_nctx++];
}

/* every scalar the device handles were built with: a change between inputs (module-mode commands, user code
 * editing the Jconf) must rebuild them */
static void cfg_key_of(RecogProcess *r, float *k)
{
  k[0] = r->config->lmp.lm_weight; k[1] = r->config->lmp.lm_penalty; k[2] = r->config->lmp.lm_penalty_trans;
  k[3] = r->config->lmp.penalty1; k[4] = (float)r->am->hmminfo->cdset_method; k[5] = (float)r->am->hmminfo->cdmax_num;
  k[6] = (float)r->am->config->gprune_method; k[7] = (float)r->am->hmmwrk.OP_gprune_num;
  k[8] = (float)(r->am->hmmwrk.OP_gshmm != NULL ? r->am->hmmwrk.my_nbest : 0); k[9] = (float)r->lmvar;
}

static boolean ctx_prepare(pass1_ctx *c, RecogProcess *r)
{
  float key[10];
  int gprune = JAMD_GPRUNE_NONE, rc;
  if (jamd_abi_version() != JAMD_ABI_VERSION) {   /* header this shim was compiled with vs the loaded library */
    jlog("ERROR: jamd: libjulius_amd.so has ABI %d, this shim was built for %d\n", jamd_abi_version(), JAMD_ABI_VERSION);
    return FALSE;
  }
  if (g_eng == NULL) {
    const char *dev = getenv("JAMD_DEVICE");
    if (jamd_engine_create(dev ? atoi(dev) : 0, &g_eng) != JAMD_OK) {
      jlog("ERROR: jamd: %s\n", jamd_last_error());
      return FALSE;
    }
  }
  /* a grammar update rebuilds the lexicon tree (multigram_build()); the sizes and the DFA pointer
   * also catch most rebuilt trees that landed on the old address */
  if (c->wchmm == r->wchmm && c->hmminfo == r->am->hmminfo && c->beam_width == r->trellis_beam_width &&
      c->bs_width == r->config->pass1.score_pruning_width && c->beam != NULL &&
      c->nnode == r->wchmm->n && c->nword == r->wchmm->winfo->num && c->dfa == (void *)r->wchmm->dfa &&
      (cfg_key_of(r, key), memcmp(key, c->cfg_key, sizeof(key)) == 0)) return TRUE;
  ctx_release(c);
  if ((r->lmtype != LM_PROB && r->lmtype != LM_DFA) || r->config->successive.enabled) {
    jlog("ERROR: jamd: the device first pass covers N-gram and grammar LMs, no -spsegment\n");
    return FALSE;
  }
  /* tie order (julius_amd.h, JAMD_ORDER_*): default = the reference's order (exact-order kernel); JAMD_ORDER_MODE =
   * fast | exact | strict chooses; multipath models: exact (their own frame-parallel frame) or strict */
  c->order_mode = -1;
  {
    const char *om = getenv("JAMD_ORDER_MODE");
    if (om != NULL && !strcmp(om, "fast")) c->order_mode = JAMD_ORDER_FAST;
    else if (om != NULL && !strcmp(om, "exact")) c->order_mode = JAMD_ORDER_EXACT;
    else if (om != NULL && !strcmp(om, "strict")) c->order_mode = JAMD_ORDER_STRICT;
  }
  c->strict = c->order_mode == JAMD_ORDER_STRICT ||
              (getenv("JAMD_STRICT_ORDER") != NULL && atoi(getenv("JAMD_STRICT_ORDER")) != 0);
  if (r->am->hmmwrk.OP_gshmm != NULL && r->am->dnn != NULL) {
    jlog("ERROR: jamd: Gaussian mixture selection (-gshmm) with a DNN-HMM is not supported\n");
    return FALSE;
  }
  if (r->am->dnn != NULL) {                    /* DNN-HMM: dnn_calc_outprob() for the whole utterance */
    jamd_flat_dnn fd;
    if (jamd_flatten_dnn(r->am->dnn, &fd) != JAMD_OK) { jlog("ERROR: jamd: cannot flatten the DNN\n"); return FALSE; }
    rc = jamd_dnn_create(g_eng, &fd.desc, &c->dnn);
    c->nstate = fd.desc.dims[fd.desc.nlayer];
    jamd_flat_dnn_free(&fd);
    if (rc != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
  } else {
    switch (r->am->config->gprune_method) {      /* jconf.h:94, values hmm_calc.h:38-45 */
    case GPRUNE_SEL_NONE: gprune = JAMD_GPRUNE_NONE; break;
    case GPRUNE_SEL_SAFE: gprune = JAMD_GPRUNE_SAFE; break;
    case GPRUNE_SEL_HEURISTIC: gprune = JAMD_GPRUNE_HEU; break;     /* tied-mixture codebooks: the reference's values under */
    case GPRUNE_SEL_BEAM: gprune = JAMD_GPRUNE_BEAM; break;         /* eager scoring (history = frame t-1 of the input)     */
    default:
      jlog("ERROR: jamd: unknown -gprune method\n");
      return FALSE;
    }
    {
      jamd_flat_gmm fg;
      if (jamd_flatten_hmminfo(r->am->hmminfo, &fg) != 0) { jlog("ERROR: jamd: cannot flatten the acoustic model\n"); return FALSE; }
      rc = jamd_gmm_create(g_eng, &fg.desc, gprune, r->am->hmmwrk.OP_gprune_num, &c->gmm);
      c->nstate = fg.desc.nstate;
      jamd_flat_gmm_free(&fg);
      if (rc != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
    }
    if (r->am->hmmwrk.OP_gshmm != NULL) {
      /* Gaussian mixture selection (gms_init(), libsent/src/phmm/gms.c:275-317): the device scores
       * every state anyway; the stage below REPLACES the scores the reference would not have
       * computed by the selection state's, so the search sees the reference's numbers */
      HMMWork *wrk = &(r->am->hmmwrk);
      jamd_flat_gmm fs;
      int *map = (int *)malloc(sizeof(int) * (size_t)c->nstate), s;
      if (map == NULL || jamd_flatten_hmminfo(wrk->OP_gshmm, &fs) != 0) { free(map); jlog("ERROR: jamd: cannot flatten the selection model\n"); return FALSE; }
      for (s = 0; s < c->nstate; s++)                  /* selection state ids are the flattened order */
        map[s] = (wrk->state2gs[s] >= 0 && wrk->state2gs[s] < wrk->gsset_num) ? wrk->gsset[wrk->state2gs[s]].state->id : -1;
      rc = jamd_gms_create(g_eng, &fs.desc, map, c->nstate, wrk->my_nbest, &c->gms);
      jamd_flat_gmm_free(&fs); free(map);
      if (rc != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
      if (c->strict || c->order_mode != JAMD_ORDER_FAST) jamd_gms_set_strict_order(c->gms, 1);   /* the reference's heap decides boundary ties */
      jlog("STAT: jamd: Gaussian mixture selection on the device (%d selection states, %d selected per frame)\n",
           wrk->gsset_num, wrk->my_nbest);
    }
  }
  {
    jamd_flat_lexicon fl;
    if ((r->am->hmminfo->multipath ? jamd_flatten_lexicon_multipath(r, &fl) : jamd_flatten_lexicon(r, &fl)) != JAMD_OK) {
      jlog("ERROR: jamd: this lexicon / LM configuration is not served by the device first pass (include/julius_amd.h, "
           "\"NOT SERVED\": a grammar without category trees, a user-defined LM, no 1-gram factoring)\n");
      return FALSE;
    }
    rc = jamd_lexicon_create(g_eng, &fl.desc, &c->lex);
    jamd_flat_lexicon_free(&fl);
    if (rc != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
  }
  /* word trellis bound: a generous per-frame share of the beam, T <= 32767 */
  rc = jamd_beam_create(g_eng, c->lex, r->trellis_beam_width, r->config->pass1.score_pruning_width, 1,
                        1 << 20, &c->beam);
  if (rc != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
  /* a multipath model runs on the exact-order kernel's multipath frame (csrc/beam_exact_mp.h) where that can serve the
   * lexicon and the beam; else (a root that reaches a word end along its own arcs, a beam too wide) in strict order */
  if (r->am->hmminfo->multipath && (jamd_beam_order_mode(c->beam) != JAMD_ORDER_EXACT || c->order_mode == JAMD_ORDER_FAST)) c->strict = 1;
  if (c->strict && jamd_beam_set_strict_order(c->beam, 1) != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
  if (!c->strict && c->order_mode == JAMD_ORDER_FAST) jamd_beam_set_order_mode(c->beam, JAMD_ORDER_FAST);
  if (!c->strict && c->order_mode == JAMD_ORDER_EXACT && jamd_beam_set_order_mode(c->beam, JAMD_ORDER_EXACT) != JAMD_OK) {
    jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE;
  }
  {
    static const char *const names[] = {"fast (canonical tie breaks)", "strict (sequential)", "exact (reference tie order)", "exact (sequential extraction)"};
    const int m = jamd_beam_order_mode(c->beam);
    jlog("Stat: jamd: first pass on HIP device %d, beam %d, tie order: %s\n", jamd_engine_device(g_eng), r->trellis_beam_width,
         (m >= 0 && m < 4) ? names[m] : "?");
    if (m == JAMD_ORDER_FAST && c->order_mode != JAMD_ORDER_FAST)
      jlog("Warning: jamd: beam %d is too wide for the exact-order kernel's LDS image (it serves beams up to about 12 000); exactly tied "
           "hypotheses are resolved canonically, not in the reference's visiting order (JAMD_ORDER_MODE=strict gives the reference's "
           "order, slowly)\n", r->trellis_beam_width);
  }
  c->wchmm = r->wchmm; c->hmminfo = r->am->hmminfo;
  c->nnode = r->wchmm->n; c->nword = r->wchmm->winfo->num; c->dfa = (void *)r->wchmm->dfa;
  cfg_key_of(r, c->cfg_key);
  c->beam_width = r->trellis_beam_width; c->bs_width = r->config->pass1.score_pruning_width;
  jlog("STAT: jamd: first pass on HIP device %d (beam %d, %d states, %d lexicon nodes)\n",
       jamd_engine_device(g_eng), c->beam_width, c->nstate, r->wchmm->n);
  return TRUE;
}

/* ---- batch of utterances ---------------------------------------------------------
 * A driver that holds many buffered inputs (a file list) queues them, lets the device decode them
 * all in one launch -- one workgroup per utterance, the regime the first-pass kernel is built for --
 * and then runs Julius' normal per-input loop: get_back_trellis_init() recognises each input by
 * length + content hash and get_back_trellis_end() hands over the stored trellis instead of
 * launching anything.  Inputs that were not queued (or whose trellis overflowed) take the normal path. */
static unsigned long long frames_hash(const float *f, size_t n)
{
  const unsigned char *p = (const unsigned char *)f;
  unsigned long long h = 1469598103934665603ull;
  size_t i;
  for (i = 0; i < n * sizeof(float); i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

int jamd_pass1_prefetch_add(RecogProcess *r, HTK_Param *param)
{
  pass1_ctx *c = ctx_get(r);
  pre_entry *e;
  if (c == NULL || param == NULL || param->samplenum <= 0 || param->is_outprob || !ctx_prepare(c, r)) return JAMD_EINVAL;
  if (c->npre == c->pre_cap) {
    c->pre_cap = c->pre_cap ? 2 * c->pre_cap : 64;
    c->pre = (pre_entry *)realloc(c->pre, sizeof(pre_entry) * c->pre_cap);
  }
  e = &c->pre[c->npre];
  memset(e, 0, sizeof(*e));
  e->T = param->samplenum; e->veclen = param->veclen;
  e->frames = jamd_pack_param(param, 0, e->T);
  if (e->frames == NULL) return JAMD_EINVAL;
  e->key = frames_hash(e->frames, (size_t)e->T * e->veclen);
  c->npre++;
  return JAMD_OK;
}

/* one launch over entries [first, first+n): score all frames, run the first pass, fetch everything */
/* keep bit 0: store the score rows for the later passes; bit 1: on failure leave the inputs queued for a retry with a
 * smaller launch instead of marking them failed */
static int prefetch_chunk(pass1_ctx *c, RecogProcess *r, int first, int n, int keepflags)
{
  const int keep = keepflags & 1;
  int retry = (keepflags & 2) != 0;
  jamd_beam *bb = NULL;
  float *frames = NULL, *d_frames = NULL, *d_scores = NULL;
  jamd_pass1_result *res = NULL;
  int *off = (int *)malloc(sizeof(int) * (n + 1)), u, rc = JAMD_EINVAL, veclen = c->pre[first].veclen;
  size_t total = 0;
  if (off == NULL) { jlog("ERROR: jamd: batch first pass: out of memory\n"); return JAMD_ENOMEM; }
  off[0] = 0;
  for (u = 0; u < n; u++) { total += (size_t)c->pre[first + u].T; off[u + 1] = (int)total; }
  frames = (float *)malloc(sizeof(float) * total * veclen);
  res = (jamd_pass1_result *)malloc(sizeof(jamd_pass1_result) * n);
  if (frames == NULL || res == NULL) { rc = JAMD_ENOMEM; goto out; }
  for (u = 0; u < n; u++)
    memcpy(frames + (size_t)off[u] * veclen, c->pre[first + u].frames, sizeof(float) * (size_t)c->pre[first + u].T * veclen);
  /* every step keeps ITS return code: only a launch that did not fit (JAMD_ENOMEM / JAMD_ELAUNCH) is worth retrying smaller */
#define JAMD_STEP(call) do { rc = (call); if (rc != JAMD_OK) goto out; } while (0)
  JAMD_STEP(jamd_beam_create(g_eng, c->lex, c->beam_width, c->bs_width, n, 1 << 19, &bb));
  if (c->strict) JAMD_STEP(jamd_beam_set_strict_order(bb, 1));
  if (!c->strict && c->order_mode >= 0) JAMD_STEP(jamd_beam_set_order_mode(bb, c->order_mode));
  JAMD_STEP(jamd_malloc(g_eng, sizeof(float) * total * veclen, (void **)&d_frames));
  JAMD_STEP(jamd_malloc(g_eng, sizeof(float) * total * c->nstate, (void **)&d_scores));
  JAMD_STEP(jamd_memcpy_h2d(g_eng, d_frames, frames, sizeof(float) * total * veclen));
  JAMD_STEP(c->dnn ? jamd_dnn_outprob_dev(c->dnn, d_frames, (int)total, d_scores, NULL)
                   : jamd_gmm_outprob_utts_dev(c->gmm, d_frames, off, n, d_scores, NULL));
  if (c->gms) JAMD_STEP(jamd_gms_apply_dev(c->gms, d_frames, (int)total, off, n, d_scores, NULL));
  JAMD_STEP(jamd_beam_pass1_dev(bb, d_scores, c->nstate, off, n, NULL));
  JAMD_STEP(jamd_engine_sync(g_eng));
  JAMD_STEP(jamd_beam_results(bb, res, n));
#undef JAMD_STEP
  rc = JAMD_EINVAL;
  for (u = 0; u < n; u++) {
    pre_entry *e = &c->pre[first + u];
    e->res = res[u];
    e->done = -1;                                     /* unusable unless everything below succeeds */
    if (res[u].status == JAMD_PASS1_OVERFLOW) continue;   /* the normal path has the larger trellis area */
    e->natom = res[u].natom;
    e->atoms = (jamd_trellis_atom *)malloc(sizeof(jamd_trellis_atom) * (e->natom > 0 ? e->natom : 1));
    if (e->atoms == NULL || jamd_beam_trellis(bb, u, e->atoms, e->natom, &e->natom) != JAMD_OK) continue;
    if (keep) {
      e->scores = (float *)malloc(sizeof(float) * (size_t)e->T * c->nstate);
      if (e->scores == NULL ||
          jamd_memcpy_d2h(g_eng, e->scores, d_scores + (size_t)off[u] * c->nstate,
                          sizeof(float) * (size_t)e->T * c->nstate) != JAMD_OK) continue;
    }
    e->done = 1;
  }
  rc = JAMD_OK;
out:
  if (rc != JAMD_ENOMEM && rc != JAMD_ELAUNCH) retry = 0;     /* a bad model or argument fails the same way at any size */
  if (rc != JAMD_OK && retry) jlog("STAT: jamd: a launch of %d queued inputs did not fit (%s): trying half of it\n", n, jamd_last_error());
  else if (rc != JAMD_OK) jlog("ERROR: jamd: batch first pass failed (rc %d): %s\n", rc, jamd_last_error());
  for (u = 0; u < n; u++) {
    if (rc != JAMD_OK && retry) { c->pre[first + u].done = 0; continue; }          /* frames stay for the retry */
    free(c->pre[first + u].frames); c->pre[first + u].frames = NULL;
    if (rc != JAMD_OK) c->pre[first + u].done = -1;
  }
  if (bb) jamd_beam_destroy(bb);
  if (d_frames) jamd_free(g_eng, d_frames);
  if (d_scores) jamd_free(g_eng, d_scores);
  free(frames); free(res); free(off);
  return rc;
}

int jamd_pass1_prefetch_run(RecogProcess *r)
{
  pass1_ctx *c = ctx_get(r);
  const int keep = !r->config->compute_only_1pass && getenv("JAMD_NO_CACHE_FILL") == NULL;
  int first = 0, rc = JAMD_OK, nrun = 0;
  if (c == NULL || c->beam == NULL) return JAMD_EINVAL;
  while (first < c->npre) {                           /* launches of <= 512 utterances (two per CU: the exact-order kernel's
                                                       * half shape, jamd_beam_set_workgroup_shape()) / 2^20 frames */
    int n = 0; size_t fr = 0;
    if (c->pre[first].done != 0) { first++; continue; }
    while (first + n < c->npre && c->pre[first + n].done == 0 && n < 512 &&
           c->pre[first + n].veclen == c->pre[first].veclen &&
           (n == 0 || fr + (size_t)c->pre[first + n].T <= ((size_t)1 << 20))) { fr += (size_t)c->pre[first + n].T; n++; }
    /* a launch that does not fit the device (score matrix + work area of n utterances: sized for an MI355X) is retried
     * in halves rather than given up: prefetch_chunk() leaves the inputs queued (done = 0) when told so */
    while (n > 1) {
      const int crc = prefetch_chunk(c, r, first, n, keep | 2);
      if (crc == JAMD_OK) break;
      if (crc != JAMD_ENOMEM && crc != JAMD_ELAUNCH) { rc = crc; break; }   /* reported at once; the inputs are marked failed */
      n = (n + 1) / 2;
    }
    if (n == 1 && c->pre[first].done == 0) { const int crc = prefetch_chunk(c, r, first, 1, keep); if (crc != JAMD_OK) rc = crc; }
    first += n; nrun += n;
  }
  jlog("STAT: jamd: batch first pass over %d queued inputs\n", nrun);
  return rc;
}

/* number of queued inputs whose stored first pass has been handed to Julius so far */
int jamd_pass1_prefetch_served(RecogProcess *r)
{
  pass1_ctx *c = ctx_get(r);
  int i, n = 0;
  for (i = 0; c != NULL && i < c->npre; i++) n += c->pre[i].used;
  return n;
}

void jamd_pass1_prefetch_clear(RecogProcess *r)
{
  pass1_ctx *c = ctx_get(r);
  if (c != NULL) pre_clear(c);
}

boolean get_back_trellis_init(HTK_Param *param, RecogProcess *r)
{
  pass1_ctx *c = ctx_get(r);
  FSBeam *d = &(r->pass1);
  if (c == NULL || !ctx_prepare(c, r)) return FALSE;
  bt_prepare(r->backtrellis);                         /* beam.c:1845 */
  d->bos.wid = WORD_INVALID;                          /* init_nodescore(), beam.c:1581-1584 */
  d->bos.begintime = d->bos.endtime = -1;
  outprob_style_cache_init(r->wchmm);                 /* beam.c:1595: the 2nd pass reuses these caches */
  r->have_interim = FALSE;
  /* interval of the progressive output in frames (-progout / -proginterval), beam.c:1886-1887 */
  r->config->output.progout_interval_frame =
      (int)((float)r->config->output.progout_interval / ((float)param->header.wshift / 10000.0));
  if (r->config->output.progout_interval_frame < 1) r->config->output.progout_interval_frame = 1;
  c->chunk = getenv("JAMD_STREAM_CHUNK") ? atoi(getenv("JAMD_STREAM_CHUNK")) : 0;
  if (c->strict) c->chunk = 0;                        /* one final push */
  c->whole_input = c->gms != NULL ||
                   (c->gmm != NULL && r->am->hmminfo->is_tied_mixture &&
                    (r->am->config->gprune_method == GPRUNE_SEL_HEURISTIC || r->am->config->gprune_method == GPRUNE_SEL_BEAM));
  if (c->whole_input) c->chunk = 0;
  c->pushed = 0; c->failed = 0; c->hit = -1;
  if (c->npre > 0 && param->samplenum > 0) {          /* decoded ahead in a batch? */
    float *fr = jamd_pack_param(param, 0, param->samplenum);
    if (fr != NULL) {
      const unsigned long long key = frames_hash(fr, (size_t)param->samplenum * param->veclen);
      int i;
      for (i = 0; i < c->npre; i++)
        if (c->pre[i].done == 1 && !c->pre[i].used && c->pre[i].T == param->samplenum &&
            c->pre[i].veclen == param->veclen && c->pre[i].key == key) { c->hit = i; break; }
      free(fr);
    }
    if (c->hit >= 0) return TRUE;
  }
  if (jamd_beam_stream_begin(c->beam, 1) != JAMD_OK) { jlog("ERROR: jamd: %s\n", jamd_last_error()); return FALSE; }
  return TRUE;
}

/* Score frames [c->pushed, upto) and advance the device search by them; `final` also runs
 * get_back_trellis_end() + traceback on the device. */
static boolean push_frames(pass1_ctx *c, RecogProcess *r, HTK_Param *param, int upto, int final)
{
  int n = upto - c->pushed, off[2];
  float *frames = NULL, *d_frames = NULL, *d_scores = NULL;
  boolean ok = FALSE;
  const boolean keep = !r->config->compute_only_1pass && getenv("JAMD_NO_CACHE_FILL") == NULL;
  off[0] = 0; off[1] = n;
  if (n > 0 && param->is_outprob) {
    /* -input outprob: the vectors ARE the state scores (outprob_state() returns parvec[t][id],
     * libsent/src/phmm/outprob.c:209-216); they go to the device search as they are, and the 2nd
     * pass reads them from the parameter itself */
    if (param->veclen != c->nstate) { jlog("ERROR: jamd: outprob vector size %d != %d states\n", param->veclen, c->nstate); goto out; }
    frames = jamd_pack_param(param, c->pushed, upto);
    if (frames == NULL ||
        jamd_malloc(g_eng, sizeof(float) * (size_t)n * c->nstate, (void **)&d_scores) != JAMD_OK ||
        jamd_memcpy_h2d(g_eng, d_scores, frames, sizeof(float) * (size_t)n * c->nstate) != JAMD_OK) goto out;
  } else if (n > 0) {
    frames = jamd_pack_param(param, c->pushed, upto);
    if (frames == NULL ||
        jamd_malloc(g_eng, sizeof(float) * (size_t)n * param->veclen, (void **)&d_frames) != JAMD_OK ||
        jamd_malloc(g_eng, sizeof(float) * (size_t)n * c->nstate, (void **)&d_scores) != JAMD_OK ||
        jamd_memcpy_h2d(g_eng, d_frames, frames, sizeof(float) * (size_t)n * param->veclen) != JAMD_OK ||
        (c->dnn ? jamd_dnn_outprob_dev(c->dnn, d_frames, n, d_scores, NULL)
                : jamd_gmm_outprob_dev(c->gmm, d_frames, n, d_scores, NULL)) != JAMD_OK ||
        (c->gms && jamd_gms_apply_dev(c->gms, d_frames, n, NULL, 0, d_scores, NULL) != JAMD_OK)) goto out;
    if (keep) {                                      /* rows for the reference's outprob cache (2nd pass) */
      if (upto > c->host_cap) {                      /* page-locked: the D2H copy of the rows runs at the PCIe rate */
        const int ncap = upto + (upto > 4096 ? upto / 2 : 2048);
        void *nb = NULL;
        if (jamd_host_alloc(g_eng, sizeof(float) * (size_t)ncap * c->nstate, &nb) != JAMD_OK) goto out;
        if (c->host_scores != NULL) {
          memcpy(nb, c->host_scores, sizeof(float) * (size_t)c->pushed * c->nstate);
          jamd_host_free(g_eng, c->host_scores);
        }
        c->host_scores = (float *)nb; c->host_cap = ncap;
      }
      if (c->host_scores == NULL ||
          jamd_memcpy_d2h(g_eng, c->host_scores + (size_t)c->pushed * c->nstate, d_scores,
                          sizeof(float) * (size_t)n * c->nstate) != JAMD_OK) goto out;
    }
  }
  if (jamd_beam_stream_push_dev(c->beam, d_scores, c->nstate, off, 1, final, NULL) != JAMD_OK) goto out;
  if (jamd_engine_sync(g_eng) != JAMD_OK) goto out;
  c->pushed = upto;
  ok = TRUE;
out:
  if (!ok) { jlog("ERROR: jamd: first pass failed: %s\n", jamd_last_error()); c->failed = 1; }
  if (d_frames) jamd_free(g_eng, d_frames);
  if (d_scores) jamd_free(g_eng, d_scores);
  free(frames);
  return ok;
}

/* bt_current_max(), beam.c:877-914: the best word sequence ending at frame t, from the trellis words the device
 * has emitted so far.  The reference walks backtrellis->list (newest first) over the atoms with endtime == t and
 * keeps the first strictly better one: among equal scores the LAST created.  The exact-order kernel emits the
 * atoms of a frame in the reference's creation order, so the same rule applies to the indices. */
static void interim_result(RecogProcess *r, const jamd_trellis_atom *atoms, int natom, int t)
{
  int i, best = -1;
  LOGPROB maxscore = LOG_ZERO;
  for (i = natom - 1; i >= 0; i--) {
    if (atoms[i].endtime != t) { if (atoms[i].endtime < t) break; else continue; }
    if (maxscore < atoms[i].backscore) { maxscore = atoms[i].backscore; best = i; }
  }
  r->result.status = J_RESULT_STATUS_SUCCESS;
  r->result.num_frame = t;
  if (best < 0) { r->result.pass1.word_num = 0; return; }
  if (r->lmvar == LM_DFA_WORD) {
    r->result.pass1.word[0] = (WORD_ID)atoms[best].wid;
    r->result.pass1.word_num = 1;
    r->result.pass1.score = atoms[best].backscore;
    r->result.pass1.score_lm = 0.0;
    r->result.pass1.score_am = atoms[best].backscore;
  } else {                                            /* trace_backptr(), beam.c:294-340 */
    WORD_ID rev[MAXSEQNUM];
    LOGPROB lm = 0.0;
    int n = 0, a = best;
    for (;;) {
      lm += atoms[a].lscore;
      rev[n++] = (WORD_ID)atoms[a].wid;
      if (atoms[a].begintime <= 0 || atoms[a].last_tre < 0 || n >= MAXSEQNUM) break;
      a = atoms[a].last_tre;
    }
    for (i = 0; i < n; i++) r->result.pass1.word[i] = rev[n - 1 - i];
    r->result.pass1.word_num = n;
    r->result.pass1.score = atoms[best].backscore;
    r->result.pass1.score_lm = lm;
    r->result.pass1.score_am = atoms[best].backscore;
  }
}

boolean get_back_trellis_proceed(int t, HTK_Param *param, RecogProcess *r, boolean final_for_multipath)
{
  /* Frame t is available.  With JAMD_STREAM_CHUNK = n the device search advances every n frames
   * (live input); by default everything is pushed at get_back_trellis_end() (buffered input). */
  pass1_ctx *c = ctx_get(r);
  r->have_interim = FALSE;
  if (c == NULL || c->beam == NULL || c->failed) return FALSE;
  /* -progout (beam.c:2983-2992): every progout_interval_frame frames the caller fires
   * CALLBACK_RESULT_PASS1_INTERIM with the best path so far.  The device search is advanced up to this frame
   * (the atoms ending at t-1 are emitted while frame t is processed) and the trellis so far is read back.  The
   * strict-order kernel and the selection stage need the whole input in one piece: no interim results there. */
  if (r->config->output.progout_flag && t > 0 && ((t - 1) % r->config->output.progout_interval_frame) == 0 &&
      (c->hit >= 0 || (!c->strict && !c->whole_input))) {
    if (c->hit >= 0) {
      interim_result(r, c->pre[c->hit].atoms, c->pre[c->hit].natom, t - 1);
      r->have_interim = TRUE;
    } else {
      jamd_pass1_result res;
      int natom = 0;
      if (c->pushed < t + 1 && !push_frames(c, r, param, t + 1, 0)) return FALSE;
      if (jamd_beam_results(c->beam, &res, 1) != JAMD_OK) return FALSE;
      if (res.status == JAMD_PASS1_DIED) {
        jlog("ERROR: jamd: frame %d: no nodes left in beam, now terminates search\n", res.died_at);
        return FALSE;
      }
      if (c->iatoms == NULL || res.natom > c->iatom_cap) {
        const int ncap = res.natom + 65536;
        jamd_trellis_atom *na = (jamd_trellis_atom *)realloc(c->iatoms, sizeof(jamd_trellis_atom) * (size_t)ncap);
        if (na == NULL) jlog("Warning: jamd: out of memory for the interim trellis (%d atoms): no interim result at frame %d\n", ncap, t - 1);
        else { c->iatoms = na; c->iatom_cap = ncap; }
      }
      if (c->iatoms != NULL && res.natom <= c->iatom_cap && jamd_beam_trellis(c->beam, 0, c->iatoms, res.natom, &natom) == JAMD_OK) {
        interim_result(r, c->iatoms, natom < res.natom ? natom : res.natom, t - 1);
        r->have_interim = TRUE;
      }
    }
    if (c->hit >= 0) return TRUE;
  }
  if (c->hit >= 0) return TRUE;                       /* served from the batch at _end() */
  if (c->chunk > 0 && t + 1 - c->pushed >= c->chunk) {
    jamd_pass1_result res;
    if (!push_frames(c, r, param, t + 1, 0)) return FALSE;
    if (jamd_beam_results(c->beam, &res, 1) == JAMD_OK && res.status == JAMD_PASS1_DIED) {
      jlog("ERROR: jamd: frame %d: no nodes left in beam, now terminates search\n", res.died_at);
      return FALSE;                                   /* beam.c:3012-3015: the caller segments the input */
    }
  }
  return TRUE;
}

void get_back_trellis_end(HTK_Param *param, RecogProcess *r)
{
  pass1_ctx *c = ctx_get(r);
  FSBeam *d = &(r->pass1);
  int T = param->samplenum, natom = 0, i;
  jamd_pass1_result res;
  jamd_trellis_atom *atoms = NULL;
  TRELLIS_ATOM **made = NULL;

  r->result.status = J_RESULT_STATUS_FAIL;            /* until proven otherwise */
  d->wordend_best_score = LOG_ZERO;
  if (c == NULL || c->beam == NULL || T <= 0 || c->failed) return;
  if (c->hit >= 0) {                                  /* decoded ahead by jamd_pass1_prefetch_run() */
    pre_entry *e = &c->pre[c->hit];
    res = e->res; natom = e->natom;
    atoms = e->atoms; e->atoms = NULL;                /* ownership moves here (freed below) */
    e->used = 1;
    if (res.status == JAMD_PASS1_DIED)
      jlog("ERROR: jamd: frame %d: no nodes left in beam, now terminates search\n", res.died_at);
    if (e->scores != NULL) {
      jamd_fill_outprob_cache(&(r->am->hmmwrk), e->scores, 0, T, c->nstate);
      free(e->scores); e->scores = NULL;
    }
    made = (TRELLIS_ATOM **)malloc(sizeof(TRELLIS_ATOM *) * (natom > 0 ? natom : 1));
  } else {
    if (!push_frames(c, r, param, T, 1) || jamd_beam_results(c->beam, &res, 1) != JAMD_OK) goto done;
    if (res.status == JAMD_PASS1_DIED)
      jlog("ERROR: jamd: frame %d: no nodes left in beam, now terminates search\n", res.died_at);
    if (c->host_scores != NULL && !r->config->compute_only_1pass && getenv("JAMD_NO_CACHE_FILL") == NULL)
      jamd_fill_outprob_cache(&(r->am->hmmwrk), c->host_scores, 0, T, c->nstate);   /* 2nd pass = cache hits */
    if (res.status == JAMD_PASS1_OVERFLOW) { jlog("ERROR: jamd: word trellis overflow\n"); goto done; }
    natom = res.natom;
    atoms = (jamd_trellis_atom *)malloc(sizeof(jamd_trellis_atom) * (natom > 0 ? natom : 1));
    made = (TRELLIS_ATOM **)malloc(sizeof(TRELLIS_ATOM *) * (natom > 0 ? natom : 1));
    if (jamd_beam_trellis(c->beam, 0, atoms, natom, &natom) != JAMD_OK) { natom = 0; goto done; }
  }
  /* save_trellis(), beam.c:2209-2247, for every atom the device emitted */
  for (i = 0; i < natom; i++) {
    TRELLIS_ATOM *tre = bt_new(r->backtrellis);
    tre->wid = (WORD_ID)atoms[i].wid;
    tre->backscore = atoms[i].backscore;
    tre->begintime = atoms[i].begintime;
    tre->endtime = atoms[i].endtime;
    tre->last_tre = (atoms[i].last_tre < 0) ? &(d->bos) : made[atoms[i].last_tre];
    tre->lscore = atoms[i].lscore;
    tre->dfa_state = -1;
    bt_store(r->backtrellis, tre);
    made[i] = tre;
  }
  /* what find_1pass_result() (beam.c:372-510) leaves behind */
  if (res.status == JAMD_PASS1_OK) {
    /* with -progout the result record keeps the last interim result unless -v is given (beam.c:498) */
    const boolean fill = verbose_flag || !r->config->output.progout_flag;
    r->result.status = J_RESULT_STATUS_SUCCESS;
    if (fill) r->result.num_frame = T;
    for (i = 0; i < res.wnum; i++) { r->pass1_wseq[i] = (WORD_ID)res.wseq[i]; if (fill) r->result.pass1.word[i] = (WORD_ID)res.wseq[i]; }
    r->pass1_wnum = res.wnum;
    r->pass1_score = res.score;
    if (fill) { r->result.pass1.word_num = res.wnum; r->result.pass1.score = res.score; }
    if (fill) {                                         /* trace_backptr(): total LM score */
      LOGPROB lm = 0.0; int a;
      for (a = natom - 1; a >= 0; a--)     /* the atom the device traced back from: the sentence's last word */
        if (atoms[a].wid == res.wseq[res.wnum - 1] && atoms[a].backscore == res.score) break;
      for (; a >= 0; a = atoms[a].last_tre) { lm += atoms[a].lscore; if (atoms[a].begintime <= 0) break; }
      r->result.pass1.score_lm = lm;
      r->result.pass1.score_am = res.score - lm;
    }
  }
done:
  free(atoms); free(made);
}

/* qsort order of the word-mode N-best list: the reference compares the truncated score difference
 * (compare_backscore(), beam.c:548-551), so words less than 1.0 apart keep qsort's arrangement */
static int by_backscore(const void *a, const void *b)
{
  return (int)((*(TRELLIS_ATOM *const *)b)->backscore - (*(TRELLIS_ATOM *const *)a)->backscore);
}

/* Isolated word recognition (-w): the first pass is the whole recognition; its N best words on the
 * last frame become the final result (find_1pass_result_word(), beam.c:561-698). */
static void word_mode_result(RecogProcess *r, int len)
{
  BACKTRELLIS *bt = r->backtrellis;
  TRELLIS_ATOM *best = NULL, **order;
  LOGPROB top = LOG_ZERO;
  int t, i, n, want;
#ifdef CONFIDENCE_MEASURE
  LOGPROB sum = 0.0;
#endif
  for (t = len - 1; t >= 0; t--) {
    for (i = 0; i < bt->num[t]; i++)
      if (top < bt->rw[t][i]->backscore) { top = bt->rw[t][i]->backscore; best = bt->rw[t][i]; }
    if (top != LOG_ZERO) break;
  }
  if (t < 0) {
    jlog("WARNING: %02d %s: no word survived on the last frame, search failed\n", r->config->id, r->config->name);
    r->result.status = J_RESULT_STATUS_FAIL;
    return;
  }
  n = bt->num[t];
#ifdef CONFIDENCE_MEASURE
  for (i = 0; i < n; i++) sum += pow(10, r->config->annotate.cm_alpha * (bt->rw[t][i]->backscore - top));
#endif
  r->result.status = J_RESULT_STATUS_SUCCESS;
  want = r->config->output.output_hypo_maxnum > 1 ? r->config->output.output_hypo_maxnum : 1;
  if (want > n) want = n;
  order = (TRELLIS_ATOM **)malloc(sizeof(TRELLIS_ATOM *) * n);
  if (r->config->output.output_hypo_maxnum > 1) {
    for (i = 0; i < n; i++) order[i] = bt->rw[t][i];
    qsort(order, n, sizeof(TRELLIS_ATOM *), by_backscore);
  } else order[0] = best;
  result_sentence_malloc(r, want);
  r->result.sentnum = want;
  for (i = 0; i < want; i++) {
    Sentence *s = &(r->result.sent[i]);
    s->word_num = 1;
    s->word[0] = order[i]->wid;
#ifdef CONFIDENCE_MEASURE
    s->confidence[0] = pow(10, r->config->annotate.cm_alpha * (order[i]->backscore - top)) / sum;
#endif
    s->score = order[i]->backscore;
    s->score_lm = 0.0;
    s->score_am = order[i]->backscore;
    s->gram_id = multigram_get_all_num(r->lm) > 0 ? multigram_get_gram_from_wid(s->word[0], r->lm) : 0;
  }
  free(order);
  memcpy(&(r->result.pass1), &(r->result.sent[0]), sizeof(Sentence));
  r->result.pass1.align = NULL;
}

void finalize_1st_pass(RecogProcess *r, int len)
{
  BACKTRELLIS *backtrellis = r->backtrellis;
  int status = r->result.status;
  backtrellis->framelen = len;                        /* beam.c:3139-3145 */
  bt_relocate_rw(backtrellis);
  bt_sort_rw(backtrellis);
  if (backtrellis->num == NULL) {
    if (backtrellis->framelen > 0)
      jlog("WARNING: %02d %s: input processed, but no survived word found\n", r->config->id, r->config->name);
    r->result.status = J_RESULT_STATUS_FAIL;
    return;
  }
  if (r->lmvar == LM_DFA_WORD) {                      /* beam.c:3157-3158; the device's choice is re-derived */
    word_mode_result(r, len);                         /* from the rebuilt trellis (needs the N-best list)  */
    return;
  }
  if (status != J_RESULT_STATUS_SUCCESS) {
    if (r->lmtype == LM_DFA)
      jlog("WARNING: %02d %s: no sentence-end word survived on last beam\n", r->config->id, r->config->name);
    else
      jlog("WARNING: %02d %s: no tail silence word survived on the last frame, search failed\n",
           r->config->id, r->config->name);
    r->result.status = J_RESULT_STATUS_FAIL;
  }
}

void fsbeam_free(FSBeam *d)
{
  int i;
  for (i = 0; i < g_nctx; i++)
    if (&(g_ctx[i].r->pass1) == d) { ctx_release(&g_ctx[i]); g_ctx[i] = g_ctx[--g_nctx]; break; }
  if (d->pausemodelnames != NULL) { free(d->pausemodelnames); free(d->pausemodel); }
  if (d->boslist != NULL) free(d->boslist);
}
