/*
 * jamd_flatten.h -- reference-side half of the drop-in boundary.
 *
 * These functions are compiled INSIDE the Julius tree (against the reference's
 * own <sent/...> / <julius/...> headers) and turn the pointer-linked model
 * structures the unmodified loaders build into the flat arrays that the C ABI
 * of include/julius_amd.h takes.  See INTEGRATION.md.
 */
#ifndef JAMD_FLATTEN_H
#define JAMD_FLATTEN_H

#include <sent/stddefs.h>
#include <sent/htk_hmm.h>
#include <sent/htk_param.h>
#include <sent/hmm.h>
#include <sent/hmm_calc.h>
#include <sent/dnn.h>
#include "julius_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Owned flat copy of an HTK_HMM_INFO's state pool (all arrays malloc'ed). */
typedef struct {
  jamd_gmm_desc desc;       /* points into the arrays below */
  float *mean, *ivar, *gconst;
  int *st_off, *ent_dens;
  float *ent_logw;
  int *st_book;
  int book_size_max;
} jamd_flat_gmm;

/* Walk hmminfo->ststart (HTK_HMM_State.id order), de-duplicate HTK_HMM_Dens by
 * pointer, and fill `out`.  Variances must already be inverted
 * (hmminfo->variance_inversed, outprob_init.c:74-79).  Returns 0 or JAMD_EINVAL
 * (multi-stream model, inconsistent vector lengths). */
int  jamd_flatten_hmminfo(HTK_HMM_INFO *hmminfo, jamd_flat_gmm *out);
void jamd_flat_gmm_free(jamd_flat_gmm *f);
/* Write the flattened model as a blob of named arrays ("JAMDGMM1"; reader:
 * julius_amd/lexblob.py::load_gmm) for workers that do not link Julius. */
int  jamd_gmm_save(const jamd_gmm_desc *d, const char *path);
int  jamd_gms_save(const jamd_gmm_desc *gs, const int *state2gs, int nstate, int nbest, const char *path);
int  jamd_rejgmm_save(const jamd_gmm_desc *gmm, const int *model_state, int nmodel, int gprune_num,
                      const unsigned char *is_voice, const char *const *names, const char *path);

/* Pack HTK_Param rows (each parvec[t] is a separate allocation,
 * libsent/src/anlz/param_malloc.c:52-77) into one [T][veclen] block. */
float *jamd_pack_param(const HTK_Param *param, int t0, int t1);

/* DNNData (libsent/include/sent/dnn.h:41-74) -> jamd_dnn_desc.  The weight matrices are
 * already contiguous [out][in] float arrays, so the descriptor points straight into the
 * DNNData (nothing is copied; `f` only owns the small pointer tables). */
typedef struct {
  jamd_dnn_desc desc;
  int *dims; const float **w; const float **b;
} jamd_flat_dnn;
int  jamd_flatten_dnn(DNNData *dnn, jamd_flat_dnn *out);
void jamd_flat_dnn_free(jamd_flat_dnn *f);

/* Copy n rows of [.][S] device scores into wrk->outprob_cache starting at frame t0 (growing
 * the cache like the reference's static outprob_cache_extend()), so that later outprob_state()
 * calls for those frames are cache hits. */
int  jamd_fill_outprob_cache(HMMWork *wrk, const float *scores, int t0, int n, int S);


/* ---- first pass: tree lexicon + LM tables (jamd_flatten_lex.c) ------------- */
#ifdef JAMD_WITH_LIBJULIUS   /* needs <julius/julius.h>; the GMM part above only needs libsent */
#include <julius/julius.h>

/* Owned flat copy of everything get_back_trellis_*() reads: all arrays malloc'ed,
 * `desc` points into them. */
typedef struct {
  jamd_lexicon_desc desc;
  float *self_a, *next_a; int *ac_off, *ac_to; float *ac_a;
  int *stend, *scid; unsigned char *out_kind; int *out_id;
  int *lc_tab, *word_lc, *set_off, *set_states;
  int *startnode, *start2isolate;
  float *wordend_a; int *wton; float *cprob; unsigned char *is_transparent; int *word_head;
  float *fscore; int *scword;
  float *ng_uni_prob, *ng_uni_bo; int *ng_bi_bgn, *ng_bi_num, *ng_bi_wid; float *ng_bi_prob;
  unsigned char *cat_pair; int *start2wid, *init_node; float *init_lscore;
  int *fwd_off, *fwd_label, *fwd_to, *init_to_state;      /* forward DFA (NULL without one) */
} jamd_flat_lexicon;

/* Walk r->wchmm (after j_final_fusion()) and fill `out`.  JAMD_EINVAL for the
 * configurations the device beam does not cover (grammars
 * without per-category trees, user LM plugin, 24-bit
 * compacted 2-gram index). */
int  jamd_flatten_lexicon(RecogProcess *r, jamd_flat_lexicon *out);
/* The same for a multipath acoustic model (lm_type | JAMD_LM_MULTIPATH); not yet accepted by
 * jamd_lexicon_create() -- used by the oracle's checks of the multipath search. */
int  jamd_flatten_lexicon_multipath(RecogProcess *r, jamd_flat_lexicon *out);
void jamd_flat_lexicon_free(jamd_flat_lexicon *f);
/* Write the descriptor as a self-describing blob of named arrays (the format
 * julius_amd/lexblob.py and jamd_lexicon_load() (include/julius_amd.h) read). */
int  jamd_lexicon_save(const jamd_lexicon_desc *d, const char *path);
int  jamd_lexicon_append_ngram_names(const char *path, const NGRAM_INFO *ng);
int  jamd_lexicon_append_separation(const char *path, int separate_wnum);

/* ---- batch of buffered inputs through the first-pass shim (jamd_pass1_shim.c) ----------------
 * Replaces the one-launch-per-utterance pattern of get_back_trellis() / decode_proceed()
 * (libjulius/src/pass1.c:111-317, :615-736) for file lists: queue every input, decode them all
 * in one device launch, then run Julius' usual per-input loop, which finds each first pass done.
 *   for (f in files) { j_open_stream(recog, f); jamd_pass1_prefetch_add(r, recog->mfcclist->param); }
 *   jamd_pass1_prefetch_run(r);
 *   for (f in files) { j_open_stream(recog, f); j_recognize_stream(recog); }       // unchanged loop
 * Inputs are recognised by length + content hash, so un-queued inputs simply take the normal path. */
int  jamd_pass1_prefetch_add(RecogProcess *r, HTK_Param *param);
int  jamd_pass1_prefetch_run(RecogProcess *r);
int  jamd_pass1_prefetch_served(RecogProcess *r);   /* inputs served from the batch so far */
void jamd_pass1_prefetch_clear(RecogProcess *r);
#endif

#ifdef __cplusplus
}
#endif
#endif
